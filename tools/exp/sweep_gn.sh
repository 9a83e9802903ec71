# sweep of the row-major GroupNorm block shapes (per-call times from bench.py --dump-calls; event pair ~4.5 us included)
for sr in 16 32 64; do for br in 16 32 64 128; do for ar in 8 16 32; do
  EMRT_GN_STAT_ROWS=$sr EMRT_GN_BWD_STAT_ROWS=$br EMRT_GN_APPLY_ROWS=$ar python bench.py --no-cpu-baseline --steps 6 --warmup 2 --dump-calls /tmp/c.txt >/dev/null 2>&1
  f=$(grep groupnorm_levels_fwd /tmp/c.txt | awk '{s+=$2} END {printf "%.1f", s*1000/NR}')
  b=$(grep groupnorm_levels_bwd /tmp/c.txt | awk '{s+=$2} END {printf "%.1f", s*1000/NR}')
  echo "stat_rows $sr bwd_stat_rows $br apply_rows $ar: fwd $f us bwd $b us"
done; done; done
