cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
for k in 1 0; do
  EMRT_FFN_DROPOUT_FUSED=$k timeout 300 python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs --dump-calls $O/calls_fused$k.txt > $O/run10_bench$k.json 2> $O/run10_bench$k.err
  python3 -c "import json; d=json.loads(open('$O/run10_bench$k.json').read().strip().splitlines()[-1]); print('fused=$k', d['value'], d['ms_per_step'])"
  grep -A14 "per-launch HIP-event" $O/run10_bench$k.err | head -16
done
