cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5/stat31; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > /dev/null 2> $O/err.txt
f=$(find $O -name "*kernel_stats.csv" | head -1)
grep -E "gn_rows|ln_bwd|ln_fwd" $f | awk -F'",' '{print $1 "  " $2}' | cut -c1-200
find $O -name "*kernel_trace.csv" -delete
