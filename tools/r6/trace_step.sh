# Per-dispatch kernel trace of the default bench command (hipGraph replay): per-family totals in step order.  usage: bash tools/r6/trace_step.sh <tag>
TAG=${1:-r6t}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-other-configs > $OUT/bench_under_trace.json 2> $OUT/trace.err
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 tools/r5/trace_summary.py $f $OUT/${TAG}_timeline.txt > $OUT/${TAG}_timeline_summary.txt
find $OUT/trace -name "*.csv" -delete
head -1 $OUT/${TAG}_timeline_summary.txt
grep -E "ln_bwd|colsum_levels|igemm8p|igemm_drop" $OUT/${TAG}_timeline_summary.txt | cut -c1-140
