# Per-dispatch kernel trace of the default bench command (hipGraph replay): durations AND the gaps between consecutive kernels, in step order.
TAG=${1:-r4a}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-other-configs > $OUT/bench_under_trace.json 2> $OUT/trace.err
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 tools/r5/trace_summary.py $f $OUT/${TAG}_timeline.txt > $OUT/${TAG}_timeline_summary.txt
find $OUT/trace -name "*.csv" -delete
python3 bench.py --steps 30 --warmup 10 --dump-calls $OUT/calls.txt > $OUT/bench.json 2> $OUT/bench.err
tail -3 $OUT/${TAG}_timeline_summary.txt
cat $OUT/bench.json | head -c 600
