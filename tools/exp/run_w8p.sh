mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_bench_shapes.py -x -q -k "wgrad" 2>&1 | tail -15 > gpurun_out/r20_tests.log
timeout 600 python tools/bench_conv.py wbig > gpurun_out/r20_wbig.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_w8p -o w8p -- python3 $GRAFT_REPO_ROOT/tools/bench_conv.py wbig > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/prof_w8p/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.reader(open(f)))
    for r in rows[:12]:
        print([c[:70] for c in r[:4]])
PY
