mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_bench_shapes.py tests/test_gpu_kernels.py tests/test_gpu_fp16.py tests/test_gpu_msda_fuzz.py -x -q -k "msda or fp16" 2>&1 | tail -3
python3 tools/exp/msda_f16_vs_bf16.py 2>&1 | grep -v amdgpu
for cfg in "cfg3 bf16"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pm_x -- python3 tools/bench_msda.py $cfg > /dev/null 2>&1
  python3 - <<PY
import csv, glob
for f in glob.glob("gpurun_out/pm_x/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "msda_fwd" in r["Name"]:
            print("   $cfg", r["Name"][:60], r["Calls"], "avg %.1f us min %.1f" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
  rm -rf gpurun_out/pm_x
done
python bench.py --config cfg5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('cfg5', j['value'], j['roofline_msda']['avg_launch_us'], j['roofline_msda']['frac'])"
python bench.py --config cfg3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('cfg3', j['value'], j['roofline_msda']['avg_launch_us'], j['roofline_msda']['frac'])"
