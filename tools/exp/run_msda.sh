mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in new old; do
  if [ $v = old ]; then export EMRT_HIP_LIB=$GRAFT_REPO_ROOT/tools/exp/libemrt_hip_old.so; fi
  for cfg in "cfg2 bf16" "cfg5 fp16" "cfg3 bf16"; do
    python3 tools/bench_msda.py $cfg 2>&1 | grep -v amdgpu
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pm_$v -- python3 tools/bench_msda.py $cfg > /dev/null 2>&1
    python3 - <<PY
import csv, glob
for f in glob.glob("gpurun_out/pm_$v/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "msda_fwd" in r["Name"]:
            print("   $v $cfg", r["Name"][:60], r["Calls"], "avg %.1f us min %.1f" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
    rm -rf gpurun_out/pm_$v
  done
done
