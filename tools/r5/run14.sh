cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_msda_fuzz.py -q -m gpu -p no:xdist -x > $O/run14_msda.txt 2>&1; grep -E "passed|failed|^E  " $O/run14_msda.txt | tail -8
for v in "A=1" "BENCH_MSDA_PROBE=256" "BENCH_MSDA_PROBE=512" "BENCH_MSDA_PROBE=768" "BENCH_MSDA_PROBE=1024" "EMRT_MSDA_SCATTER_MFMA=0"; do
  echo "[$v]"; env $v timeout 300 python3 tools/bench_msda.py cfg2 2>&1 | grep -v amdgpu.ids | head -1
done
for v in "A=1" "BENCH_MSDA_PROBE=256" "BENCH_MSDA_PROBE=512" "BENCH_MSDA_PROBE=768"; do
  echo "[$v] cfg3"; env $v timeout 300 python3 tools/bench_msda.py cfg3 2>&1 | grep -v amdgpu.ids | head -1
done
