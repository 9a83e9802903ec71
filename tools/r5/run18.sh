cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_msda_fuzz.py tests/test_gpu_kernels.py tests/test_gpu_bench_shapes.py -q -m gpu -k "msda or scatter or deform" -p no:xdist -s > $O/run18_msda.txt 2>&1; grep -E "passed|failed|^E  |matrix product|scatter_mfma = " $O/run18_msda.txt | tail -50
for v in "A=1" "EMRT_MSDA_SCATTER_MFMA=0"; do
  echo "[$v] cfg3"; env $v timeout 300 python3 tools/bench_msda.py cfg3 2>&1 | grep -v amdgpu.ids | head -1
done
bash tools/r5/suite.sh > $O/run18_suite.log 2>&1; tail -25 $O/run18_suite.log
