cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_msda_fuzz.py tests/test_gpu_bench_shapes.py tests/test_gpu_kernels.py -q -m gpu -k "msda or scatter or deform or layer_norm" -p no:xdist > $O/run21_msda.txt 2>&1; grep -E "passed|failed|^E  " $O/run21_msda.txt | tail -8
echo "[mf]"; timeout 300 python3 tools/bench_msda.py cfg2 2>&1 | grep -v amdgpu.ids | head -1
for v in "A=default" "A=default2"; do
  env $v timeout 300 python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs > $O/run21_bench.json 2> $O/run21_bench.err
  python3 -c "import json,sys; d=json.loads(open('$O/run21_bench.json').read().strip().splitlines()[-1]); print('[$v]', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_max'], d['roofline']['frac'], d['final_loss'])"
  grep -E "layernorm_bwd|msda_bwd" $O/run21_bench.err | head -2
done
