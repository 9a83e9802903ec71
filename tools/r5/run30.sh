cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_gn_levels.py -q -m gpu -k "group_norm or gn or layer_norm" -p no:xdist > $O/run30_kern.txt 2>&1; grep -E "passed|failed|^E  " $O/run30_kern.txt | tail -5
EMRT_LN_BWD_THREADS=1024 EMRT_LN_BWD_ROWS=48 timeout 600 python3 -m pytest tests/test_gpu_kernels.py -q -m gpu -k "layer_norm" -p no:xdist > $O/run30_ln.txt 2>&1; grep -E "passed|failed|^E  " $O/run30_ln.txt | tail -3
for v in "A=default" "EMRT_LN_BWD_THREADS=1024 EMRT_LN_BWD_ROWS=48" "EMRT_LN_BWD_THREADS=1024 EMRT_LN_BWD_ROWS=96" "A=default2"; do
  env $v timeout 300 python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs > $O/run30_bench.json 2> $O/run30_bench.err
  python3 -c "import json,sys; d=json.loads(open('$O/run30_bench.json').read().strip().splitlines()[-1]); print('[$v]', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_max'], d['roofline']['frac'], d['final_loss'])"
  grep -E "layernorm_bwd|groupnorm_levels_bwd" $O/run30_bench.err | head -2
done
