cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
EMRT_LN_BWD_THREADS=512 timeout 600 python3 -m pytest tests/test_gpu_kernels.py -q -m gpu -k "layer_norm" -p no:xdist > $O/run20_kern.txt 2>&1; grep -E "passed|failed|^E  " $O/run20_kern.txt | tail -4
for v in "A=default" "EMRT_LN_BWD_THREADS=512" "EMRT_LN_BWD_THREADS=512 EMRT_LN_BWD_ROWS=64" "EMRT_LN_BWD_THREADS=512 EMRT_LN_BWD_ROWS=64 EMRT_LN_BWD_MAX_BLOCKS=256" "EMRT_LN_BWD_ROWS=16 EMRT_LN_BWD_MAX_BLOCKS=1024"; do
  env $v timeout 300 python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs > $O/run20_bench.json 2> $O/run20_bench.err
  python3 -c "import json,sys; d=json.loads(open('$O/run20_bench.json').read().strip().splitlines()[-1]); print('[$v]', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_max'], d['roofline']['frac'], d['final_loss'])"
  grep -E "layernorm_bwd" $O/run20_bench.err | head -2
done
