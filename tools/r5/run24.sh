cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -q -m gpu -k "channel_dropout or add_f32row or ffn or mha" -p no:xdist > $O/run24_kern.txt 2>&1; grep -E "passed|failed|^E  " $O/run24_kern.txt | tail -8
timeout 1500 python3 -m pytest tests/test_gpu_model.py tests/test_gpu_captured_step.py tests/test_gpu_dp2.py -q -m gpu -x > $O/run24_model.txt 2>&1; grep -E "passed|failed|^E  " $O/run24_model.txt | tail -8
for v in "A=default" "A=default2"; do
  env $v timeout 300 python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs > $O/run24_bench.json 2> $O/run24_bench.err
  python3 -c "import json,sys; d=json.loads(open('$O/run24_bench.json').read().strip().splitlines()[-1]); print('[$v]', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_max'], d['roofline']['frac'], d['final_loss'])"
done
grep -E "launches" $O/run24_bench.err | head -2
