cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
timeout 600 python3 -m pytest tests/test_gpu_kernels.py -q -m gpu -k "mha or ffn or linear_group or softmax_ce" -p no:xdist > $O/run11_kern.txt 2>&1; grep -E "passed|failed|^E  " $O/run11_kern.txt | tail -5
for v in "A=default" "EMRT_FFN_DROPOUT_FUSED=0" "A=default2"; do
  env $v timeout 300 python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs > $O/run11_bench.json 2> $O/run11_bench.err
  python3 -c "import json,sys; d=json.loads(open('$O/run11_bench.json').read().strip().splitlines()[-1]); print('[$v]', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_max'], d['roofline']['frac'], d['final_loss'])"
done
bash tools/r5/profile_round.sh r5a > $O/profile_round.log 2>&1; tail -12 $O/profile_round.log
