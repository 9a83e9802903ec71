cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
T0=$(date +%s)
timeout 1500 python3 -m pytest tests/test_gpu_captured_step.py tests/test_gpu_wgrad_group.py tests/test_gpu_dp2.py tests/test_gpu_conv_fuzz.py -q -m gpu --durations=12 -s > $O/run2_tests.txt 2>&1
echo "tests wall seconds: $(( $(date +%s) - T0 ))" | tee -a $O/run2_tests.txt
grep -E "passed|failed" $O/run2_tests.txt | tail -3
grep -E "CAPTURED|step 1 vs" $O/run2_tests.txt
timeout 600 python3 tools/bench_conv.py xk > $O/run2_xk.txt 2>&1
cat $O/run2_xk.txt
for k in 0 -1; do
  EMRT_XK=$k timeout 300 python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs > $O/run2_bench_xk$k.json 2> $O/run2_bench_xk$k.err
  python3 -c "import json,sys; d=json.loads(open('$O/run2_bench_xk$k.json').read().strip().splitlines()[-1]); print('xk=$k', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_max'], d['roofline']['frac'], d['loss_check'])"
done
