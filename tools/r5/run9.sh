cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
T0=$(date +%s)
timeout 2000 python3 -m pytest tests -q -m gpu --durations=12 > $O/run9_suite.txt 2>&1
echo "suite wall seconds (4 workers): $(( $(date +%s) - T0 ))" | tee -a $O/run9_suite.txt
grep -E "passed|failed" $O/run9_suite.txt | tail -3
grep -E "^[0-9.]+s call" $O/run9_suite.txt | head -8
grep -E "^FAILED|^ERROR" $O/run9_suite.txt | head
for v in "A=default" "EMRT_FFN_DROPOUT_FUSED=0" "EMRT_GROUP_ATTN_PROJ=0" "EMRT_XK=-1" "A=default2"; do
  env $v timeout 300 python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs > $O/run9_bench.json 2> $O/run9_bench.err
  python3 -c "import json,sys; d=json.loads(open('$O/run9_bench.json').read().strip().splitlines()[-1]); print('[$v]', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_max'], d['roofline']['frac'], d['final_loss'])"
done
