cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
T0=$(date +%s)
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "mha or ffn or dropout" --durations=5 -s -p no:xdist > $O/run3_kern.txt 2>&1
echo "kernel tests wall: $(( $(date +%s) - T0 ))"; grep -E "passed|failed|MHA|dropout in" $O/run3_kern.txt | tail -12
T0=$(date +%s)
timeout 2000 python3 -m pytest tests -q -m gpu --durations=25 > $O/run3_suite.txt 2>&1
echo "suite wall seconds (4 workers): $(( $(date +%s) - T0 ))" | tee -a $O/run3_suite.txt
grep -E "passed|failed" $O/run3_suite.txt | tail -3
grep -E "^[0-9.]+s call" $O/run3_suite.txt | head -12
grep -E "^FAILED|^ERROR" $O/run3_suite.txt | head
for v in "" "EMRT_FFN_DROPOUT_FUSED=0" "EMRT_MHA_VALU=1" "EMRT_XK=-1"; do
  env $v timeout 300 python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs > $O/run3_bench.json 2> $O/run3_bench.err
  python3 -c "import json,sys; d=json.loads(open('$O/run3_bench.json').read().strip().splitlines()[-1]); print('[$v]', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_max'], d['roofline']['frac'])"
done
