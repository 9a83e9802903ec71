cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
for v in "A=default" "EMRT_FFN_DROPOUT_FUSED=0" "EMRT_MHA_VALU=1" "EMRT_XK=-1" "A=default2"; do
  env $v timeout 300 python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs > $O/run5_bench.json 2> $O/run5_bench.err
  python3 -c "import json,sys; d=json.loads(open('$O/run5_bench.json').read().strip().splitlines()[-1]); print('[$v]', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_max'], d['roofline']['frac'], d['final_loss'])"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/run5_stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/run5_bench_prof.json 2> $O/run5_stats.err
f=$(find $O/run5_stats -name "*kernel_stats.csv" | head -1); cp $f $O/run5_kernel_stats.csv; head -40 $O/run5_kernel_stats.csv | cut -c1-150
find $O/run5_stats -name "*.csv" -delete
for k in 0 1; do EMRT_MSDA_SCATTER_MERGE=$k python3 tools/bench_msda.py cfg2 bf16 2>&1 | grep -v amdgpu.ids | head -2; done
T0=$(date +%s)
timeout 2000 python3 -m pytest tests -q -m gpu --durations=12 > $O/run5_suite.txt 2>&1
echo "suite wall seconds (4 workers): $(( $(date +%s) - T0 ))" | tee -a $O/run5_suite.txt
grep -E "passed|failed" $O/run5_suite.txt | tail -3
grep -E "^[0-9.]+s call" $O/run5_suite.txt | head -8
grep -E "^FAILED|^ERROR" $O/run5_suite.txt | head
