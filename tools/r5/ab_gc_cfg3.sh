cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
for i in 1 2 3 4; do
  for v in "--keep-gc" ""; do
    timeout 300 python3 bench.py --config cfg3 --steps 30 --warmup 8 --no-cpu-baseline $v > $O/run26_bench.json 2> $O/run26_bench.err
    python3 -c "import json,sys; d=json.loads(open('$O/run26_bench.json').read().strip().splitlines()[-1]); print('[cfg3 $v]', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_max'], d.get('slowest_step'))"
  done
done
T0=$(date +%s); timeout 900 python3 bench.py > $O/run26_bench_default.json 2> $O/run26_bench_default.err; echo "default bench wall seconds: $(( $(date +%s) - T0 ))"
python3 -c "
import json
d=json.loads(open('$O/run26_bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_max'], d.get('slowest_step'), d['roofline']['frac'], d['roofline'].get('traffic'), d['final_loss'])
for k,v in d['other_configs'].items(): print(k, v['value'], v['ms_per_step'], v.get('ms_per_step_median'), v.get('ms_per_step_max'), v.get('slowest_step'))
"
