cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
T0=$(date +%s); timeout 900 python3 bench.py > $O/run29_bench_default.json 2> $O/run29_bench_default.err; echo "default bench wall seconds: $(( $(date +%s) - T0 ))"
python3 -c "
import json
d=json.loads(open('$O/run29_bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_max'], d.get('slowest_step'), d['roofline']['frac'], d['roofline'].get('frac_net_of_event_cost'), d['roofline'].get('traffic'), d['final_loss'], d['cpu_baseline']['value'])
for k,v in d['other_configs'].items(): print(k, v['value'], v['ms_per_step'], v.get('ms_per_step_median'), v.get('ms_per_step_max'), v.get('slowest_step'), v['roofline']['frac'], v['roofline'].get('traffic_over_algorithmic'))
"
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/r5/suite.sh suite29 -x > $O/run29_suite.log 2>&1; grep -E "passed|failed|suite wall" $O/run29_suite.log | tail -3
