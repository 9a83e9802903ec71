cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
bash tools/r5/profile_round.sh r5c > $O/profile_round_r5c.log 2>&1; tail -4 $O/profile_round_r5c.log
mkdir -p profiles_tmp; cp gpurun_out/r5c/keep/r5c_pmc_traffic*.json profiles/ 2>/dev/null      # the default bench line below reads the fresh counters
T0=$(date +%s)
timeout 900 python3 bench.py > $O/run25_bench_default.json 2> $O/run25_bench_default.err
echo "default bench wall seconds: $(( $(date +%s) - T0 ))"
python3 -c "
import json
d=json.loads(open('$O/run25_bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_max'], d['roofline']['frac'], d['roofline'].get('frac_net_of_event_cost'), d['roofline'].get('traffic'), d['roofline'].get('traffic_note'), d['final_loss'], d['cpu_baseline']['value'])
for k,v in d['other_configs'].items(): print(k, v['value'], v['ms_per_step'], v.get('ms_per_step_median'), v.get('ms_per_step_max'), v['roofline']['frac'], v['roofline'].get('traffic_over_algorithmic'), v['roofline'].get('traffic_note'))
"
bash tools/r5/suite.sh suite25 > $O/run25_suite.log 2>&1; grep -E "passed|failed|suite wall" $O/run25_suite.log | tail -3
