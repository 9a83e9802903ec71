cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4final
T0=$(date +%s); timeout 600 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r4final/bench_default.json 2> gpurun_out/r4final/bench_default.err; echo "Elapsed $(( $(date +%s) - T0 )) s"
python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r4final/bench_default.json') if l.startswith('{')][-1])
print(d['metric'], d['value'], d['unit'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_net_of_event_cost'], d['cpu_baseline'])
print({k:(v['value'], v['ms_per_step']) for k,v in d['other_configs'].items()})
print(d['roofline_msda'] if 'roofline_msda' in d else '')
"
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
