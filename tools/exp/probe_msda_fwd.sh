# phases of the LDS-staged MSDA forward switched off through the probe knob (results WRONG), rocprofv3 kernel averages:
#   1 = no gather, 2 = no slab staging, 4 = no per-sample preparation (combinations add)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in cfg2 cfg5; do for pr in 0 1 2 3 4 5 6 7; do
  BENCH_MSDA_PROBE=$pr rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/q$cfg$pr -- python3 tools/bench_msda.py $cfg $([ $cfg = cfg5 ] && echo fp16) > /dev/null 2>&1
  python3 - $cfg $pr <<'PY'
import csv, glob, sys
for f in glob.glob('/tmp/q%s%s/*/*kernel_stats.csv' % (sys.argv[1], sys.argv[2])):
    for r in csv.DictReader(open(f)):
        if 'msda_fwd' in r['Name']:
            print(sys.argv[1], 'probe', sys.argv[2], r['Name'][:40], 'calls', r['Calls'], 'avg %.1f us' % (float(r['AverageNs']) / 1e3))
PY
done; done
