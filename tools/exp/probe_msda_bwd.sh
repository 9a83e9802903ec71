# phases of the LDS value-gradient scatter (msda_bwd_value_lds_kernel) switched off through the probe knob (results WRONG):
#   16 = no |g| max scan, 32 = no sample loop, 48 = neither (zero fill + write-out only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for pr in 0 16 32 48; do
  BENCH_MSDA_PROBE=$pr rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p$pr -- python3 tools/bench_msda.py ${1:-cfg2} > /dev/null 2>&1
  python3 - $pr <<'PY'
import csv, glob, sys
for f in glob.glob('/tmp/p%s/*/*kernel_stats.csv' % sys.argv[1]):
    for r in csv.DictReader(open(f)):
        if 'msda_bwd' in r['Name']:
            print('probe', sys.argv[1], r['Name'][:60], 'calls', r['Calls'], 'avg %.1f us' % (float(r['AverageNs']) / 1e3))
PY
done
