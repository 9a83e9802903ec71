cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for sc in "1 3 6 8" "3 6 8" "1" "8" "6" "3"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python3 tools/exp/pool_probe.py $sc > /dev/null 2>&1
  python3 - "$sc" <<'PY'
import csv, glob, sys
for f in glob.glob('/tmp/pp/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'adaptive_pool' in r['Name']:
            print('scales', sys.argv[1], 'calls', r['Calls'], 'avg %.1f us' % (float(r['AverageNs']) / 1e3))
PY
  rm -rf /tmp/pp
done
