TAG=${1:-r4tp}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-other-configs --two-phase > $OUT/bench.json 2> $OUT/trace.err
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 tools/r4/trace_summary.py $f $OUT/${TAG}_timeline.txt > $OUT/${TAG}_summary.txt
python3 - <<PY
import re
rows=[]
for ln in open("$OUT/${TAG}_timeline.txt"):
    m=re.match(r'\s*([\d.]+) us\s+dur\s+([\d.]+)\s+gap\s+([\d.]+)\s+grid\s+\S+ wg\s+\S+\s+(.*)',ln)
    rows.append((float(m.group(1)),float(m.group(2)),float(m.group(3)),m.group(4)))
print("kernels",len(rows),"busy",sum(r[1] for r in rows),"gaps",sum(r[2] for r in rows))
for i,r in enumerate(rows):
    if r[2]>8 or 'ccl' in r[3].lower() or 'reduce' in r[3].lower() and 'col_reduce' not in r[3] and 'wgrad' not in r[3]:
        print(i, r)
PY
find $OUT/trace -name "*.csv" -delete
