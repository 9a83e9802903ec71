# Indirect Infinity-Cache evidence: average TCC->EA read latency per kernel (TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ, in TCC cycles) and the share of
# read requests marked for DRAM, from one PMC pass of the eager step.  No MALL hit counter is exposed on this stack (rocprofv3 -L).
TAG=${1:-r4lat}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum --output-format csv -d $OUT/pmc -- python3 bench.py --steps 2 --warmup 0 --no-graph --no-cpu-baseline --no-other-configs > /dev/null 2> $OUT/pmc.err
f=$(find $OUT/pmc -name "*counter_collection.csv" | head -1)
[ -n "$f" ] || { echo "no counter file"; tail -5 $OUT/pmc.err; exit 1; }
python3 - "$f" > $OUT/${TAG}_ea_read_latency.txt <<'PY'
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in rows:
    k = re.sub(r"\(.*", "", r["Kernel_Name"])[:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "TCC_EA0_RDREQ_sum":
        cnt[k] += 1
print("%-72s %6s %12s %10s %8s" % ("kernel", "disp", "EA reads", "avg lat", "DRAM %"))
out = []
for k, v in agg.items():
    rd = v.get("TCC_EA0_RDREQ_sum", 0.0)
    if rd < 1e4:
        continue
    out.append((rd, k, cnt[k], v.get("TCC_EA0_RDREQ_LEVEL_sum", 0.0) / rd, 100.0 * v.get("TCC_EA0_RDREQ_DRAM_sum", 0.0) / rd))
for rd, k, n, lat, dram in sorted(out, reverse=True)[:40]:
    print("%-72s %6d %12.3e %10.1f %8.1f" % (k, n, rd, lat, dram))
PY
head -45 $OUT/${TAG}_ea_read_latency.txt
find $OUT -name "*.csv" -delete
