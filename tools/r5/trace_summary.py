"""rocprofv3 kernel_trace.csv -> the timeline of ONE hipGraph replay of the training step (the `back`-th optimizer launch from the end, default 6): per kernel duration, the gap to its predecessor,
and totals per kernel name (developer tool)."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# a step starts at the zero-gradient memset / first kernel after sgd: find the sgd_momentum launches
idx = [i for i, n in enumerate(names) if "sgd_momentum" in n]
if len(idx) < 3:
    print("no steps found", len(rows))
    sys.exit(0)
back = int(sys.argv[3]) if len(sys.argv) > 3 else 6
a, b = idx[-back - 1] + 1, idx[-back] + 1          # a CAPTURED replay from the middle of the timed region: the last three optimizer launches of a bench.py run belong to its eager per-launch profile (round 4 summarised one of those by mistake)
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])
busy = 0
gaps = 0
per = defaultdict(lambda: [0, 0, 0])
prev_end = None
with open(sys.argv[2], "w") as f:
    for r in step:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = 0 if prev_end is None else s - prev_end
        prev_end = max(prev_end or e, e)
        busy += e - s
        gaps += max(gap, 0)
        nm = r["Kernel_Name"]
        short = nm.split("(")[0][:70]
        p = per[short]
        p[0] += 1
        p[1] += e - s
        p[2] += max(gap, 0)
        f.write("%9.2f us  dur %8.2f  gap %6.2f  grid %8s wg %5s  %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?"), short))
wall = int(step[-1]["End_Timestamp"]) - t0
print("kernels %d  wall %.3f ms  busy %.3f ms  gaps %.3f ms" % (len(step), wall / 1e6, busy / 1e6, gaps / 1e6))
for k, (n, d, g) in sorted(per.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print("%-72s %4d  dur %8.1f us  (avg %6.2f)  gap-before %7.1f us (avg %5.2f)" % (k, n, d / 1e3, d / n / 1e3, g / 1e3, g / n / 1e3))
print("kernels %d  wall %.3f ms  busy %.3f ms  gaps %.3f ms" % (len(step), wall / 1e6, busy / 1e6, gaps / 1e6))
