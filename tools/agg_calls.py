"""Aggregate a bench.py --dump-calls file per entry point and per conv shape (developer tool)."""
import collections
import re
import sys

path = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
agg = collections.OrderedDict()
per = collections.Counter()
cnt = collections.Counter()
for line in open(path):
    parts = line.split(None, 3)
    if len(parts) < 3:
        continue
    name = parts[0]
    ms = float(parts[1])
    desc = parts[3].strip() if len(parts) > 3 else ""
    per[name] += ms
    cnt[name] += 1
    if name not in ("emrt_conv2d", "emrt_conv2d_wgrad", "emrt_conv2d_bwd"):
        continue
    d = re.sub(r" gflop.*", "", desc)
    g = float(re.search(r"gflop ([\d.]+)", desc).group(1)) if "gflop" in desc else 0
    a = agg.setdefault((name[5:], d), [0, 0.0, g])
    a[0] += 1
    a[1] += ms
print("total %.3f ms" % sum(per.values()))
for k, v in per.most_common(16):
    print(f"  {k:28s} n={cnt[k]:4d} {v:7.3f} ms")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{k[0]:13s} {k[1]:52s} n={a[0]:3d} tot={a[1]:6.3f} avg={a[1]/a[0]*1e3:7.1f}us {a[2]*a[0]/a[1] if a[1] else 0:6.1f} TF")
