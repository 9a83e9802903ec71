# LDS bank-conflict share and MFMA / VALU busy per kernel over one eager step of the default bench (developer tool, GPU box)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_lds
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -- python3 bench.py --steps 2 --warmup 0 --no-graph --no-cpu-baseline "$@" > /dev/null 2> $OUT/a.err
python3 - <<'PY'
import csv, glob, collections, re
f = glob.glob("gpurun_out/pmc_lds/a/*/*counter_collection.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(f[0])):
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:64]
    agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": cnt[name] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0))
print("%-64s %6s %10s %9s %9s %9s" % ("kernel", "calls", "Mcycles", "conflict%", "mfma%", "valu%"))
for name, c in rows[:40]:
    g = c.get("GRBM_GUI_ACTIVE", 0)
    lds = c.get("SQ_LDS_IDX_ACTIVE", 0)
    print("%-64s %6d %10.2f %9.1f %9.1f %9.1f" % (name, cnt[name], g / 1e6, 100 * c.get("SQ_LDS_BANK_CONFLICT", 0) / lds if lds else 0.0,
          100 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * g) if g else 0, 100 * 4 * c.get("SQ_ACTIVE_INST_VALU", 0) / (1024 * g) if g else 0))
PY
rm -rf $OUT/a
