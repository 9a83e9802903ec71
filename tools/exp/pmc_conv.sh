cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_conv
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d gpurun_out/pmc_conv/a -- python3 tools/exp/pmc_conv.py > /dev/null 2> gpurun_out/pmc_conv/a.err
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_conv/b -- python3 tools/exp/pmc_conv.py > /dev/null 2> gpurun_out/pmc_conv/b.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc_conv/s -- python3 tools/exp/pmc_conv.py > /dev/null 2> gpurun_out/pmc_conv/s.err
python3 - <<'PY'
import csv, glob, collections
for d in ("a", "b"):
    f = glob.glob("gpurun_out/pmc_conv/%s/*/*counter_collection.csv" % d)
    if not f:
        print("no counters in", d, open("gpurun_out/pmc_conv/%s.err" % d).read()[-500:]); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
        if "igemm" in name or "wgrad" in name:
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(k)
        for cn, vals in sorted(v.items()):
            print("    %-28s %.4g (n=%d)" % (cn, sum(vals) / len(vals), len(vals)))
f = glob.glob("gpurun_out/pmc_conv/s/*/*kernel_stats.csv")
for r in csv.DictReader(open(f[0])):
    if "igemm" in r["Name"] or "wgrad" in r["Name"]:
        print(r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, "us")
PY
rm -rf gpurun_out/pmc_conv/a gpurun_out/pmc_conv/b gpurun_out/pmc_conv/s
