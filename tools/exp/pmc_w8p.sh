# SQ counters + HBM-side traffic of the 256x256 LDS-DMA 8-phase weight-gradient kernel next to the 128x128 tile (developer tool, GPU box).
# usage: bash tools/exp/pmc_w8p.sh [N H W C OC k s pad]      (default: UpHead conv_2 at batch 8)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_w8p
mkdir -p $OUT
for v in 0 8 9; do      # 0: 128x128 tile; 8: 8-phase, XCD-aware; 9: 8-phase in launch order
  export EMRT_WGRAD8P_MIN_STEPS=$([ $v = 0 ] && echo 0 || echo 8)
  export EMRT_WGRAD8P_XCD=$([ $v = 9 ] && echo 0 || echo 1)
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d $OUT/a$v -- python3 tools/exp/pmc_conv.py "$@" > /dev/null 2> $OUT/a$v.err
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/b$v -- python3 tools/exp/pmc_conv.py "$@" > /dev/null 2> $OUT/b$v.err
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f$v -- python3 tools/exp/pmc_conv.py "$@" > /dev/null 2> $OUT/f$v.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/w$v -- python3 tools/exp/pmc_conv.py "$@" > /dev/null 2> $OUT/w$v.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s$v -- python3 tools/exp/pmc_conv.py "$@" > /dev/null 2> $OUT/s$v.err
done
python3 - <<'PY'
import csv, glob, collections
OUT = "gpurun_out/pmc_w8p"
for v, label in ((0, "128x128 register-staged, fp32 atomics"), (8, "256x256 LDS-DMA 8-phase, slices pinned to XCDs, slab + reduce"), (9, "256x256 LDS-DMA 8-phase, launch order")):
    print("==== %s" % label)
    for d in ("a", "b", "f", "w"):
        f = glob.glob("%s/%s%d/*/*counter_collection.csv" % (OUT, d, v))
        if not f:
            print("no counters in", d, v, open("%s/%s%d.err" % (OUT, d, v)).read()[-500:]); continue
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f[0])):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
            if "wgrad" in name:
                agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, vv in agg.items():
            print(k)
            for cn, vals in sorted(vv.items()):
                print("    %-28s %.4g (n=%d)" % (cn, sum(vals) / len(vals), len(vals)))
    f = glob.glob("%s/s%d/*/*kernel_stats.csv" % (OUT, v))
    for r in csv.DictReader(open(f[0])):
        if "wgrad" in r["Name"]:
            print(r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, "us")
PY
echo "HBM-side traffic per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB; gfx950 counts 128-byte read requests as 64 B: MI355X_MICROARCH.md, HBM)."
echo "Algorithmic bytes of the default shape (8x128x128x256 -> 256, 3x3, bf16): x 67.1 MB + dy 67.1 MB + dW 2.4 MB (fp32, read-modify-write 4.7 MB) = 139 MB."
rm -rf $OUT/a? $OUT/b? $OUT/s? $OUT/f? $OUT/w?
