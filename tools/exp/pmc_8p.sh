# SQ counters of the 256x256 LDS-DMA 8-phase kernel next to the 128x128 tile on the UpHead conv_2 shape (developer tool, GPU box).
# usage: bash tools/exp/pmc_8p.sh [N H W C OC k s pad]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_8p
mkdir -p $OUT
for tile in 3 7; do
  export CONV_TILE=$tile
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d $OUT/a$tile -- python3 tools/exp/pmc_conv.py "$@" > /dev/null 2> $OUT/a$tile.err
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/b$tile -- python3 tools/exp/pmc_conv.py "$@" > /dev/null 2> $OUT/b$tile.err
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f$tile -- python3 tools/exp/pmc_conv.py "$@" > /dev/null 2> $OUT/f$tile.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/w$tile -- python3 tools/exp/pmc_conv.py "$@" > /dev/null 2> $OUT/w$tile.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s$tile -- python3 tools/exp/pmc_conv.py "$@" > /dev/null 2> $OUT/s$tile.err
done
python3 - <<'PY'
import csv, glob, collections
OUT = "gpurun_out/pmc_8p"
for tile in (3, 7):
    print("==== conv_tile %d (%s)" % (tile, "128x128 register-staged" if tile == 3 else "256x256 LDS-DMA 8-phase"))
    for d in ("a", "b", "f", "w"):
        f = glob.glob("%s/%s%d/*/*counter_collection.csv" % (OUT, d, tile))
        if not f:
            print("no counters in", d, tile, open("%s/%s%d.err" % (OUT, d, tile)).read()[-500:]); continue
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f[0])):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
            if "igemm" in name:
                agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            print(k)
            for cn, vals in sorted(v.items()):
                print("    %-28s %.4g (n=%d)" % (cn, sum(vals) / len(vals), len(vals)))
    f = glob.glob("%s/s%d/*/*kernel_stats.csv" % (OUT, tile))
    for r in csv.DictReader(open(f[0])):
        if "igemm" in r["Name"]:
            print(r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, "us")
PY
echo "HBM traffic per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB; gfx950 counts 128-byte read requests as 64 B: MI355X_MICROARCH.md, HBM)."
echo "Algorithmic bytes of the default shape (8x128x128x256 -> 256, 3x3, bf16): input 67.1 MB + output 67.1 MB + weights 1.2 MB = 135.4 MB (forward and dgrad alike)."
rm -rf $OUT/a3 $OUT/b3 $OUT/s3 $OUT/a7 $OUT/b7 $OUT/s7 $OUT/f3 $OUT/w3 $OUT/f7 $OUT/w7
