# SQ counters of the matrix-product value-gradient kernel (tools/bench_msda.py cfg2): where its wave cycles go.  Run on the GPU box through gpurun.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5/pmc_mf; mkdir -p $O
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/$tag -- python3 tools/bench_msda.py cfg2 > /dev/null 2> $O/$tag.err
  f=$(find $O/$tag -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "msda_bwd_value" in k or "msda_bwd_lds" in k:
        acc[k.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "dispatches", len(next(iter(d.values()))))
PY
done
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete
