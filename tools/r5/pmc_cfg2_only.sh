# FETCH / WRITE passes of the headline config only (tools/pmc_summary.py without the per-layer table): refreshes profiles/<tag>_pmc_traffic.{json,csv}
TAG=${1:-r5c}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${TAG}x; mkdir -p $OUT/keep
EAGER="--steps 2 --warmup 0 --no-graph --no-cpu-baseline --no-other-configs"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $EAGER --dump-calls $OUT/calls_eager.txt > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $EAGER > /dev/null 2> $OUT/pmc_write.err
python3 tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write $OUT/keep/${TAG}_pmc_traffic --calls $OUT/calls_eager.txt > $OUT/pmc_summary.log 2>&1
tail -30 $OUT/pmc_summary.log
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete
cp $OUT/keep/${TAG}_pmc_traffic.json profiles/
timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline'].get('traffic'), d['roofline_msda']['backward'].get('traffic'), d['roofline_msda']['backward'].get('traffic_unit'))"
