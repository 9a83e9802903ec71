cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r1k
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r1k/stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r1k/bench_under_rocprof.json 2> gpurun_out/r1k/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/r1k/pmc_fetch -- python3 bench.py --steps 2 --warmup 0 --no-graph --no-cpu-baseline > /dev/null 2> gpurun_out/r1k/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/r1k/pmc_write -- python3 bench.py --steps 2 --warmup 0 --no-graph --no-cpu-baseline > /dev/null 2> gpurun_out/r1k/pmc_write.err
python3 tools/pmc_summary.py gpurun_out/r1k/pmc_fetch gpurun_out/r1k/pmc_write gpurun_out/r1k/r1k_pmc_traffic
cp gpurun_out/r1k/r1k_pmc_traffic.json profiles/r1k_pmc_traffic.json
python3 bench.py > gpurun_out/r1k/bench.json 2> gpurun_out/r1k/bench.err
ls gpurun_out/r1k gpurun_out/r1k/stats/* | head -30
# keep the merge small: drop the big traces
find gpurun_out/r1k -name "*kernel_trace.csv" -size +20M -delete
find gpurun_out/r1k -name "*counter_collection.csv" -size +20M -delete
tail -3 gpurun_out/r1k/bench.json | cut -c1-400
