# Round profile (run on the GPU box through gpurun): rocprofv3 kernel stats of the default bench command, the two PMC passes the
# microarch guide prescribes for HBM traffic (FETCH_SIZE / WRITE_SIZE in separate runs, --kernel-trace only), per config.
TAG=${1:-r2}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 2 --warmup 0 --no-graph --no-cpu-baseline > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 2 --warmup 0 --no-graph --no-cpu-baseline > /dev/null 2> $OUT/pmc_write.err
python3 tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write $OUT/${TAG}_pmc_traffic > /dev/null
for cfg in cfg3 cfg5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$cfg -- python3 bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_${cfg}_under_rocprof.json 2> $OUT/stats_$cfg.err
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$cfg -- python3 bench.py --config $cfg --steps 2 --warmup 0 --no-graph --no-cpu-baseline > /dev/null 2> $OUT/pmc_fetch_$cfg.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$cfg -- python3 bench.py --config $cfg --steps 2 --warmup 0 --no-graph --no-cpu-baseline > /dev/null 2> $OUT/pmc_write_$cfg.err
  python3 tools/pmc_summary.py $OUT/pmc_fetch_$cfg $OUT/pmc_write_$cfg $OUT/${TAG}_pmc_traffic_$cfg > /dev/null
done
# keep the merge small: only the summaries travel back
mkdir -p $OUT/keep
for d in stats stats_cfg3 stats_cfg5; do f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/keep/${TAG}_$(echo $d | sed 's/stats_\?//; s/^$/cfg2/')_kernel_stats.csv; done
cp $OUT/${TAG}_pmc_traffic*.json $OUT/${TAG}_pmc_traffic*.csv $OUT/keep/ 2>/dev/null
cp $OUT/bench_*under_rocprof.json $OUT/keep/
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
ls -la $OUT/keep
