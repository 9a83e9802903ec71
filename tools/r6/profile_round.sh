# Round profile (run on the GPU box through gpurun): rocprofv3 kernel stats of the default bench command (hipGraph replay), then per config the
# PMC passes the microarch guide prescribes for HBM traffic (FETCH_SIZE / WRITE_SIZE in separate runs, --kernel-trace only) plus -- for cfg2 --
# an SQ pass (MFMA busy) and a plain kernel-trace pass of the same eager command, joined per layer by tools/pmc_summary.py.
TAG=${1:-r6a}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
mkdir -p $OUT $OUT/keep
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $OUT/keep/${TAG}_bench_under_rocprof.json 2> $OUT/stats.err
EAGER="--steps 2 --warmup 0 --no-graph --no-cpu-baseline --no-other-configs"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $EAGER > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $EAGER > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 bench.py $EAGER > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_eager -- python3 bench.py $EAGER --dump-calls $OUT/calls_eager.txt > /dev/null 2> $OUT/trace_eager.err
python3 tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write $OUT/keep/${TAG}_pmc_traffic $OUT/pmc_sq $OUT/trace_eager $OUT/calls_eager.txt > $OUT/pmc_summary.log 2>&1
tail -5 $OUT/pmc_summary.log
for cfg in cfg3 cfg5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$cfg -- python3 bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline > $OUT/keep/${TAG}_bench_${cfg}_under_rocprof.json 2> $OUT/stats_$cfg.err
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$cfg -- python3 bench.py --config $cfg $EAGER --dump-calls $OUT/calls_$cfg.txt > /dev/null 2> $OUT/pmc_fetch_$cfg.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$cfg -- python3 bench.py --config $cfg $EAGER > /dev/null 2> $OUT/pmc_write_$cfg.err
  python3 tools/pmc_summary.py $OUT/pmc_fetch_$cfg $OUT/pmc_write_$cfg $OUT/keep/${TAG}_pmc_traffic_$cfg --calls $OUT/calls_$cfg.txt > /dev/null 2>> $OUT/pmc_summary.log
done
# per-dispatch timeline of ONE captured replay of the headline config (tools/r5/trace_summary.py picks it from the middle of the timed region)
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_graph -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-other-configs > /dev/null 2> $OUT/trace_graph.err
f=$(find $OUT/trace_graph -name "*kernel_trace.csv" | head -1)
python3 tools/r5/trace_summary.py $f $OUT/keep/${TAG}_timeline_cfg2.txt > $OUT/keep/${TAG}_timeline_summary_cfg2.txt 2>> $OUT/pmc_summary.log
head -c 4000000 $OUT/keep/${TAG}_timeline_cfg2.txt > /dev/null
# keep the merge small: only the summaries travel back
for d in stats stats_cfg3 stats_cfg5; do f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/keep/${TAG}_$(echo $d | sed 's/stats_\?//; s/^$/cfg2/')_kernel_stats.csv; done
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
ls -la $OUT/keep
