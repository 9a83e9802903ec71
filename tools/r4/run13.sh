cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4o
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4o/p -- python3 tools/exp/pool_probe.py > /dev/null 2> gpurun_out/r4o/p.err
f=$(find gpurun_out/r4o/p -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && grep -i "pool\|fill" "$f" | cut -c1-200
find gpurun_out/r4o -name "*.csv" -delete
