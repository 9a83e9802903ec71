# full GPU validation + per-kernel stats of the default bench (developer tool, GPU box)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/full_tests.log
for i in 1 2; do
python bench.py --no-cpu-baseline --steps 60 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('cfg2', j['value'], j['ms_per_step'])"
python bench.py --config cfg3 --no-cpu-baseline --steps 40 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('cfg3', j['value'], j['ms_per_step'])"
python bench.py --config cfg5 --no-cpu-baseline --steps 60 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('cfg5', j['value'], j['ms_per_step'])"
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/st_x -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
cp $(find gpurun_out/st_x -name "*kernel_stats.csv" | head -1) gpurun_out/st_x_cfg2_kernel_stats.csv; rm -rf gpurun_out/st_x
