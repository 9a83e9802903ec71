# full GPU validation: tests, smoke, the three benchmark configurations (developer tool, GPU box)
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/full_tests.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/full_smoke.log 2>&1
timeout 600 python bench.py > gpurun_out/full_bench_cfg2.json 2> gpurun_out/full_bench_cfg2.err
timeout 600 python bench.py --config cfg3 --no-cpu-baseline > gpurun_out/full_bench_cfg3.json 2> gpurun_out/full_bench_cfg3.err
timeout 600 python bench.py --config cfg5 --no-cpu-baseline > gpurun_out/full_bench_cfg5.json 2> gpurun_out/full_bench_cfg5.err
