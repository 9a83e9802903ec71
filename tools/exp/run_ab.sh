mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_conv_fuzz.py tests/test_gpu_bench_shapes.py tests/test_gpu_kernels.py -x -q -k "conv or wgrad" 2>&1 | tail -3
for i in 1 2; do
python bench.py --no-cpu-baseline --steps 60 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('cfg2', j['value'], j['ms_per_step'])"
python bench.py --config cfg5 --no-cpu-baseline --steps 60 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('cfg5', j['value'], j['ms_per_step'])"
done
python tools/bench_conv.py conv 2>&1 | grep -v amdgpu | cut -c1-120 | head -12
