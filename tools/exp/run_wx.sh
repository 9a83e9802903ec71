mkdir -p gpurun_out
python tools/bench_conv.py wgrad > gpurun_out/wx_on.log 2>&1
EMRT_WGRAD8P_XCD=0 python tools/bench_conv.py wgrad > gpurun_out/wx_off.log 2>&1
python bench.py --config cfg3 --no-cpu-baseline --steps 30 2>/dev/null | cut -c1-200 > gpurun_out/wx_cfg3_on.json
EMRT_WGRAD8P_XCD=0 python bench.py --config cfg3 --no-cpu-baseline --steps 30 2>/dev/null | cut -c1-200 > gpurun_out/wx_cfg3_off.json
python bench.py --no-cpu-baseline --steps 30 2>/dev/null | cut -c1-200 > gpurun_out/wx_cfg2_on.json
EMRT_WGRAD8P_XCD=0 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | cut -c1-200 > gpurun_out/wx_cfg2_off.json
timeout 900 python -m pytest tests/test_gpu_bench_shapes.py tests/test_gpu_kernels.py -x -q -k "wgrad or conv" 2>&1 | tail -3 > gpurun_out/wx_tests.log
