cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4m
python3 -m pytest -x -q -m gpu tests/test_gpu_conv_fuzz.py tests/test_gpu_bench_shapes.py -k "forced_tiles or conv_bench_shape or ksplit or stride2" 2>&1 | tail -3
run() { name=$1; shift; "$@" 2> gpurun_out/r4m/$name.err | grep "^{" > gpurun_out/r4m/$name.json; python3 -c "import json;d=json.load(open('gpurun_out/r4m/$name.json'));print('$name', d['value'], d['ms_per_step'], d['final_loss'])"; }
run a python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs
run b python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs
run c5 python3 bench.py --config cfg5 --steps 20 --warmup 5 --no-cpu-baseline
