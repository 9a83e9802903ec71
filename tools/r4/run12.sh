cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4n
python3 -m pytest -x -q -m gpu tests/test_gpu_kernels.py -k "adaptive" 2>&1 | tail -3
python3 -m pytest -x -q -m gpu tests/test_gpu_model.py -k "eval or full_size_train" 2>&1 | tail -3
run() { name=$1; shift; "$@" 2> gpurun_out/r4n/$name.err | grep "^{" > gpurun_out/r4n/$name.json; python3 -c "import json;d=json.load(open('gpurun_out/r4n/$name.json'));print('$name', d['value'], d['ms_per_step'])"; }
run a python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs
run c3 python3 bench.py --config cfg3 --steps 20 --warmup 5 --no-cpu-baseline
run c5 python3 bench.py --config cfg5 --steps 20 --warmup 5 --no-cpu-baseline
