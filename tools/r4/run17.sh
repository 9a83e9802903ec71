cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4t
timeout 600 python3 -m pytest -x -q -m gpu tests/test_gpu_kernels.py -k "consumers_loads or adaptive" 2>&1 | tail -5
timeout 600 python3 -m pytest -x -q -m gpu tests/test_gpu_model.py -k "sliding or eval" > gpurun_out/r4t/model.txt 2>&1; grep -E "passed|failed" gpurun_out/r4t/model.txt | tail -3
run() { name=$1; shift; timeout 300 "$@" 2> gpurun_out/r4t/$name.err | grep "^{" > gpurun_out/r4t/$name.json; python3 -c "import json;d=json.load(open('gpurun_out/r4t/$name.json'));print('$name', d['value'], d['ms_per_step'])"; }
run a python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --dump-calls gpurun_out/r4t/calls.txt
grep -E "adaptive|resize|maxpool" gpurun_out/r4t/calls.txt | head
run c5 python3 bench.py --config cfg5 --steps 20 --warmup 5 --no-cpu-baseline
