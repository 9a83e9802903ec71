cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4p
timeout 200 python3 -m pytest -x -q -m gpu tests/test_gpu_kernels.py -k adaptive 2>&1 | tail -4
run() { name=$1; shift; timeout 300 "$@" 2> gpurun_out/r4p/$name.err | grep "^{" > gpurun_out/r4p/$name.json; python3 -c "import json;d=json.load(open('gpurun_out/r4p/$name.json'));print('$name', d['value'], d['ms_per_step'])"; }
run a python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --dump-calls gpurun_out/r4p/calls.txt
grep "adaptive" gpurun_out/r4p/calls.txt
run c3 python3 bench.py --config cfg3 --steps 20 --warmup 5 --no-cpu-baseline --dump-calls gpurun_out/r4p/calls_c3.txt
grep "adaptive" gpurun_out/r4p/calls_c3.txt
