cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4r
timeout 400 python3 -m pytest -x -q -m gpu tests/test_gpu_wgrad_group.py 2>&1 | tail -5
run() { name=$1; shift; timeout 300 "$@" 2> gpurun_out/r4r/$name.err | grep "^{" > gpurun_out/r4r/$name.json; python3 -c "import json;d=json.load(open('gpurun_out/r4r/$name.json'));print('$name', d['value'], d['ms_per_step'])"; }
for i in 1 2; do
run ow$i python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs
EMRT_WGRAD_NO_OVERWRITE=1 run at$i python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs
done
run c3 python3 bench.py --config cfg3 --steps 20 --warmup 5 --no-cpu-baseline
EMRT_WGRAD_NO_OVERWRITE=1 run c3at python3 bench.py --config cfg3 --steps 20 --warmup 5 --no-cpu-baseline
timeout 300 python3 -m pytest -x -q -m gpu tests/test_gpu_model.py -k "trajectory or train" 2>&1 | tail -3
