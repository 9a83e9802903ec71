cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4pyr
timeout 600 python3 -m pytest -x -q -m gpu tests/test_gpu_kernels.py -k "pyramid or resize" 2>&1 | grep -E "^E  |passed|failed" | head -12
run() { name=$1; shift; timeout 300 "$@" 2> gpurun_out/r4pyr/$name.err | grep "^{" > gpurun_out/r4pyr/$name.json; python3 -c "import json;d=json.load(open('gpurun_out/r4pyr/$name.json'));print('$name', d['value'], d['ms_per_step'])"; }
for i in 1 2; do
run grouped$i python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --dump-calls gpurun_out/r4pyr/calls_g.txt
EMRT_PYRAMID_GROUP=0 run perscale$i python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs
done
grep pyramid gpurun_out/r4pyr/calls_g.txt | cut -c1-60
run c3 python3 bench.py --config cfg3 --steps 20 --warmup 5 --no-cpu-baseline
EMRT_PYRAMID_GROUP=0 run c3per python3 bench.py --config cfg3 --steps 20 --warmup 5 --no-cpu-baseline
