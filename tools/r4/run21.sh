cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4x
timeout 900 python3 -m pytest -x -q -m gpu tests/test_gpu_conv_fuzz.py -k "forced_tiles" 2>&1 | tail -3
run() { name=$1; shift; timeout 300 "$@" 2> gpurun_out/r4x/$name.err | grep "^{" > gpurun_out/r4x/$name.json; python3 -c "import json;d=json.load(open('gpurun_out/r4x/$name.json'));print('$name', d['value'], d['ms_per_step'])"; }
for i in 1 2; do
run ks$i python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs
EMRT_NO_KSPLIT128=1 run no$i python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs
done
run c3 python3 bench.py --config cfg3 --steps 20 --warmup 5 --no-cpu-baseline
EMRT_NO_KSPLIT128=1 run c3no python3 bench.py --config cfg3 --steps 20 --warmup 5 --no-cpu-baseline
run c5 python3 bench.py --config cfg5 --steps 20 --warmup 5 --no-cpu-baseline
EMRT_NO_KSPLIT128=1 run c5no python3 bench.py --config cfg5 --steps 20 --warmup 5 --no-cpu-baseline
