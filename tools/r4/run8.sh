cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4j
run() { name=$1; shift; "$@" 2> gpurun_out/r4j/$name.err | grep "^{" > gpurun_out/r4j/$name.json; python3 -c "import json;d=json.load(open('gpurun_out/r4j/$name.json'));print('$name', d['value'], d['ms_per_step'], d['final_loss'])"; }
run main python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs
EMRT_WGRAD_SIDE=1 run side python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs
EMRT_WGRAD_SIDE=1 EMRT_WGRAD_BATCH=12 run side12 python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs
EMRT_WGRAD_SIDE=1 EMRT_WGRAD_BATCH=48 run side48 python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs
EMRT_WGRAD_SIDE=1 run side_c3 python3 bench.py --config cfg3 --steps 20 --warmup 5 --no-cpu-baseline
run main_c3 python3 bench.py --config cfg3 --steps 20 --warmup 5 --no-cpu-baseline
