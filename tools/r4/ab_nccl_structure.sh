cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4i
run() { name=$1; shift; "$@" 2> gpurun_out/r4i/$name.err | grep "^{" > gpurun_out/r4i/$name.json; python3 -c "import json;d=json.load(open('gpurun_out/r4i/$name.json'));print('$name', d['value'], d['ms_per_step'])"; }
run single python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs
EMRT_EXCHANGE_NOOP=1 run tp_noop python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --two-phase
EMRT_EXCHANGE_NOOP=1 run tp_noop_nosyncbn python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --two-phase --two-phase-no-syncbn
EMRT_EXCHANGE_NOOP=1 run tp_noop_noearly python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --two-phase --no-early-exchange
run tp python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --two-phase
