cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4h
run() { name=$1; shift; "$@" 2> gpurun_out/r4h/$name.err | grep "^{" > gpurun_out/r4h/$name.json; python3 -c "import json;d=json.load(open('gpurun_out/r4h/$name.json'));print('$name', d['value'], d['ms_per_step'])"; }
run single python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs
run tp python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --two-phase
run tp_nosyncbn python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --two-phase --two-phase-no-syncbn
run tp_noearly python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --two-phase --no-early-exchange
EMRT_GRAD_EXCHANGE=bf16 run tp_bf16 python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --two-phase
run tp_nograph python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --two-phase --no-graph
run single_nograph python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-graph
