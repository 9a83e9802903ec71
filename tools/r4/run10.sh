cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4l
python3 -m pytest -x -q -s -m gpu tests/test_gpu_conv_fuzz.py -k stride2 2>&1 | grep "s2-\|passed\|failed\|Error" | cut -c1-200
python3 tools/bench_conv.py s2 2>&1 | tail -8
run() { name=$1; shift; "$@" 2> gpurun_out/r4l/$name.err | grep "^{" > gpurun_out/r4l/$name.json; python3 -c "import json;d=json.load(open('gpurun_out/r4l/$name.json'));print('$name', d['value'], d['ms_per_step'], d['final_loss'])"; }
run s2 python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs
EMRT_NO_S2_DGRAD=1 run generic python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs
run s2_c3 python3 bench.py --config cfg3 --steps 20 --warmup 5 --no-cpu-baseline
EMRT_NO_S2_DGRAD=1 run generic_c3 python3 bench.py --config cfg3 --steps 20 --warmup 5 --no-cpu-baseline
