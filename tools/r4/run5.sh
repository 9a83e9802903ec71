cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4f
python3 tools/bench_conv.py wgroup mixes > gpurun_out/r4f/wgroup_mixes.txt 2>&1; cat gpurun_out/r4f/wgroup_mixes.txt
for ms in 32 16 48; do
EMRT_WGROUP_MIN_STEPS=$ms python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs 2> gpurun_out/r4f/m$ms.err | grep "^{" > gpurun_out/r4f/m$ms.json
python3 -c "import json;d=json.load(open('gpurun_out/r4f/m$ms.json'));print('cfg2 min_steps $ms', d['value'], d['ms_per_step'])"
done
for ms in 32 4; do
EMRT_WGROUP_MIN_STEPS=$ms python3 bench.py --config cfg3 --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/r4f/c3_m$ms.err | grep "^{" > gpurun_out/r4f/c3_m$ms.json
python3 -c "import json;d=json.load(open('gpurun_out/r4f/c3_m$ms.json'));print('cfg3 min_steps $ms', d['value'], d['ms_per_step'])"
done
python3 -m pytest -x -q -m gpu tests/test_gpu_wgrad_group.py 2>&1 | tail -3
