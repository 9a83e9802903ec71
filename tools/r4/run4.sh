cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4e
python3 tools/bench_conv.py wgroup mixes > gpurun_out/r4e/wgroup_mixes.txt 2>&1; cat gpurun_out/r4e/wgroup_mixes.txt
python3 -m pytest -x -q -m gpu tests/test_gpu_wgrad_group.py 2>&1 | tail -3
for b in 0 1024; do
EMRT_WGROUP_BLOCKS=$b python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs 2> gpurun_out/r4e/b$b.err | grep "^{" > gpurun_out/r4e/b$b.json
python3 -c "import json;d=json.load(open('gpurun_out/r4e/b$b.json'));print('cfg2 wgroup_blocks $b', d['value'], d['ms_per_step'])"
EMRT_WGROUP_BLOCKS=$b python3 bench.py --config cfg3 --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/r4e/c3_b$b.err | grep "^{" > gpurun_out/r4e/c3_b$b.json
python3 -c "import json;d=json.load(open('gpurun_out/r4e/c3_b$b.json'));print('cfg3 wgroup_blocks $b', d['value'], d['ms_per_step'])"
done
for f in "" "--no-early-exchange"; do
python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs --two-phase $f 2> gpurun_out/r4e/tp$f.err | grep "^{" > gpurun_out/r4e/tp$f.json
python3 -c "import json;d=json.load(open('gpurun_out/r4e/tp$f.json'));print('two-phase $f', d['value'], d['ms_per_step'])"
done
