cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4s
timeout 600 python3 -m pytest -x -q -m gpu tests/test_gpu_kernels.py -k "consumers_loads or batch_norm or bn_relu or residual_join or maxpool or resize" 2>&1 | tail -15
run() { name=$1; shift; timeout 300 "$@" 2> gpurun_out/r4s/$name.err | grep "^{" > gpurun_out/r4s/$name.json; python3 -c "import json;d=json.load(open('gpurun_out/r4s/$name.json'));print('$name', d['value'], d['ms_per_step'])"; }
for i in 1 2; do
run defer$i python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs
EMRT_BN_DEFER=0 run sep$i python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs
done
timeout 600 python3 -m pytest -x -q -m gpu tests/test_gpu_model.py > gpurun_out/r4s/model.txt 2>&1; grep -E "passed|failed" gpurun_out/r4s/model.txt | tail -3
