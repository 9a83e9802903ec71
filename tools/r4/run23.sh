cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4z
timeout 600 python3 -m pytest -x -q -m gpu tests/test_gpu_kernels.py -k "applied_by or batch_norm_train or residual_join" 2>&1 | grep -E "^E  |passed|failed" | head -12
run() { name=$1; shift; timeout 300 "$@" 2> gpurun_out/r4z/$name.err | grep "^{" > gpurun_out/r4z/$name.json; python3 -c "import json;d=json.load(open('gpurun_out/r4z/$name.json'));print('$name', d['value'], d['ms_per_step'])"; }
for i in 1 2; do
run defer$i python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs
EMRT_BN_DEFER=0 run sep$i python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs
done
run c3 python3 bench.py --config cfg3 --steps 20 --warmup 5 --no-cpu-baseline
EMRT_BN_DEFER=0 run c3sep python3 bench.py --config cfg3 --steps 20 --warmup 5 --no-cpu-baseline
timeout 900 python3 -m pytest -x -q -m gpu tests/test_gpu_model.py tests/test_gpu_bench_shapes.py tests/test_gpu_dp2.py > gpurun_out/r4z/model.txt 2>&1; grep -E "passed|failed" gpurun_out/r4z/model.txt | tail -3
