cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4aa
timeout 600 python3 -m pytest -x -q -m gpu tests/test_gpu_kernels.py -k "applied_by or batch_norm_train or residual_join or bn_relu" 2>&1 | grep -E "^E  |passed|failed" | head -12
run() { name=$1; shift; timeout 300 "$@" 2> gpurun_out/r4aa/$name.err | grep "^{" > gpurun_out/r4aa/$name.json; python3 -c "import json;d=json.load(open('gpurun_out/r4aa/$name.json'));print('$name', d['value'], d['ms_per_step'])"; grep -E "emrt_bn_bwd_dx|emrt_bn_apply |emrt_bn_bwd_reduce" gpurun_out/r4aa/$name.err; }
for i in 1 2; do
run defer$i python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs
EMRT_BN_DEFER=0 run sep$i python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs
done
