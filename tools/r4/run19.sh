cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4v
timeout 600 python3 -m pytest -x -q -m gpu tests/test_gpu_kernels.py -k "classifiers_loads or consumers_loads or bn_relu_conv" -s 2>&1 | grep -E "^E  |passed|failed|relative errors" | head -30
run() { name=$1; shift; timeout 300 "$@" 2> gpurun_out/r4v/$name.err | grep "^{" > gpurun_out/r4v/$name.json; python3 -c "import json;d=json.load(open('gpurun_out/r4v/$name.json'));print('$name', d['value'], d['ms_per_step'])"; }
for i in 1 2; do
run defer$i python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --dump-calls gpurun_out/r4v/calls_defer.txt
EMRT_BN_DEFER=0 run sep$i python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --dump-calls gpurun_out/r4v/calls_sep.txt
done
grep -E "pointwise" gpurun_out/r4v/calls_defer.txt | cut -c1-60
grep -E " 8 128 128 256 256 1048576|emrt_conv2d_bwd .* 6 " gpurun_out/r4v/calls_sep.txt | cut -c1-80 | head
