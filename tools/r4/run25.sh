cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4ab
timeout 600 python3 -m pytest -x -q -m gpu tests/test_gpu_kernels.py -k "applied_by or maxpool or resize" 2>&1 | grep -E "^E  |passed|failed" | head -12
run() { name=$1; shift; timeout 300 "$@" 2> gpurun_out/r4ab/$name.err | grep "^{" > gpurun_out/r4ab/$name.json; python3 -c "import json;d=json.load(open('gpurun_out/r4ab/$name.json'));print('$name', d['value'], d['ms_per_step'])"; }
for i in 1 2; do
run fold$i python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --dump-calls gpurun_out/r4ab/calls_fold.txt
EMRT_BN_FOLD_BWD=0 run nofold$i python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --dump-calls gpurun_out/r4ab/calls_nofold.txt
done
grep -E "emrt_bn_maxpool_bwd|emrt_bn_resize_bilinear_bwd" gpurun_out/r4ab/calls_fold.txt | cut -c1-60
grep -E "emrt_maxpool_bwd|emrt_resize_bilinear_bwd .* 256 (32|64) " gpurun_out/r4ab/calls_nofold.txt | cut -c1-90
timeout 900 python3 -m pytest -x -q -m gpu tests/test_gpu_model.py -k "train or trajectory or gradients" > gpurun_out/r4ab/model.txt 2>&1; grep -E "passed|failed" gpurun_out/r4ab/model.txt | tail -3
