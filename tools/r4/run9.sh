cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4k
python3 -m pytest -x -q -m gpu tests/test_gpu_kernels.py -k "bn or batch_norm or BatchNorm" 2>&1 | tail -3
python3 -m pytest -x -q -m gpu tests/test_gpu_bn_fold.py tests/test_gpu_model.py::test_train_forward_and_gradients_match_oracle 2>&1 | tail -3
run() { name=$1; shift; "$@" 2> gpurun_out/r4k/$name.err | grep "^{" > gpurun_out/r4k/$name.json; python3 -c "import json;d=json.load(open('gpurun_out/r4k/$name.json'));print('$name', d['value'], d['ms_per_step'], d['final_loss'])"; }
run main python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs
run main2 python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs
run c3 python3 bench.py --config cfg3 --steps 20 --warmup 5 --no-cpu-baseline
for kb in 4 16; do
EMRT_BN_BLOCK_KB=$kb run kb$kb python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs
done
