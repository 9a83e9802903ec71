cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
timeout 600 python3 -m pytest tests/test_gpu_kernels.py -q -m gpu -k "sgd or momentum or optim" -p no:xdist -x > $O/run16_sgd.txt 2>&1; grep -E "passed|failed|^E  " $O/run16_sgd.txt | tail -4
for v in "EMRT_MSDA_SCATTER_MFMA=0" "EMRT_MSDA_MF_BANDS=512" "EMRT_MSDA_SCATTER_MFMA=0 EMRT_SGD_NT=0" "EMRT_MSDA_SCATTER_MFMA=0 A=again"; do
  env $v timeout 300 python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs > $O/run16_bench.json 2> $O/run16_bench.err
  python3 -c "import json,sys; d=json.loads(open('$O/run16_bench.json').read().strip().splitlines()[-1]); print('[$v]', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_max'], d['roofline']['frac'], d['final_loss'])"
  grep -E "sgd_momentum|msda_bwd" $O/run16_bench.err | head -3
done
