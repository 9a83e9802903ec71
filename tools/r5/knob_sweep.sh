# End-of-round sweep of the dispatcher knobs around their defaults (bench.py cfg2, 40 steps); run on the GPU box through gpurun.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
run() {
  env $1 timeout 300 python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs > $O/sweep_bench.json 2> $O/sweep_bench.err
  python3 -c "import json,sys; d=json.loads(open('$O/sweep_bench.json').read().strip().splitlines()[-1]); print('%-44s %8.2f tiles/s  mean %.3f  median %.3f  max %.3f ms' % ('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_max']))"
}
for v in "A=default" "EMRT_WGROUP_BLOCKS=768" "EMRT_WGROUP_BLOCKS=1536" "EMRT_WGROUP_MIN_STEPS=24" "EMRT_WGROUP_MIN_STEPS=48" "EMRT_WGROUP_MAX=16" \
         "EMRT_BN_BLOCK_KB=4" "EMRT_BN_BLOCK_KB=16" "EMRT_PAIR_MAX=512" "EMRT_PAIR_MAX=1024" "EMRT_IGEMM8P_MIN_BLOCKS=128" "EMRT_IGEMM8P_MIN_BLOCKS=96" \
         "EMRT_WGRAD8P_MIN_STEPS=6" "EMRT_WGRAD8P_MIN_STEPS=12" "A=default2" "EMRT_MSDA_FWD_CHUNKS=3" "EMRT_MSDA_FWD_CHUNKS=5" "EMRT_GN_BWD_STAT_ROWS=16" "EMRT_GN_BWD_STAT_ROWS=64" \
         "EMRT_GN_APPLY_ROWS=4" "EMRT_GN_APPLY_ROWS=16" "EMRT_GN_STAT_ROWS=16" "EMRT_GN_STAT_ROWS=64" "EMRT_IGEMM64_NST=4" "EMRT_NO_KSPLIT128=1" "EMRT_THIN_BLOCKS=256" "EMRT_BN_OPERAND_BLOCKS=512" "A=default3"; do
  run "$v"
done
