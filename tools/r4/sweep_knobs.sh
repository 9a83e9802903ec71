# One-box sweep of dispatcher knobs around their defaults on the default workload (cfg2): prints tiles/s per setting.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4knobs
run() { name=$1; shift; env "$@" timeout 300 python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-other-configs 2> /dev/null | grep "^{" > gpurun_out/r4knobs/$name.json; python3 -c "import json;d=json.load(open('gpurun_out/r4knobs/$name.json'));print('%-40s %8.2f tiles/s  %.3f ms' % ('$name', d['value'], d['ms_per_step']))"; }
run base0 X=0
run wgroup_blocks_768 EMRT_WGROUP_BLOCKS=768
run wgroup_blocks_1536 EMRT_WGROUP_BLOCKS=1536
run wgroup_min_steps_16 EMRT_WGROUP_MIN_STEPS=16
run wgroup_min_steps_48 EMRT_WGROUP_MIN_STEPS=48
run wgrad_batch_16 EMRT_WGRAD_BATCH=16
run wgrad_batch_32 EMRT_WGRAD_BATCH=32
run base1 X=0
run bn_block_kb_4 EMRT_BN_BLOCK_KB=4
run bn_block_kb_16 EMRT_BN_BLOCK_KB=16
run igemm8p_min_blocks_128 EMRT_IGEMM8P_MIN_BLOCKS=128
run igemm8p_min_blocks_200 EMRT_IGEMM8P_MIN_BLOCKS=200
run wgrad8p_min_steps_6 EMRT_WGRAD8P_MIN_STEPS=6
run wgrad8p_min_steps_12 EMRT_WGRAD8P_MIN_STEPS=12
run thin_blocks_256 EMRT_THIN_BLOCKS=256
run thin_blocks_64 EMRT_THIN_BLOCKS=64
run gn_stat_rows_16 EMRT_GN_STAT_ROWS=16
run msda_fwd_threads_512 EMRT_MSDA_FWD_THREADS=512
run base2 X=0
