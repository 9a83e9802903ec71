cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4y
run() { name=$1; shift; timeout 300 "$@" 2> gpurun_out/r4y/$name.err | grep "^{" > gpurun_out/r4y/$name.json; python3 -c "import json;d=json.load(open('gpurun_out/r4y/$name.json'));print('$name', d['value'], d['ms_per_step'])"; grep -E "emrt_layernorm_bwd" gpurun_out/r4y/$name.err; }
run base python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs
EMRT_LN_BWD_ROWS=16 EMRT_LN_BWD_MAX_BLOCKS=1024 run r16 python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs
EMRT_LN_BWD_ROWS=16 EMRT_LN_BWD_MAX_BLOCKS=512 run r16c python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs
EMRT_LN_BWD_ROWS=64 run r64 python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs
EMRT_LN_ATOMIC=0 run noatom python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs
