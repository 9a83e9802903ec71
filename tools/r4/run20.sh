cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4w
run() { name=$1; shift; timeout 300 "$@" 2> gpurun_out/r4w/$name.err | grep "^{" > gpurun_out/r4w/$name.json; python3 -c "import json;d=json.load(open('gpurun_out/r4w/$name.json'));print('$name', d['value'], d['ms_per_step'])"; }
for b in 512 1024 2048 4096; do
EMRT_BN_OPERAND_BLOCKS=$b run b$b python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs --dump-calls gpurun_out/r4w/calls_$b.txt
grep -E "emrt_bn_resize|emrt_bn_maxpool|pointwise_fwd" gpurun_out/r4w/calls_$b.txt | cut -c1-50
done
