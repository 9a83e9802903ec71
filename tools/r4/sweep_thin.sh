cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4thin
run() { name=$1; shift; env "$@" timeout 300 python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-other-configs --dump-calls gpurun_out/r4thin/$name.txt 2> /dev/null | grep "^{" > gpurun_out/r4thin/$name.json; python3 -c "import json;d=json.load(open('gpurun_out/r4thin/$name.json'));print('%-28s %8.2f tiles/s' % ('$name', d['value']))"; grep -E "pointwise" gpurun_out/r4thin/$name.txt | cut -c1-50; }
run base X=0
run ch4 EMRT_THIN_CH=4
run cblk32 EMRT_THIN_CBLK=32
run cblk128 EMRT_THIN_CBLK=128
run cblk256 EMRT_THIN_CBLK=256
run ch4_cblk128 EMRT_THIN_CH=4 EMRT_THIN_CBLK=128
run blocks512 EMRT_THIN_BLOCKS=512
