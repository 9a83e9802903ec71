cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
# background load: two other processes hammering the same GPU (what the xdist suite does to a test)
(for i in 1 2 3 4 5 6 7 8; do python3 bench.py --steps 400 --warmup 5 --no-cpu-baseline --no-other-configs > /dev/null 2>&1; done) &
BG1=$!
(for i in 1 2 3 4 5 6 7 8; do python3 bench.py --config cfg5 --steps 800 --warmup 5 --no-cpu-baseline > /dev/null 2>&1; done) &
BG2=$!
sleep 20
for v in "A=default" "EMRT_XK=-1" "EMRT_MHA_VALU=1" "EMRT_WGRAD8P_SLAB=0" "EMRT_XK=-1 EMRT_MHA_VALU=1 EMRT_WGRAD8P_SLAB=0 EMRT_GROUP_ATTN_PROJ=0" "A=default_again"; do
  echo "== $v"; env $v timeout 400 python3 tools/r5/bisect_split.py single single 2>&1 | grep -E "^(eager|single|split) "
done
kill $BG1 $BG2 2>/dev/null
wait 2>/dev/null
