cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(for i in $(seq 1 14); do python3 bench.py --steps 400 --warmup 5 --no-cpu-baseline --no-other-configs > /dev/null 2>&1; done) &
(for i in $(seq 1 14); do python3 bench.py --config cfg5 --steps 800 --warmup 5 --no-cpu-baseline > /dev/null 2>&1; done) &
sleep 20
for v in "A=sep1" "A=sep2" "A=sep3" "EMRT_WGRAD8P_SLAB=0" "EMRT_WGRAD8P_SLAB=0" "EMRT_XK=-1" "EMRT_XK=-1" "EMRT_MHA_VALU=1" "EMRT_MHA_VALU=1"; do
  echo "== $v"; env $v timeout 400 python3 tools/r5/bisect_split.py single single 2>&1 | grep -E "^(eager|single|split) " | awk '{print $1, $2, $3, $6, $10, $14, $18, $22, $25, $26, $27}'
done
