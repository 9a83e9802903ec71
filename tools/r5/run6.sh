cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
for v in "A=default" "EMRT_XK=-1" "EMRT_WGRAD8P_SLAB=0" "EMRT_MHA_VALU=1" "EMRT_GROUP_ATTN_PROJ=0" "EMRT_WGRAD_NO_OVERWRITE=1"; do
  echo "== $v"; env $v timeout 300 python3 tools/r5/bisect_split.py 2>&1 | grep -E "^(eager|single|split) "
done
