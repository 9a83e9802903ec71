# which of the round's bf16 changes breaks training?  short runs of the convergence proxy's recipe under one knob each
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
python3 - <<'PY'
import importlib.util, os
spec = importlib.util.spec_from_file_location("m", "tools/make_fake_potsdam.py"); mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
d = mod.make("/tmp/proxyset", learnable=True, n_train=96, n_val=24, size=256, seed=7)
open("/tmp/proxy.yaml", "w").write('BASE: ["%s"]\nDATA: {DATA_PATH: "%s", NUM_WORKERS: 8}\nTRAIN: {ITERS: 400}\nSAVE_FREQ_CHECKPOINT: 400\nLOGGING_INFO_FREQ: 100\n' % (os.path.abspath("emrt_amd/configs/EMRT/EMRT_256x256_160k_potsdam.yaml"), d))
PY
for v in "A=default" "EMRT_MHA_VALU=1" "EMRT_FFN_DROPOUT_FUSED=0" "EMRT_XK=-1" "EMRT_MHA_VALU=1 EMRT_FFN_DROPOUT_FUSED=0 EMRT_XK=-1"; do
  env $v timeout 600 python3 -m emrt_amd.train --config /tmp/proxy.yaml --data dataset --dtype bf16 --iters 400 --seed 1234 --save_dir /tmp/out_bisect > $O/bisect.log 2>&1
  echo "[$v] $(grep -oE 'iter: [0-9]+/[0-9]+, loss: [0-9.]+' $O/bisect.log | tr '\n' ' ') $(grep -oE 'In this val: mIoU [0-9.]+' $O/bisect.log | tail -1)"
done
