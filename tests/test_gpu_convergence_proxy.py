"""-m gpu: a convergence PROXY for the `metric`'s accuracy half ("Potsdam mIoU within +-0.2 of the reference").  The Potsdam tiles, the
ImageNet ResNet-50 weights and the reference's trained weights do not exist offline, so the claim itself cannot be tested.  What can be:
that the bf16 HIP path -- the dtype the throughput number is quoted in -- TRAINS like the fp32 HIP path (the reference's precision) through
the reference's own loop and recipe (train.py:141-195: SGD momentum 0.9, lr 0.01 poly 0.9, weight decay 1e-4, clip 1.0, CE + 0.4 aux CE,
batch 8, crop 256, the transform chain of transforms/__init__.py:25-56; evaluation as val.py:124-209) on a seeded LEARNABLE tile set in
the Potsdam directory layout (tools/make_fake_potsdam.py --learnable: the label is a function of the image).  Both runs start from the same
initial weights and see the same batches; the validation mIoU of the two must agree and be far above chance."""
import importlib.util
import os
import re

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG_DIR = os.path.join(ROOT, "emrt_amd/configs/EMRT")


def _make_set(root, n_train, n_val):
    spec = importlib.util.spec_from_file_location("make_fake_potsdam", os.path.join(ROOT, "tools/make_fake_potsdam.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.make(str(root), learnable=True, n_train=n_train, n_val=n_val, size=256, seed=7)


def test_bf16_training_reaches_the_fp32_miou_on_a_learnable_tile_set(tmp_path, capsys):
    from emrt_amd import train
    iters = int(os.environ.get("EMRT_PROXY_ITERS", "1200"))
    data = _make_set(tmp_path / "potsdam_like", 96, 24)
    cfg = str(tmp_path / "proxy.yaml")
    with open(cfg, "w") as f:
        f.write('BASE: ["%s"]\n' % os.path.relpath(os.path.join(CFG_DIR, "EMRT_256x256_160k_potsdam.yaml"), str(tmp_path)))
        f.write('DATA: {DATA_PATH: "%s", NUM_WORKERS: 8}\n' % data)
        f.write("TRAIN: {ITERS: %d}\n" % iters)
        f.write("SAVE_FREQ_CHECKPOINT: %d\nLOGGING_INFO_FREQ: 200\n" % iters)        # one evaluation, at the end (train.py:187-195)
    miou, loss = {}, {}
    for dtype in ("bf16", "fp32"):
        train.main(["--config", cfg, "--data", "dataset", "--dtype", dtype, "--iters", str(iters), "--seed", "1234", "--save_dir", str(tmp_path / ("out_" + dtype))])
        out = capsys.readouterr().out
        m = re.findall(r"In this val: mIoU ([\d.]+),\s+Acc: ([\d.]+)", out)
        ls = re.findall(r"iter: (\d+)/\d+, loss: ([\d.]+)", out)
        assert m and ls, out[-2000:]
        miou[dtype], loss[dtype] = float(m[-1][0]), [(int(a), float(b)) for a, b in ls]
        print("PROXY %s: %d iterations, validation mIoU %.4f Acc %s; loss %s" % (dtype, iters, miou[dtype], m[-1][1], ["%d:%.3f" % t for t in loss[dtype]]))
    gap = abs(miou["bf16"] - miou["fp32"])
    print("PROXY mIoU: bf16 %.4f  fp32 %.4f  |difference| %.2f points (chance on 6 classes: ~0.09)" % (miou["bf16"], miou["fp32"], 100 * gap))
    assert loss["bf16"][-1][1] < 0.5 * loss["bf16"][0][1] and loss["fp32"][-1][1] < 0.5 * loss["fp32"][0][1], "the loss did not come down"
    assert miou["bf16"] > 0.5 and miou["fp32"] > 0.5, miou            # well above chance: the set was learned, by both
    assert gap <= 0.01, "bf16 and fp32 training end %.2f mIoU points apart" % (100 * gap)
