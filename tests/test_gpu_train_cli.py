"""-m gpu: the reference's train.py / val.py surface end to end on the HIP path (train.py:141-229, val.py:66-209):
periodic evaluation, iter_N_model_state.pdparams rotation, best_model.pdparams, an unusable SAVE_DIR, and the `.pdparams`
importer feeding the GPU path."""
import os
import pickle

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CFG_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "emrt_amd/configs/EMRT")


def _yaml(tmp_path, save_dir, iters=8):
    """Potsdam recipe shrunk to a ResNet-18 on 64x64 tiles (BASE inheritance as the reference's yamls use it)."""
    path = str(tmp_path / "tiny.yaml")
    with open(path, "w") as f:
        f.write('BASE: ["%s"]\n' % os.path.relpath(os.path.join(CFG_DIR, "EMRT_256x256_160k_potsdam.yaml"), str(tmp_path)))
        f.write('DATA: {CROP_SIZE: "(64, 64)", BATCH_SIZE: 2}\n')
        f.write('MODEL: {ENCODER: {TYPE: "resnet18"}}\n')
        f.write("TRAIN: {ITERS: %d}\n" % iters)
        f.write("VAL: {IMAGE_BASE_SIZE: 64, CROP_SIZE: [64, 64], STRIDE_SIZE: [64, 64]}\n")
        f.write("SAVE_FREQ_CHECKPOINT: 4\nLOGGING_INFO_FREQ: 2\nKEEP_CHECKPOINT_MAX: 1\n")
        f.write('SAVE_DIR: "%s"\n' % save_dir)
    return path


def test_train_cli_evaluates_checkpoints_and_keeps_the_best_model(tmp_path, capsys):
    from emrt_amd import train, val
    save = str(tmp_path / "out")
    cfg = _yaml(tmp_path, save)
    model = train.main(["--config", cfg, "--dtype", "fp32", "--val_tiles", "4"])
    out = capsys.readouterr().out
    assert "[TRAIN] Epochs:" in out and "ips:" in out                       # the reference's log line (train.py:177-181)
    assert out.count("In this val: mIoU") == 2 and "Current best_mIoU" in out   # evaluated at iterations 4 and 8 (:187-195)
    assert "The model with the best validation mIoU" in out
    files = sorted(os.listdir(save))
    # KEEP_CHECKPOINT_MAX = 1: only the newest checkpoint pair survives, both files of the older one are gone (:210-213)
    assert files == ["best_model.pdparams", "iter_8_model_state.pdparams", "iter_8_state.pt"], files
    with open(os.path.join(save, "best_model.pdparams"), "rb") as f:
        raw = pickle.load(f)
    assert raw["model.encoder.layers.0.linear1.weight"].shape == (256, 1024)   # Paddle layout on disk: Linear [in, out]
    # the final weights round-trip through the Paddle-format file into a fresh model ON THE GPU PATH and reproduce the logits
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 3, 64, 64, generator=g).cuda()
    model.eval()
    want = model(x)[0].clone()
    from emrt_amd.config import get_config, update_config
    from emrt_amd.src.models import get_model
    from emrt_amd.src.utils.checkpoint import load_entire_model
    from emrt_amd.runtime import F32
    import argparse
    fresh = get_model(update_config(get_config(), argparse.Namespace(cfg=cfg)))
    n = load_entire_model(fresh, os.path.join(save, "iter_8_model_state.pdparams"))
    assert n == len(fresh.state_dict())
    fresh.to_hip("cuda:0", F32)
    fresh.eval()
    got = fresh(x)[0]
    assert torch.equal(got, want), (got - want).abs().max().item()
    # val.py's CLI on the Paddle-format file (val.py:76-81 load_entire_model)
    miou = val.main(["--config", cfg, "--model_path", os.path.join(save, "best_model.pdparams")])
    assert 0.0 <= miou <= 1.0
    assert "[EVAL] Images:" in capsys.readouterr().out


def test_unusable_save_dir_is_reported_and_replaced(tmp_path, capsys, monkeypatch):
    from emrt_amd import train
    blocker = tmp_path / "file"
    blocker.write_text("x")                                  # a path UNDER a regular file can never be created
    cfg = _yaml(tmp_path, str(blocker / "sub" / "run7"), iters=4)
    monkeypatch.chdir(tmp_path)
    train.main(["--config", cfg, "--dtype", "fp32", "--val_tiles", "2"])
    out = capsys.readouterr().out
    assert "[WARNING] SAVE_DIR" in out and "checkpoints go to" in out
    assert os.path.exists(os.path.join(str(tmp_path), "output", "run7", "iter_4_model_state.pdparams"))
