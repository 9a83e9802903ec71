"""Data pipeline, CPU: readers over a synthetic directory tree in the reference's layout, transforms."""
import argparse
import os
import random

import numpy as np
import torch
from PIL import Image

from emrt_amd.config import get_config, update_config
from emrt_amd.distributed import DistributedTileSampler
from emrt_amd.src import transforms as T
from emrt_amd.src.datasets import get_dataset, TileLoader

CFG_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "emrt_amd/configs/EMRT")


def _make_potsdam(root, n=6, size=80):
    rng = np.random.RandomState(0)
    for sub in ("train", "test"):
        os.makedirs(os.path.join(root, sub)); os.makedirs(os.path.join(root, sub + "_convert_labels"))
        for i in range(n):
            Image.fromarray(rng.randint(0, 256, (size, size, 3), dtype=np.uint8)).save(os.path.join(root, sub, "%d.tif" % (10 * i + 3)))
            Image.fromarray(rng.randint(0, 6, (size, size), dtype=np.uint8)).save(os.path.join(root, sub + "_convert_labels", "%d.png" % (10 * i + 3)))


def test_resize_matches_half_pixel_bilinear_and_floor_nearest():
    g = np.random.RandomState(1)
    img = g.rand(13, 17, 3).astype(np.float32) * 255
    out = T.resize_bilinear(img, 29, 21)
    ref = torch.nn.functional.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None], size=(21, 29), mode="bilinear", align_corners=False)
    assert np.abs(out - ref[0].permute(1, 2, 0).numpy()).max() < 1e-3
    lab = g.randint(0, 6, (13, 17)).astype(np.uint8)
    nn_ = T.resize_nearest(lab, 34, 26)
    assert nn_.shape == (26, 34) and (nn_[::2, ::2] == lab).all()      # exact 2x: every source pixel is hit by floor(dst/2)


def test_potsdam_reader_and_train_transforms(tmp_path):
    root = str(tmp_path / "potsdam")
    _make_potsdam(root)
    cfg = update_config(get_config(), argparse.Namespace(cfg=os.path.join(CFG_DIR, "EMRT_256x256_160k_potsdam.yaml")))
    cfg.DATA.DATA_PATH = root
    cfg.DATA.CROP_SIZE = [64, 64]
    ds = get_dataset(cfg, T.get_transforms(cfg), "train")
    assert len(ds) == 6 and [os.path.basename(p[0]) for p in ds.file_list][:3] == ["3.tif", "13.tif", "23.tif"]      # numeric sort
    np.random.seed(5); random.seed(5)
    img, lab = ds[2]
    assert img.shape == (3, 64, 64) and img.dtype == np.float32 and lab.shape == (64, 64) and lab.dtype == np.uint8
    assert set(np.unique(lab)) <= set(range(6)) | {255}
    np.random.seed(5); random.seed(5)
    img2, lab2 = ds[2]
    assert np.array_equal(img, img2) and np.array_equal(lab, lab2)            # same seeds -> same augmentation decisions
    # normalisation constants: a zero-padded pixel maps to -mean/std
    tr = T.Compose([T.RandomPaddingCrop((96, 96), (0, 0, 0), 255), T.Normalize(T._MEAN, T._STD)])
    x, y = tr(np.full((80, 80, 3), 255.0, np.float32), np.zeros((80, 80), np.uint8))
    assert x.shape == (3, 96, 96) and (y == 255).sum() == 96 * 96 - 80 * 80
    pad = x[:, y == 255]
    assert np.allclose(pad[:, 0], [-123.675 / 58.395, -116.28 / 57.12, -103.53 / 57.375], atol=1e-5)
    val = get_dataset(cfg, T.get_val_transforms(cfg), "val")
    vi, vl = val[0]
    base = cfg.VAL.IMAGE_BASE_SIZE          # val images are resized to VAL.IMAGE_BASE_SIZE (train.py:89-91), labels are not
    assert vi.shape == (3, base, base) and vl.shape == (1, 80, 80)


def test_loveda_label_shift(tmp_path):
    root = str(tmp_path / "loveda")
    rng = np.random.RandomState(2)
    for sub in ("Train", "Val"):
        os.makedirs(os.path.join(root, sub, "images_png")); os.makedirs(os.path.join(root, sub, "masks_png"))
        for i in range(3):
            Image.fromarray(rng.randint(0, 256, (32, 32, 3), dtype=np.uint8)).save(os.path.join(root, sub, "images_png", "%d.png" % i))
            Image.fromarray(rng.randint(0, 8, (32, 32), dtype=np.uint8)).save(os.path.join(root, sub, "masks_png", "%d.png" % i))
    cfg = update_config(get_config(), argparse.Namespace(cfg=os.path.join(CFG_DIR, "EMRT_256x256_160k_loveda.yaml")))
    cfg.DATA.DATA_PATH = root
    ds = get_dataset(cfg, T.get_transforms(cfg), "train")
    raw = np.asarray(Image.open(ds.file_list[1][1]))
    _, lab = ds[1]
    assert ((raw == 0) == (lab == 255)).all() and (lab[raw > 0] == raw[raw > 0] - 1).all()      # 0 = ignore -> 255, classes 1..7 -> 0..6


def test_tile_loader_batches(tmp_path):
    root = str(tmp_path / "potsdam")
    _make_potsdam(root, n=8, size=48)
    cfg = update_config(get_config(), argparse.Namespace(cfg=os.path.join(CFG_DIR, "EMRT_256x256_160k_potsdam.yaml")))
    cfg.DATA.DATA_PATH = root
    cfg.DATA.CROP_SIZE = [32, 32]
    ds = get_dataset(cfg, T.get_transforms(cfg), "train")
    sampler = DistributedTileSampler(len(ds), 4, 0, 1, shuffle=True, drop_last=True, seed=1)
    it = TileLoader(ds, sampler, torch.device("cpu"), workers=2, prefetch=2).epochs()
    for _ in range(5):                       # crosses an epoch boundary (2 batches per epoch)
        x, y = next(it)
        assert x.shape == (4, 3, 32, 32) and x.dtype == torch.float32 and y.shape == (4, 32, 32) and y.dtype == torch.int64
    it.close()


def test_tile_loader_never_holds_more_than_prefetch_ready_batches(tmp_path):
    """A slow consumer must not let the reader threads pile up finished batches (they become device memory in training)."""
    import time
    root = str(tmp_path / "potsdam")
    _make_potsdam(root, n=8, size=48)
    cfg = update_config(get_config(), argparse.Namespace(cfg=os.path.join(CFG_DIR, "EMRT_256x256_160k_potsdam.yaml")))
    cfg.DATA.DATA_PATH = root
    cfg.DATA.CROP_SIZE = [32, 32]
    ds = get_dataset(cfg, T.get_transforms(cfg), "train")
    sampler = DistributedTileSampler(len(ds), 2, 0, 1, shuffle=False, drop_last=True, seed=1)
    loader = TileLoader(ds, sampler, torch.device("cpu"), workers=3, prefetch=2)
    built = []
    orig = loader._batch
    loader._batch = lambda idx: (built.append(1), orig(idx))[1]
    it = loader.epochs()
    next(it)
    time.sleep(1.0)                          # consumer stalls: the workers may only fill the free slots
    assert len(built) <= 1 + 2, "%d batches were built while at most prefetch = 2 may wait" % len(built)
    seen = [next(it) for _ in range(6)]      # and the stream keeps flowing in order afterwards
    assert all(x.shape == (2, 3, 32, 32) for x, _ in seen)
    it.close()
