"""Synthetic tile sets in the reference's Potsdam directory layout (src/datasets/potsdam.py:50-66):
    <root>/train/<n>.tif + <root>/train_convert_labels/<n>.png ; <root>/test/<n>.tif + <root>/test_convert_labels/<n>.png

    python tools/make_fake_potsdam.py [root] [--learnable] [--train 64] [--val 16] [--size 256]

default: uniform noise images and labels (plumbing tests: nothing to learn).
--learnable: the label is a deterministic function of the image, so a segmentation network can be TRAINED on it and a validation mIoU means
something (the convergence proxy, tests/test_gpu_convergence_proxy.py): every tile is a set of random rectangles and discs of the six
classes; a class has its own mean colour, except that classes 4 and 5 share one and differ only in texture (a 4-pixel checkerboard against
flat), so colour alone does not solve the task; per-pixel Gaussian noise and a smooth illumination ramp on top; 1.5 % of the label pixels
are set to 255 (ignore) as in real annotations.  No dataset or pretrained weights exist offline: this is a PROXY for "bf16 trains like fp32",
not a Potsdam result."""
import os
import sys

import numpy as np
from PIL import Image

COLOURS = np.array([[200, 60, 60], [60, 180, 70], [60, 80, 200], [210, 200, 70], [128, 128, 128], [128, 128, 128]], dtype=np.float32)


def learnable_tile(rng, size, ncls=6):
    lab = np.full((size, size), rng.randint(0, ncls), dtype=np.uint8)
    yy, xx = np.mgrid[0:size, 0:size]
    for _ in range(rng.randint(6, 14)):
        c = rng.randint(0, ncls)
        if rng.rand() < 0.5:
            y0, x0 = rng.randint(0, size - 8, 2)
            h, w = rng.randint(size // 10, size // 2, 2)
            lab[y0:y0 + h, x0:x0 + w] = c
        else:
            cy, cx, r = rng.randint(0, size), rng.randint(0, size), rng.randint(size // 12, size // 4)
            lab[(yy - cy) ** 2 + (xx - cx) ** 2 < r * r] = c
    img = COLOURS[lab]
    checker = (((yy // 4) + (xx // 4)) % 2).astype(np.float32) * 2.0 - 1.0
    img = img + (lab == 5)[..., None] * checker[..., None] * 45.0
    ramp = (xx * rng.uniform(-0.15, 0.15) + yy * rng.uniform(-0.15, 0.15))[..., None]
    img = img + ramp + rng.normal(0.0, 18.0, img.shape)
    img = np.clip(img, 0, 255).astype(np.uint8)
    lab = lab.copy()
    lab[rng.rand(size, size) < 0.015] = 255
    return img, lab


def make(root, learnable=False, n_train=24, n_val=4, size=256, seed=0):
    rng = np.random.RandomState(seed)
    for sub, n in (("train", n_train), ("test", n_val)):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
        os.makedirs(os.path.join(root, sub + "_convert_labels"), exist_ok=True)
        for i in range(n):
            if learnable:
                img, lab = learnable_tile(rng, size)
            else:
                img, lab = rng.randint(0, 256, (size, size, 3), dtype=np.uint8), rng.randint(0, 6, (size, size), dtype=np.uint8)
            Image.fromarray(img).save(os.path.join(root, sub, "%d.tif" % i))
            Image.fromarray(lab).save(os.path.join(root, sub + "_convert_labels", "%d.png" % i))
    return root


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("root", nargs="?", default="gpurun_out/fake_potsdam")
    ap.add_argument("--learnable", action="store_true")
    ap.add_argument("--train", type=int, default=24)
    ap.add_argument("--val", type=int, default=4)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    make(a.root, a.learnable, a.train, a.val, a.size, a.seed)
