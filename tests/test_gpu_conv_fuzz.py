"""-m gpu: seeded random convolution geometries through EVERY kernel variant of the conv dispatchers, against torch (fp32 CPU) on the
same rounded inputs.  The parametrised cases elsewhere are the model's own shapes and the edges thought of in advance; this sweep is for
the edges nobody thought of: odd map sizes, ragged tiles, channel counts on and off the vector path, stride / dilation / padding mixes.
Forward, data gradient, weight gradient and bias gradient (reference operators: nn.Conv2D, paddle_vision_resnet.py:108-123; resnet.py:102-160).
"""
import random

import pytest
import torch

pytestmark = pytest.mark.gpu

from emrt_amd import _lib                                   # noqa: E402
from tests.test_gpu_bench_shapes import _conv_case_vs_torch  # noqa: E402


def _cases(seed, n, big_channels):
    rng = random.Random(seed)
    out = []
    while len(out) < n:
        k = rng.choice([1, 1, 3, 3, 3, 5])
        stride = rng.choice([1, 1, 1, 2])
        dil = rng.choice([1, 1, 1, 2]) if k > 1 else 1
        pad = rng.choice([0, dil * (k // 2), dil * (k // 2)])
        H, W = rng.randint(5, 40), rng.randint(5, 40)
        if (H + 2 * pad - dil * (k - 1) - 1) < 0 or (W + 2 * pad - dil * (k - 1) - 1) < 0:
            continue
        if big_channels:        # on the vector path, sizes the 128x128 / 256x256 / split-K variants accept
            Cin, Cout = rng.choice([64, 128, 192, 256]), rng.choice([64, 96, 128, 256, 320])
        else:                   # anything: channel counts off the 8-element grid take the scalar loaders
            Cin, Cout = rng.choice([3, 8, 20, 24, 40, 64, 72]), rng.choice([6, 8, 30, 32, 48, 64, 100])
        N = rng.randint(1, 4)
        out.append(("fuzz%d-%d" % (seed, len(out)), N, H, W, Cin, Cout, k, stride, pad, rng.random() < 0.5, dil))
    return out


def _wgrad8p_cases(seed, n):
    """geometries the 256x256 weight-gradient kernel accepts: C, OC multiples of 256, OH * OW a multiple of 64"""
    rng = random.Random(seed)
    out = []
    while len(out) < n:
        k = rng.choice([1, 3, 3])
        dil = rng.choice([1, 1, 2]) if k > 1 else 1
        stride = rng.choice([1, 1, 2])
        OH, OW = rng.choice([(8, 8), (16, 8), (8, 16), (4, 16), (16, 16), (8, 24), (2, 32), (1, 64), (64, 3)])
        pad = dil * (k // 2)
        H = (OH - 1) * stride + dil * (k - 1) + 1 - 2 * pad + rng.randint(0, stride - 1)      # any H that gives OH
        W = (OW - 1) * stride + dil * (k - 1) + 1 - 2 * pad + rng.randint(0, stride - 1)
        if H < 1 or W < 1:
            continue
        out.append(("w8fuzz%d-%d" % (seed, len(out)), rng.randint(1, 5), H, W, rng.choice([256, 512]), rng.choice([256, 512]), k, stride, pad, rng.random() < 0.5, dil))
    return out


SMALL = _cases(101, 14, False)
BIG = _cases(202, 10, True)
WG8 = _wgrad8p_cases(303, 8)


@pytest.mark.parametrize("case", SMALL, ids=[c[0] for c in SMALL])
def test_conv_random_geometry_default_dispatch(case):
    _conv_case_vs_torch(case, dilation=case[10])


# conv_tile: 1 = 64x64, 3 = 128x128, 4 = 128x32 (thin OC), 5 / 6 = in-block K split by 2 / 4, 7 = 256x256 LDS-DMA 8-phase (C % 64 == 0)
@pytest.mark.parametrize("tile", [1, 3, 4, 5, 6, 7])
@pytest.mark.parametrize("case", BIG, ids=[c[0] for c in BIG])
def test_conv_random_geometry_forced_tiles(case, tile):
    L_ = _lib.lib()
    # the data gradient and the weight gradient as their own launches (through the forced tile / the 128x128 weight-gradient kernel);
    # wgrad8p_force: the 256x256 weight-gradient kernel wherever its shape conditions hold (C, OC % 256 == 0, OH * OW % 64 == 0)
    old = [(k, L_.set_tuning(k, v)) for k, v in (("conv_tile", tile), ("pair_max", 0), ("wgrad8p_force", 1))]
    try:
        _conv_case_vs_torch(case, dilation=case[10])
    finally:
        for k, v in old:
            L_.set_tuning(k, v)


@pytest.mark.parametrize("case", BIG[:6], ids=[c[0] for c in BIG[:6]])
def test_conv_random_geometry_pair_kernel(case):
    from emrt_amd.runtime import ctx
    L_ = _lib.lib()
    old = L_.set_tuning("pair_max", 1 << 30)          # dgrad + wgrad tiles in ONE launch whatever the grid size
    ctx().wgrad_batch = 0                             # (layer-by-layer backward: with batching on, the data gradient goes out alone)
    try:
        _conv_case_vs_torch(case, dilation=case[10])
    finally:
        L_.set_tuning("pair_max", old)
        ctx().wgrad_batch = 24


@pytest.mark.parametrize("slab", [1, 0], ids=["slab", "atomics"])
@pytest.mark.parametrize("case", WG8, ids=[c[0] for c in WG8])
def test_wgrad_8phase_random_geometry(case, slab):
    L_ = _lib.lib()
    old = [(k, L_.set_tuning(k, v)) for k, v in (("pair_max", 0), ("wgrad8p_force", 1), ("wgrad8p_slab", slab))]
    try:
        _conv_case_vs_torch(case, dilation=case[10])
    finally:
        for k, v in old:
            L_.set_tuning(k, v)
