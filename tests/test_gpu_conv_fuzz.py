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


# conv_tile: 1 = 64x64, 3 = 128x128, 4 = 128x32 (thin OC), 5 / 6 = in-block K split by 2 / 4, 7 = 256x256 LDS-DMA 8-phase (C % 64 == 0),
# 8 = 128x128 with K split over two wave groups
@pytest.mark.parametrize("tile", [1, 3, 4, 5, 6, 7, 8])
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


# stride-2 data gradients through the parity-class kernel (csrc/conv.hip: igemm_body S2): 1x1 / 3x3 / 5x5 taps, pad 0 / k//2, non-square
# maps, several 64-channel k-tiles, two N tiles -- against torch, and bit for bit against the generic kernel (the skipped k-tiles are exact zeros)
S2_CASES = [("s2-ds-1x1", 8, 32, 32, 512, 1024, 1, 2, 0, False, 1), ("s2-3x3", 2, 16, 16, 64, 64, 3, 2, 1, False, 1),
            ("s2-3x3-nonsquare", 1, 32, 16, 128, 192, 3, 2, 1, False, 1), ("s2-1x1-small", 4, 8, 8, 256, 64, 1, 2, 0, False, 1),
            ("s2-5x5", 2, 16, 32, 64, 128, 5, 2, 2, False, 1), ("s2-2x2-nopad", 2, 16, 16, 64, 128, 2, 2, 0, False, 1),
            ("s2-layer2-conv2", 8, 64, 64, 128, 128, 3, 2, 1, False, 1)]


@pytest.mark.parametrize("case", S2_CASES, ids=[c[0] for c in S2_CASES])
def test_stride2_data_gradient_parity_class_kernel(case):
    import math
    import torch.nn.functional as F
    from emrt_amd import nn as hnn
    from emrt_amd.runtime import BF16, Tape
    from tests.hip_utils import Holder, dev_map, host_map, init, rnd
    name, N, H, W, Cin, Cout, k, stride, pad = case[:9]
    L_ = _lib.lib()
    c = init(BF16)
    g = torch.Generator().manual_seed(3)
    x = rnd(torch.randn(N, Cin, H, W, generator=g))
    conv = hnn.Conv2D(Cin, Cout, k, stride, pad, bias=False)
    with torch.no_grad():
        conv.weight.copy_(rnd(torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)))
    wref = conv.weight.detach().clone()
    Holder(conv=conv).place()
    xr = x.clone().requires_grad_(True)
    yr = F.conv2d(xr, wref, None, stride=stride, padding=pad)
    dy = rnd(torch.randn(yr.shape, generator=g))
    yr.backward(dy)
    got = []
    for knob in (-1, 1):
        # knob -1: the parity-class kernel for every eligible shape (the dispatcher leaves 1x1 kernels on the generic path: no gain there);
        # knob 1: the generic 64x64 tile without k split (conv_tile = 1) -- the same k order per accumulator as the parity-class kernel, which
        # only leaves out k-tiles that are exact zeros; the dispatcher's own choice for a shape may be a k-split variant (another fp32 order)
        old = L_.set_tuning("no_s2_dgrad", knob)
        oldt = L_.set_tuning("conv_tile", 1 if knob == 1 else 0)
        try:
            xd = dev_map(x)
            tape = Tape()
            c.tape = tape
            y = conv(xd)
            c.tape = None
            tape.watch(xd)
            tape.add_grad(y, dev_map(dy))
            L_.start_record()
            tape.backward()
            rec = L_.stop_record()
            torch.cuda.synchronize()
            got.append(host_map(tape.result(xd)))
        finally:
            L_.set_tuning("no_s2_dgrad", old)
            L_.set_tuning("conv_tile", oldt)
    rel = ((got[0] - xr.grad).norm() / xr.grad.norm()).item()
    print("%s: dx vs torch rel %.2e; parity-class kernel == generic kernel: %s" % (name, rel, torch.equal(got[0], got[1])))
    assert rel < 4e-3, rel
    assert torch.equal(got[0], got[1]), (got[0] - got[1]).abs().max().item()


# cross-block K split (csrc/conv.hip: igemm_body XK): S copies of the 64x64 tile grid, partial tiles through the registered scratch, the last
# block to arrive sums them and runs the epilogue.  Forced on every fuzz geometry (ragged M / OC / K tails, stride, dilation, bias), forward
# and data gradient (pair_max 0, batched weight gradients: the data gradient is its own launch), S = 2 / 3 / 8 (3: copies with unequal k ranges;
# 8 on the short-K cases: the dispatcher cuts the copies down to ceil(k-tiles / ceil(k-tiles / 8)) so that each owns at least one k-tile -- round 5
# launched trailing copies without any, which only arrived with zeros)
@pytest.mark.parametrize("copies", [2, 3, 8])
@pytest.mark.parametrize("case", BIG, ids=[c[0] for c in BIG])
def test_conv_random_geometry_cross_block_k_split(case, copies):
    L_ = _lib.lib()
    old = [(k, L_.set_tuning(k, v)) for k, v in (("xk", copies), ("pair_max", 0))]
    try:
        from tests.hip_utils import init
        from emrt_amd.runtime import BF16
        init(BF16)
        L_.start_record()
        _conv_case_vs_torch(case, dilation=case[10])
        names = [n for n, _ in L_.stop_record()]
        assert "emrt_conv2d" in names
    finally:
        for k, v in old:
            L_.set_tuning(k, v)


def test_cross_block_k_split_is_bit_reproducible_under_uneven_load():
    """The hand-off (sc1 partial stores -> every wave drains -> ticket -> the last arriver's agent-scope acquire -> partial loads) must not
    depend on dispatch order, timing or placement: the same layer is launched 300 times while OTHER kernels of varying size run before it
    (the consuming CUs' L1 then holds lines of the scratch from the previous launch: a missing acquire reads stale partials), with the
    BatchNorm-statistics epilogue on; every launch's output must be bit-identical to the first (its fp64 sums equal to 1e-12), and equal the unsplit kernel's
    result to fp32 summation-order noise.  Layer4's 3x3 (8 x 8 x 8 x 512 -> 512: 64 tiles, 72 k-tiles) and the auxiliary head's 16 x 16 x 1024 3x3."""
    import ctypes
    from tests.hip_utils import init
    from emrt_amd.runtime import BF16
    c = init(BF16)
    L_ = _lib.lib()
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    g = torch.Generator().manual_seed(5)
    for (N, H, W, C, OC, k, S) in ((8, 8, 8, 512, 512, 3, 8), (8, 16, 16, 1024, 256, 3, 4), (3, 7, 9, 256, 320, 3, 5)):
        x = torch.randn(N, H, W, C, generator=g).cuda().bfloat16()
        w = (torch.randn(OC, k, k, C, generator=g) / (k * k * C) ** 0.5).cuda().bfloat16()
        y = torch.empty(N, H, W, OC, device="cuda", dtype=torch.bfloat16)
        stats = torch.zeros(8 * 2 * OC, device="cuda", dtype=torch.float64)
        junk = [torch.randn(n, device="cuda") for n in (1 << 12, 1 << 18, 1 << 22, 1 << 24)]

        def run():
            stats.zero_()
            L_.call("emrt_conv2d", P(x), P(w), P(y), None, None, N, H, W, C, C, H * W * C, H, W, OC, OC, H * W * OC, 0, 0,
                    k, k, 1, k // 2, 0, 1, 0, P(stats), None, 0, 0, 1, None, BF16, c.stream)
        old = L_.set_tuning("xk", -1)
        run()
        torch.cuda.synchronize()
        ref, ref_stats = y.float().clone(), stats.clone()
        L_.set_tuning("xk", S)
        try:
            L_.start_record()
            run()
            assert [n for n, _ in L_.stop_record()] == ["emrt_conv2d"]
            torch.cuda.synchronize()
            first, first_stats = y.clone(), stats.clone()
            rel = ((first.float() - ref).norm() / ref.norm()).item()
            assert rel < 2e-3, rel           # bf16 outputs of two fp32 summation orders: the occasional last-bit flip
            assert ((first_stats - ref_stats).abs().max() / ref_stats.abs().max()).item() < 1e-3
            bad = 0
            for i in range(300):
                junk[i % 4].mul_(1.0001)                       # another kernel of a different size in front: uneven load, warm L1s
                if i % 3 == 0:
                    junk[(i + 1) % 4].add_(1e-6)
                run()
                if i % 10 == 9 or i < 20:
                    torch.cuda.synchronize()
                    # (the statistics are fp64 atomics of several tiles' column sums into 8 replicas: their ORDER is free, as in every conv kernel)
                    bad += int(not torch.equal(y, first)) + int(not torch.allclose(stats, first_stats, rtol=1e-12, atol=1e-9))
            assert bad == 0, "%d of the checked launches differed from the first (N%d %dx%dx%d->%d, S=%d)" % (bad, N, H, W, C, OC, S)
        finally:
            L_.set_tuning("xk", old)
