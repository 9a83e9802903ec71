"""Sweep the implicit-GEMM tile choice / wgrad split over the layer shapes of the EMRT step (developer tool, GPU).

usage: python tools/bench_conv.py            # prints one line per shape: time per tile config
"""
import ctypes
import os
import sys

sys.path.insert(0, ".")
import torch

from emrt_amd import _lib

L = _lib.lib()
dev = torch.device("cuda:0")
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None

SHAPES = [  # N, H, W, C, OC, k, stride, pad
    (8, 32, 32, 256, 256, 3, 1, 1),
    (8, 16, 16, 256, 256, 3, 1, 1),
    (8, 8, 8, 256, 256, 3, 1, 1),
    (8, 1, 1344, 256, 1024, 1, 1, 0),
    (8, 1, 1344, 1024, 256, 1, 1, 0),
    (8, 1, 1344, 256, 256, 1, 1, 0),
    (8, 1, 1344, 256, 432, 1, 1, 0),
    (8, 64, 64, 64, 256, 1, 1, 0),
    (8, 64, 64, 256, 64, 1, 1, 0),
    (8, 64, 64, 64, 64, 3, 1, 1),
    (8, 32, 32, 128, 128, 3, 1, 1),
    (8, 32, 32, 512, 128, 1, 1, 0),
    (8, 32, 32, 128, 512, 1, 1, 0),
    (8, 16, 16, 1024, 256, 1, 1, 0),
    (8, 16, 16, 256, 1024, 1, 1, 0),
    (8, 8, 8, 512, 512, 3, 1, 1),
    (8, 8, 8, 2048, 512, 1, 1, 0),
    (8, 8, 8, 512, 2048, 1, 1, 0),
    (8, 64, 64, 256, 256, 3, 1, 1),
    (8, 128, 128, 64, 64, 3, 1, 1),
    (8, 128, 128, 256, 256, 3, 1, 1),
    (8, 32, 32, 1536, 512, 3, 1, 1),
    (8, 32, 32, 512, 256, 3, 1, 1),
    (8, 16, 16, 1024, 256, 3, 1, 1),
    (8, 16, 16, 512, 512, 3, 1, 1),
]


def timed(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def thin():
    """The thin backward kernel (OC <= 8, 1x1) on the classifier shape: block count / channels per thread sweep."""
    N, H, W, C, OC = 8, 128, 128, 256, 6
    x = torch.randn(N, H, W, C, device=dev).bfloat16()
    dy = torch.randn(N, H, W, OC, device=dev).bfloat16()
    wb = (torch.randn(C, OC, device=dev) * 0.05).bfloat16()
    dx = torch.empty_like(x)
    dw = torch.zeros(OC, C, device=dev, dtype=torch.float32)
    db = torch.zeros(OC, device=dev, dtype=torch.float32)
    stats = torch.zeros(8 * 2 * C, device=dev, dtype=torch.float64)

    def run(mask):
        L._raw_emrt_conv2d_bwd(P(x), P(dy), P(wb), P(dx), C, H * W * C, 0, P(dw), P(db), N, H, W, C, C, H * W * C, H, W, OC, OC, H * W * OC,
                               1, 1, 1, 0, P(stats) if mask else None, P(x) if mask else None, C if mask else 0, H * W * C if mask else 0, 1.0, None, 0, 0, None, 0, 0, 1, 1, stream)
    for ch in (8, 4):
        for cblk in (64, 128, 256):
            for blocks in (64, 128, 256, 512):
                L.set_tuning("thin_ch", ch), L.set_tuning("thin_blocks", blocks), L.set_tuning("thin_cblk", cblk)
                print("CH %d cblk %3d pixel chunks %4d: masked+stats %.1f us  plain %.1f us" % (ch, cblk, blocks, timed(lambda: run(True)), timed(lambda: run(False))), flush=True)


BIG = [  # the layers that carry the FLOPs, at the sizes of BASELINE configs[1] (batch 8, 256x256), [2] (batch 4, 512x512), [4] (16 windows)
    ("cfg2 uphead conv_2", 8, 128, 128, 256, 256, 3), ("cfg2 uphead conv_1", 8, 64, 64, 256, 256, 3), ("cfg2 uphead conv_0", 8, 32, 32, 256, 256, 3),
    ("cfg2 cls_psp.0", 8, 32, 32, 1536, 512, 3), ("cfg2 cls_psp.3", 8, 32, 32, 512, 256, 3),
    ("cfg3 uphead conv_2", 4, 256, 256, 256, 256, 3), ("cfg3 uphead conv_1", 4, 128, 128, 256, 256, 3), ("cfg3 uphead conv_0", 4, 64, 64, 256, 256, 3),
    ("cfg3 cls_psp.0", 4, 64, 64, 1536, 512, 3), ("cfg3 cls_psp.3", 4, 64, 64, 512, 256, 3),
    ("cfg5 uphead conv_2", 16, 128, 128, 256, 256, 3), ("cfg5 cls_psp.0", 16, 32, 32, 1536, 512, 3),
]


def big():
    """128 x 128 register-staged tile (conv_tile 3) against the 256 x 256 LDS-DMA 8-phase kernel (conv_tile 7): time and agreement."""
    for (name, N, H, W, C, OC, k) in BIG:
        pad = k // 2
        x = torch.randn(N, H, W, C, device=dev).bfloat16()
        wf = (torch.randn(OC, k, k, C, device=dev) / (k * k * C) ** 0.5).bfloat16()
        wb = (torch.randn(C, k, k, OC, device=dev) / (k * k * OC) ** 0.5).bfloat16()
        dy = torch.randn(N, H, W, OC, device=dev).bfloat16()
        y = torch.empty(N, H, W, OC, device=dev, dtype=torch.bfloat16)
        dx = torch.empty_like(x)
        stats = torch.zeros(8 * 2 * max(C, OC), device=dev, dtype=torch.float64)
        gf = 2.0 * N * H * W * OC * k * k * C / 1e9

        def fwd():
            L._raw_emrt_conv2d(P(x), P(wf), P(y), None, None, N, H, W, C, C, H * W * C, H, W, OC, OC, H * W * OC, 0, 0,
                               k, k, 1, pad, 0, 0, 0, P(stats), None, 0, 0, 1, None, 1, stream)

        def dgrad():
            L._raw_emrt_conv2d(P(dy), P(wb), P(dx), None, None, N, H, W, OC, OC, H * W * OC, H, W, C, C, H * W * C, 0, 0,
                               k, k, 1, pad, 1, 0, 0, None, None, 0, 0, 1, None, 1, stream)
        line = "%-20s N%d %dx%dx%d->%d k%d %7.2f GF |" % (name, N, H, W, C, OC, k, gf)
        for nm, fn, out in (("fwd", fwd, y), ("dgrad", dgrad, dx)):
            res, outs = [], []
            for tile in (0, 3, 7):
                L.set_tuning("conv_tile", tile)
                res.append(min(timed(fn, 20), timed(fn, 20)))
                outs.append(out.float().clone())
            L.set_tuning("conv_tile", 0)
            rel = ((outs[2] - outs[1]).norm() / outs[1].norm()).item()
            line += " %s auto %.1f us (%.0f TF/s)  128x128 %.1f  8-phase %.1f us (%.0f TF/s)  rel diff %.1e |" % (
                nm, res[0], gf / res[0] * 1e3, res[1], res[2], gf / res[2] * 1e3, rel)
        print(line, flush=True)


def wbig():
    """Weight gradient of the big layers: 128 x 128 register-staged tile with fp32 atomics against the 256 x 256 LDS-DMA 8-phase kernel
    (partial tiles through the scratch slabs + reduce launch, or fp32 atomics), and the 8-phase kernel's slice count."""
    scratch = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    L.call("emrt_set_scratch", P(scratch), ctypes.c_size_t(scratch.numel()), stream)
    extra = [("cfg2 layer3 3x3", 8, 16, 16, 256, 256, 3), ("cfg2 layer4 3x3", 8, 8, 8, 512, 512, 3), ("cfg3 layer3 3x3", 4, 32, 32, 256, 256, 3),
             ("cfg2 ffn linear1", 8, 1, 1344, 256, 1024, 1), ("cfg2 ffn linear2", 8, 1, 1344, 1024, 256, 1), ("cfg2 attn proj", 8, 1, 1344, 256, 256, 1),
             ("cfg3 ffn linear1", 4, 1, 5376, 256, 1024, 1), ("cfg3 ffn linear2", 4, 1, 5376, 1024, 256, 1), ("cfg3 attn proj", 4, 1, 5376, 256, 256, 1)]
    only = sys.argv[2] if len(sys.argv) > 2 else ""
    for (name, N, H, W, C, OC, k) in [c for c in BIG + extra if only in c[0]]:
        pad = k // 2
        x = torch.randn(N, H, W, C, device=dev).bfloat16()
        dy = torch.randn(N, H, W, OC, device=dev).bfloat16()
        dw = torch.zeros(OC, k, k, C, device=dev, dtype=torch.float32)
        db = torch.zeros(OC, device=dev, dtype=torch.float32)
        gf = 2.0 * N * H * W * OC * k * k * C / 1e9

        def wgrad():
            L._raw_emrt_conv2d_wgrad(P(x), P(dy), P(dw), N, H, W, C, C, H * W * C, H, W, OC, OC, H * W * OC, k, k, 1, pad, P(db), 1, 1, stream)

        def one(**knobs):
            old = [(kk, L.set_tuning(kk, v)) for kk, v in knobs.items()]
            dw.zero_()
            wgrad()
            out = dw.clone()
            t = min(timed(wgrad, 20), timed(wgrad, 20))
            for kk, v in old:
                L.set_tuning(kk, v)
            return t, out
        t0, ref = one(wgrad8p_min_steps=0)
        line = "%-20s N%d %dx%dx%d->%d k%d %7.2f GF | 128x128 %.1f us (%.0f TF/s) |" % (name, N, H, W, C, OC, k, gf, t0, gf / t0 * 1e3)
        ta, oa = one()
        line += " auto %.1f us (%.0f TF/s) |" % (ta, gf / ta * 1e3)
        for label, knobs in (("8p slab", dict(wgrad8p_force=1)), ("8p launch-order", dict(wgrad8p_force=1, wgrad8p_xcd=0)), ("8p atomics", dict(wgrad8p_force=1, wgrad8p_slab=0)),
                             ("8p slab S/2", dict(wgrad8p_force=1, wgrad_split=max(1, 128 // (k * k * (C // 256) * (OC // 256))))),
                             ("8p 1 slice", dict(wgrad8p_force=1, wgrad_split=1))):
            t, o = one(**knobs)
            line += " %s %.1f us (%.0f TF/s) rel %.1e |" % (label, t, gf / t * 1e3, ((o - ref).norm() / ref.norm()).item())
        print(line, flush=True)


class _WD(ctypes.Structure):       # EmrtWgradDesc
    _fields_ = [("x", ctypes.c_void_p), ("dy", ctypes.c_void_p), ("dw", ctypes.c_void_p), ("dbias", ctypes.c_void_p),
                ("N", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int), ("C", ctypes.c_int), ("ldx", ctypes.c_int),
                ("x_bs", ctypes.c_longlong), ("OH", ctypes.c_int), ("OW", ctypes.c_int), ("OC", ctypes.c_int), ("lddy", ctypes.c_int),
                ("dy_bs", ctypes.c_longlong), ("KH", ctypes.c_int), ("KW", ctypes.c_int), ("stride", ctypes.c_int), ("pad", ctypes.c_int),
                ("dilation", ctypes.c_int), ("dw_is_zero", ctypes.c_int)]


def wgroup8():
    """The step's weight-gradient batches with the problems that fit the 256x256 LDS-DMA kernel grouped on it (wgrad8p_group_kernel, knob wgroup8) against
    all of them on the 128x128 group kernel; block-count sweep of the grouped 256x256 launch.  dW declared zero (as after zero_grad in the step): one-slice
    problems store their tiles."""
    scratch = torch.empty((64 << 20) + (8 << 20) + 65536, dtype=torch.uint8, device=dev)
    L.call("emrt_set_scratch", P(scratch), ctypes.c_size_t(scratch.numel()), stream)

    def problems(shapes):
        keep, arr = [], (_WD * len(shapes))()
        gf = 0.0
        for d, (N, H, W, C, OC, k, s, pad) in zip(arr, shapes):
            OH, OW = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
            x = torch.randn(N, H, W, C, device=dev).bfloat16()
            dy = torch.randn(N, OH, OW, OC, device=dev).bfloat16()
            dw = torch.zeros(OC, k, k, C, device=dev, dtype=torch.float32)
            keep += [x, dy, dw]
            d.x, d.dy, d.dw, d.dbias = x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None
            d.N, d.H, d.W, d.C, d.ldx, d.x_bs = N, H, W, C, C, H * W * C
            d.OH, d.OW, d.OC, d.lddy, d.dy_bs = OH, OW, OC, OC, OH * OW * OC
            d.KH, d.KW, d.stride, d.pad, d.dilation, d.dw_is_zero = k, k, s, pad, 1, 1
            gf += 2.0 * N * OH * OW * OC * k * k * C / 1e9
        return arr, keep, gf

    enc1 = [(8, 1, 1344, 256, 256, 1, 1, 0), (8, 1, 1344, 256, 432, 1, 1, 0), (8, 1, 1344, 256, 256, 1, 1, 0), (8, 1, 1344, 256, 1024, 1, 1, 0), (8, 1, 1344, 1024, 256, 1, 1, 0),
            (8, 32, 32, 256, 256, 3, 1, 1), (8, 16, 16, 256, 256, 3, 1, 1), (8, 8, 8, 256, 256, 3, 1, 1)]
    mixes = (("encoder layers x24", enc1 * 3),
             ("resnet layer3 x18+2", [(8, 16, 16, 1024, 256, 1, 1, 0), (8, 16, 16, 256, 256, 3, 1, 1), (8, 16, 16, 256, 1024, 1, 1, 0)] * 6 + [(8, 32, 32, 512, 1024, 1, 2, 0), (8, 32, 32, 256, 256, 3, 2, 1)]),
             ("resnet layer4 x9+2", [(8, 8, 8, 2048, 512, 1, 1, 0), (8, 8, 8, 512, 512, 3, 1, 1), (8, 8, 8, 512, 2048, 1, 1, 0)] * 3 + [(8, 16, 16, 1024, 2048, 1, 2, 0), (8, 16, 16, 512, 512, 3, 2, 1)]),
             ("decoder + heads", [(8, 1, 110, 256, 256, 1, 1, 0), (8, 1, 110, 256, 512, 1, 1, 0), (8, 1, 110, 256, 1024, 1, 1, 0), (8, 1, 110, 1024, 256, 1, 1, 0)] * 2 +
              [(8, 32, 32, 512, 256, 3, 1, 1), (8, 16, 16, 1024, 256, 3, 1, 1), (8, 32, 32, 256, 256, 3, 1, 1), (8, 16, 16, 256, 256, 3, 1, 1)]))
    for name, mix in mixes:
        arr, keep, gf = problems(mix)
        run = lambda: L._raw_emrt_conv2d_wgrad_group(arr, len(mix), 1, stream)
        old = L.set_tuning("wgroup8", 0)
        t0 = min(timed(run, 20), timed(run, 20))
        L.set_tuning("wgroup8", 1)
        t1 = min(timed(run, 20), timed(run, 20))
        line = "%-22s %6.1f GF | 128x128 group %6.1f us | + 256x256 group %6.1f us | blocks:" % (name, gf, t0, t1)
        for blocks in (128, 192, 256, 320, 384, 512, 768):
            ob = L.set_tuning("wgroup8_blocks", blocks)
            t = min(timed(run, 20), timed(run, 20))
            L.set_tuning("wgroup8_blocks", ob)
            line += " %4d: %6.1f |" % (blocks, t)
        ob = L.set_tuning("wgrad8p_slab", 0)
        t = min(timed(run, 20), timed(run, 20))
        L.set_tuning("wgrad8p_slab", ob)
        line += " atomics: %6.1f |" % t
        L.set_tuning("wgroup8", old)
        print(line, flush=True)


def wgroup():
    """Batched weight gradients (emrt_conv2d_wgrad_group) in the throughput regime: n copies of one layer shape in one call, against n single
    launches; and the step's real mixes (a ResNet stage, an encoder layer).  What the 128x128 weight-gradient body delivers when the machine is full."""
    def problems(shapes):
        keep, arr = [], (_WD * len(shapes))()
        gf = 0.0
        for d, (N, H, W, C, OC, k, s, pad) in zip(arr, shapes):
            OH, OW = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
            x = torch.randn(N, H, W, C, device=dev).bfloat16()
            dy = torch.randn(N, OH, OW, OC, device=dev).bfloat16()
            dw = torch.zeros(OC, k, k, C, device=dev, dtype=torch.float32)
            keep += [x, dy, dw]
            d.x, d.dy, d.dw, d.dbias = x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None
            d.N, d.H, d.W, d.C, d.ldx, d.x_bs = N, H, W, C, C, H * W * C
            d.OH, d.OW, d.OC, d.lddy, d.dy_bs = OH, OW, OC, OC, OH * OW * OC
            d.KH, d.KW, d.stride, d.pad, d.dilation = k, k, s, pad, 1
            gf += 2.0 * N * OH * OW * OC * k * k * C / 1e9
        return arr, keep, gf

    def run(arr, n):
        return lambda: L._raw_emrt_conv2d_wgrad_group(arr, n, 1, stream)
    one = [(8, 32, 32, 256, 256, 3, 1, 1), (8, 16, 16, 256, 256, 3, 1, 1), (8, 16, 16, 1024, 256, 1, 1, 0), (8, 16, 16, 256, 1024, 1, 1, 0), (8, 8, 8, 512, 512, 3, 1, 1),
           (8, 1, 1344, 256, 1024, 1, 1, 0), (8, 1, 1344, 1024, 256, 1, 1, 0), (8, 1, 1344, 256, 256, 1, 1, 0), (8, 64, 64, 64, 256, 1, 1, 0), (8, 64, 64, 256, 64, 1, 1, 0),
           (8, 64, 64, 64, 64, 3, 1, 1), (8, 32, 32, 128, 128, 3, 1, 1), (8, 32, 32, 512, 128, 1, 1, 0)]
    for shp in ([] if len(sys.argv) > 2 and sys.argv[2] == "mixes" else one):
        line = "N%d %dx%dx%d->%d k%d |" % shp[:6]
        for n in (1, 4, 12, 24):
            arr, keep, gf = problems([shp] * n)
            t = min(timed(run(arr, n), 20), timed(run(arr, n), 20))
            line += " x%-2d %7.1f us = %5.1f us each, %4.0f TF/s |" % (n, t, t / n, gf / t * 1e3)
        print(line, flush=True)
    layer3 = [(8, 16, 16, 1024, 256, 1, 1, 0), (8, 16, 16, 256, 256, 3, 1, 1), (8, 16, 16, 256, 1024, 1, 1, 0)] * 6
    layer1 = [(8, 64, 64, 256, 64, 1, 1, 0), (8, 64, 64, 64, 64, 3, 1, 1), (8, 64, 64, 64, 256, 1, 1, 0)] * 3
    enc = [(8, 1, 1344, 256, 256, 1, 1, 0), (8, 1, 1344, 256, 432, 1, 1, 0), (8, 1, 1344, 256, 256, 1, 1, 0), (8, 1, 1344, 256, 1024, 1, 1, 0), (8, 1, 1344, 1024, 256, 1, 1, 0),
           (8, 32, 32, 256, 256, 3, 1, 1), (8, 16, 16, 256, 256, 3, 1, 1), (8, 8, 8, 256, 256, 3, 1, 1)] * 3
    layer4 = [(8, 8, 8, 2048, 512, 1, 1, 0), (8, 8, 8, 512, 512, 3, 1, 1), (8, 8, 8, 512, 2048, 1, 1, 0)] * 3
    layer2 = [(8, 32, 32, 512, 128, 1, 1, 0), (8, 32, 32, 128, 128, 3, 1, 1), (8, 32, 32, 128, 512, 1, 1, 0)] * 4
    dec = [(8, 1, 110, 256, 256, 1, 1, 0), (8, 1, 110, 256, 512, 1, 1, 0), (8, 1, 110, 256, 1024, 1, 1, 0), (8, 1, 110, 1024, 256, 1, 1, 0), (8, 1, 110, 256, 432, 1, 1, 0), (8, 1, 1344, 256, 256, 1, 1, 0)] * 2
    for name, mix in (("resnet layer3 x18", layer3), ("resnet layer4 x9", layer4), ("resnet layer2 x12", layer2), ("resnet layer1 x9", layer1), ("encoder layers x24", enc), ("decoder x12", dec)):
        arr, keep, gf = problems(mix)
        t = min(timed(run(arr, len(mix)), 20), timed(run(arr, len(mix)), 20))
        line = "%-20s %6.1f GF | default plan %6.1f us | forced block counts:" % (name, gf, t)
        for blocks in (128, 256, 384, 512, 768, 1024, 2048):
            old, oldm = L.set_tuning("wgroup_blocks", blocks), L.set_tuning("wgroup_min_steps", 1)
            t = min(timed(run(arr, len(mix)), 20), timed(run(arr, len(mix)), 20))
            L.set_tuning("wgroup_blocks", old), L.set_tuning("wgroup_min_steps", oldm)
            line += " %4d: %6.1f us |" % (blocks, t)
        print(line, flush=True)


def s2():
    """The stride-2 data gradients of the ResNet-50 step: generic kernels (no_s2_dgrad = 1) against the parity-class kernel."""
    shapes = [(8, 64, 64, 256, 512, 1, 2, 0), (8, 64, 64, 128, 128, 3, 2, 1), (8, 32, 32, 512, 1024, 1, 2, 0), (8, 32, 32, 256, 256, 3, 2, 1),
              (8, 16, 16, 1024, 2048, 1, 2, 0), (8, 16, 16, 512, 512, 3, 2, 1)]
    tot = [0.0, 0.0]
    for (N, H, W, C, OC, k, s, pad) in shapes:
        OH, OW = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
        wb = (torch.randn(C, k, k, OC, device=dev) * 0.05).bfloat16()
        dy = torch.randn(N, OH, OW, OC, device=dev).bfloat16()
        dx = torch.empty(N, H, W, C, device=dev, dtype=torch.bfloat16)

        def dgrad():
            L._raw_emrt_conv2d(P(dy), P(wb), P(dx), None, None, N, OH, OW, OC, OC, OH * OW * OC, H, W, C, C, H * W * C, 0, 0,
                               k, k, s, pad, 1, 0, 0, None, None, 0, 0, 1, None, 1, stream)
        res = []
        for knob in (1, 0):
            old = L.set_tuning("no_s2_dgrad", knob)
            res.append(min(timed(dgrad), timed(dgrad)))
            L.set_tuning("no_s2_dgrad", old)
        tot[0] += res[0]
        tot[1] += res[1]
        print("N%d %dx%dx%d<-%d k%d s2 dgrad: generic %.1f us  parity-class %.1f us" % (N, H, W, C, OC, k, res[0], res[1]), flush=True)
    print("sum: generic %.1f us  parity-class %.1f us" % tuple(tot))


def xk():
    """Cross-block K split (igemm_body XK) on the step's few-tile, long-K layers: the old choice (xk = -1: in-block wave-group split) against S = 2 / 4 / 8
    copies of the tile grid, forward and data gradient; agreement with the old kernel and bit-reproducibility of the split over 20 launches."""
    scratch = torch.empty((64 << 20) + (8 << 20) + 65536, dtype=torch.uint8, device=dev)
    L.call("emrt_set_scratch", P(scratch), ctypes.c_size_t(scratch.numel()), stream)
    shapes = [(8, 16, 16, 256, 256, 3, 1, 1), (8, 8, 8, 512, 512, 3, 1, 1), (8, 8, 8, 2048, 512, 1, 1, 0), (8, 8, 8, 512, 2048, 1, 1, 0),
              (8, 16, 16, 1024, 256, 3, 1, 1), (8, 16, 16, 1024, 256, 1, 1, 0), (8, 16, 16, 256, 1024, 1, 1, 0), (8, 16, 16, 512, 512, 3, 2, 1),
              (8, 8, 8, 256, 256, 3, 1, 1), (8, 32, 32, 256, 256, 3, 1, 1), (8, 32, 32, 512, 128, 1, 1, 0), (8, 1, 110, 1024, 256, 1, 1, 0),
              (16, 8, 8, 512, 512, 3, 1, 1), (16, 16, 16, 256, 256, 3, 1, 1), (4, 16, 16, 512, 512, 3, 1, 1), (4, 32, 32, 256, 256, 3, 1, 1)]
    for (N, H, W, C, OC, k, s, pad) in shapes:
        OH, OW = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
        x = torch.randn(N, H, W, C, device=dev).bfloat16()
        wf = (torch.randn(OC, k, k, C, device=dev) / (k * k * C) ** 0.5).bfloat16()
        wb = (torch.randn(C, k, k, OC, device=dev) / (k * k * OC) ** 0.5).bfloat16()
        dy = torch.randn(N, OH, OW, OC, device=dev).bfloat16()
        y = torch.empty(N, OH, OW, OC, device=dev, dtype=torch.bfloat16)
        dx = torch.empty_like(x)
        stats = torch.zeros(8 * 2 * max(C, OC), device=dev, dtype=torch.float64)
        gf = 2.0 * N * OH * OW * OC * k * k * C / 1e9

        def fwd():
            L._raw_emrt_conv2d(P(x), P(wf), P(y), None, None, N, H, W, C, C, H * W * C, OH, OW, OC, OC, OH * OW * OC, 0, 0,
                               k, k, s, pad, 0, 0, 0, P(stats), None, 0, 0, 1, None, 1, stream)

        def dgrad():
            L._raw_emrt_conv2d(P(dy), P(wb), P(dx), None, None, N, OH, OW, OC, OC, OH * OW * OC, H, W, C, C, H * W * C, 0, 0,
                               k, k, s, pad, 1, 0, 0, None, None, 0, 0, 1, None, 1, stream)
        tiles = ((N * OH * OW + 63) // 64) * ((OC + 63) // 64)
        line = "N%d %dx%dx%d->%d k%d s%d %6.2f GF, %4d fwd tiles, %3d k-tiles |" % (N, H, W, C, OC, k, s, gf, tiles, k * k * C // 64)
        for nm, fn, out in (("fwd", fwd, y), ("dgrad", dgrad, dx)):
            res, outs = [], []
            for knob in (-1, 0, 2, 4, 8):
                old = L.set_tuning("xk", knob)
                fn()
                torch.cuda.synchronize()
                outs.append(out.float().clone())
                res.append(min(timed(fn, 20), timed(fn, 20)))
                same = True
                if knob > 0:
                    for _ in range(20):
                        fn()
                        same = same and torch.equal(out.float(), outs[-1])
                L.set_tuning("xk", old)
                res[-1] = (res[-1], same)
            rel = max(((o - outs[0]).norm() / outs[0].norm()).item() for o in outs[1:])
            line += " %s old %.1f auto %.1f S2 %.1f S4 %.1f S8 %.1f us  rel %.1e %s |" % (nm, res[0][0], res[1][0], res[2][0], res[3][0], res[4][0], rel,
                                                                                          "reproducible" if all(r[1] for r in res) else "NOT REPRODUCIBLE")
        print(line, flush=True)


def mid():
    """The step's mid-size GEMMs (10 752 token rows; the encoder's per-level 3x3 convs): every tile the dispatcher can be forced to, forward and data
    gradient, with the achieved TFLOP/s -- which tile bounds them (round 6: are they L2-operand-traffic bound on 64x64, latency bound on 128x128?)."""
    scratch = torch.empty((64 << 20) + (8 << 20) + 65536, dtype=torch.uint8, device=dev)
    L.call("emrt_set_scratch", P(scratch), ctypes.c_size_t(scratch.numel()), stream)
    shapes = [(8, 1, 1344, 256, 1024, 1, 1, 0), (8, 1, 1344, 1024, 256, 1, 1, 0), (8, 1, 1344, 256, 256, 1, 1, 0), (8, 1, 1344, 256, 432, 1, 1, 0),
              (8, 32, 32, 256, 256, 3, 1, 1), (8, 16, 16, 256, 256, 3, 1, 1), (8, 64, 64, 64, 256, 1, 1, 0), (8, 64, 64, 256, 64, 1, 1, 0),
              (8, 32, 32, 128, 512, 1, 1, 0), (8, 32, 32, 512, 128, 1, 1, 0)]
    for (N, H, W, C, OC, k, s, pad) in shapes:
        OH, OW = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
        x = torch.randn(N, H, W, C, device=dev).bfloat16()
        wf = (torch.randn(OC, k, k, C, device=dev) * 0.05).bfloat16()
        wb = (torch.randn(C, k, k, OC, device=dev) * 0.05).bfloat16()
        y = torch.empty(N, OH, OW, OC, device=dev, dtype=torch.bfloat16)
        dx = torch.empty_like(x)
        gf = 2.0 * N * OH * OW * OC * k * k * C / 1e9

        def fwd():
            L._raw_emrt_conv2d(P(x), P(wf), P(y), None, None, N, H, W, C, C, H * W * C, OH, OW, OC, OC, OH * OW * OC, 0, 0,
                               k, k, s, pad, 0, 0, 0, None, None, 0, 0, 1, None, 1, stream)

        def dgrad():
            L._raw_emrt_conv2d(P(y), P(wb), P(dx), None, None, N, OH, OW, OC, OC, OH * OW * OC, H, W, C, C, H * W * C, 0, 0,
                               k, k, s, pad, 1, 0, 0, None, None, 0, 0, 1, None, 1, stream)
        line = "N%d %dx%dx%d->%d k%d %6.2f GF |" % (N, H, W, C, OC, k, gf)
        for name, fn in (("fwd", fwd), ("dgrad", dgrad)):
            res = []
            for tile in (0, 1, 2, 3, 8, 7):
                L.set_tuning("conv_tile", tile)
                res.append(min(timed(fn), timed(fn)))
            L.set_tuning("conv_tile", 0)
            line += " %s auto %.1f (%.0f TF/s) 64x64 %.1f 128x64 %.1f 128x128 %.1f 128x128-k2 %.1f 256x256-8p %.1f |" % ((name, res[0], gf / res[0] * 1e3) + tuple(res[1:]))
        print(line, flush=True)


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which == "mid":
        return mid()
    if which == "xk":
        return xk()
    if which == "s2":
        return s2()
    if which == "wgroup":
        return wgroup()
    if which == "wgroup8":
        return wgroup8()
    if which == "wbig":
        return wbig()
    if which == "thin":
        return thin()
    if which == "big":
        return big()
    for (N, H, W, C, OC, k, s, pad) in SHAPES:
        OH = (H + 2 * pad - k) // s + 1
        OW = (W + 2 * pad - k) // s + 1
        x = torch.randn(N, H, W, C, device=dev).bfloat16()
        wf = (torch.randn(OC, k, k, C, device=dev) * 0.05).bfloat16()
        wb = (torch.randn(C, k, k, OC, device=dev) * 0.05).bfloat16()
        y = torch.empty(N, OH, OW, OC, device=dev, dtype=torch.bfloat16)
        dx = torch.empty_like(x)
        dw = torch.zeros(OC, k, k, C, device=dev, dtype=torch.float32)
        gf = 2.0 * N * OH * OW * OC * k * k * C / 1e9

        def fwd():
            L._raw_emrt_conv2d(P(x), P(wf), P(y), None, None, N, H, W, C, C, H * W * C, OH, OW, OC, OC, OH * OW * OC, 0, 0,
                               k, k, s, pad, 0, 0, 0, None, None, 0, 0, 1, None, 1, stream)

        def dgrad():
            L._raw_emrt_conv2d(P(y), P(wb), P(dx), None, None, N, OH, OW, OC, OC, OH * OW * OC, H, W, C, C, H * W * C, 0, 0,
                               k, k, s, pad, 1, 0, 0, None, None, 0, 0, 1, None, 1, stream)

        def wgrad():
            L._raw_emrt_conv2d_wgrad(P(x), P(y), P(dw), N, H, W, C, C, H * W * C, OH, OW, OC, OC, OH * OW * OC,
                                     k, k, s, pad, None, 1, 1, stream)

        def bwd():
            L._raw_emrt_conv2d_bwd(P(x), P(y), P(wb), P(dx), C, H * W * C, 0, P(dw), None, N, H, W, C, C, H * W * C, OH, OW, OC, OC, OH * OW * OC,
                                   k, k, s, pad, None, None, 0, 0, 1.0, None, 0, 0, None, 0, 0, 1, 1, stream)

        line = "N%d %dx%dx%d->%d k%d  %6.2f GF |" % (N, H, W, C, OC, k, gf)
        if which == "bwd":
            res = []
            for pm in (0, 1 << 30):
                old = L.set_tuning("pair_max", pm)
                res.append(timed(bwd))
                L.set_tuning("pair_max", old)
            Md = N * H * W
            nd = ((Md + 63) // 64) * ((C + 63) // 64)
            line += " bwd separate %.1f paired %.1f   (dgrad 64x64 tiles %d)" % (res[0], res[1], nd)
        if which in ("all", "conv"):
            for name, fn in (("fwd", fwd), ("dgrad", dgrad)):
                res = []
                for tile in (0, 1, 5, 6, 3, 8):
                    L.set_tuning("conv_tile", tile)
                    res.append(timed(fn))
                L.set_tuning("conv_tile", 0)
                line += " %s auto %.1f 64x64 %.1f ksplit2 %.1f ksplit4 %.1f 128x128 %.1f 128x128-ksplit2 %.1f |" % (name, *res)
        if which in ("all", "wgrad"):
            res = []
            for sp in (0, 1, 2, 4, 8, 16, 32):
                L.set_tuning("wgrad_split", sp)
                res.append(timed(wgrad))
            L.set_tuning("wgrad_split", 0)
            line += " wgrad auto %.1f S1 %.1f S2 %.1f S4 %.1f S8 %.1f S16 %.1f S32 %.1f" % tuple(res)
        print(line, flush=True)


if __name__ == "__main__":
    main()
