"""BatchNorm + ReLU applied by the consuming convolution's loads (emrt_conv2d_bna) against emrt_bn_apply + emrt_conv2d, per layer shape of the
ResNet-50 step at batch 8, 256 x 256 (developer tool, GPU).  usage: python tools/r6/bench_bna.py"""
import ctypes
import sys

sys.path.insert(0, ".")
import torch

from emrt_amd import _lib

L = _lib.lib()
dev = torch.device("cuda:0")
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None

SHAPES = [  # name, N, H, W, C, OC, k
    ("layer1 conv2 3x3", 8, 64, 64, 64, 64, 3), ("layer1 conv3 1x1", 8, 64, 64, 64, 256, 1),
    ("layer2 conv2 3x3", 8, 32, 32, 128, 128, 3), ("layer2 conv3 1x1", 8, 32, 32, 128, 512, 1),
    ("layer3 conv2 3x3", 8, 16, 16, 256, 256, 3), ("layer3 conv3 1x1", 8, 16, 16, 256, 1024, 1),
    ("layer4 conv2 3x3", 8, 8, 8, 512, 512, 3), ("layer4 conv3 1x1", 8, 8, 8, 512, 2048, 1),
    ("cls_psp.3 3x3", 8, 32, 32, 512, 256, 3), ("EFP 32x32 3x3", 8, 32, 32, 256, 256, 3), ("EFP 16x16 3x3", 8, 16, 16, 256, 256, 3),
]


def timed(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    scratch = torch.empty((64 << 20) + (8 << 20) + 65536, dtype=torch.uint8, device=dev)
    L.call("emrt_set_scratch", P(scratch), ctypes.c_size_t(scratch.numel()), stream)
    tot = [0.0, 0.0, 0.0]
    for (name, N, H, W, C, OC, k) in SHAPES:
        pad = k // 2
        raw = torch.randn(N, H, W, C, device=dev).bfloat16()
        a = torch.empty_like(raw)
        a2 = torch.empty_like(raw)
        wf = (torch.randn(OC, k, k, C, device=dev) / (k * k * C) ** 0.5).bfloat16()
        y, y2 = torch.empty(N, H, W, OC, device=dev, dtype=torch.bfloat16), torch.empty(N, H, W, OC, device=dev, dtype=torch.bfloat16)
        M = N * H * W
        sums = torch.zeros(8, 2, C, device=dev, dtype=torch.float64)
        sums[0, 0] = raw.double().sum((0, 1, 2))
        sums[0, 1] = (raw.double() ** 2).sum((0, 1, 2))
        st2 = torch.zeros(8 * 2 * OC, device=dev, dtype=torch.float64)
        mean, invstd = torch.empty(C, device=dev), torch.empty(C, device=dev)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        gam, bet = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1

        def apply():
            L._raw_emrt_bn_apply(P(raw), C, None, 0, P(a), C, P(sums), float(M), 1e-5, 0.9, P(mean), P(invstd), P(rm), P(rv), P(gam), P(bet), M, C, 1, 1, stream)

        def conv():
            L._raw_emrt_conv2d(P(a), P(wf), P(y), None, None, N, H, W, C, C, H * W * C, H, W, OC, OC, H * W * OC, 0, 0, k, k, 1, pad, 0, 0, 0, P(st2), None, 0, 0, 1, None, 1, stream)

        def two():
            apply()
            conv()

        args = (P(raw), P(wf), P(y2), None, None, N, H, W, C, C, H * W * C, H, W, OC, OC, H * W * OC, 0, 0, k, k, 1, pad, 0, 0, P(st2), 1, P(sums), float(M), 1e-5, 0.9,
                P(mean), P(invstd), P(rm), P(rv), P(gam), P(bet), 1, P(a2), 1, stream)
        L.set_tuning("no_bna", -1)          # (every shape, also the ones the dispatcher leaves to the separate launch)
        ok = L._raw_emrt_conv2d_bna_supported(*args)

        def fused():
            L._raw_emrt_conv2d_bna(*args)
        t_apply, t_conv, t_two = timed(apply), timed(conv), timed(two)
        line = "%-20s N%d %dx%dx%d->%d k%d | bn_apply %.1f + conv %.1f = pair %.1f us |" % (name, N, H, W, C, OC, k, t_apply, t_conv, t_two)
        if ok:
            t_f = timed(fused)
            two()
            fused()
            torch.cuda.synchronize()
            same = torch.equal(y, y2) and torch.equal(a, a2)
            line += " fused %.1f us (%+.1f) %s" % (t_f, t_f - t_two, "bit-identical" if same else "DIFFERENT")
            tot[0] += t_two
            tot[1] += t_f
        else:
            line += " not supported"
        print(line, flush=True)
    print("sum over supported shapes: pair %.1f us, fused %.1f us" % (tot[0], tot[1]))


if __name__ == "__main__":
    main()
