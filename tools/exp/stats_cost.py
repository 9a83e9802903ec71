"""Experiment: what do the BatchNorm-statistics atomics in the conv epilogue cost?  Forward conv with and without `bn_stats` (developer tool, GPU)."""
import ctypes, sys
sys.path.insert(0, ".")
import torch
from emrt_amd import _lib
L = _lib.lib()
dev = torch.device("cuda:0")
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None

def timed(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for (N, H, W, C, OC, k) in [(8, 128, 128, 256, 256, 3), (8, 64, 64, 256, 256, 3), (8, 64, 64, 64, 256, 1), (8, 64, 64, 256, 64, 1), (8, 32, 32, 128, 512, 1), (8, 16, 16, 256, 1024, 1), (8, 16, 16, 256, 256, 3), (8, 8, 8, 512, 2048, 1)]:
    pad = k // 2
    x = torch.randn(N, H, W, C, device=dev).bfloat16()
    wf = (torch.randn(OC, k, k, C, device=dev) * 0.05).bfloat16()
    y = torch.empty(N, H, W, OC, device=dev, dtype=torch.bfloat16)
    stats = torch.zeros(8 * 2 * OC, device=dev, dtype=torch.float64)
    def fwd(st):
        L._raw_emrt_conv2d(P(x), P(wf), P(y), None, None, N, H, W, C, C, H * W * C, H, W, OC, OC, H * W * OC, 0, 0,
                           k, k, 1, pad, 0, 0, 0, P(stats) if st else None, None, 0, 0, 1, None, 1, stream)
    a = min(timed(lambda: fwd(False)), timed(lambda: fwd(False)))
    b = min(timed(lambda: fwd(True)), timed(lambda: fwd(True)))
    print("N%d %dx%dx%d->%d k%d: without stats %.1f us, with %.1f us (+%.1f)" % (N, H, W, C, OC, k, a, b, b - a), flush=True)
