"""Timing split of the 256x256 8-phase kernel's loop (developer tool, GPU, needs a -DEMRT_8P_PROBES build of conv.hip):
the kernel with its DMA issue (1), fragment reads (2) and MFMAs (4) switched off in turn.  Results of the probe variants are wrong."""
import ctypes
import sys
sys.path.insert(0, ".")
import torch
from emrt_amd import _lib
L = _lib.lib()
dev = torch.device("cuda:0")
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


SHAPES = [(8, 128, 128, 256, 256, 3), (8, 128, 128, 512, 256, 3), (8, 64, 64, 256, 256, 3), (8, 32, 32, 1536, 1536, 1),
          (8, 128, 128, 64, 256, 1), (8, 128, 128, 128, 256, 1), (8, 64, 64, 64, 256, 1)]      # last three: 1-2 k-tiles = the kernel's fixed cost
for (N, H, W, C, OC, k) in SHAPES:
    pad = k // 2
    x = torch.randn(N, H, W, C, device=dev).bfloat16()
    wf = (torch.randn(OC, k, k, C, device=dev) / (k * k * C) ** 0.5).bfloat16()
    y = torch.empty(N, H, W, OC, device=dev, dtype=torch.bfloat16)

    def fwd():
        L._raw_emrt_conv2d(P(x), P(wf), P(y), None, None, N, H, W, C, C, H * W * C, H, W, OC, OC, H * W * OC, 0, 0,
                           k, k, 1, pad, 0, 0, 0, None, None, 0, 0, 1, None, 1, stream)
    L.set_tuning("conv_tile", 7)
    line = "N%d %dx%dx%d->%d k%d (%d k-tiles, %d blocks):" % (N, H, W, C, OC, k, k * k * C // 64, (N * H * W // 256) * ((OC + 255) // 256))
    for pr in (0, 32, 0, 32, 8):
        L.set_tuning("igemm8p_probe", pr)
        line += "  probe %d %.1f us" % (pr, timed(fn=fwd))
    L.set_tuning("igemm8p_probe", 0)
    L.set_tuning("conv_tile", 0)
    print(line, flush=True)
