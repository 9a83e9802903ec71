"""Experiment: what does a small conv launch pay for COLD weights (HBM) compared with weights that sit in L2 / Infinity Cache?
In the training step every weight matrix is touched once per direction and ~4 GB of other traffic pass before its next use, so it
comes from HBM each time; micro-benchmarks replay the same launch and see it hot.  (developer tool, GPU)"""
import ctypes
import sys

sys.path.insert(0, ".")
import torch

from emrt_amd import _lib

L = _lib.lib()
dev = torch.device("cuda:0")
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
flush_src = torch.empty(768 << 20, dtype=torch.uint8, device=dev)
flush_dst = torch.empty_like(flush_src)

SHAPES = [(8, 16, 16, 1024, 256, 1), (8, 16, 16, 256, 256, 3), (8, 8, 8, 512, 512, 3), (8, 8, 8, 2048, 512, 1), (8, 32, 32, 512, 128, 1), (8, 1, 1344, 256, 1024, 1)]
for (N, H, W, C, OC, k) in SHAPES:
    pad = k // 2
    x = torch.randn(N, H, W, C, device=dev).bfloat16()
    wf = (torch.randn(OC, k, k, C, device=dev) * 0.05).bfloat16()
    y = torch.empty(N, H, W, OC, device=dev, dtype=torch.bfloat16)

    def fwd():
        L._raw_emrt_conv2d(P(x), P(wf), P(y), None, None, N, H, W, C, C, H * W * C, H, W, OC, OC, H * W * OC, 0, 0,
                           k, k, 1, pad, 0, 0, 0, None, None, 0, 0, 1, None, 1, stream)

    def one(flush_w, flush_x):
        ts = []
        for _ in range(12):
            if flush_w or flush_x:
                flush_dst.copy_(flush_src)          # 1.5 GB of traffic: L2 and the 256 MB Infinity Cache now hold the copy
                if not flush_x:
                    x.add_(0)                        # re-touch the activations (they would be hot in the step: just written by the producer)
                if not flush_w:
                    wf.add_(0)
            else:
                fwd()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fwd()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        return ts[len(ts) // 2]
    print("N%d %dx%dx%d->%d k%d: hot %.1f us | cold weights, hot activations %.1f us | everything cold %.1f us   (median of 12, event pair included)" % (
        N, H, W, C, OC, k, one(False, False), one(True, False), one(True, True)), flush=True)
