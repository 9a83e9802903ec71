"""MSDA forward / backward micro-benchmark at the encoder shapes (developer tool, GPU).  Run under
`rocprofv3 --kernel-trace --stats` to split the backward into its gradient and scatter kernels.

    python tools/bench_msda.py [cfg2|cfg3|cfg5] [bf16|fp16|fp32]
"""
import sys

sys.path.insert(0, ".")
import torch

from emrt_amd import functional as Fn, _lib
from emrt_amd.runtime import BF16, F16, F32, Tape
from emrt_amd.src.models.emrt import encoder_reference_points
from tests.hip_utils import init, dev

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
dt = {"bf16": BF16, "fp16": F16, "fp32": F32}[sys.argv[2] if len(sys.argv) > 2 else "bf16"]
B, shapes = {"cfg2": (8, [(32, 32), (16, 16), (8, 8)]), "cfg3": (4, [(64, 64), (32, 32), (16, 16)]), "cfg5": (16, [(32, 32), (16, 16), (8, 8)])}[cfg]
c = init(dt)
g = torch.Generator().manual_seed(0)
M, L, Pn = 8, 3, 6
Lv = sum(h * w for h, w in shapes)
Lq = Lv
tp = M * L * Pn
value = dev(torch.randn(B, Lv, 256, generator=g))
offw = dev(torch.cat([torch.randn(B, Lq, 2 * tp, generator=g) * 2, torch.randn(B, Lq, tp, generator=g)], -1), torch.float32)
ref = encoder_reference_points(shapes).cuda()
dy = dev(torch.randn(B, Lq, 256, generator=g))
Lb = _lib.lib()
import os
if os.environ.get("BENCH_MSDA_PROBE"):       # timing experiments (tools/exp/probe_msda_*.sh): parts of the kernels switched off, results WRONG
    Lb.set_tuning("msda_fwd_probe", int(os.environ["BENCH_MSDA_PROBE"]))
esz = 4 if dt == F32 else 2
alg = B * (Lv * 256 * esz + Lq * tp * 3 * 4 + Lq * 256 * esz)

c.keepalive = []
tape = Tape()
c.tape = tape
Lb.start_record()
y = Fn.msda(value, offw, ref, shapes, M, Pn)
c.tape = None
if dt != F16:
    tape.add_grad(y, dy)
    tape.backward()
rec = Lb.stop_record()
torch.cuda.synchronize()
for _ in range(5):
    Lb.replay(rec)
res = {}
for _ in range(10):
    for name, a, ms in Lb.replay(rec, timed=True):
        res.setdefault(name, []).append(ms * 1e3)
print(cfg, sys.argv[2:] or "bf16", {k: round(min(v), 1) for k, v in res.items()}, "us (min of 10, HIP events around each launch)")
t = min(res["emrt_msda_fwd"])
print("forward: %.1f MB algorithmic / %.1f us = %.0f GB/s = %.3f of 8 TB/s" % (alg / 1e6, t, alg / t / 1e3, alg / t / 1e3 / 8000))
