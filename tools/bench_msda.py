"""MSDA forward / backward micro-benchmark at the encoder shape (developer tool, GPU).  Run under
`rocprofv3 --kernel-trace --stats` to split the backward into its gradient and scatter kernels."""
import sys

sys.path.insert(0, ".")
import torch

from emrt_amd import functional as Fn, _lib
from emrt_amd.runtime import BF16, Tape
from emrt_amd.src.models.emrt import encoder_reference_points
from tests.hip_utils import init, dev

c = init(BF16)
g = torch.Generator().manual_seed(0)
B, M, L, Pn = 8, 8, 3, 6
shapes = [(32, 32), (16, 16), (8, 8)]
Lv = sum(h * w for h, w in shapes)
Lq = Lv
tp = M * L * Pn
value = dev(torch.randn(B, Lv, 256, generator=g))
offw = dev(torch.cat([torch.randn(B, Lq, 2 * tp, generator=g) * 2, torch.randn(B, Lq, tp, generator=g)], -1), torch.float32)
ref = encoder_reference_points(shapes).cuda()
dy = dev(torch.randn(B, Lq, 256, generator=g))
Lb = _lib.lib()

c.keepalive = []
tape = Tape()
c.tape = tape
Lb.start_record()
y = Fn.msda(value, offw, ref, shapes, M, Pn)
c.tape = None
tape.add_grad(y, dy)
tape.backward()
rec = Lb.stop_record()
torch.cuda.synchronize()
for _ in range(5):
    Lb.replay(rec)
res = {}
for _ in range(5):
    for name, a, ms in Lb.replay(rec, timed=True):
        res.setdefault(name, []).append(ms * 1e3)
print({k: round(min(v), 1) for k, v in res.items()}, "us (min of 5)")
