import sys, ctypes, os; sys.path.insert(0, '.')
import torch
from emrt_amd import functional as Fn, _lib
from emrt_amd.runtime import ctx, BF16, F32, Tape
from tests.hip_utils import init, dev
dt = BF16
c = init(dt)
g = torch.Generator().manual_seed(0)
B, M, L, Pn = 8, 8, 3, 6
shapes = [(32, 32), (16, 16), (8, 8)]
Lv = sum(h * w for h, w in shapes); Lq = Lv; tp = M * L * Pn
value = dev(torch.randn(B, Lv, 256, generator=g))
offw = dev(torch.cat([torch.randn(B, Lq, 2 * tp, generator=g) * 2, torch.randn(B, Lq, tp, generator=g)], -1), torch.float32)
from emrt_amd.src.models.emrt import encoder_reference_points
ref = encoder_reference_points(shapes).cuda()
dy = dev(torch.randn(B, Lq, 256, generator=g))
L_ = _lib.lib()
def run(n=20):
    tape = Tape(); c.tape = tape; y = Fn.msda(value, offw, ref, shapes, M, Pn); c.tape = None
    tape.add_grad(y, dy)
    L_.start_profile(); 
    for _ in range(1): pass
    # time backward calls
    import time
    torch.cuda.synchronize()
    tape.backward()
    calls = L_.stop_profile()
    return {n_: ms for n_, a, ms in calls}
for i in range(3):
    r = run()
print("EMRT_MSDA_DBG=%s" % os.environ.get("EMRT_MSDA_DBG", "0"), {k: round(v * 1e3, 1) for k, v in r.items()}, "us")
