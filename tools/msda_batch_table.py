"""MSDA encoder call (Lq = Lv = 1344, 8 heads x 3 levels x 6 points, bf16) per launch batch B = 8 ... 64: forward and backward against the HBM roofline
(algorithmic bytes of SURVEY 8d / 8 TB/s) and the forward's LDS gather against 128 B/clk/CU.  Run under rocprofv3 --kernel-trace --stats for the kernel
clock (tools/r6/msda_table.sh); stand-alone it prints the back-to-back event clock (the recorded call re-issued 40 times behind a backlogged queue).

    python tools/msda_batch_table.py [out.txt]
"""
import sys

sys.path.insert(0, ".")
import torch

from emrt_amd import functional as Fn, _lib
from emrt_amd.runtime import BF16, Tape
from emrt_amd.src.models.emrt import encoder_reference_points
from tests.hip_utils import init, dev

shapes = [(32, 32), (16, 16), (8, 8)]
M, L, Pn = 8, 3, 6
Lv = sum(h * w for h, w in shapes)
Lq, tp = Lv, M * L * Pn
c = init(BF16)
Lb = _lib.lib()
lines = ["MSDA encoder call, bf16, Lq = Lv = 1344, M = 8, L = 3, P = 6 on one MI355X: us per launch, 40 back-to-back C-ABI calls between two events, min of 10 (the queue stays backlogged: the host issues a call in less than a launch lasts)",
         "forward algorithmic bytes = B (Lv 256 e + Lq 144 3 4 + Lq 256 e); backward = forward's + dout + dvalue (e) + doffw (e per value): SURVEY 8d",
         "LDS gather bytes = B Lq M 18 samples 4 corners 64 B; LDS peak = 256 CUs x 128 B/clk x 2.4 GHz = 78.6 TB/s",
         "   B   fwd us   fwd MB   GB/s  HBM frac  LDS frac |   bwd us   bwd MB   GB/s  HBM frac"]
for B in (8, 16, 32, 64):
    g = torch.Generator().manual_seed(0)
    value = dev(torch.randn(B, Lv, 256, generator=g))
    offw = dev(torch.cat([torch.randn(B, Lq, 2 * tp, generator=g) * 2, torch.randn(B, Lq, tp, generator=g)], -1), torch.float32)
    ref = encoder_reference_points(shapes).cuda()
    dy = dev(torch.randn(B, Lq, 256, generator=g))
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
    fwd = [r for r in rec if r[0] == "emrt_msda_fwd"]
    bwd = [r for r in rec if r[0] == "emrt_msda_bwd"]
    assert len(fwd) == 1 and len(bwd) == 1, [r[0] for r in rec]

    def clock(one):
        for _ in range(5):
            Lb.replay(one)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40):
                Lb.replay(one)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / 40)
        return best
    tf, tb = clock(fwd), clock(bwd)
    af = B * (Lv * 256 * 2 + Lq * tp * 3 * 4 + Lq * 256 * 2)
    ab = af + B * (Lq * 256 * 2 + Lv * 256 * 2 + Lq * tp * 3 * 2)
    lds = B * Lq * M * 18 * 4 * 64
    lines.append("%4d  %7.1f  %7.1f  %5.0f     %.3f     %.3f | %8.1f  %7.1f  %5.0f     %.3f" % (
        B, tf, af / 1e6, af / tf / 1e3, af / tf / 1e3 / 8000, lds / tf / 1e3 / 78643, tb, ab / 1e6, ab / tb / 1e3, ab / tb / 1e3 / 8000))
    print(lines[-1], flush=True)
out = "\n".join(lines) + "\n"
print(out)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(out)
