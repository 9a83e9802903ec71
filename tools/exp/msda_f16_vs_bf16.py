"""Experiment: the LDS-staged MSDA forward at the cfg5 encoder shape (B=16) in fp16 and bf16, with the kernel's parts switched off
(msda_fwd_probe: 1 = no gather, 4 = no preparation, 5 = neither).  (developer tool, GPU)"""
import sys
sys.path.insert(0, ".")
import torch
from emrt_amd import functional as Fn, _lib
from emrt_amd.runtime import BF16, F16
from emrt_amd.src.models.emrt import encoder_reference_points
from tests.hip_utils import init, dev

M, L, Pn = 8, 3, 6
Lb = _lib.lib()
B, shapes = 16, [(32, 32), (16, 16), (8, 8)]
for name, dt in (("bf16", BF16), ("fp16", F16), ("bf16", BF16), ("fp16", F16)):
    c = init(dt)
    g = torch.Generator().manual_seed(0)
    Lv = sum(h * w for h, w in shapes)
    tp = M * L * Pn
    value = dev(torch.randn(B, Lv, 256, generator=g))
    offw = dev(torch.cat([torch.randn(B, Lv, 2 * tp, generator=g) * 2, torch.randn(B, Lv, tp, generator=g)], -1), torch.float32)
    ref = encoder_reference_points(shapes).cuda()

    def timed():
        c.keepalive = []
        Lb.start_record()
        Fn.msda(value, offw, ref, shapes, M, Pn)
        rec = Lb.stop_record()
        torch.cuda.synchronize()
        for _ in range(3):
            Lb.replay(rec)
        ts = []
        for _ in range(20):
            ts += [ms * 1e3 for _, _, ms in Lb.replay(rec, timed=True)]
        c.keepalive = None
        ts.sort()
        return ts[0], ts[len(ts) // 2]
    line = "%s:" % name
    for probe in (0, 1, 4, 5):
        Lb.set_tuning("msda_fwd_probe", probe)
        line += "  p%d min %.1f med %.1f" % ((probe,) + timed())
    Lb.set_tuning("msda_fwd_probe", 0)
    print(line, flush=True)
