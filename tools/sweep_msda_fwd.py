"""Sweep of the LDS-staged MSDA forward's launch shape and timing probes (developer tool, GPU)."""
import sys
sys.path.insert(0, ".")
import torch
from emrt_amd import functional as Fn, _lib
from emrt_amd.runtime import BF16
from emrt_amd.src.models.emrt import encoder_reference_points
from tests.hip_utils import init, dev

c = init(BF16)
g = torch.Generator().manual_seed(0)
M, L, Pn = 8, 3, 6
Lb = _lib.lib()
for cfg, B, shapes in (("cfg2", 8, [(32, 32), (16, 16), (8, 8)]), ("cfg5", 16, [(32, 32), (16, 16), (8, 8)])):
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
        for _ in range(10):
            ts += [ms * 1e3 for _, _, ms in Lb.replay(rec, timed=True)]
        c.keepalive = None
        return min(ts)

    for threads in (1024, 768, 512, 256):
        Lb.set_tuning("msda_fwd_threads", threads)
        for chunks in (1, 2, 3, 4, 6, 8):
            Lb.set_tuning("msda_fwd_chunks", chunks)
            line = "%s threads %4d chunks %d:" % (cfg, threads, chunks)
            for probe in (0, 1, 2, 3, 4, 5, 7):
                Lb.set_tuning("msda_fwd_probe", probe)
                line += "  p%d %.1f" % (probe, timed())
            Lb.set_tuning("msda_fwd_probe", 0)
            print(line, flush=True)
    Lb.set_tuning("msda_fwd_threads", 1024)
    Lb.set_tuning("msda_fwd_chunks", 0)
    Lb.set_tuning("msda_fwd_global", 1)
    print(cfg, "global-gather kernel: %.1f us" % timed())
    Lb.set_tuning("msda_fwd_global", 0)
