"""-m gpu: seeded random pyramids through the deformable-attention kernels (LDS-staged, row-band and global-gather forward; LDS gradient,
band gradient, integer LDS scatter with and without query split, global-atomic backward -- whichever the dispatchers pick for the geometry)
against the oracle's torch expression of deformable_attention_core_func (EMRT_utils/utils.py:64-97) on the same rounded inputs.
Odd and non-square level sizes, levels that do not halve, few and many queries, offsets from sub-pixel to far outside the map."""
import random

import pytest
import torch

pytestmark = pytest.mark.gpu

from emrt_amd import functional as Fn                    # noqa: E402
from emrt_amd.runtime import F32, BF16, Tape              # noqa: E402
from tests.hip_utils import init, dev, host, rnd          # noqa: E402
from tests.test_gpu_kernels import _msda_ref, run_bwd     # noqa: E402


def _cases(seed, n):
    rng = random.Random(seed)
    out = []
    for i in range(n):
        h0, w0 = rng.randint(6, 72), rng.randint(6, 72)
        if rng.random() < 0.5:      # a pyramid that halves (rounded up), as the backbone produces
            shapes = [(h0, w0), ((h0 + 1) // 2, (w0 + 1) // 2), ((h0 + 3) // 4, (w0 + 3) // 4)]
        else:                       # three unrelated maps
            shapes = [(h0, w0), (rng.randint(2, 40), rng.randint(2, 40)), (rng.randint(1, 20), rng.randint(1, 20))]
        out.append(dict(name="msda-fuzz%d-%d" % (seed, i), B=rng.randint(1, 6), shapes=shapes,
                        Lq=None if rng.random() < 0.6 else rng.randint(1, 400), sigma=rng.choice([0.3, 2.5, 2.5, 8.0, 40.0]),
                        dtype=rng.choice([BF16, BF16, F32]), seed=seed * 100 + i))
    return out


CASES = _cases(77, 16)


@pytest.mark.parametrize("cfg", CASES, ids=[c["name"] for c in CASES])
def test_msda_random_pyramid_vs_oracle(cfg):
    dt = cfg["dtype"]
    c = init(dt)
    g = torch.Generator().manual_seed(cfg["seed"])
    M, L, Pn = 8, 3, 6
    shapes, B = cfg["shapes"], cfg["B"]
    Lv = sum(h * w for h, w in shapes)
    Lq = cfg["Lq"] or Lv
    tp = M * L * Pn
    r = rnd if dt == BF16 else (lambda t: t)
    value = r(torch.randn(B, Lv, M * 32, generator=g))
    offw = torch.cat([torch.randn(B, Lq, 2 * tp, generator=g) * cfg["sigma"], torch.randn(B, Lq, tp, generator=g)], -1)
    if cfg["Lq"] is None:
        from emrt_amd.src.models.emrt import encoder_reference_points
        ref = encoder_reference_points(shapes)
    else:
        ref = torch.rand(1, Lq, 1, 2, generator=g)
    vr, orq = value.clone().requires_grad_(True), offw.clone().requires_grad_(True)
    out_r = _msda_ref(vr, orq, ref, shapes, M, L, Pn)
    dy = r(torch.randn(out_r.shape, generator=g))
    out_r.backward(dy)
    vd, od, rd = dev(value), dev(offw, torch.float32), dev(ref, torch.float32)
    tape = Tape()
    c.tape = tape
    y = Fn.msda(vd, od, rd, shapes, M, Pn)
    c.tape = None
    tape.watch(vd)
    tape.watch(od)
    den = out_r.detach().norm().item() or 1.0
    rel = ((host(y) - out_r.detach()).norm() / den).item()
    dv, do = run_bwd(tape, [(y, dev(dy))], [vd, od])
    gv, go = vr.grad, orq.grad
    rel_v = ((host(dv) - gv).norm() / (gv.norm().item() or 1.0)).item()
    if dt == BF16:      # the other value-gradient kernel too: matrix product forced wherever its plan exists / never (csrc/msda.hip: knob msda_scatter_mfma)
        from emrt_amd import _lib
        for knob in (2, 0):
            old = _lib.lib().set_tuning("msda_scatter_mfma", knob)
            try:
                tape2 = Tape()
                c.tape = tape2
                y2 = Fn.msda(vd, od, rd, shapes, M, Pn)
                c.tape = None
                tape2.watch(vd)
                dv2, = run_bwd(tape2, [(y2, dev(dy))], [vd])
            finally:
                _lib.lib().set_tuning("msda_scatter_mfma", old)
            rel_k = ((host(dv2) - gv).norm() / (gv.norm().item() or 1.0)).item()
            print("    msda_scatter_mfma = %d: dvalue rel %.2e" % (knob, rel_k))
            rel_v = max(rel_v, rel_k)
    rel_o = ((host(do) - go).norm() / (go.norm().item() or 1.0)).item()
    print("%s %s B=%d Lq=%d sigma=%.1f %s: fwd rel %.2e, dvalue rel %.2e, doffw rel %.2e" % (
        cfg["name"], shapes, B, Lq, cfg["sigma"], "bf16" if dt == BF16 else "fp32", rel, rel_v, rel_o))
    # bf16: one rounding of each stored result (2^-9) + the gather's weights rounded to bf16; fp32: summation order only
    # (the value gradient's integer scatter quantises to 2^-30 of Lq * max|g| per addend)
    tol, tol_v = (4e-3, 6e-3) if dt == BF16 else (2e-5, 2e-4)
    assert rel < tol and rel_v < tol_v and rel_o < (6e-3 if dt == BF16 else 2e-4), (rel, rel_v, rel_o)
