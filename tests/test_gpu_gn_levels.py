"""Multi-level GroupNorm (+GELU, +residual) of the encoder's conv branch and of input_proj: emrt_groupnorm_levels_fwd / _bwd
against torch's group_norm + autograd per level (reference: transformer_encoder_decoder.py:125-144,163-182,375-379).

Both implementations are exercised: the row-major statistics + apply pair (default, 8 channels per group) and the older
one-block-per-(image, group) kernel (tuning knob gn_group_blocks).  Level shapes include the ragged ones of a non-square tile
(12x20 / 6x10 / 3x5: not multiples of any block size) and the benchmark's 32^2 / 16^2 / 8^2."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from emrt_amd import _lib
from emrt_amd.functional import P
from emrt_amd.runtime import F32, BF16, F16
from tests.hip_utils import init, dev

pytestmark = pytest.mark.gpu


def _run(dtype, B, hws, gelu, with_res, group_blocks, seed=0):
    c = init(dtype)
    L, C, G = len(hws), 256, 32
    Lv = sum(hws)
    g = torch.Generator().manual_seed(seed)
    tdt = c.tdtype
    x = (torch.randn(B, Lv, C, generator=g) * 1.5 + 0.3).to(tdt)
    res = torch.randn(B, Lv, C, generator=g).to(tdt) if with_res else None
    dy = torch.randn(B, Lv, C, generator=g).to(tdt)
    gam = [torch.rand(C, generator=g) + 0.5 for _ in range(L)]
    bet = [torch.randn(C, generator=g) * 0.2 for _ in range(L)]
    # ---- reference: fp64 on the dtype-rounded inputs
    xr = x.double().requires_grad_(True)
    gr = [t.double().requires_grad_(True) for t in gam]
    br = [t.double().requires_grad_(True) for t in bet]
    outs, s0 = [], 0
    for l, n in enumerate(hws):
        xl = xr[:, s0:s0 + n].transpose(1, 2)                         # [B, C, n]
        o = F.group_norm(xl, G, gr[l], br[l], 1e-5)
        if gelu:
            o = F.gelu(o)
        o = o.transpose(1, 2)
        if with_res:
            o = o + res.double()[:, s0:s0 + n]
        outs.append(o)
        s0 += n
    want = torch.cat(outs, 1)
    want.backward(dy.double())
    # ---- HIP
    Lb = _lib.lib()
    old = Lb.set_tuning("gn_group_blocks", int(group_blocks))
    try:
        xd, rd, dyd = dev(x), (dev(res) if with_res else None), dev(dy)
        out = torch.empty_like(xd)
        dx = torch.empty_like(xd)
        gd = [dev(t, torch.float32) for t in gam]
        bd = [dev(t, torch.float32) for t in bet]
        dgd = [torch.zeros(C, device="cuda") for _ in range(L)]
        dbd = [torch.zeros(C, device="cuda") for _ in range(L)]
        starts = (ctypes.c_int * L)(*[sum(hws[:l]) for l in range(L)])
        hw = (ctypes.c_int * L)(*hws)
        arr = lambda ts: (ctypes.c_void_p * L)(*[t.data_ptr() for t in ts])
        mean = torch.empty(L * B * G, device="cuda")
        rstd = torch.empty(L * B * G, device="cuda")
        ws1 = torch.zeros(L * B * G * 2, dtype=torch.float64, device="cuda")
        ws2 = torch.zeros(L * B * G * 2, dtype=torch.float64, device="cuda")
        Lb.call("emrt_groupnorm_levels_fwd", P(xd), C, Lv * C, P(rd), C if with_res else 0, Lv * C if with_res else 0, P(out), C, Lv * C,
                arr(gd), arr(bd), P(mean), P(rstd), starts, hw, L, B, C, G, 1e-5, int(gelu), P(ws1), dtype, c.stream)
        if dtype != F16:
            Lb.call("emrt_groupnorm_levels_bwd", P(xd), C, Lv * C, P(dyd), C, Lv * C, P(dx), C, Lv * C, arr(gd), arr(bd), P(mean), P(rstd),
                    arr(dgd), arr(dbd), starts, hw, L, B, C, G, int(gelu), P(ws2), dtype, c.stream)
        torch.cuda.synchronize()
    finally:
        Lb.set_tuning("gn_group_blocks", old)
    return dict(out=out.float().cpu(), dx=dx.float().cpu(), dgam=[t.cpu() for t in dgd], dbet=[t.cpu() for t in dbd], mean=mean.cpu(),
                rstd=rstd.cpu(), want=want.detach().float(), wdx=xr.grad.float(), wdg=[t.grad.float() for t in gr], wdb=[t.grad.float() for t in br])


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("B,hws", [(2, (240, 60, 15)), (8, (1024, 256, 64)), (1, (1, 7))])
@pytest.mark.parametrize("gelu,with_res", [(True, True), (False, False)])
@pytest.mark.parametrize("group_blocks", [False, True])
def test_groupnorm_levels_match_torch(dtype, B, hws, gelu, with_res, group_blocks):
    r = _run(dtype, B, hws, gelu, with_res, group_blocks)
    tol = 2e-4 if dtype == F32 else 3e-2           # bf16: one rounding of an O(4) output (2^-8 relative)
    assert (r["out"] - r["want"]).abs().max().item() < tol * max(1.0, r["want"].abs().max().item())
    scale = max(1.0, r["wdx"].abs().max().item())
    assert (r["dx"] - r["wdx"]).abs().max().item() < tol * scale
    for l in range(len(hws)):
        for got, want in ((r["dgam"][l], r["wdg"][l]), (r["dbet"][l], r["wdb"][l])):
            assert (got - want).abs().max().item() < 2e-3 * max(1.0, want.abs().max().item()), l


def test_groupnorm_levels_fp16_forward_and_both_kernels_agree():
    a = _run(F16, 4, (1024, 256, 64), True, True, False)
    b = _run(F16, 4, (1024, 256, 64), True, True, True)
    assert (a["out"] - a["want"]).abs().max().item() < 4e-3 * max(1.0, a["want"].abs().max().item())
    assert (a["out"] - b["out"]).abs().max().item() < 4e-3 * max(1.0, a["want"].abs().max().item())
    assert (a["mean"] - b["mean"]).abs().max().item() < 1e-5 and ((a["rstd"] - b["rstd"]).abs() / b["rstd"]).max().item() < 1e-5
