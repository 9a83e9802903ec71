"""Helpers for the -m gpu parity tests: host<->device layout conversion and a tiny parameter store for single layers."""
import torch
import torch.nn as tnn

from emrt_amd import nn as hnn
from emrt_amd.runtime import ctx, F32, BF16, Tape

TOL = {F32: dict(atol=2e-4, rtol=2e-4), BF16: dict(atol=6e-2, rtol=6e-2)}


def init(dtype, seed=0):
    c = ctx()
    c.init_device("cuda:0", dtype, seed)
    c.ensure_scratch()          # (a training context: the model's train-mode forward does this; bf16 only)
    c.training = True
    c.tape = None
    c.world_size = 1
    return c


def dev_map(t_nchw, dtype=None):
    """CPU [N,C,H,W] float -> device [N,H,W,C] in the compute dtype."""
    c = ctx()
    return t_nchw.permute(0, 2, 3, 1).contiguous().to(device=c.device, dtype=dtype or c.tdtype)


def host_map(t_nhwc):
    return t_nhwc.float().cpu().permute(0, 3, 1, 2).contiguous()


def dev(t, dtype=None):
    c = ctx()
    return t.contiguous().to(device=c.device, dtype=dtype or c.tdtype)


def host(t):
    return t.float().cpu()


def rnd(x):
    """Round a CPU fp32 tensor through the compute dtype so CPU reference and GPU kernel see identical inputs."""
    return x.to(ctx().tdtype).float()


class Holder(tnn.Module):
    """Wraps HIP layers so ParamStore / bind_all can place their parameters on the device."""

    def __init__(self, **layers):
        super().__init__()
        for k, v in layers.items():
            self.add_module(k, v)

    def place(self):
        c = ctx()
        self.store = hnn.ParamStore(self, c.device, c.dtype)
        hnn.bind_all(self, self.store)
        self.store.pack()
        return self


def close(name, got, ref, dtype, scale=1.0, atol=None, rtol=None):
    tol = dict(TOL[dtype])
    if atol is not None:
        tol["atol"] = atol
    if rtol is not None:
        tol["rtol"] = rtol
    got, ref = got.float().cpu(), ref.float().cpu()
    assert got.shape == ref.shape, "%s: shape %s vs %s" % (name, tuple(got.shape), tuple(ref.shape))
    err = (got - ref).abs()
    bound = tol["atol"] * scale + tol["rtol"] * ref.abs()
    bad = err > bound
    if bad.any():
        idx = torch.nonzero(bad)[0].tolist()
        raise AssertionError("%s: %d/%d elements off; max |diff| %.4g (ref max %.4g); first at %s got %.6g ref %.6g" % (
            name, int(bad.sum()), bad.numel(), err.max().item(), ref.abs().max().item(), idx,
            got[tuple(idx)].item(), ref[tuple(idx)].item()))
    assert torch.isfinite(got).all(), "%s: non-finite values" % name


def close_gemm(name, got, ref, dtype, out_bits=None):
    """Tolerance of a GEMM-shaped kernel (convolution / linear forward, data gradient, weight gradient) from what can actually differ, instead of
    the flat 6e-2 of TOL[BF16].  The reference is fp32 torch on the SAME rounded operands, the kernel multiplies them exactly (bf16 x bf16 fits
    fp32) and accumulates in fp32, so the only differences are
      * the rounding of the stored result: 2^-9 relative for a bf16 output (out_bits = 8 significant bits), none for an fp32 output;
      * the fp32 summation order over the contraction: ~2^-23 sqrt(K) of the terms' magnitude, i.e. parts in 10^5..10^6 of the output's RMS.
    bound = 2^-out_bits |ref| + 2^-(out_bits + 2) rms(ref) for a rounded output (twice the rounding; the RMS floor covers outputs that cancel to
    ~0), 2^-12 (|ref| + rms(ref)) for an fp32 output (fp32 atomics / split reductions in any order).  A mis-weighted tap of a 3x3 kernel moves an
    output by ~rms(ref) / 3: 40 to 100 times these bounds (the flat tolerance let it pass at kernel level)."""
    got, ref = got.float().cpu(), ref.float().cpu()
    assert got.shape == ref.shape, "%s: shape %s vs %s" % (name, tuple(got.shape), tuple(ref.shape))
    assert torch.isfinite(got).all(), "%s: non-finite values" % name
    rms = ref.pow(2).mean().sqrt().item()
    if dtype == F32:
        bound = 2e-4 * ref.abs() + 2e-4 * max(rms, 1e-30)
    elif out_bits is None:          # fp32 output of bf16 operands
        bound = 2.0 ** -12 * (ref.abs() + rms)
    else:
        bound = 2.0 ** -out_bits * ref.abs() + 2.0 ** -(out_bits + 2) * rms
    err = (got - ref).abs()
    bad = err > bound
    if bad.any():
        idx = torch.nonzero(bad)[0].tolist()
        raise AssertionError("%s: %d/%d elements beyond the contraction-derived bound; max |diff| %.4g at rms(ref) %.4g; first at %s got %.6g ref %.6g" % (
            name, int(bad.sum()), bad.numel(), err.max().item(), rms, idx, got[tuple(idx)].item(), ref[tuple(idx)].item()))
