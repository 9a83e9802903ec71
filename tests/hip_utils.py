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
