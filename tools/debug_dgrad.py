import sys, math, ctypes; sys.path.insert(0, '.')
import torch, torch.nn.functional as F
from emrt_amd import nn as hnn, functional as Fn
from emrt_amd.runtime import ctx, F32, BF16, Tape
from tests.hip_utils import init, dev_map, host_map, host, Holder, rnd
from tests.test_gpu_kernels import CONV_CASES, run_bwd
for dtype in (F32, BF16):
  for case in CONV_CASES:
    N, H, W, Cin, Cout, k, stride, pad, bias = case
    c = init(dtype)
    g = torch.Generator().manual_seed(1)
    x = rnd(torch.randn(N, Cin, H, W, generator=g))
    conv = hnn.Conv2D(Cin, Cout, k, stride, pad, bias=bias)
    with torch.no_grad():
        conv.weight.copy_(rnd(torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)))
        if bias: conv.bias.copy_(torch.randn(Cout, generator=g))
    w_ref, b_ref = conv.weight.detach().clone(), (conv.bias.detach().clone() if bias else None)
    Holder(conv=conv).place()
    xr = x.clone().requires_grad_(True); wr = w_ref.clone().requires_grad_(True)
    br = b_ref.clone().requires_grad_(True) if bias else None
    yr = F.conv2d(xr, wr, br, stride=stride, padding=pad)
    dy = rnd(torch.randn(yr.shape, generator=g)); yr.backward(dy)
    xd = dev_map(x)
    tape = Tape(); c.tape = tape; y = conv(xd); c.tape = None; tape.watch(xd)
    e_f = (host_map(y) - yr.detach()).abs().max().item()
    dyd = dev_map(dy)
    dx, = run_bwd(tape, [(y, dyd)], [xd])
    e_d = (host_map(dx) - xr.grad).abs().max().item()
    e_w = (host(conv.weight.grad) - wr.grad).abs().max().item()
    # direct dgrad again, after the fact
    OH, OW = yr.shape[2], yr.shape[3]
    dx2 = c.empty((N, H, W, Cin))
    Fn._L().call("emrt_conv2d", Fn.P(dyd), ctypes.c_void_p(conv.gw.bwd_ptr), Fn.P(dx2), None, None, N, OH, OW, Cout, Cout, OH*OW*Cout, H, W, Cin, Cin, H*W*Cin, 0, 0, k, k, stride, pad, 1, 0, 0, c.dtype, c.stream)
    e_d2 = (host_map(dx2) - xr.grad).abs().max().item()
    print("dtype", dtype, case, "fwd %.3g dgrad(tape) %.3g dgrad(direct) %.3g wgrad %.3g | refmax y %.3g dx %.3g dw %.3g" % (e_f, e_d, e_d2, e_w, yr.abs().max().item(), xr.grad.abs().max().item(), wr.grad.abs().max().item()), flush=True)
