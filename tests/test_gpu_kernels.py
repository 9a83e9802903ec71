"""-m gpu parity tests, kernel level: every libemrt_hip.so op (through the C-ABI, via emrt_amd.functional) against the
torch-CPU expression the oracle uses for the same reference operator.  fp32: tight tolerances; bf16: inputs are
rounded through bf16 on both sides and tolerances reflect bf16 storage of the outputs.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from emrt_amd import functional as Fn          # noqa: E402
from emrt_amd import nn as hnn                  # noqa: E402
from emrt_amd.runtime import ctx, F32, BF16, Tape   # noqa: E402
from tests.hip_utils import close_gemm, init, dev_map, host_map, dev, host, rnd, Holder, close   # noqa: E402

DTYPES = [F32, BF16]


def run_bwd(tape, outs_and_grads, watched):
    for o, g in outs_and_grads:
        tape.add_grad(o, g)
    tape.backward()
    return [tape.result(w) for w in watched]


# -----------------------------------------------------------------------------------------------------------------
CONV_CASES = [
    # N, H, W, Cin, Cout, k, stride, pad, bias
    (2, 16, 16, 64, 64, 3, 1, 1, False),
    (2, 16, 16, 128, 128, 3, 2, 1, False),
    (2, 12, 20, 256, 512, 1, 1, 0, True),
    (3, 16, 16, 64, 256, 1, 2, 0, False),
    (2, 32, 32, 3, 64, 7, 2, 3, False),
    (2, 16, 16, 3, 64, 3, 1, 1, False),
    (2, 24, 24, 256, 6, 1, 1, 0, True),
    (3, 17, 19, 512, 7, 1, 1, 0, True),      # thin_bwd_kernel: ragged pixel count, 7 classes
    (2, 48, 48, 256, 6, 1, 1, 0, True),      # >= 4096 pixels: thin_fwd_kernel too (bf16), 32 lanes per pixel
    (1, 65, 64, 512, 7, 1, 1, 0, True),      # thin_fwd_kernel with 64 lanes per pixel, ragged tail of the 4-pixel groups
    (1, 80, 80, 64, 6, 1, 1, 0, False),      # thin_fwd_kernel with 8 lanes per pixel, no bias
    (1, 40, 40, 64, 192, 3, 1, 1, True),
    (2, 8, 8, 512, 128, 3, 1, 1, False),
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_fwd_dgrad_wgrad(dtype, case):
    N, H, W, Cin, Cout, k, stride, pad, bias = case
    c = init(dtype)
    g = torch.Generator().manual_seed(1)
    x = rnd(torch.randn(N, Cin, H, W, generator=g))
    conv = hnn.Conv2D(Cin, Cout, k, stride, pad, bias=bias)
    with torch.no_grad():
        conv.weight.copy_(rnd(torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)))
        if bias:
            conv.bias.copy_(torch.randn(Cout, generator=g))
    w_ref, b_ref = conv.weight.detach().clone(), (conv.bias.detach().clone() if bias else None)
    Holder(conv=conv).place()
    xr = x.clone().requires_grad_(True)
    wr = w_ref.clone().requires_grad_(True)
    br = b_ref.clone().requires_grad_(True) if bias else None
    yr = F.conv2d(xr, wr, br, stride=stride, padding=pad)
    dy = rnd(torch.randn(yr.shape, generator=g))
    yr.backward(dy)

    xd = dev_map(x)
    tape = Tape()
    c.tape = tape
    y = conv(xd)
    c.tape = None
    tape.watch(xd)
    # tolerances from what can differ (tests/hip_utils.close_gemm): the output's rounding (bf16: 8 significant bits) and the fp32 summation order
    close_gemm("conv fwd", host_map(y), yr.detach(), dtype, out_bits=8)
    dx, = run_bwd(tape, [(y, dev_map(dy))], [xd])
    close_gemm("conv dgrad", host_map(dx), xr.grad, dtype, out_bits=8)
    close_gemm("conv wgrad", host(conv.weight.grad), wr.grad, dtype)          # fp32 gradients of bf16 operands: summation order only
    if bias:
        close_gemm("conv bias grad", host(conv.bias.grad), br.grad, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv2d_views_residual_relu(dtype):
    """token-slab input view, concat-slice output view, fused bias + residual + relu epilogue, fp32 output."""
    c = init(dtype)
    g = torch.Generator().manual_seed(2)
    B, h, w, C, OC = 2, 8, 8, 64, 128
    Lv = h * w + 16
    tokens = rnd(torch.randn(B, Lv, C, generator=g))
    conv = hnn.Conv2D(C, OC, 3, 1, 1, bias=True)
    with torch.no_grad():
        conv.weight.copy_(rnd(torch.randn(OC, C, 3, 3, generator=g) / 24))
        conv.bias.copy_(torch.randn(OC, generator=g))
    wref, bref = conv.weight.detach().clone(), conv.bias.detach().clone()
    Holder(conv=conv).place()
    res = rnd(torch.randn(B, OC, h, w, generator=g))
    td = dev(tokens)
    xin = Fn.tokens_as_map(td.narrow(1, 16, h * w), h, w)
    cat = c.zeros((B, h, w, 3 * OC))
    out = cat.narrow(3, OC, OC)
    y = Fn.conv2d(xin, conv.gw, 1, 1, relu=True, residual=dev_map(res), out=out)
    xr = tokens[:, 16:].transpose(1, 2).reshape(B, C, h, w)
    yr = F.relu(F.conv2d(xr, wref, bref, padding=1) + res)
    close("conv view fwd", host_map(cat[..., OC:2 * OC].contiguous()), yr, dtype)
    assert float(cat[..., :OC].float().abs().max()) == 0 and float(cat[..., 2 * OC:].float().abs().max()) == 0
    y32 = Fn.conv2d(xin, conv.gw, 1, 1, out_f32=True)
    assert y32.dtype == torch.float32
    close("conv f32 out", host_map(y32), F.conv2d(xr, wref, bref, padding=1), dtype, atol=(2e-4 if dtype == F32 else 2e-2))


@pytest.mark.parametrize("dtype", DTYPES)
def test_linear_relu_backward(dtype):
    c = init(dtype)
    g = torch.Generator().manual_seed(3)
    B, L, C, OC = 2, 37, 256, 1024
    x = rnd(torch.randn(B, L, C, generator=g))
    lin = hnn.Linear(C, OC)
    with torch.no_grad():
        lin.weight.copy_(rnd(torch.randn(OC, C, generator=g) / 16))
        lin.bias.copy_(torch.randn(OC, generator=g) * 0.1)
    wref, bref = lin.weight.detach().clone(), lin.bias.detach().clone()
    Holder(lin=lin).place()
    xr = x.clone().requires_grad_(True)
    wr, br = wref.clone().requires_grad_(True), bref.clone().requires_grad_(True)
    yr = F.relu(F.linear(xr, wr, br))
    dy = rnd(torch.randn(yr.shape, generator=g))
    yr.backward(dy)
    xd = dev(x)
    tape = Tape()
    c.tape = tape
    y = lin(xd, relu=True)
    c.tape = None
    tape.watch(xd)
    close("linear fwd", host(y), yr.detach(), dtype)
    dx, = run_bwd(tape, [(y, dev(dy))], [xd])
    close("linear dx", host(dx), xr.grad, dtype, 8.0)
    close("linear dw", host(lin.weight.grad), wr.grad, dtype, 8.0)
    close("linear db", host(lin.bias.grad), br.grad, dtype, 8.0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("k,stride,pad", [(7, 2, 3), (3, 1, 1)])
def test_conv_on_the_padded_image(dtype, k, stride, pad):
    """Conv2D(3, 64, pad_cin=8) fed the 8-channel image map (ResNet stem, spatial_branch.Enc0: paddle_vision_resnet.py:196,
    paddle_EMRT.py:84-91): forward and weight gradient equal the 3-channel convolution; the parameter keeps its [64, 3, k, k]
    shape, the stored padding channels and their gradients stay zero."""
    c = init(dtype)
    g = torch.Generator().manual_seed(21)
    N, H, W = 2, 32, 40
    img = torch.randn(N, 3, H, W, generator=g)
    conv = hnn.Conv2D(3, 64, k, stride, pad, bias=False, need_dx=False, pad_cin=8)
    with torch.no_grad():
        conv.weight.copy_(rnd(torch.randn(64, 3, k, k, generator=g) / math.sqrt(3 * k * k)))
    w_ref = conv.weight.detach().clone()
    h = Holder(conv=conv).place()
    assert tuple(conv.weight.shape) == (64, 3, k, k) and conv.gw.C == 8
    assert torch.equal(conv.weight.detach().cpu(), w_ref)
    xr = rnd(img)
    wr = w_ref.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, stride=stride, padding=pad)
    dy = rnd(torch.randn(yr.shape, generator=g))
    yr.backward(dy)
    xd = Fn.nchw_to_nhwc(img.cuda(), c_out=8)
    tape = Tape()
    c.tape = tape
    y = conv(xd)
    c.tape = None
    close("padded conv fwd", host_map(y), yr.detach(), dtype)
    run_bwd(tape, [(y, dev_map(dy))], [])
    wscale = math.sqrt(N * yr.shape[2] * yr.shape[3])
    close("padded conv wgrad", host(conv.weight.grad), wr.grad, dtype, wscale * (1.0 if dtype == F32 else 0.3))
    a, cnt = h.store.views["conv.weight"]
    stored = h.store.master[a:a + cnt].view(64, k, k, 8)
    sgrad = h.store.grad[a:a + cnt].view(64, k, k, 8)
    assert float(stored[..., 3:].abs().max()) == 0.0 and float(sgrad[..., 3:].abs().max()) == 0.0


# -----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(4, 16, 16, 64), (2, 8, 8, 512), (8, 3, 3, 256), (2, 32, 32, 2048)])
@pytest.mark.parametrize("relu,with_res", [(True, True), (True, False), (False, False)])
def test_batch_norm_train(dtype, shape, relu, with_res):
    N, H, W, C = shape
    c = init(dtype)
    g = torch.Generator().manual_seed(4)
    x = rnd(torch.randn(N, C, H, W, generator=g) * 2 + 0.5)
    res = rnd(torch.randn(N, C, H, W, generator=g)) if with_res else None
    bn = hnn.BatchNorm2D(C)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(C, generator=g) * 0.2)
    gam, bet = bn.weight.detach().clone(), bn.bias.detach().clone()
    Holder(bn=bn).place()
    xr = x.clone().requires_grad_(True)
    rr = res.clone().requires_grad_(True) if with_res else None
    gr, br_ = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    o = F.batch_norm(xr, None, None, gr, br_, True, 0.1, 1e-5)
    if with_res:
        o = o + rr
    if relu:
        o = F.relu(o)
    dy = rnd(torch.randn(o.shape, generator=g))
    o.backward(dy)
    xd = dev_map(x)
    rd = dev_map(res) if with_res else None
    tape = Tape()
    c.tape = tape
    y = bn(xd, relu=relu, residual=rd)
    c.tape = None
    tape.watch(xd)
    if rd is not None:
        tape.watch(rd)
    close("bn fwd", host_map(y), o.detach(), dtype)
    outs = run_bwd(tape, [(y, dev_map(dy))], [xd] + ([rd] if rd is not None else []))
    close("bn dx", host_map(outs[0]), xr.grad, dtype, 2.0)
    if with_res:
        close("bn dres", host_map(outs[1]), rr.grad, dtype)
    sc = math.sqrt(N * H * W)
    close("bn dgamma", host(bn.weight.grad), gr.grad, dtype, sc)
    close("bn dbeta", host(bn.bias.grad), br_.grad, dtype, sc)
    # running statistics: Paddle convention (momentum 0.9, biased variance)
    mean = x.mean(dim=(0, 2, 3))
    var = x.var(dim=(0, 2, 3), unbiased=False)
    close("bn run_mean", host(bn._buffers["_mean"]), 0.1 * mean, F32, atol=1e-4)
    close("bn run_var", host(bn._buffers["_variance"]), 0.9 + 0.1 * var, F32, atol=1e-3)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(2, 16, 16, 64, 64, 3, 1, True), (2, 12, 12, 256, 128, 1, 0, True), (2, 16, 16, 64, 64, 3, 1, False),
                                  (2, 24, 24, 256, 6, 1, 0, True), (3, 13, 11, 64, 7, 1, 0, True), (2, 9, 9, 128, 6, 1, 0, False)])   # last three: thin_bwd_kernel (OC <= 8)
def test_bn_relu_conv_chain_backward_fused_in_dgrad(dtype, case):
    """x -> BatchNorm -> ReLU -> conv: the conv's dgrad applies the ReLU mask and accumulates the BatchNorm's backward sums
    (sum dy', sum dy'*y; xhat = (y - beta) / gamma where y > 0), so emrt_bn_bwd_reduce is not launched.  With a second
    consumer of the BatchNorm output (sole=False) the separate reduction must be used and give the same answer."""
    from emrt_amd import _lib
    N, H, W, C, OC, k, pad, sole = case
    c = init(dtype)
    g = torch.Generator().manual_seed(31)
    x = rnd(torch.randn(N, C, H, W, generator=g) * 1.5 + 0.2)
    bn = hnn.BatchNorm2D(C)
    conv = hnn.Conv2D(C, OC, k, 1, pad, bias=False)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(C, generator=g) * 0.3)
        conv.weight.copy_(rnd(torch.randn(OC, C, k, k, generator=g) / math.sqrt(C * k * k)))
    gam, bet, wref = bn.weight.detach().clone(), bn.bias.detach().clone(), conv.weight.detach().clone()
    Holder(bn=bn, conv=conv).place()
    xr = x.clone().requires_grad_(True)
    gr, br_, wr = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True), wref.clone().requires_grad_(True)
    yb = F.relu(F.batch_norm(xr, None, None, gr, br_, True, 0.1, 1e-5))
    o = F.conv2d(yb, wr, None, padding=pad)
    dy = rnd(torch.randn(o.shape, generator=g))
    extra = rnd(torch.randn(yb.shape, generator=g)) if not sole else None
    loss = (o * dy).sum() + ((yb * extra).sum() if extra is not None else 0.0)
    loss.backward()

    xd = dev_map(x)
    tape = Tape()
    c.tape = tape
    yd = bn(xd, relu=True)
    od = conv(yd)
    c.tape = None
    tape.watch(xd)
    L = _lib.lib()
    L.start_record()
    grads = [(od, dev_map(dy))] + ([(yd, dev_map(extra))] if extra is not None else [])
    dx, = run_bwd(tape, grads, [xd])
    names = [n for n, _ in L.stop_record()]
    assert ("emrt_bn_bwd_reduce" in names) == (not sole), names
    sc = math.sqrt(N * H * W)
    ksc = math.sqrt(OC * k * k)
    close("chain dx", host_map(dx), xr.grad, dtype, 2.0 * ksc * (1.0 if dtype == F32 else 0.3))
    close("chain dgamma", host(bn.weight.grad), gr.grad, dtype, sc * ksc * (1.0 if dtype == F32 else 0.5))
    close("chain dbeta", host(bn.bias.grad), br_.grad, dtype, sc * ksc * (1.0 if dtype == F32 else 0.5))
    close("chain dw", host(conv.weight.grad), wr.grad, dtype, sc * (1.0 if dtype == F32 else 0.3))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("scenario", ["addend", "two_convs", "late_add"])
def test_residual_join_backward_fused_into_consumer_dgrad(dtype, scenario):
    """out = relu(BatchNorm(x) + r) with several consumers (a ResNet block boundary).  The dgrad of a conv that consumes
    `out` folds the contributions made so far in (addend / in place), masks with out > 0 and accumulates the join's
    BatchNorm sums against x; when it was the last contribution the join's backward skips its reduction pass.
      addend    : an external gradient (the next block's identity path) arrives first, then the conv's   -> fused, addend
      two_convs : two convs consume `out`; the second to run accumulates in place                        -> fused
      late_add  : a non-conv consumer contributes AFTER the conv                                         -> falls back, same numbers"""
    from emrt_amd import _lib
    N, H, W, C, OC = 2, 12, 12, 64, 128
    c = init(dtype)
    g = torch.Generator().manual_seed(41)
    x = rnd(torch.randn(N, C, H, W, generator=g) * 1.3 + 0.1)
    r = rnd(torch.randn(N, C, H, W, generator=g))
    other = rnd(torch.randn(N, C, H, W, generator=g))
    bn = hnn.BatchNorm2D(C)
    conv_b = hnn.Conv2D(C, OC, 1, 1, 0, bias=False)
    conv_c = hnn.Conv2D(C, OC, 3, 1, 1, bias=False)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(C, generator=g) * 0.3)
        conv_b.weight.copy_(rnd(torch.randn(OC, C, 1, 1, generator=g) / math.sqrt(C)))
        conv_c.weight.copy_(rnd(torch.randn(OC, C, 3, 3, generator=g) / math.sqrt(9 * C)))
    gam, bet, wb, wc = (t.detach().clone() for t in (bn.weight, bn.bias, conv_b.weight, conv_c.weight))
    Holder(bn=bn, conv_b=conv_b, conv_c=conv_c).place()
    dyb = rnd(torch.randn(N, OC, H, W, generator=g))
    dyc = rnd(torch.randn(N, OC, H, W, generator=g))
    extra = rnd(torch.randn(N, C, H, W, generator=g))

    xr, rr = x.clone().requires_grad_(True), r.clone().requires_grad_(True)
    gr, br_, wbr, wcr = (t.clone().requires_grad_(True) for t in (gam, bet, wb, wc))
    out_r = F.relu(F.batch_norm(xr, None, None, gr, br_, True, 0.1, 1e-5) + rr)
    loss = (F.conv2d(out_r, wbr) * dyb).sum()
    if scenario == "addend":
        loss = loss + (out_r * extra).sum()
    elif scenario == "two_convs":
        loss = loss + (F.conv2d(out_r, wcr, padding=1) * dyc).sum()
    else:
        loss = loss + ((out_r + other) * extra).sum()
    loss.backward()

    xd, rd = dev_map(x), dev_map(r)
    tape = Tape()
    c.tape = tape
    out = bn(xd, relu=True, residual=rd)
    grads = []
    if scenario == "late_add":
        s_ = Fn.add(out, dev_map(other))            # recorded before conv_b => its gradient reaches `out` after conv_b's
        grads.append((s_, dev_map(extra)))
    if scenario == "two_convs":
        oc = conv_c(out)
        grads.append((oc, dev_map(dyc)))
    ob = conv_b(out)
    grads.append((ob, dev_map(dyb)))
    if scenario == "addend":
        grads.append((out, dev_map(extra)))
    c.tape = None
    tape.watch(xd)
    tape.watch(rd)
    L = _lib.lib()
    L.start_record()
    dx, dr = run_bwd(tape, grads, [xd, rd])
    calls = L.stop_record()
    names = [n for n, _ in calls]
    assert ("emrt_bn_bwd_reduce" in names) == (scenario == "late_add"), names
    bwd = [a for n, a in calls if n == "emrt_conv2d_bwd"]
    assert all(a[29] is not None for a in bwd)                       # every consumer conv tries the join fusion
    if scenario == "addend":
        assert bwd[0][32] is not None and "emrt_acc3d" not in names and "emrt_add" not in names
    if scenario == "two_convs":
        assert bwd[0][6] == 0 and bwd[1][6] == 1
    sc = math.sqrt(N * H * W)
    ksc = math.sqrt(OC) * (2.0 if scenario == "two_convs" else 1.0)
    f = 1.0 if dtype == F32 else 0.3
    close("join dx", host_map(dx), xr.grad, dtype, 3.0 * ksc * f)
    close("join dres", host_map(dr), rr.grad, dtype, 2.0 * ksc * f)
    close("join dgamma", host(bn.weight.grad), gr.grad, dtype, sc * ksc * (1.0 if dtype == F32 else 0.5))
    close("join dbeta", host(bn.bias.grad), br_.grad, dtype, sc * ksc * (1.0 if dtype == F32 else 0.5))
    close("join dw_b", host(conv_b.weight.grad), wbr.grad, dtype, sc * f)


@pytest.mark.parametrize("dtype", DTYPES)
def test_batch_norm_eval_and_slice_output(dtype):
    c = init(dtype)
    c.training = False
    g = torch.Generator().manual_seed(5)
    N, H, W, C = 2, 8, 8, 256
    x = rnd(torch.randn(N, C, H, W, generator=g))
    bn = hnn.BatchNorm2D(C)
    with torch.no_grad():
        bn._mean.copy_(torch.randn(C, generator=g) * 0.3)
        bn._variance.copy_(torch.rand(C, generator=g) + 0.5)
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(C, generator=g))
    rm, rv, gam, bet = bn._mean.clone(), bn._variance.clone(), bn.weight.detach().clone(), bn.bias.detach().clone()
    Holder(bn=bn).place()
    cat = c.zeros((N, H, W, 3 * C))
    bn(dev_map(x), relu=True, out=cat.narrow(3, C, C))
    ref = F.relu(F.batch_norm(x, rm, rv, gam, bet, False, 0.1, 1e-5))
    close("bn eval slice", host_map(cat[..., C:2 * C].contiguous()), ref, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("gelu,with_res", [(True, True), (False, False)])
def test_group_norm(dtype, gelu, with_res):
    c = init(dtype)
    g = torch.Generator().manual_seed(6)
    N, H, W, C = 3, 16, 16, 256
    x = rnd(torch.randn(N, C, H, W, generator=g) * 1.5 + 0.3)
    res = rnd(torch.randn(N, C, H, W, generator=g)) if with_res else None
    gn = hnn.GroupNorm(32, C)
    with torch.no_grad():
        gn.weight.copy_(torch.rand(C, generator=g) + 0.5)
        gn.bias.copy_(torch.randn(C, generator=g) * 0.3)
    gam, bet = gn.weight.detach().clone(), gn.bias.detach().clone()
    Holder(gn=gn).place()
    xr = x.clone().requires_grad_(True)
    gr, br_ = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    o = F.group_norm(xr, 32, gr, br_, 1e-5)
    if gelu:
        o = F.gelu(o)
    rr = None
    if with_res:
        rr = res.clone().requires_grad_(True)
        o = o + rr
    dy = rnd(torch.randn(o.shape, generator=g))
    o.backward(dy)
    xd = dev_map(x)
    rd = dev_map(res) if with_res else None
    tape = Tape()
    c.tape = tape
    y = gn(xd, gelu=gelu, residual=rd)
    c.tape = None
    tape.watch(xd)
    if rd is not None:
        tape.watch(rd)
    close("gn fwd", host_map(y), o.detach(), dtype)
    outs = run_bwd(tape, [(y, dev_map(dy))], [xd] + ([rd] if rd is not None else []))
    close("gn dx", host_map(outs[0]), xr.grad, dtype, 2.0)
    if with_res:
        close("gn dres", host_map(outs[1]), rr.grad, dtype)
    sc = math.sqrt(N * H * W)
    close("gn dgamma", host(gn.weight.grad), gr.grad, dtype, sc)
    close("gn dbeta", host(gn.bias.grad), br_.grad, dtype, sc)


@pytest.mark.parametrize("dtype", DTYPES)
def test_layer_norm_residual_post(dtype):
    c = init(dtype)
    g = torch.Generator().manual_seed(7)
    B, L, C = 3, 113, 256
    a, b, post = (rnd(torch.randn(B, L, C, generator=g)) for _ in range(3))
    ln = hnn.LayerNorm(C)
    with torch.no_grad():
        ln.weight.copy_(torch.rand(C, generator=g) + 0.5)
        ln.bias.copy_(torch.randn(C, generator=g) * 0.3)
    gam, bet = ln.weight.detach().clone(), ln.bias.detach().clone()
    Holder(ln=ln).place()
    ar, br_, pr = (t.clone().requires_grad_(True) for t in (a, b, post))
    gr, ber = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    zr = rnd((ar + br_).detach()) if dtype == BF16 else None
    o = F.layer_norm(ar + br_, (C,), gr, ber, 1e-5) + pr
    dy = rnd(torch.randn(o.shape, generator=g))
    o.backward(dy)
    ad, bd, pd = dev(a), dev(b), dev(post)
    tape = Tape()
    c.tape = tape
    y = ln(ad, bd, post=pd)
    c.tape = None
    for t in (ad, bd, pd):
        tape.watch(t)
    close("ln fwd", host(y), o.detach(), dtype)
    da, db, dp = run_bwd(tape, [(y, dev(dy))], [ad, bd, pd])
    close("ln da", host(da), ar.grad, dtype, 2.0)
    close("ln db", host(db), br_.grad, dtype, 2.0)
    close("ln dpost", host(dp), pr.grad, dtype)
    sc = math.sqrt(B * L)
    close("ln dgamma", host(ln.weight.grad), gr.grad, dtype, sc)
    close("ln dbeta", host(ln.bias.grad), ber.grad, dtype, sc)


@pytest.mark.parametrize("dtype", DTYPES)
def test_layer_norm_with_fused_branch_dropout(dtype):
    """LN(a + dropout(b)) with the dropout inside the LayerNorm kernels must equal the composition of the stand-alone
    dropout kernel (same device seed and salt => same mask) and the plain LayerNorm, forward and backward."""
    c = init(dtype)
    c.training = True
    g = torch.Generator().manual_seed(8)
    B, L, C = 2, 57, 256
    a, b = (rnd(torch.randn(B, L, C, generator=g)) for _ in range(2))
    dy = rnd(torch.randn(B, L, C, generator=g))
    res = {}
    for mode in ("fused", "composed"):
        ln = hnn.LayerNorm(C)
        with torch.no_grad():
            ln.weight.copy_(torch.linspace(0.5, 1.5, C))
            ln.bias.copy_(torch.linspace(-0.2, 0.2, C))
        Holder(ln=ln).place()
        ad, bd = dev(a), dev(b)
        tape = Tape()
        c.tape = tape
        y = ln(ad, bd, drop_p=0.3, drop_salt=11) if mode == "fused" else ln(ad, Fn.dropout(bd, 0.3, 11))
        c.tape = None
        tape.watch(ad)
        tape.watch(bd)
        da, db = run_bwd(tape, [(y, dev(dy))], [ad, bd])
        res[mode] = [host(t) for t in (y, da, db)] + [host(ln.weight.grad), host(ln.bias.grad)]
    assert (res["fused"][2] == 0).float().mean() > 0.2                       # the branch gradient really is masked
    for u, v, name in zip(res["fused"], res["composed"], ("y", "da", "db", "dgamma", "dbeta")):
        assert (u - v).abs().max().item() <= 1e-6 * max(1.0, v.abs().max().item()), name


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("with_post", [False, True])
def test_layer_norm_writes_the_next_query_and_sums_its_gradient(dtype, with_post):
    """with_pos_embed(LN(...), pos) (transformer_encoder_decoder.py:186,283-289) from the LayerNorm launch itself: layer_norm(q_pos=) returns (out, q)
    with q = out + pos broadcast over the batch, bit-identical to the separate add launch (the stored out, rounded, plus pos); backward sums the two
    output gradients as it loads them (dy2), hands q's gradient to the embedding's callback and gives `post` the sum (dysum).  Against the two-launch
    form: the same gradients (fp32: to the last bits of one reassociated add; bf16: one rounding fewer), one add and one accumulate launch less."""
    from emrt_amd import _lib
    c = init(dtype)
    c.training = True
    g = torch.Generator().manual_seed(19)
    B, L, C = 3, 110, 256
    a, b, post = (rnd(torch.randn(B, L, C, generator=g)) for _ in range(3))
    pos = rnd(torch.randn(L, C, generator=g))
    dy1, dyq = (rnd(torch.randn(B, L, C, generator=g)) for _ in range(2))
    res, calls, seen = {}, {}, {}
    for mode in ("fused", "plain"):
        ln = hnn.LayerNorm(C)
        with torch.no_grad():
            ln.weight.copy_(torch.linspace(0.5, 1.5, C))
            ln.bias.copy_(torch.linspace(-0.2, 0.2, C))
        Holder(ln=ln).place()
        ad, bd, pd, posd = dev(a), dev(b), dev(post) if with_post else None, dev(pos)
        got = []
        tape = Tape()
        c.tape = tape
        _lib.lib().start_record()
        if mode == "fused":
            y, q = ln(ad, bd, post=pd, drop_p=0.2, drop_salt=5, q_pos=posd, q_bgrad=lambda gq: got.append(gq))
        else:
            y = ln(ad, bd, post=pd, drop_p=0.2, drop_salt=5)
            q = Fn.add(y, posd, period=L * C, bgrad=lambda gq: got.append(gq))
        fwd = [n for n, _ in _lib.lib().stop_record()]
        c.tape = None
        watch = [ad, bd] + ([pd] if with_post else [])
        for t in watch:
            tape.watch(t)
        _lib.lib().start_record()
        grads = run_bwd(tape, [(y, dev(dy1)), (q, dev(dyq))], watch)
        calls[mode] = fwd + [n for n, _ in _lib.lib().stop_record()]
        assert len(got) == 1
        res[mode] = [host(y), host(q)] + [host(t) for t in grads] + [host(ln.weight.grad), host(ln.bias.grad), host(got[0])]
    assert calls["fused"] == ["emrt_layernorm_fwd", "emrt_layernorm_bwd"], calls["fused"]
    assert calls["plain"].count("emrt_add") == 1 and calls["plain"].count("emrt_acc3d") + calls["plain"].count("emrt_add3d") == 1, calls["plain"]
    names = ["out", "q", "da", "db"] + (["dpost"] if with_post else []) + ["dgamma", "dbeta", "dq handed to the embedding"]
    for u, v, name in zip(res["fused"], res["plain"], names):
        if name in ("out", "q", "dq handed to the embedding"):
            assert torch.equal(u, v), name
        else:
            tol = 2e-2 if dtype == BF16 else 2e-6
            assert (u - v).abs().max().item() <= tol * max(1.0, v.abs().max().item()), (name, (u - v).abs().max().item())
    want_q = rnd(res["plain"][0]) + pos          # q really is out + pos
    close("q = out + pos", res["fused"][1], want_q, dtype, atol=1e-6 if dtype == F32 else None)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("p", [0.0, 0.3])
def test_layer_norm_backward_sums_an_identity_contribution(dtype, p):
    """The encoder layer's shape (transformer_encoder_decoder.py:184-204): t = f(a) + a is computed first, LN1(a + dropout(b)) second, and t is consumed
    AFTER it (as norm2's `post`).  layer_norm(identity_from=t) lets LN1's backward kernel add d t into d a (dz_addend) and t's producer skip its own
    accumulate: same d a as the two-launch form (one rounding fewer in bf16), d b untouched by the addend, and the accumulate launch is gone."""
    from emrt_amd import _lib
    c = init(dtype)
    c.training = True
    g = torch.Generator().manual_seed(9)
    B, L, C = 2, 77, 256
    a, b, w = (rnd(torch.randn(B, L, C, generator=g)) for _ in range(3))
    dy1, dyt = (rnd(torch.randn(B, L, C, generator=g)) for _ in range(2))
    res, calls = {}, {}
    for mode in ("fused", "plain"):
        ln = hnn.LayerNorm(C)
        with torch.no_grad():
            ln.weight.copy_(torch.linspace(0.5, 1.5, C))
            ln.bias.copy_(torch.linspace(-0.2, 0.2, C))
        Holder(ln=ln).place()
        ad, bd, wd = dev(a), dev(b), dev(w)
        tape = Tape()
        c.tape = tape
        t = Fn.add_maps(ad, wd)                      # stands for level_conv_gn: an earlier consumer of a with an identity path

        def t_bwd(t=t, ad=ad, tape=tape):            # its producer's backward, written like level_conv_gn's: skip the identity when a later consumer summed it
            dt, dt_n = tape.pop_grad(t, with_count=True)
            if id(t) in tape.identity_done:
                summed, summed_n = tape.identity_done.pop(id(t))
                assert summed is dt and summed_n == dt_n
            else:
                tape.add_grad(ad, dt)
        tape.ops[-1] = t_bwd                         # (replaces add_maps' own backward)
        y = ln(ad, bd, drop_p=p, drop_salt=5, identity_from=t if mode == "fused" else None)
        c.tape = None
        tape.watch(ad)
        tape.watch(bd)
        _lib.lib().start_record()
        da, db = run_bwd(tape, [(t, dev(dyt)), (y, dev(dy1))], [ad, bd])
        calls[mode] = [n for n, _ in _lib.lib().stop_record()]
        res[mode] = [host(da), host(db)]
    # (without branch dropout a and b share ONE gradient tensor, so the addend cannot be folded in: nothing changes then)
    assert calls["plain"].count("emrt_acc3d") + calls["plain"].count("emrt_add3d") == calls["fused"].count("emrt_acc3d") + calls["fused"].count("emrt_add3d") + (1 if p > 0 else 0), calls
    tol = 2e-2 if dtype == BF16 else 1e-6            # bf16: d a is rounded once (fused) instead of twice
    assert (res["fused"][0] - res["plain"][0]).abs().max().item() <= tol * max(1.0, res["plain"][0].abs().max().item())
    assert torch.equal(res["fused"][1], res["plain"][1]), "the branch gradient must not see the addend"


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("k2", [1, 3])
def test_channel_dropout_mask_fused_into_the_consumers_dgrad(dtype, k2):
    """conv_k2(Dropout2D(relu(conv3x3(x)))) as in the two heads (fcn_head.py:62-66, paddle_EMRT.py:209-213: nn.Dropout2D in front of the classifier /
    UpHead): with sole_consumer_is_linear=True the consumer's data gradient applies the channel mask (stored value > 0, scale 1 / (1 - p)) and no
    emrt_mask_bwd runs for the dropout; must equal the composed form (same seed and salt => same mask) in outputs and every gradient."""
    from emrt_amd import _lib
    c = init(dtype)
    c.training = True
    p = 0.25
    g = torch.Generator().manual_seed(79)
    N, H, W, C, Cm, Co = 3, 12, 10, 32, 64, (8 if k2 == 1 else 32)
    x = rnd(torch.randn(N, H, W, C, generator=g))
    dy = rnd(torch.randn(N, H, W, Co, generator=g))
    w1 = rnd(torch.randn(Cm, C, 3, 3, generator=g) / math.sqrt(9 * C))
    w2 = rnd(torch.randn(Co, Cm, k2, k2, generator=g) / math.sqrt(k2 * k2 * Cm))
    res, names = {}, {}
    for mode in ("fused", "composed"):
        c1, c2 = hnn.Conv2D(C, Cm, 3, 1, 1, bias=False), hnn.Conv2D(Cm, Co, k2, 1, k2 // 2)
        with torch.no_grad():
            c1.weight.copy_(w1)
            c2.weight.copy_(w2)
            c2.bias.zero_()
        Holder(c1=c1, c2=c2).place()
        xd = dev(x)
        tape = Tape()
        c.tape = tape
        h = Fn.conv2d(xd, c1.gw, 1, 1, relu=True)
        hd = Fn.dropout(h, p, 29, mode=1, hw=H * W, sole_consumer_is_linear=(mode == "fused"))
        o = c2(hd)
        c.tape = None
        tape.watch(xd)
        L = _lib.lib()
        L.start_record()
        dx, = run_bwd(tape, [(o, dev(dy))], [xd])
        names[mode] = [n for n, _ in L.stop_record()]
        res[mode] = [host(t) for t in (o, dx, c1.weight.grad, c2.weight.grad, c2.bias.grad)] + [host(h), host(hd)]
    assert names["fused"].count("emrt_mask_bwd") == 0 and names["composed"].count("emrt_mask_bwd") == 2, names      # (dropout mask + conv1's ReLU mask)
    hh, hdd = res["fused"][5], res["fused"][6]
    dropped = ((hdd == 0) & (hh > 0)).float().sum((1, 2))                  # per (image, channel): all of its positive pixels or none
    positive = (hh > 0).float().sum((1, 2))
    assert torch.all((dropped == 0) | (dropped == positive)) and 0.1 < (dropped > 0).float().mean() < 0.4
    for u, v, name in zip(res["fused"][:5], res["composed"][:5], ("o", "dx", "dw1", "dw2", "db2")):
        close("channel dropout fused vs composed " + name, u, v, dtype, max(1.0, v.abs().max().item()) * (0.01 if dtype == F32 else 1.0))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("dims", [(2, 57, 256, 1024), (3, 110, 64, 96)])
def test_ffn_dropout_relu_masks_fused_into_linear2_dgrad(dtype, dims):
    """linear2(dropout(relu(linear1(x)))) with sole_consumer_is_linear=True: linear2's dgrad applies both masks (one test
    on its input, scale 1/(1-p)) and no emrt_mask_bwd runs.  Reference: torch autograd with the mask read back from the
    device's own dropout output, and the unfused composition on the device (identical seed => identical mask)."""
    from emrt_amd import _lib
    B, Lq, C, Hd = dims
    c = init(dtype)
    c.training = True
    p = 0.3
    g = torch.Generator().manual_seed(77)
    x = rnd(torch.randn(B, Lq, C, generator=g))
    dy = rnd(torch.randn(B, Lq, C, generator=g))
    w1, b1 = rnd(torch.randn(Hd, C, generator=g) / math.sqrt(C)), torch.randn(Hd, generator=g) * 0.1
    w2, b2 = rnd(torch.randn(C, Hd, generator=g) / math.sqrt(Hd)), torch.randn(C, generator=g) * 0.1
    res, names = {}, {}
    for mode in ("fused", "composed"):
        l1, l2 = hnn.Linear(C, Hd), hnn.Linear(Hd, C)
        with torch.no_grad():
            l1.weight.copy_(w1)
            l1.bias.copy_(b1)
            l2.weight.copy_(w2)
            l2.bias.copy_(b2)
        Holder(l1=l1, l2=l2).place()
        xd = dev(x)
        tape = Tape()
        c.tape = tape
        h = l1(xd, relu=True)
        hd = Fn.dropout(h, p, 23, sole_consumer_is_linear=(mode == "fused"))
        o = l2(hd)
        c.tape = None
        tape.watch(xd)
        L = _lib.lib()
        L.start_record()
        dx, = run_bwd(tape, [(o, dev(dy))], [xd])
        names[mode] = [n for n, _ in L.stop_record()]
        res[mode] = [host(t) for t in (o, dx, l1.weight.grad, l1.bias.grad, l2.weight.grad, l2.bias.grad)] + [host(h), host(hd)]
    assert "emrt_mask_bwd" not in names["fused"] and names["composed"].count("emrt_mask_bwd") == 2
    for u, v, name in zip(res["fused"][:6], res["composed"][:6], ("o", "dx", "dw1", "db1", "dw2", "db2")):
        close("ffn fused vs composed " + name, u, v, dtype, max(1.0, v.abs().max().item()) * (0.01 if dtype == F32 else 1.0))
    # torch reference with the device's mask
    hh, hdd = res["fused"][6], res["fused"][7]
    keep = ((hdd != 0) | (hh <= 0)).float()
    assert 0.6 < keep[hh > 0].mean() < 0.8
    xr = x.clone().requires_grad_(True)
    W1 = w1.clone().requires_grad_(True)
    B1, W2, B2 = b1.clone().requires_grad_(True), w2.clone().requires_grad_(True), b2.clone().requires_grad_(True)
    hr = F.relu(F.linear(xr, W1, B1))
    if dtype == BF16:
        hr = hr + (rnd(hr.detach()) - hr.detach())            # the device stores h in bf16 before the dropout scales it
    orf = F.linear(hr * keep / (1 - p), W2, B2)
    (orf * dy).sum().backward()
    sc = math.sqrt(B * Lq)
    close("ffn o", res["fused"][0], orf.detach(), dtype, 2.0)
    close("ffn dx", res["fused"][1], xr.grad, dtype, 3.0 * (1.0 if dtype == F32 else 0.3))
    close("ffn db1", res["fused"][3], B1.grad, dtype, sc * (1.0 if dtype == F32 else 0.5))
    close("ffn db2", res["fused"][5], B2.grad, dtype, sc)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("dims", [(2, 57, 256, 1024), (3, 110, 64, 96), (8, 1344, 256, 1024)])
def test_ffn_dropout_drawn_in_linear1_epilogue(dtype, dims):
    """linear2(dropout(relu(linear1(x)))) with the dropout drawn in linear1's GEMM epilogue (emrt_conv2d_drop; reference: nn.Linear ->
    F.relu -> nn.Dropout -> nn.Linear, transformer_encoder_decoder.py:157-161): ONE forward launch for the first half, no dropout launch, no
    mask launch in backward.  Checked: every stored activation is 0 or relu(linear1) / (1 - p); the dropped fraction among the positive ones
    is p (binomial bounds), evenly over rows, channels and the 8 lanes of a mask group; a second seed gives another mask; forward and every
    gradient equal torch autograd with the device's own mask read back from the stored activation."""
    from emrt_amd import _lib
    B, Lq, C, Hd = dims
    c = init(dtype)
    c.training = True
    p = 0.1
    g = torch.Generator().manual_seed(78)
    x = rnd(torch.randn(B, Lq, C, generator=g))
    dy = rnd(torch.randn(B, Lq, C, generator=g))
    w1, b1 = rnd(torch.randn(Hd, C, generator=g) / math.sqrt(C)), torch.randn(Hd, generator=g) * 0.1
    w2, b2 = rnd(torch.randn(C, Hd, generator=g) / math.sqrt(Hd)), torch.randn(C, generator=g) * 0.1
    l1, l2 = hnn.Linear(C, Hd), hnn.Linear(Hd, C)
    with torch.no_grad():
        l1.weight.copy_(w1)
        l1.bias.copy_(b1)
        l2.weight.copy_(w2)
        l2.bias.copy_(b2)
    Holder(l1=l1, l2=l2).place()
    xd = dev(x)
    L = _lib.lib()
    tape = Tape()
    c.tape = tape
    L.start_record()
    hd = l1(xd, relu=True, drop=(p, 23))
    o = l2(hd)
    fwd_names = [n for n, _ in L.stop_record()]
    c.tape = None
    tape.watch(xd)
    assert fwd_names == ["emrt_conv2d_drop", "emrt_conv2d"], fwd_names
    L.start_record()
    dx, = run_bwd(tape, [(o, dev(dy))], [xd])
    bwd_names = [n for n, _ in L.stop_record()]
    assert "emrt_mask_bwd" not in bwd_names and "emrt_dropout_fwd" not in bwd_names
    hdd = host(hd)
    # the mask, read back: a stored value is positive iff it was kept and past the ReLU
    xr = x.clone().requires_grad_(True)
    W1 = w1.clone().requires_grad_(True)
    B1, W2, B2 = b1.clone().requires_grad_(True), w2.clone().requires_grad_(True), b2.clone().requires_grad_(True)
    hr = F.relu(F.linear(xr, W1, B1))
    pos = hr.detach() > (1e-2 if dtype == BF16 else 1e-4)          # (clear of the ReLU's edge: rounding may put a tiny value on either side)
    keep = (hdd > 0).float()
    dropped = 1.0 - keep[pos].mean().item()
    n_pos = int(pos.sum())
    sd = 5.0 * math.sqrt(p * (1 - p) / n_pos)
    print("dropout in the GEMM epilogue %s: %.4f of %d positive activations dropped (p = %.2f, 5 sigma = %.4f)" % (dims, dropped, n_pos, p, sd))
    assert abs(dropped - p) < sd + 2e-4          # (+ the 16-bit threshold's resolution)
    k3 = keep.reshape(-1, Hd)
    p3 = pos.reshape(-1, Hd)
    for lane in range(8):                            # the eight 16-bit fields of a mask group
        sel = p3[:, lane::8]
        frac = 1.0 - k3[:, lane::8][sel].mean().item()
        assert abs(frac - p) < 5.0 * math.sqrt(p * (1 - p) / max(1, int(sel.sum()))) + 2e-4, (lane, frac)
    rows = (1.0 - (k3 * p3).sum(1) / p3.sum(1).clamp(min=1))
    assert rows[p3.sum(1) > 16].std().item() < 3.0 * math.sqrt(p * (1 - p) / max(8.0, p3.sum(1).float().mean().item())) + 0.02
    scale = 1.0 / (1.0 - p)
    want_h = hr.detach() * keep * scale
    close("ffn hidden = relu(linear1) * mask / (1 - p)", hdd, want_h, dtype, 2.0)
    if dtype == BF16:
        hr = hr + (rnd(hr.detach() * scale) / scale - hr.detach())         # the device rounds the SCALED activation to bf16 once
    orf = F.linear(hr * keep * scale, W2, B2)
    (orf * dy).sum().backward()
    sc = math.sqrt(B * Lq)
    close("ffn o", host(o), orf.detach(), dtype, 2.0)
    close("ffn dx", host(dx), xr.grad, dtype, 3.0 * (1.0 if dtype == F32 else 0.3))
    close("ffn dw1", host(l1.weight.grad), W1.grad, dtype, sc * (1.0 if dtype == F32 else 0.5))
    close("ffn db1", host(l1.bias.grad), B1.grad, dtype, sc * (1.0 if dtype == F32 else 0.5))
    close("ffn dw2", host(l2.weight.grad), W2.grad, dtype, sc)
    close("ffn db2", host(l2.bias.grad), B2.grad, dtype, sc)
    # another step (the device seed advances): another mask
    L.call("emrt_counter_add", Fn.P(c._seed), 0x2545F4914F6CDD1D & 0x7FFFFFFFFFFFFFFF, c.stream)
    hd2 = host(l1(xd, relu=True, drop=(p, 23)))
    differ = ((hd2 > 0) != (hdd > 0)).float().mean().item()
    assert 0.5 * 2 * p * (1 - p) * pos.float().mean().item() < differ < 1.5 * 2 * p * (1 - p) + 0.01, differ
    # a gradient that does NOT come from a masking consumer (the output is used directly): the layer masks and scales it itself
    tape = Tape()
    c.tape = tape
    hd3 = l1(xd, relu=True, drop=(p, 23))
    c.tape = None
    tape.watch(xd)
    gdy = rnd(torch.randn(B, Lq, Hd, generator=g))
    l1.weight.grad.zero_()
    L.start_record()
    dx3, = run_bwd(tape, [(hd3, dev(gdy))], [xd])
    assert "emrt_mask_bwd" in [n for n, _ in L.stop_record()]
    keep3 = (host(hd3) > 0).float()
    xr3 = x.clone().requires_grad_(True)
    (F.relu(F.linear(xr3, w1, b1)) * keep3 * scale * gdy).sum().backward()
    close("dropout(relu(linear1)) backward on its own", host(dx3), xr3.grad, dtype, 3.0 * (1.0 if dtype == F32 else 0.3))
    # the A/B knob's other arm (Context.fuse_ffn_dropout = False: linear1, then the stand-alone dropout launch) must still run linear1's OWN backward
    # (round 5: the first version returned before recording it -- a faster, wrong step)
    c.fuse_ffn_dropout = False
    try:
        l1.weight.grad.zero_()
        tape = Tape()
        c.tape = tape
        L.start_record()
        hd4 = l1(xd, relu=True, drop=(p, 23))
        o4 = l2(hd4)
        assert [n for n, _ in L.stop_record()] == ["emrt_conv2d", "emrt_dropout_fwd", "emrt_conv2d"]
        c.tape = None
        tape.watch(xd)
        dx4, = run_bwd(tape, [(o4, dev(dy))], [xd])
        keep4 = (host(hd4) > 0).float()
        xr4 = x.clone().requires_grad_(True)
        W14 = w1.clone().requires_grad_(True)
        (F.linear(F.relu(F.linear(xr4, W14, b1)) * keep4 * scale, w2, b2) * dy).sum().backward()
        close("unfused arm dx", host(dx4), xr4.grad, dtype, 3.0 * (1.0 if dtype == F32 else 0.3))
        close("unfused arm dw1", host(l1.weight.grad), W14.grad, dtype, sc * (1.0 if dtype == F32 else 0.5))
    finally:
        c.fuse_ffn_dropout = True


# -----------------------------------------------------------------------------------------------------------------
def _msda_ref(value, offw, ref, shapes, M, L, Pn):
    from oracle.emrt_torch import deformable_attention_core_func
    B, Lq = offw.shape[:2]
    tp = M * L * Pn
    off = offw[..., :2 * tp].reshape(B, Lq, M, L, Pn, 2)
    aw = torch.softmax(offw[..., 2 * tp:3 * tp].reshape(B, Lq, M, L * Pn), -1).reshape(B, Lq, M, L, Pn)
    norm = torch.tensor([[w, h] for h, w in shapes], dtype=torch.float32).reshape(1, 1, 1, L, 1, 2)
    rl = ref.expand(B, Lq, L if ref.shape[2] == L else 1, 2)
    if rl.shape[2] == 1:
        rl = rl.expand(B, Lq, L, 2)
    loc = rl.reshape(B, Lq, 1, L, 1, 2) + off / norm
    return deformable_attention_core_func(value.reshape(B, -1, M, 32), shapes, loc, aw)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [dict(B=2, Lq=None, shapes=[(8, 8), (4, 4), (2, 2)], refL=1, shared=True),
                                 dict(B=3, Lq=37, shapes=[(16, 12), (8, 6), (4, 3)], refL=1, shared=True),
                                 dict(B=2, Lq=21, shapes=[(8, 8), (4, 4), (2, 2)], refL=3, shared=False),
                                 # the decoder's cross-attention at the benchmark size (B*M*Lq = 7040: LDS-staged kernels in bf16,
                                 # reference-point gradient by atomics over the heads), per-level and shared reference points
                                 dict(B=8, Lq=110, shapes=[(32, 32), (16, 16), (8, 8)], refL=3, shared=False),
                                 dict(B=8, Lq=110, shapes=[(32, 32), (16, 16), (8, 8)], refL=1, shared=True)])
def test_msda_fwd_bwd(dtype, cfg):
    c = init(dtype)
    g = torch.Generator().manual_seed(8)
    M, L, Pn = 8, 3, 6
    shapes = cfg["shapes"]
    Lv = sum(h * w for h, w in shapes)
    B = cfg["B"]
    Lq = cfg["Lq"] or Lv
    tp = M * L * Pn
    value = rnd(torch.randn(B, Lv, M * 32, generator=g))
    offw = torch.cat([torch.randn(B, Lq, 2 * tp, generator=g) * 2.0, torch.randn(B, Lq, tp, generator=g)], -1)
    ref = torch.rand(1 if cfg["shared"] else B, Lq, cfg["refL"], 2, generator=g) * 1.2 - 0.1    # some samples fall outside
    vr, orq, rr = value.clone().requires_grad_(True), offw.clone().requires_grad_(True), ref.clone().requires_grad_(True)
    out_r = _msda_ref(vr, orq, rr, shapes, M, L, Pn)
    dy = rnd(torch.randn(out_r.shape, generator=g))
    out_r.backward(dy)
    vd, od, rd = dev(value), dev(offw, torch.float32), dev(ref, torch.float32)
    tape = Tape()
    c.tape = tape
    y = Fn.msda(vd, od, rd, shapes, M, Pn, need_dref=True)
    c.tape = None
    for t in (vd, od, rd):
        tape.watch(t)
    close("msda fwd", host(y), out_r.detach(), dtype)
    dv, do, dr = run_bwd(tape, [(y, dev(dy))], [vd, od, rd])
    close("msda dvalue", host(dv), vr.grad, dtype, 2.0)
    close("msda doffw", host(do), orq.grad, dtype, 4.0, atol=(5e-4 if dtype == F32 else 6e-2))
    close("msda dref", host(dr), rr.grad, dtype, 64.0, atol=(5e-4 if dtype == F32 else 6e-2))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M", [4, 2])
def test_msda_bwd_fewer_heads_long_query(dtype, M):
    """M < 8 heads with Lq >= 1024 and the global-gather gradient kernel: the value-gradient scatter then takes its fixed-point
    scale from msda_absmax_kernel, whose row-lane reduction used to cover only 8 of the 256 / (4 M) row lanes (max |dout|
    under-estimated -> int32 overflow in the scatter -> silently wrong dvalue).  The model always runs M = 8."""
    from emrt_amd import _lib
    c = init(dtype)
    g = torch.Generator().manual_seed(21 + M)
    L, Pn = 3, 6
    shapes = [(32, 32), (16, 16), (8, 8)]
    Lv = sum(h * w for h, w in shapes)
    B, Lq = 2, Lv
    tp = M * L * Pn
    value = rnd(torch.randn(B, Lv, M * 32, generator=g))
    offw = torch.cat([torch.randn(B, Lq, 2 * tp, generator=g) * 2.0, torch.randn(B, Lq, tp, generator=g)], -1)
    ref = torch.rand(1, Lq, 1, 2, generator=g)
    vr, orq = value.clone().requires_grad_(True), offw.clone().requires_grad_(True)
    out_r = _msda_ref(vr, orq, ref, shapes, M, L, Pn)
    # one very large |dout| per (batch, head), on a row whose lane (row % (256 / (4 M))) is >= 8: exactly what the old reduction
    # dropped -- the scale 2^30 / (Lq max|g|) was then taken from the 0.01-sized rest and the spike's contributions overflowed int32
    dy = torch.randn(out_r.shape, generator=g) * 0.01
    rpp = 256 // (4 * M)
    for b in range(B):
        for m in range(M):
            dy[b, 64 * (3 + m) + rpp - 1, m * 32 + 5] = 1000.0
    dy = rnd(dy)
    out_r.backward(dy)
    vd, od, rd = dev(value), dev(offw, torch.float32), dev(ref, torch.float32)
    L_ = _lib.lib()
    old = L_.set_tuning("msda_bwd_global", 1)
    try:
        tape = Tape()
        c.tape = tape
        y = Fn.msda(vd, od, rd, shapes, M, Pn)
        c.tape = None
        for t in (vd, od):
            tape.watch(t)
        close("msda fwd M%d" % M, host(y), out_r.detach(), dtype)
        dv, do = run_bwd(tape, [(y, dev(dy))], [vd, od])
    finally:
        L_.set_tuning("msda_bwd_global", old)
    rel = (host(dv) - vr.grad).norm() / vr.grad.norm()
    # (the scatter accumulates in 32-bit fixed point with resolution Lq * max|g| / 2^30 = 1.3e-3 here: a bound on the ABSOLUTE error)
    # measured 1.4e-3 (fp32): the quantisation of ~70 contributions per element; an overflowed scatter is O(1) wrong
    assert rel < (5e-3 if dtype == F32 else 8e-3), "msda dvalue M=%d: relative L2 error %g" % (M, rel)
    assert (host(dv) - vr.grad).abs().max().item() < (0.2 if dtype == F32 else 4.0)
    close("msda doffw M%d" % M, host(do), orq.grad, dtype, 4.0, atol=(2e-3 if dtype == F32 else 0.2))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("L", [110, 128, 37])      # 110: the EMRT decoder; 128: backward without the LDS copy of P; 37: ragged quads
def test_mha_fwd_bwd(dtype, L):
    c = init(dtype)
    c.training = False   # no dropout
    g = torch.Generator().manual_seed(9)
    B, E, Mh = 3, 256, 8
    qk = rnd(torch.randn(B, L, 2 * E, generator=g))
    v = rnd(torch.randn(B, L, E, generator=g))
    qkr, vr = qk.clone().requires_grad_(True), v.clone().requires_grad_(True)
    q = qkr[..., :E].reshape(B, L, Mh, 32).transpose(1, 2)
    k = qkr[..., E:].reshape(B, L, Mh, 32).transpose(1, 2)
    vv = vr.reshape(B, L, Mh, 32).transpose(1, 2)
    w = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(32.0), -1)
    o = (w @ vv).transpose(1, 2).reshape(B, L, E)
    dy = rnd(torch.randn(o.shape, generator=g))
    o.backward(dy)
    qd, vd = dev(qk), dev(v)
    tape = Tape()
    c.tape = tape
    y = Fn.mha(qd, vd, Mh, 0.1, 3)
    c.tape = None
    tape.watch(qd)
    tape.watch(vd)
    close("mha fwd", host(y), o.detach(), dtype)
    dqk, dv = run_bwd(tape, [(y, dev(dy))], [qd, vd])
    close("mha dqk", host(dqk), qkr.grad, dtype)
    close("mha dv", host(dv), vr.grad, dtype)


def test_mha_dropout_backward_consistent_with_forward():
    """Dropout on the attention weights (layers.py:297): with the mask fixed (same device seed and salt) the output is
    linear in V, so <dy, o(V + d) - o(V)> must equal <dV, d> exactly -- this ties the backward's re-derived mask to the
    forward's."""
    c = init(F32)
    c.training = True
    g = torch.Generator().manual_seed(19)
    B, L, E, Mh = 2, 110, 256, 8
    qk, v, d = (torch.randn(B, L, n, generator=g) for n in (2 * E, E, E))
    dy = torch.randn(B, L, E, generator=g)
    qd, vd, v2d = dev(qk), dev(v), dev(v + d)
    tape = Tape()
    c.tape = tape
    y = Fn.mha(qd, vd, Mh, 0.5, 7)
    c.tape = None
    tape.watch(vd)
    y2 = Fn.mha(qd, v2d, Mh, 0.5, 7)
    y0 = None
    c.training = False
    y0 = Fn.mha(qd, vd, Mh, 0.5, 7)
    assert (host(y) - host(y0)).abs().max() > 1e-2                  # the mask really drops something
    dv, = run_bwd(tape, [(y, dev(dy))], [vd])
    lhs = ((host(y2) - host(y)) * dy).sum().item()
    rhs = (host(dv) * d).sum().item()
    assert abs(lhs - rhs) < 2e-3 * max(1.0, abs(lhs)), (lhs, rhs)


@pytest.mark.parametrize("dtype", DTYPES)
def test_linear_group_equals_separate_linears(dtype):
    """Fn.linear_group (value_proj(value) | offsets-logits projection of the query in ONE launch each way, t_e_d.py:83-92): the grouped launch
    runs the same 64x64 tile with the same k order as the separate launches -- outputs (one of them fp32), data gradients and weight / bias
    gradients bit-identical; the second data gradient accumulates into an existing gradient of its input."""
    c = init(dtype)
    g = torch.Generator().manual_seed(31)
    B, Lv, Lq, C = 2, 1344, 110, 256
    value, query = rnd(torch.randn(B, Lv, C, generator=g)), rnd(torch.randn(B, Lq, C, generator=g))
    dv_, dq_ = rnd(torch.randn(B, Lv, 256, generator=g)), rnd(torch.randn(B, Lq, 432, generator=g))
    prior = rnd(torch.randn(B, Lv, C, generator=g))
    res = {}
    for mode in ("group", "separate"):
        l1, l2 = hnn.Linear(C, 256), hnn.Linear(C, 432)
        gg = torch.Generator().manual_seed(32)
        with torch.no_grad():
            l1.weight.copy_(rnd(torch.randn(256, C, generator=gg) / 16))
            l2.weight.copy_(rnd(torch.randn(432, C, generator=gg) / 16))
            l1.bias.copy_(torch.randn(256, generator=gg))
            l2.bias.copy_(torch.randn(432, generator=gg))
        Holder(l1=l1, l2=l2).place()
        vd, qd = dev(value), dev(query)
        tape = Tape()
        c.tape = tape
        if mode == "group":
            a, b = Fn.linear_group([(vd, l1.gw, False), (qd, l2.gw, True)])
        else:
            a, b = Fn.linear(vd, l1.gw), Fn.linear(qd, l2.gw, out_f32=True)
        c.tape = None
        assert b.dtype == torch.float32 and a.dtype == c.tdtype
        tape.watch(vd)
        tape.watch(qd)
        tape.add_grad(vd, dev(prior), owned=True)          # an earlier contribution: the value gradient accumulates into it
        gx, gq = run_bwd(tape, [(a, dev(dv_)), (b, dev(dq_))], [vd, qd])
        res[mode] = [host(t) for t in (a, b, gx, gq, l1.weight.grad, l1.bias.grad, l2.weight.grad, l2.bias.grad)]
    for u, v, name in zip(res["group"][:4], res["separate"][:4], ("value_proj out", "offsets|logits out", "d value", "d query")):
        assert torch.equal(u, v), name
    for u, v, name in zip(res["group"][4:], res["separate"][4:], ("dW1", "db1", "dW2", "db2")):      # (batched weight gradients: fp32 atomics, order-dependent last bits)
        assert ((u - v).norm() / v.norm()).item() < 2e-6, name
    want = F.linear(value, l1.weight.detach().cpu().float(), None)          # sanity against torch (weights read back)
    assert res["group"][0].shape == want.shape


@pytest.mark.parametrize("p", [0.0, 0.1], ids=["no-dropout", "dropout"])
def test_mha_backward_split_over_two_blocks_is_bit_identical(p):
    """mha_bwd_mfma_kernel as two blocks per (batch, head) -- dq | dk, dv, both re-deriving the row dots (knob mha_bwd_split, default on) -- against the
    one-block form: the same arithmetic on the same lanes, so every gradient bit for bit (layers.py:283-303), at the decoder's L = 110 and ragged lengths."""
    from emrt_amd import _lib
    L_ = _lib.lib()
    c = init(BF16)
    g = torch.Generator().manual_seed(33)
    E, Mh = 256, 8
    for L in (110, 128, 37, 2):
        B = 3
        c.training = p > 0.0
        qk, v = rnd(torch.randn(B, L, 2 * E, generator=g)), rnd(torch.randn(B, L, E, generator=g))
        dy = rnd(torch.randn(B, L, E, generator=g))
        outs = {}
        for knob in (0, 1):
            old = L_.set_tuning("mha_bwd_split", knob)
            try:
                c.salt_counter = 100
                qd, vd = dev(qk), dev(v)
                tape = Tape()
                c.tape = tape
                y = Fn.mha(qd, vd, Mh, p, 3)
                c.tape = None
                tape.watch(qd)
                tape.watch(vd)
                dqk, dv = run_bwd(tape, [(y, dev(dy))], [qd, vd])
                outs[knob] = [host(t) for t in (y, dqk, dv)]
            finally:
                L_.set_tuning("mha_bwd_split", old)
        for u, w_, name in zip(outs[0], outs[1], ("out", "dqk", "dv")):
            assert torch.equal(u, w_), (L, name, (u - w_).abs().max().item())
            assert torch.isfinite(u).all()
    c.training = True


def test_mha_mfma_kernels_dropout_and_agreement_with_the_valu_kernels():
    """bf16 takes the MFMA kernels (csrc/attn.hip: mha_fwd_mfma_kernel / mha_bwd_mfma_kernel; layers.py:283-303).  (1) Without dropout they
    must agree with the VALU kernels (knob mha_valu) on the same bf16 inputs to bf16 rounding, forward and all three gradients, at the
    decoder's L = 110 and at ragged / full tile counts.  (2) With dropout on the attention weights (layers.py:297) the forward's mask and the
    backward's two re-derivations of it (row pass: dq; column pass: dk, dv) must be ONE mask: with the mask fixed the output is linear in V,
    so <dy, o(V + d) - o(V)> == <dV, d>; and it is linear in the scores' gradient direction too: a finite-difference check of dq / dk along a
    random direction ties the column pass's mask to the forward's.  The dropped fraction is p."""
    from emrt_amd import _lib
    L_ = _lib.lib()
    c = init(BF16)
    g = torch.Generator().manual_seed(21)
    E, Mh = 256, 8
    for L in (110, 128, 37, 16, 2):
        B = 2
        c.training = False
        qk, v = rnd(torch.randn(B, L, 2 * E, generator=g)), rnd(torch.randn(B, L, E, generator=g))
        dy = rnd(torch.randn(B, L, E, generator=g))
        outs = {}
        for knob in (0, 1):
            old = L_.set_tuning("mha_valu", knob)
            try:
                qd, vd = dev(qk), dev(v)
                tape = Tape()
                c.tape = tape
                y = Fn.mha(qd, vd, Mh, 0.1, 3)
                c.tape = None
                tape.watch(qd)
                tape.watch(vd)
                dqk, dv = run_bwd(tape, [(y, dev(dy))], [qd, vd])
                outs[knob] = [host(t) for t in (y, dqk, dv)]
            finally:
                L_.set_tuning("mha_valu", old)
        for u, w_, name in zip(outs[0], outs[1], ("out", "dqk", "dv")):
            rel = ((u - w_).norm() / w_.norm()).item()
            assert rel < 6e-3, ("L=%d %s: MFMA vs VALU relative L2 %.3g" % (L, name, rel))          # two bf16 roundings (probabilities as MFMA operands)
    # ---- nothing may depend on what an earlier kernel left in LDS: the backward multiplies the rows of the padding tiles by a zero probability,
    # and round 5's first version read their row sums from uninitialised LDS -- 0 x NaN = NaN in dk / dv whenever a NaN bit pattern happened to
    # lie there (seen only beside other processes on the GPU).  Here every CU's LDS is filled with NaNs first (the staged value slab of a
    # deformable-attention forward on an all-NaN value tensor) and the gradients must come out bit-identical to the clean run
    c.training = False
    B, L = 8, 110
    qk, v = rnd(torch.randn(B, L, 2 * E, generator=g)), rnd(torch.randn(B, L, E, generator=g))
    dy = rnd(torch.randn(B, L, E, generator=g))
    shapes = [(32, 32), (16, 16), (8, 8)]
    Lv = sum(h * w for h, w in shapes)
    nanv = dev(torch.full((8, Lv, 256), float("nan")))
    offw = dev(torch.zeros(8, Lv, 3 * 8 * 3 * 6), torch.float32)
    from emrt_amd.src.models.emrt import encoder_reference_points
    refp = dev(encoder_reference_points(shapes), torch.float32)
    got = []
    for poison in (False, True, True):
        qd, vd = dev(qk), dev(v)
        tape = Tape()
        c.tape = tape
        y = Fn.mha(qd, vd, Mh, 0.0, 3)
        c.tape = None
        tape.watch(qd)
        tape.watch(vd)
        if poison:
            for _ in range(3):
                Fn.msda(nanv, offw, refp, shapes, 8, 6)
        dqk, dv = run_bwd(tape, [(y, dev(dy))], [qd, vd])
        got.append((host(dqk), host(dv)))
    for a_, b_ in zip(got[0], got[1]):
        assert torch.isfinite(b_).all() and torch.equal(a_, b_), "the attention backward depends on stale LDS contents"
    # ---- dropout: one mask in three places -------------------------------------------------------------------------------------
    c.training = True
    B, L, p = 2, 110, 0.5
    qk, v, d = (rnd(torch.randn(B, L, n, generator=g)) for n in (2 * E, E, E))
    d = rnd(d * 0.25)
    dy = rnd(torch.randn(B, L, E, generator=g))
    qd, vd, v2d = dev(qk), dev(v), dev(rnd(v + d))
    tape = Tape()
    c.tape = tape
    y = Fn.mha(qd, vd, Mh, p, 7)
    c.tape = None
    tape.watch(vd)
    tape.watch(qd)
    y2 = Fn.mha(qd, v2d, Mh, p, 7)
    c.training = False
    y0 = Fn.mha(qd, vd, Mh, p, 7)
    c.training = True
    assert (host(y) - host(y0)).abs().max() > 1e-2
    dqk, dv = run_bwd(tape, [(y, dev(dy))], [qd, vd])
    dvh = host(dv)
    lhs = ((host(y2) - host(y)) * dy).sum().item()
    rhs = (dvh * (host(v2d) - host(vd))).sum().item()
    print("MHA MFMA dropout: <dy, o(V+d) - o(V)> = %.4f, <dV, d> = %.4f" % (lhs, rhs))
    # both sides are sums of 56 320 products: the bf16 rounding of the two outputs alone puts ~0.07 of noise on lhs (2^-9 |y| |dy| sqrt(n)), a backward
    # that used another mask moves rhs by ~|dV| |d| sqrt(n) ~ 6; 0.3 separates the two (round 5's 0.02 sat inside the noise and held by the luck of one
    # mask: it failed on the first other one).  The comparison against torch autograd with the device's own mask below is the sharp check.
    assert abs(lhs - rhs) < 0.3 * max(1.0, abs(lhs)), (lhs, rhs)
    # dq / dk / dv against torch autograd with the DEVICE'S OWN mask: the dropped probabilities are read out of the forward itself with one-hot
    # values (V_j = e_{j mod 32} for the keys of one block of 32: out_i[d] is then Pd[i][that key]), once with and once without dropout
    def read_probs(pp, train):
        c.training = train
        cols = []
        for blk in range((L + 31) // 32):
            vv = torch.zeros(B, L, Mh, 32)
            for jj in range(blk * 32, min(L, blk * 32 + 32)):
                vv[:, jj, :, jj - blk * 32] = 1.0
            o = host(Fn.mha(qd, dev(vv.reshape(B, L, E)), Mh, pp, 7)).reshape(B, L, Mh, 32)
            cols.append(o[..., :min(32, L - blk * 32)])
        c.training = True
        return torch.cat(cols, -1).permute(0, 2, 1, 3)          # [B, Mh, L(query), L(key)]
    pd_dev, p_dev = read_probs(p, True), read_probs(p, False)
    mask = (pd_dev > 0).float()
    frac = 1.0 - mask[p_dev > 1e-4].mean().item()
    print("MHA MFMA dropout: %.4f of the attention weights dropped (p = %.2f)" % (frac, p))
    assert abs(frac - p) < 0.01, frac
    assert ((pd_dev - p_dev * mask / (1 - p)).abs().max() < 2e-2 * p_dev.max()).item()          # kept weights are the undropped ones / (1 - p), to bf16
    qkr, vr = qk.clone().requires_grad_(True), v.clone().requires_grad_(True)
    q = qkr[..., :E].reshape(B, L, Mh, 32).transpose(1, 2)
    k = qkr[..., E:].reshape(B, L, Mh, 32).transpose(1, 2)
    w = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(32.0), -1) * mask / (1 - p)
    o = (w @ vr.reshape(B, L, Mh, 32).transpose(1, 2)).transpose(1, 2).reshape(B, L, E)
    o.backward(dy)
    close("mha mfma dropout fwd", host(y), o.detach(), BF16, 2.0)
    for got, want, name in ((host(dqk), qkr.grad, "dqk"), (dvh, vr.grad, "dv")):
        rel = ((got - want).norm() / want.norm()).item()
        print("MHA MFMA dropout: %s vs torch with the device's mask: relative L2 %.3g" % (name, rel))
        assert rel < 1.5e-2, (name, rel)          # bf16 probabilities / dS as MFMA operands: two roundings of 2^-9
    # the dropped fraction: zeros among the probabilities cannot be read from outside, but with V = ones(.) every output channel is the
    # kept probability mass / (1 - p): its mean over queries is 1, its variance that of a p-thinned sum
    ones = dev(torch.ones(B, L, E))
    mass = host(Fn.mha(qd, ones, Mh, p, 7))
    assert abs(mass.mean().item() - 1.0) < 0.03, mass.mean().item()
    assert 0.05 < mass.std().item() < 0.6, mass.std().item()


# -----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(2, 8, 8, 64, 16, 16, True), (2, 8, 8, 64, 16, 16, False), (2, 3, 3, 256, 32, 32, True),
                                  (1, 6, 6, 128, 32, 32, True), (2, 1, 1, 64, 8, 8, True), (2, 16, 12, 6, 64, 48, False),
                                  (2, 10, 10, 32, 10, 10, True), (2, 2, 2, 64, 16, 16, False), (1, 8, 8, 512, 32, 32, False)])
def test_resize_bilinear(dtype, case):
    N, IH, IW, C, OH, OW, ac = case
    c = init(dtype)
    g = torch.Generator().manual_seed(10)
    x = rnd(torch.randn(N, C, IH, IW, generator=g))
    add_t = rnd(torch.randn(N, C, OH, OW, generator=g))
    xr, ar = x.clone().requires_grad_(True), add_t.clone().requires_grad_(True)
    o = F.interpolate(xr, size=(OH, OW), mode="bilinear", align_corners=ac) + ar
    dy = rnd(torch.randn(o.shape, generator=g))
    o.backward(dy)
    xd, ad = dev_map(x), dev_map(add_t)
    tape = Tape()
    c.tape = tape
    y = Fn.resize_bilinear(xd, OH, OW, ac, add_t=ad)
    c.tape = None
    tape.watch(xd)
    tape.watch(ad)
    close("resize fwd", host_map(y), o.detach(), dtype)
    dx, da = run_bwd(tape, [(y, dev_map(dy))], [xd, ad])
    close("resize dx", host_map(dx), xr.grad, dtype, float(max(1, (OH // IH) * (OW // IW))) ** 0.5)
    close("resize dadd", host_map(da), ar.grad, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_resize_to_nchw_logits(dtype):
    c = init(dtype)
    g = torch.Generator().manual_seed(11)
    N, IH, IW, C = 2, 16, 16, 6
    x = rnd(torch.randn(N, C, IH, IW, generator=g))
    xr = x.clone().requires_grad_(True)
    o = F.interpolate(xr, size=(32, 32), mode="bilinear", align_corners=False)
    dy = torch.randn(o.shape, generator=g)
    o.backward(dy)
    xd = dev_map(x)
    tape = Tape()
    c.tape = tape
    y = Fn.resize_bilinear(xd, 32, 32, False, out_nchw_f32=True)
    c.tape = None
    tape.watch(xd)
    assert y.dtype == torch.float32 and tuple(y.shape) == (N, C, 32, 32)
    close("resize nchw fwd", y.cpu(), o.detach(), dtype, atol=1e-5 if dtype == F32 else None)
    dx, = run_bwd(tape, [(y, dev(dy, torch.float32))], [xd])
    close("resize nchw dx", host_map(dx), xr.grad, dtype, 2.0)


@pytest.mark.parametrize("dtype", DTYPES)
def test_adaptive_pool_and_maxpool(dtype):
    c = init(dtype)
    g = torch.Generator().manual_seed(12)
    N, H, W, C = 2, 32, 32, 64
    x = rnd(torch.randn(N, C, H, W, generator=g))
    xr = x.clone().requires_grad_(True)
    scales = [1, 3, 6, 8]
    toks = torch.cat([F.adaptive_avg_pool2d(xr, k).reshape(N, C, -1) for k in scales], -1).transpose(1, 2)
    dy = rnd(torch.randn(toks.shape, generator=g))
    toks.backward(dy)
    xd = dev_map(x)
    tape = Tape()
    c.tape = tape
    y = Fn.adaptive_avgpool_tokens(xd, scales)
    c.tape = None
    tape.watch(xd)
    close("pool fwd", host(y), toks.detach(), dtype)
    dx, = run_bwd(tape, [(y, dev(dy))], [xd])
    close("pool dx", host_map(dx), xr.grad, dtype)
    # max pool, with ties (post-ReLU style input); C = 64 takes the 8-channel kernel, C = 3 (the image itself, spatial_branch) the scalar one
    for Cm in (64, 3):
        x2 = rnd(F.relu(torch.randn(N, Cm, 17, 19, generator=g)))
        x2r = x2.clone().requires_grad_(True)
        m = F.max_pool2d(x2r, 3, 2, 1)
        dm = rnd(torch.randn(m.shape, generator=g))
        m.backward(dm)
        x2d = dev_map(x2)
        tape = Tape()
        c.tape = tape
        y2 = Fn.maxpool(x2d, 3, 2, 1)
        c.tape = None
        tape.watch(x2d)
        close("maxpool fwd C=%d" % Cm, host_map(y2), m.detach(), dtype, atol=0, rtol=0)
        dx2, = run_bwd(tape, [(y2, dev_map(dm))], [x2d])
        close("maxpool dx C=%d" % Cm, host_map(dx2), x2r.grad, dtype)


def test_nchw_ingest_and_elementwise():
    for dtype in DTYPES:
        c = init(dtype)
        g = torch.Generator().manual_seed(13)
        img = torch.randn(2, 3, 16, 24, generator=g)
        y = Fn.nchw_to_nhwc(img.cuda())
        close("ingest", host_map(y), rnd(img), dtype, atol=0, rtol=0)
        y8 = Fn.nchw_to_nhwc(img.cuda(), c_out=8)        # the image as an 8-channel map, channels 3..7 zero
        assert tuple(y8.shape) == (2, 16, 24, 8) and float(y8[..., 3:].float().abs().max()) == 0.0
        close("ingest padded", host_map(y8[..., :3].contiguous()), rnd(img), dtype, atol=0, rtol=0)
        y4 = Fn.nchw_to_nhwc(img.cuda(), c_out=4)
        assert float(y4[..., 3:].float().abs().max()) == 0.0
        close("ingest padded to 4", host_map(y4[..., :3].contiguous()), rnd(img), dtype, atol=0, rtol=0)
        y5 = Fn.nchw_to_nhwc(img.cuda(), c_out=5)        # neither 4 nor 8: element-wise kernel
        close("ingest padded to 5", host_map(y5[..., :3].contiguous()), rnd(img), dtype, atol=0, rtol=0)
        a, b = rnd(torch.randn(2, 21, 64, generator=g)), rnd(torch.randn(21, 64, generator=g))
        s = Fn.add(dev(a), dev(b), period=21 * 64)
        close("add bcast", host(s), a + b, dtype)
        # strided accumulate into a token slab and a channel slice
        base = c.zeros((2, 21, 64))
        Fn.add_into(base.narrow(1, 5, 16), dev(a[:, :16]))
        ref = torch.zeros(2, 21, 64)
        ref[:, 5:] += a[:, :16]
        close("acc slab", host(base), ref, dtype)
        cat = c.zeros((2, 4, 4, 192))
        part = rnd(torch.randn(2, 4, 4, 64, generator=g))
        Fn.add_into(cat.narrow(3, 64, 64), dev(part))
        ref = torch.zeros(2, 4, 4, 192)
        ref[..., 64:128] = part
        close("acc slice", host(cat), ref, dtype)
        sg = Fn.sigmoid_f32(dev(torch.linspace(-4, 4, 220), torch.float32))
        close("sigmoid", sg.cpu(), torch.sigmoid(torch.linspace(-4, 4, 220)), F32, atol=1e-6)


@pytest.mark.parametrize("dtype", DTYPES)
def test_concat_tokens_and_its_split_backward(dtype):
    """Pyramid-pooling tokens (1 + 9 + 36 + 64 per image, paddle_EMRT.py:70-78): concat in one launch, backward = one split launch."""
    c = init(dtype)
    g = torch.Generator().manual_seed(23)
    B, C = 3, 64
    parts = [rnd(torch.randn(B, n, C, generator=g)) for n in (1, 9, 36, 64)]
    pd = [dev(p_) for p_ in parts]
    tape = Tape()
    c.tape = tape
    y = Fn.concat_tokens(pd)
    c.tape = None
    for p_ in pd:
        tape.watch(p_)
    want = torch.cat(parts, 1)
    close("concat", host(y), want, dtype, atol=0, rtol=0)
    dy = rnd(torch.randn(want.shape, generator=g))
    grads = run_bwd(tape, [(y, dev(dy))], pd)
    s0 = 0
    for p_, gp in zip(parts, grads):
        close("split", host(gp), dy[:, s0:s0 + p_.shape[1]], dtype, atol=0, rtol=0)
        s0 += p_.shape[1]


def test_dropout_statistics_and_determinism():
    c = init(F32)
    x = torch.ones(64, 1024)
    xd = dev(x)
    tape = Tape()
    c.tape = tape
    y = Fn.dropout(xd, 0.1, 5)
    c.tape = None
    tape.watch(xd)
    yh = host(y)
    keep = (yh != 0).float().mean().item()
    assert abs(keep - 0.9) < 0.01, keep
    assert torch.allclose(yh[yh != 0], torch.tensor(1.0 / 0.9))
    dx, = run_bwd(tape, [(y, dev(torch.ones(64, 1024)))], [xd])
    assert torch.equal(host(dx), yh)                      # same mask, same scale in backward
    y2 = Fn.dropout(xd, 0.1, 6)
    assert not torch.equal(host(y2), yh)                  # different salt -> different mask
    # Dropout2D: whole (image, channel) planes
    x4 = dev(torch.ones(4, 5, 5, 64))
    y4 = host(Fn.dropout(x4, 0.3, 7, mode=1, hw=25))
    per = y4.reshape(4, 25, 64)
    assert ((per == 0).all(1) | (per != 0).all(1)).all()


@pytest.mark.parametrize("dtype", DTYPES)
def test_level_embedding_gradient_from_every_layers_query_gradient_in_one_launch(dtype):
    """emrt_colsum_levels_multi: dst[l][c] += sum over T tensors, the batch and the tokens of level l (transformer_encoder_decoder.py:447-448: the level
    embedding is added to the query of every encoder layer, so its gradient sums every layer's query gradient over the level's tokens)."""
    import ctypes
    from emrt_amd import _lib
    c = init(dtype)
    g = torch.Generator().manual_seed(31)
    B, C = 3, 256
    spans = [(0, 20 * 20), (400, 10 * 10), (500, 5 * 5)]
    Lv = 525
    xs = [rnd(torch.randn(B, Lv, C, generator=g)) for _ in range(4)]
    xd = [dev(x) for x in xs]
    dst0 = torch.randn(3, C, generator=g)
    dst = dst0.clone().cuda()
    ptrs = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in xd])
    st_ = (ctypes.c_int * 3)(*[a for a, _ in spans])
    cn_ = (ctypes.c_int * 3)(*[n for _, n in spans])
    _lib.lib().call("emrt_colsum_levels_multi", ptrs, 4, st_, cn_, 3, B, Lv, C, ctypes.c_void_p(dst.data_ptr()), c.dtype, c.stream)
    want = dst0.clone()
    for l, (a, n) in enumerate(spans):
        want[l] += sum(x[:, a:a + n].double().sum((0, 1)) for x in xs).float()
    close("level sums", dst.cpu(), want, F32, scale=math.sqrt(4 * B * 400))          # fp32 sums of (rounded) inputs, any order


def _assert_masks_independent(name, m1, m2, p, lanes):
    """two keep masks ([rows, cols] bool, the column index is the position inside the hash's group modulo `lanes`) drawn at rate 1 - p: each
    keeps 1 - p, and they agree on (1 - p)^2 + p^2 of the elements at EVERY position of the group (a shared hash word shows up as agreement 1)"""
    want = (1 - p) ** 2 + p ** 2
    for e in range(lanes):
        a, b = m1[:, e::lanes], m2[:, e::lanes]
        n = a.numel()
        keep = a.float().mean().item()
        agree = (a == b).float().mean().item()
        sd_k, sd_a = 5.0 * math.sqrt(p * (1 - p) / n), 5.0 * math.sqrt(want * (1 - want) / n)
        assert abs(keep - (1 - p)) < sd_k + 1e-3, "%s: position %d keeps %.4f" % (name, e, keep)
        assert abs(agree - want) < sd_a + 1e-3, "%s: position %d of the group: two sites agree on %.4f of the elements, independent masks on %.4f" % (name, e, agree, want)


def test_dropout_masks_of_two_sites_are_independent():
    """Every nn.Dropout of the reference draws its own mask (transformer_encoder_decoder.py:118-121,157-161,259-262; layers.py:297).  Here one
    step's sites share the device seed and differ by their salt: the salt has to reach EVERY word of a group's draw (round 5's cheap hashes
    left it out of the first word, so elements 0 and 1 of every quad -- 2 of 8 in the GEMM epilogue's groups, 2 of 4 attention columns -- were
    dropped together at every site of a step).  All three hashes: drop_quad (element dropout / LayerNorm's branch dropout), drop_words8
    (emrt_conv2d_drop) and mha_drop4 (attention weights)."""
    c = init(BF16)
    c.training = True
    p = 0.3
    # (1) element mode: quads of the flat index
    xd = dev(torch.ones(512, 1024))
    m = [host(Fn.dropout(xd, p, s)) != 0 for s in (5, 6)]
    _assert_masks_independent("drop_quad", m[0], m[1], p, 4)
    # (2) the FFN's dropout drawn in linear1's epilogue: groups of 8 output channels; all pre-activations positive, so the stored sign IS the mask
    C, Hd, rows = 64, 256, 2048
    lin = hnn.Linear(C, Hd)
    with torch.no_grad():
        lin.weight.fill_(1.0 / C)
        lin.bias.fill_(0.5)
    Holder(l1=lin).place()
    xin = dev(torch.rand(1, rows, C) + 0.5)
    m = [host(lin(xin, relu=True, drop=(p, s))).reshape(rows, Hd) > 0 for s in (11, 12)]
    _assert_masks_independent("drop_words8", m[0], m[1], p, 8)
    # (3) attention-weight dropout: quads of key columns; the dropped probabilities read out with one-hot values (L = 32 keys = the head dim)
    B, L, E, Mh = 4, 32, 256, 8
    g = torch.Generator().manual_seed(3)
    qk = rnd(torch.randn(B, L, 2 * E, generator=g) * 0.1)          # near-uniform attention: every probability is far from 0
    vv = torch.zeros(B, L, Mh, 32)
    for j in range(L):
        vv[:, j, :, j] = 1.0
    qd, vd = dev(qk), dev(vv.reshape(B, L, E))
    m = [(host(Fn.mha(qd, vd, Mh, p, s)).reshape(B, L, Mh, 32).permute(0, 2, 1, 3).reshape(B * Mh * L, 32) > 0) for s in (7, 8)]
    _assert_masks_independent("mha_drop4", m[0], m[1], p, 4)


# -----------------------------------------------------------------------------------------------------------------
def test_softmax_ce_and_optimizer():
    c = init(F32)
    g = torch.Generator().manual_seed(14)
    N, C, H, W = 3, 6, 16, 20
    logits = torch.randn(N, C, H, W, generator=g) * 2
    labels = torch.randint(0, C, (N, H, W), generator=g)
    labels[torch.rand(N, H, W, generator=g) < 0.1] = 255
    lr_ = logits.clone().requires_grad_(True)
    ref = F.cross_entropy(lr_, labels, ignore_index=255)
    (0.4 * ref).backward()
    ld = dev(logits, torch.float32)
    tape = Tape()
    c.tape = tape
    res = Fn.softmax_ce(ld, labels.cuda(), 255, weight=0.4)
    c.tape = None
    tape.watch(ld)
    tape.backward()
    close("ce loss", res[:1].cpu(), ref.detach().reshape(1), F32, atol=1e-5)
    assert int(res[1].item()) == int((labels != 255).sum())
    close("ce dlogits", tape.result(ld).cpu(), lr_.grad, F32, atol=1e-7)


def test_softmax_ce_pair_equals_the_two_single_head_calls():
    """MixSoftmaxCrossEntropyLoss's two heads in one pass (emrt_softmax_ce_pair_fwd / _bwd; mix_softmax_cross_entropy_loss.py:29-35,44-51):
    losses, counts, the weighted total and both gradients bit-identical to two emrt_softmax_ce calls + the scalar axpby, and equal to torch."""
    from emrt_amd.src.models.losses import MixSoftmaxCrossEntropyLoss
    from emrt_amd.runtime import ctx
    c = init(F32)
    g = torch.Generator().manual_seed(24)
    N, C, H, W = 4, 7, 24, 40
    la, lb = torch.randn(N, C, H, W, generator=g) * 2, torch.randn(N, C, H, W, generator=g) * 3
    labels = torch.randint(0, C, (N, H, W), generator=g)
    labels[torch.rand(N, H, W, generator=g) < 0.15] = 255
    ra_, rb_ = la.clone().requires_grad_(True), lb.clone().requires_grad_(True)
    ref = F.cross_entropy(ra_, labels, ignore_index=255) + 0.4 * F.cross_entropy(rb_, labels, ignore_index=255)
    ref.backward()
    lab = labels.cuda()
    ad, bd = dev(la, torch.float32), dev(lb, torch.float32)
    tape = Tape()
    c.tape = tape
    sa = Fn.softmax_ce(ad, lab, 255, 1.0)
    sb = Fn.softmax_ce(bd, lab, 255, 0.4)
    c.tape = None
    tape.watch(ad)
    tape.watch(bd)
    tape.backward()
    single = [sa.cpu(), sb.cpu(), tape.result(ad).cpu(), tape.result(bd).cpu()]
    ad2, bd2 = dev(la, torch.float32), dev(lb, torch.float32)
    tape = Tape()
    c.tape = tape
    pa, pb, total = Fn.softmax_ce_pair(ad2, bd2, lab, 255, 1.0, 0.4)
    c.tape = None
    tape.watch(ad2)
    tape.watch(bd2)
    tape.backward()
    pair = [pa.cpu(), pb.cpu(), tape.result(ad2).cpu(), tape.result(bd2).cpu()]
    for u, v, name in zip(pair, single, ("loss a", "loss b", "dlogits a", "dlogits b")):
        assert torch.equal(u, v), name
    assert abs(total.item() - (single[0][0].item() + 0.4 * single[1][0].item())) < 1e-6
    close("ce pair total", total.cpu(), ref.detach().reshape(1), F32, atol=1e-5)
    close("ce pair dlogits a", pair[2], ra_.grad, F32, atol=1e-7)
    close("ce pair dlogits b", pair[3], rb_.grad, F32, atol=1e-7)
    # the loss object of the recipe takes the pair path (one forward call) and an all-ignored batch gives 0, not 0 / 0
    class _Out(tuple):
        tape = None
    L_ = __import__("emrt_amd._lib", fromlist=["lib"]).lib()
    L_.start_record()
    val = MixSoftmaxCrossEntropyLoss(ignore_index=255, aux=True, aux_weight=0.4)(_Out((ad, bd)), lab)
    names = [n for n, _ in L_.stop_record()]
    assert names == ["emrt_softmax_ce_pair_fwd"] and abs(val.item() - ref.item()) < 1e-5
    _, _, t0 = Fn.softmax_ce_pair(ad, bd, torch.full_like(lab, 255), 255, 1.0, 0.4)
    assert t0.item() == 0.0


def test_sgd_momentum_matches_reference_optimizer():
    from oracle.train_ref import MomentumRef, poly_lr
    from emrt_amd.src.models.solver import Momentum, PolynomialDecay
    c = init(F32)
    g = torch.Generator().manual_seed(15)

    class Tiny(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = hnn.Linear(8, 12)
            self.sampling_offsets = hnn.Linear(8, 4)     # lr_mult 0.1 by name
            self.b = hnn.LayerNorm(12)

    m = Tiny()
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn(p.shape, generator=g))
    refp = [(n, p.detach().clone().requires_grad_(True)) for n, p in m.named_parameters()]
    store = hnn.ParamStore(m, c.device, F32, lr_mult_names=["sampling_offsets.weight", "sampling_offsets.bias"])
    hnn.bind_all(m, store)
    m.store = store
    opt = Momentum(m, PolynomialDecay(0.01, 100, 0.0, 0.9), 0.9, 1e-4, 1.0)
    ropt = MomentumRef(refp, 0.9, 1e-4, 1.0)
    for step in range(3):
        grads = {n: torch.randn(p.shape, generator=g) * (3.0 if step == 0 else 0.05) for n, p in refp}
        for n, p in refp:
            p.grad = grads[n].clone()
        for n, p in m.named_parameters():
            p.grad.copy_(grads[n].cuda())
        ropt.step(poly_lr(step, 0.01, 0.0, 100, 0.9))
        opt.step()
        assert abs(opt.grad_norm() - ropt.last_grad_norm) < 1e-4 * max(1.0, ropt.last_grad_norm)
        for (n, p), (_, q) in zip(refp, m.named_parameters()):
            close("sgd step %d %s" % (step, n), q.detach().cpu(), p.detach(), F32, atol=1e-6, rtol=1e-5)
    assert int(c.step_counter.item()) == 3


def test_optimizer_pass_non_temporal_arm_is_bit_identical():
    """Knob sgd_nt (csrc/loss_optim.hip): the non-temporal loads / stores of the fp32 streams change caching, not arithmetic -- master weights, velocity
    and the bf16 mirror after two steps are bit-identical with the knob on and off (odd length: the scalar tail runs too)."""
    import ctypes
    from emrt_amd import _lib
    init(BF16)
    L_ = _lib.lib()
    n = 4 * 70001 + 3
    g = torch.Generator().manual_seed(31)
    p0, v0 = torch.randn(n, generator=g), torch.randn(n, generator=g) * 0.1
    grads = [torch.randn(n, generator=g) for _ in range(2)]
    res = {}
    for knob in (1, 0):
        old = L_.set_tuning("sgd_nt", knob)
        try:
            p, v = p0.cuda(), v0.cuda()
            mirror = torch.empty(n, dtype=torch.bfloat16, device="cuda")
            step = torch.zeros(1, dtype=torch.int64, device="cuda")
            ranges = (ctypes.c_longlong * 2)(1000, 5003)
            for gr in grads:
                gd = gr.cuda()
                L_.call("emrt_sgd_momentum_step", ctypes.c_void_p(p.data_ptr()), ctypes.c_void_p(gd.data_ptr()), ctypes.c_void_p(v.data_ptr()), n, None,
                        ctypes.c_void_p(step.data_ptr()), 0.01, 0.0, 0.9, 100, 0.9, 1e-4, ranges, 1, 0.1, None, ctypes.c_void_p(mirror.data_ptr()), 1,
                        ctx().stream)
            torch.cuda.synchronize()
            res[knob] = (p.cpu(), v.cpu(), mirror.float().cpu())
        finally:
            L_.set_tuning("sgd_nt", old)
    for a, b, name in zip(res[1], res[0], ("master", "velocity", "mirror")):
        bad = (a != b).nonzero().flatten()
        assert bad.numel() == 0, (name, bad.numel(), bad[:8].tolist(), (a - b).abs().max().item())
    assert not torch.equal(res[1][0], p0)


@pytest.mark.parametrize("dtype", DTYPES)
def test_layer_norm_backward_block_shapes_agree(dtype):
    """Knob ln_bwd_threads (csrc/norm.hip): 512-thread blocks of 64 rows (default) against 256-thread blocks of 32 rows -- dz / dz_branch per row do not
    depend on the block shape (bit-identical); dgamma / dbeta are sums of per-block partials added with fp32 atomics (order differs: tolerance)."""
    from emrt_amd import _lib
    c = init(dtype)
    c.training = True
    g = torch.Generator().manual_seed(33)
    B, L, C = 3, 211, 256
    a, b, dy = (rnd(torch.randn(B, L, C, generator=g)) for _ in range(3))
    res = {}
    for thr in (512, 256):
        old = _lib.lib().set_tuning("ln_bwd_threads", thr)
        try:
            ln = hnn.LayerNorm(C)
            with torch.no_grad():
                ln.weight.copy_(torch.linspace(0.5, 1.5, C))
            Holder(ln=ln).place()
            ad, bd = dev(a), dev(b)
            tape = Tape()
            c.tape = tape
            y = ln(ad, bd, drop_p=0.2, drop_salt=3)
            c.tape = None
            tape.watch(ad)
            tape.watch(bd)
            da, db = run_bwd(tape, [(y, dev(dy))], [ad, bd])
            res[thr] = [host(da), host(db), host(ln.weight.grad), host(ln.bias.grad)]
        finally:
            _lib.lib().set_tuning("ln_bwd_threads", old)
    assert torch.equal(res[512][0], res[256][0]) and torch.equal(res[512][1], res[256][1])
    for i in (2, 3):
        assert (res[512][i] - res[256][i]).abs().max().item() <= 1e-4 * max(1.0, res[256][i].abs().max().item())


@pytest.mark.parametrize("dtype", DTYPES)
def test_add_f32row_levels_equals_the_per_level_adds(dtype):
    """emrt_add_f32row_levels (pos = sine + level_embed[l] for every level in one launch, transformer_encoder_decoder.py:447-448) against one emrt_add_f32row
    per level: bit-identical (same arithmetic per element)."""
    import ctypes
    from emrt_amd import _lib
    c = init(dtype)
    g = torch.Generator().manual_seed(37)
    spans, C = [(0, 35), (35, 12), (47, 5), (52, 1)], 64
    Lv = 53
    a = dev(rnd(torch.randn(Lv, C, generator=g)))
    rows = torch.randn(len(spans), C, generator=g).cuda()
    one, per = torch.empty_like(a), torch.empty_like(a)
    L_ = _lib.lib()
    starts = (ctypes.c_int * len(spans))(*[s0 for s0, _ in spans])
    L_.call("emrt_add_f32row_levels", Fn.P(a), Fn.P(rows), Fn.P(one), starts, len(spans), Lv, C, c.dtype, c.stream)
    for l, (s0, n) in enumerate(spans):
        L_.call("emrt_add_f32row", Fn.P(a[s0:s0 + n]), Fn.P(rows[l]), Fn.P(per[s0:s0 + n]), n * C, C, c.dtype, c.stream)
    torch.cuda.synchronize()
    assert torch.equal(one, per)
    ref = host(a) + rows.cpu().repeat_interleave(torch.tensor([n for _, n in spans]), 0)
    close("add rows levels", host(one), ref, dtype)


@pytest.mark.parametrize("label_dtype", [torch.int64, torch.int32])
def test_segmentation_areas_kernel_equals_the_torch_expression(label_dtype):
    """emrt_segmentation_areas (metrics.calculate_area on device tensors; reference src/utils/metrics.py:20-59) against the torch bincount expression the CPU
    path keeps: ignore_index pixels dropped, predictions / labels outside [0, ncls) counted nowhere, exact integer counts."""
    from emrt_amd.src.utils import metrics
    init(F32)
    g = torch.Generator().manual_seed(41)
    ncls, n = 7, 3 * 517 * 389
    pred = torch.randint(-1, ncls + 1, (n,), generator=g).to(torch.int32)          # includes -1 and ncls: out of range
    lab = torch.randint(0, ncls + 2, (n,), generator=g)                            # includes ncls and ncls + 1
    lab[torch.rand(n, generator=g) < 0.1] = 255
    want = metrics.calculate_area(pred, lab, ncls, 255)                            # CPU tensors: the torch expression
    got = metrics.calculate_area(pred.cuda().reshape(3, 517, 389), lab.to(label_dtype).cuda().reshape(3, 517, 389), ncls, 255)
    torch.cuda.synchronize()
    for a, b, name in zip(got, want, ("intersect", "pred", "label")):
        assert a.is_cuda and torch.equal(a.cpu(), b), name
    assert int(want[2].sum()) > 0 and int(want[0].sum()) > 0


def test_memcpy_entry_point_stages_a_batch():
    """emrt_memcpy (engine.TrainEngine._stage, infer.SlidingWindowEngine): device -> device on the context's stream; zero bytes and dst == src are no-ops."""
    import ctypes
    from emrt_amd import _lib
    init(F32)
    src = torch.arange(100003, dtype=torch.int64, device="cuda")
    dst = torch.zeros_like(src)
    _lib.lib().call("emrt_memcpy", ctypes.c_void_p(dst.data_ptr()), ctypes.c_void_p(src.data_ptr()), src.numel() * 8, ctx().stream)
    _lib.lib().call("emrt_memcpy", ctypes.c_void_p(dst.data_ptr()), ctypes.c_void_p(dst.data_ptr()), src.numel() * 8, ctx().stream)
    _lib.lib().call("emrt_memcpy", None, None, 0, ctx().stream)
    torch.cuda.synchronize()
    assert torch.equal(dst, src)


@pytest.mark.parametrize("dtype", DTYPES)
def test_pack_weights_layouts(dtype):
    c = init(dtype)
    g = torch.Generator().manual_seed(16)
    conv = hnn.Conv2D(40, 72, 3, 1, 1, bias=False)
    with torch.no_grad():
        conv.weight.copy_(rnd(torch.randn(72, 40, 3, 3, generator=g)))
    w = conv.weight.detach().clone()
    h = Holder(conv=conv).place()
    esz = 4 if dtype == F32 else 2
    base = h.store.packed.data_ptr()
    if dtype == BF16:
        off = (conv.gw.fwd_ptr - base) // esz
        fwd = h.store.packed[off:off + w.numel()].float().cpu().view(72, 3, 3, 40)
        assert torch.equal(fwd, w.permute(0, 2, 3, 1))
    off = (conv.gw.bwd_ptr - base) // esz
    bwd = h.store.packed[off:off + w.numel()].float().cpu().view(40, 3, 3, 72)
    assert torch.equal(bwd, w.permute(1, 2, 3, 0))


@pytest.mark.parametrize("dtype", DTYPES)
def test_pack_bwd_only_rebuilds_the_dgrad_copies(dtype):
    """The per-step form of the weight pack (64x64 tiles transposed FROM the forward operand the optimizer keeps current) must
    reproduce exactly what the full pack writes: shapes with ragged tiles (C = 3, 40; OC = 6, 72, 432), 1x1 / 3x3 / 7x7 taps."""
    c = init(dtype)
    g = torch.Generator().manual_seed(17)
    layers = dict(a=hnn.Conv2D(40, 72, 3, 1, 1, bias=False), b=hnn.Conv2D(3, 64, 7, 2, 3, bias=False), c=hnn.Conv2D(256, 6, 1),
                  d=hnn.Linear(256, 432), e=hnn.Conv2D(128, 128, 3, 1, 1, bias=False))
    with torch.no_grad():
        for m in layers.values():
            m.weight.copy_(rnd(torch.randn(m.weight.shape, generator=g)))
    h = Holder(**layers).place()
    st = h.store
    torch.cuda.synchronize()
    want = st.packed.clone()
    lo = st.mirror_elems if dtype != F32 else 0
    st.packed[lo:].fill_(7.0)                     # the dgrad copies are garbage; the forward mirror (bf16) / master (fp32) is intact
    st.pack(bwd_only=True)
    torch.cuda.synchronize()
    assert any(g_.bwd_ptr is not None for g_ in st.gemms)
    for g_ in st.gemms:
        if g_.bwd_ptr is None:
            continue
        esz = 4 if dtype == F32 else 2
        o = (g_.bwd_ptr - st.packed.data_ptr()) // esz
        n = g_.OC * g_.KH * g_.KW * g_.C
        assert torch.equal(st.packed[o:o + n], want[o:o + n]), (g_.OC, g_.C, g_.KH)


@pytest.mark.parametrize("dtype", DTYPES)
def test_msda_golden_vectors(dtype):
    """The committed golden vectors (numpy-f64 oracle) through the HIP kernel: locations enter as offsets from ref = 0."""
    import os
    import numpy as np
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "msda_core.npz"))
    c = init(dtype)
    shapes = [tuple(int(v) for v in s) for s in z["shapes"]]
    value, loc, aw = torch.from_numpy(z["value"]), torch.from_numpy(z["loc"]), torch.from_numpy(z["aw"])
    B, Lq, M, L, Pn, _ = loc.shape
    norm = torch.tensor([[w, h] for h, w in shapes], dtype=torch.float32).reshape(1, 1, 1, L, 1, 2)
    off = (loc * norm).reshape(B, Lq, M * L * Pn * 2)          # loc = 0 + off / (W, H)
    logits = torch.log(aw.reshape(B, Lq, M * L * Pn))           # softmax(log p) = p (p sums to 1 per head)
    offw = torch.cat([off, logits], -1)
    ref = torch.zeros(1, Lq, 1, 2)
    y = Fn.msda(dev(rnd(value).reshape(B, -1, M * 32)), dev(offw, torch.float32), dev(ref, torch.float32), shapes, M, Pn)
    want = torch.from_numpy(z["out"]) if dtype == F32 else None
    if dtype == F32:
        close("msda golden", host(y), want, dtype, atol=2e-5, rtol=1e-4)
    else:
        close("msda golden", host(y), torch.from_numpy(z["out"]), dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("H,W,C,Ctot", [(16, 16, 256, 256), (64, 64, 256, 1536), (48, 80, 128, 128), (21, 30, 64, 64)],
                         ids=["small-map-one-block-per-bin", "512-tile-map-slice-of-concat", "non-square", "odd-size"])
def test_adaptive_pool_split_bins(dtype, H, W, C, Ctot):
    """AdaptiveAvgPool2D pyramid (paddle_EMRT.py:62,70-78) on maps below and above the size at which a bin is pooled by several blocks
    (csrc/spatial.hip: adaptive_pool_part_kernel), on a channel slice of a wider buffer as in the model, non-square and odd sizes."""
    c = init(dtype)
    g = torch.Generator().manual_seed(5)
    N = 3
    scales = [1, 3, 6, 8]
    x = rnd(torch.randn(N, Ctot, H, W, generator=g))
    xr = x[:, :C].clone().requires_grad_(True)
    toks = torch.cat([F.adaptive_avg_pool2d(xr, k).reshape(N, C, -1) for k in scales], -1).transpose(1, 2)
    dy = rnd(torch.randn(toks.shape, generator=g))
    toks.backward(dy)
    xd = dev_map(x)[..., :C]
    tape = Tape()
    c.tape = tape
    y = Fn.adaptive_avgpool_tokens(xd, scales)
    c.tape = None
    tape.watch(xd)
    close("pool fwd %dx%d" % (H, W), host(y), toks.detach(), dtype)
    # several blocks per bin, but no atomics: partial sums go to their own slots and are added in part order, so a second evaluation
    # (other block scheduling) gives the same bits -- inference evaluates a tile twice (slide_inference / ss_inference) and compares argmax
    for _ in range(3):
        assert torch.equal(Fn.adaptive_avgpool_tokens(xd, scales), y)
    dx, = run_bwd(tape, [(y, dev(dy))], [xd])          # (the bin-membership table kernel: rows / columns that lie in two overlapping bins)
    close("pool dx %dx%d" % (H, W), host_map(dx), xr.grad, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_pyramid_pooling_branches_as_grouped_launches(dtype):
    """The four pyramid-pooling branches conv1x1 -> BatchNorm(train) -> ReLU on 1 / 9 / 36 / 64 pooled tokens per image (paddle_EMRT.py:61-66,70-78) as ONE
    grouped launch per pass (functional.conv_bn_small_group: emrt_conv2d_group with the statistics in its epilogue, emrt_bn_group_apply; backward
    emrt_bn_group_bwd + emrt_conv2d_bwd_group) against one launch per branch and pass: forward bit-identical (same arithmetic), every gradient and the
    running statistics equal to reduction-order noise; and against torch."""
    from emrt_amd import _lib
    B, C = 8, 256
    scales = [1, 3, 6, 8]
    g = torch.Generator().manual_seed(55)
    tok = torch.randn(B, sum(k * k for k in scales), C, generator=g)
    ws = [torch.randn(C, C, 1, 1, generator=g) / math.sqrt(C) for _ in scales]
    gams = [torch.rand(C, generator=g) + 0.5 for _ in scales]
    bets = [torch.randn(C, generator=g) * 0.3 for _ in scales]
    dys = [torch.randn(B, k * k, C, generator=g) for k in scales]
    L = _lib.lib()

    def run(grouped):
        c = init(dtype)
        c.bn_small_group = grouped
        convs = [hnn.Conv2D(C, C, 1, bias=False) for _ in scales]
        bns = [hnn.BatchNorm2D(C) for _ in scales]
        with torch.no_grad():
            for cv, b, w_, ga, be in zip(convs, bns, ws, gams, bets):
                cv.weight.copy_(rnd(w_))
                b.weight.copy_(ga)
                b.bias.copy_(be)
        Holder(**{"c%d" % i: m_ for i, m_ in enumerate(convs)}, **{"b%d" % i: m_ for i, m_ in enumerate(bns)}).place()
        td = dev(rnd(tok))
        tape = Tape()
        c.tape = tape
        L.start_record()
        slices, s0 = [], 0
        for k in scales:
            slices.append(Fn.narrow(td, 1, s0, k * k))
            s0 += k * k
        outs = Fn.conv_bn_group(convs, bns, slices)
        names = [n for n, _ in L.stop_record() if n != "emrt_memset"]      # (outside a training step the fp64 sums are zeroed one by one)
        c.tape = None
        tape.watch(td)
        L.start_record()
        dtok, = run_bwd(tape, [(o, dev(rnd(d_))) for o, d_ in zip(outs, dys)], [td])
        names_b = [n for n, _ in L.stop_record()]
        torch.cuda.synchronize()
        c.bn_small_group = True
        return names, names_b, ([host(o) for o in outs] + [host(dtok)] + [host(cv.weight.grad) for cv in convs] + [host(b.weight.grad) for b in bns]
                                + [host(b.bias.grad) for b in bns] + [host(b._buffers["_mean"]) for b in bns] + [host(b._buffers["_variance"]) for b in bns])

    n1, nb1, r1 = run(True)
    n0, nb0, r0 = run(False)
    assert n1 == ["emrt_conv2d_group", "emrt_bn_group_apply"], n1
    assert n0.count("emrt_conv2d") == 4 and n0.count("emrt_bn_apply") == 4, n0
    assert nb1.count("emrt_bn_group_bwd") == 1 and nb1.count("emrt_conv2d_bwd_group") == 1 and "emrt_bn_bwd_dx" not in nb1, nb1
    assert nb0.count("emrt_bn_bwd_dx") == 4, nb0
    for i in range(4):
        assert torch.equal(r1[i], r0[i]), "branch %d: grouped forward differs from emrt_bn_apply's" % i
    tol = 2e-2 if dtype == BF16 else 2e-5
    for u, v in zip(r1[4:], r0[4:]):
        assert torch.isfinite(u).all()
        assert (u - v).abs().max().item() <= tol * max(1.0, v.abs().max().item()), (u - v).abs().max().item()
    # ... and torch (fp32): branch 2
    if dtype == F32:
        a, b_ = sum(k * k for k in scales[:2]), sum(k * k for k in scales[:3])
        xr = tok[:, a:b_].clone().requires_grad_(True)
        y = F.conv2d(xr.permute(0, 2, 1).unsqueeze(-1), ws[2])
        o = F.relu(F.batch_norm(y, None, None, gams[2], bets[2], True, 0.1, 1e-5)).squeeze(-1).permute(0, 2, 1)
        o.backward(dys[2])
        close("grouped branch vs torch", r1[2], o.detach(), dtype, 4.0)
        close("grouped d tokens vs torch", r1[4][:, a:b_], xr.grad, dtype, 8.0)


@pytest.mark.parametrize("dtype", DTYPES)
def test_efp_conv2d_blocks_level_by_level_in_grouped_launches(dtype):
    """EFP's three Conv2dBlocks (paddle_EMRT.py:13-48: relu(bn(conv3x3(relu(bn(conv3x3(x)))))) + x on the three pyramid levels, whose inputs are strided
    level slabs of the token tensor) as grouped launches -- conv1 of all levels | BatchNorm + ReLU | conv2 | BatchNorm + ReLU + x (the residual added by the
    BatchNorm launch, the ReLU mask re-derived from the raw map in backward) -- against one Conv2dBlock at a time, and against torch in fp32."""
    from emrt_amd import _lib
    from emrt_amd.src.models.emrt import EFP
    B, C = 2, 256
    shapes = [(16, 16), (8, 8), (4, 4)]
    Lv = sum(h * w for h, w in shapes)
    g = torch.Generator().manual_seed(77)
    mem = torch.randn(B, Lv, C, generator=g)
    wts = [torch.randn(C, C, 3, 3, generator=g) / math.sqrt(9 * C) for _ in range(6)]
    dy = torch.randn(B, C, 16, 16, generator=g)
    L = _lib.lib()

    def run(grouped):
        c = init(dtype)
        c.bn_small_group = grouped
        efp = EFP(C, C)
        convs = [efp.conv0.conv1[0], efp.conv0.conv2[0], efp.conv1.conv1[0], efp.conv1.conv2[0], efp.conv2.conv1[0], efp.conv2.conv2[0]]
        with torch.no_grad():
            for cv, w_ in zip(convs, wts):
                cv.weight.copy_(rnd(w_))
        Holder(efp=efp).place()
        md = dev(rnd(mem))
        tape = Tape()
        c.tape = tape
        maps, s0 = [], 0
        for h, w_ in shapes:
            maps.append(Fn.tokens_as_map(Fn.narrow(md, 1, s0, h * w_), h, w_))
            s0 += h * w_
        out = c.empty((B, 16, 16, C))
        L.start_record()
        y = efp(maps[0], maps[1], maps[2], out=out)
        names = [n for n, _ in L.stop_record() if n != "emrt_memset"]
        c.tape = None
        tape.watch(md)
        dmem, = run_bwd(tape, [(y, dev_map(rnd(dy)))], [md])
        torch.cuda.synchronize()
        c.bn_small_group = True
        return names, [host_map(y), host(dmem)] + [host(cv.weight.grad) for cv in convs]

    n1, r1 = run(True)
    n0, r0 = run(False)
    assert n1.count("emrt_conv2d_group") == 2 and n1.count("emrt_bn_group_apply") == 2 and "emrt_add3d" not in n1 and "emrt_bn_apply" not in n1, n1
    assert "emrt_conv2d_group" not in n0 and n0.count("emrt_add3d") == 3, n0
    for u, v, lab in zip(r1, r0, ["out", "d memory"] + ["dW%d" % i for i in range(6)]):
        assert torch.isfinite(u).all(), lab
        if dtype == F32:
            assert (u - v).abs().max().item() <= 2e-5 * max(1.0, v.abs().max().item()), (lab, (u - v).abs().max().item())
        else:
            # bf16: the grouped form rounds relu(bn(.)) + x once where the separate add rounds twice; with 32 ... 512 rows per BatchNorm a last-bit change
            # flips a few ReLUs, and each flip moves single weight-gradient elements by O(1) (both forms are ~25 % max-norm from fp32 torch here, and equal
            # to each other in the norm): the sharp comparison is the fp32 one
            rel = ((u - v).norm() / v.norm().clamp_min(1e-20)).item()
            assert rel < 0.2, (lab, rel)
    if dtype == F32:
        xr = mem.clone().requires_grad_(True)
        lv, s0 = [], 0
        for h, w_ in shapes:
            lv.append(xr[:, s0:s0 + h * w_].transpose(1, 2).reshape(B, C, h, w_))
            s0 += h * w_

        def blk(x, wa, wb):
            a = F.relu(F.batch_norm(F.conv2d(x, wa, padding=1), None, None, None, None, True, 0.1, 1e-5))
            return F.relu(F.batch_norm(F.conv2d(a, wb, padding=1), None, None, None, None, True, 0.1, 1e-5)) + x
        o0, o1, o2 = blk(lv[0], wts[0], wts[1]), blk(lv[1], wts[2], wts[3]), blk(lv[2], wts[4], wts[5])
        x21 = F.interpolate(o2, size=(8, 8), mode="bilinear", align_corners=True) + o1
        o = F.interpolate(x21, size=(16, 16), mode="bilinear", align_corners=True) + o0
        o.backward(dy)
        close("EFP grouped vs torch", r1[0], o.detach(), dtype, 8.0)
        close("EFP grouped d memory vs torch", r1[1], xr.grad, dtype, 40.0)


# -----------------------------------------------------------------------------------------------------------------
# BatchNorm + ReLU between two convolutions applied by the CONSUMING convolution's operand loads (emrt_conv2d_bna; csrc/conv.hip: igemm_body BNA)
# -----------------------------------------------------------------------------------------------------------------
BNA_CASES = [
    # name, N, H, W, Cin, C (the BatchNorm's channels), OC, k, dilation, knobs forcing the kernel variant, consumer ReLU
    ("layer1 bn2->conv3 1x1, plain tile", 2, 32, 32, 64, 64, 256, 1, 1, {}, False),
    ("layer1 bn1->conv2 3x3, plain tile", 2, 32, 32, 32, 64, 64, 3, 1, {}, False),
    ("ragged M, 3x3, two wave groups", 3, 7, 9, 64, 128, 128, 3, 1, {"conv_tile": 5}, False),
    ("ragged M, 3x3, four wave groups", 3, 7, 9, 64, 256, 256, 3, 1, {"conv_tile": 6}, False),
    ("layer4-like 3x3, cross-block K split x3", 2, 8, 8, 64, 512, 512, 3, 1, {"xk": 3}, False),
    ("1x1, cross-block K split x2", 2, 8, 8, 64, 512, 128, 1, 1, {"xk": 2}, False),
    ("dilated 3x3 (resnet50c)", 2, 12, 12, 64, 128, 128, 3, 2, {}, False),
    ("1x1 with ragged OC and consumer ReLU", 1, 10, 14, 64, 128, 72, 1, 1, {}, True),
    ("auto dispatch at a layer3 shape (3x3, 256 ch, 16x16)", 8, 16, 16, 64, 256, 256, 3, 1, {}, False),
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", BNA_CASES, ids=[c_[0] for c_ in BNA_CASES])
def test_batchnorm_relu_applied_by_the_consuming_convolutions_loads(dtype, case):
    """conv -> BatchNorm(train) -> ReLU -> conv -> BatchNorm (paddle_vision_resnet.py:129-149 bn1 -> relu -> conv2, bn2 -> relu -> conv3; paddle_EMRT.py:16-23,
    201-209) with the first BatchNorm applied by the SECOND convolution's operand loads: the raw map goes through relu(x * scale + shift) between the
    global load and the LDS write, the first tile column writes the normalised map on the way, and no emrt_bn_apply is launched.  The arithmetic is the
    separate launch's (same fmaf, max, rounding), so EVERYTHING must be bit-identical to the two-launch form: the consumer's output and its batch sums
    (through the second BatchNorm's output), the normalised map (through every gradient of the backward, which reads it), saved mean / invstd and the
    running statistics.  Negative gammas, padding taps (zeros of the NORMALISED map), ragged M / OC, dilation, every kernel variant forced in turn."""
    from emrt_amd import _lib
    name, N, H, W, Cin, C, OC, k, dil, knobs, relu2 = case
    g = torch.Generator().manual_seed(123)
    x = torch.randn(N, Cin, H, W, generator=g)
    w1 = torch.randn(C, Cin, 1, 1, generator=g) / math.sqrt(Cin)
    w2 = torch.randn(OC, C, k, k, generator=g) / math.sqrt(C * k * k)
    gam = torch.rand(C, generator=g) + 0.5
    gam[::3] *= -1.0
    bet = torch.randn(C, generator=g) * 0.3
    dy = torch.randn(N, OC, H, W, generator=g)
    L = _lib.lib()
    with_bn2 = 256 % (OC // 4) == 0 if OC % 4 == 0 else False

    def run(fused):
        c = init(dtype)
        c.bn_conv = fused
        conv1, bn1 = hnn.Conv2D(Cin, C, 1, bias=False), hnn.BatchNorm2D(C)
        conv2, bn2 = hnn.Conv2D(C, OC, k, 1, dil * (k // 2), bias=False, dilation=dil), hnn.BatchNorm2D(OC)
        with torch.no_grad():
            conv1.weight.copy_(rnd(w1))
            conv2.weight.copy_(rnd(w2))
            bn1.weight.copy_(gam)
            bn1.bias.copy_(bet)
        Holder(conv1=conv1, bn1=bn1, conv2=conv2, bn2=bn2).place()
        xd = dev_map(rnd(x))
        old = [(kk, L.set_tuning(kk, v)) for kk, v in list(knobs.items()) + [("no_bna", -1)]]      # (-1: also the long-k 3x3 layers the dispatcher declines)
        try:
            tape = Tape()
            c.tape = tape
            L.start_record()
            a = Fn.conv_bn(conv1, bn1, xd, relu=True, defer="conv")
            if with_bn2:
                out = Fn.conv_bn(conv2, bn2, a, relu=relu2)
            else:      # (a channel count emrt_bn_apply does not take: the consumer alone, with its ReLU in the GEMM epilogue)
                out = Fn.conv2d(a, conv2.gw, 1, dil * (k // 2), relu=relu2, dilation=dil)
            names = [n for n, _ in L.stop_record()]
            c.tape = None
            tape.watch(xd)
            dx, = run_bwd(tape, [(out, dev_map(rnd(dy)))], [xd])
            torch.cuda.synchronize()
        finally:
            for kk, v in old:
                L.set_tuning(kk, v)
            c.bn_conv = True
        return names, [host_map(out), host_map(dx), host(conv1.weight.grad), host(conv2.weight.grad), host(bn1.weight.grad), host(bn1.bias.grad),
                       host(bn2.weight.grad), host(bn2.bias.grad), host(bn1._buffers["_mean"]), host(bn1._buffers["_variance"]),
                       host(bn2._buffers["_mean"]), host(bn2._buffers["_variance"])]

    names, res = run(True)
    names0, res0 = run(False)
    assert names.count("emrt_conv2d_bna") == 1 and names.count("emrt_bn_apply") == int(with_bn2), names          # (the one left is the second BatchNorm's)
    assert names0.count("emrt_conv2d_bna") == 0 and names0.count("emrt_bn_apply") == 1 + int(with_bn2), names0
    labels = ("out", "dx", "dW1", "dW2", "dgamma1", "dbeta1", "dgamma2", "dbeta2", "running mean 1", "running var 1", "running mean 2", "running var 2")
    for u, v, lab in zip(res, res0, labels):
        assert torch.isfinite(u).all(), lab
        if lab in ("dW1", "dW2"):      # (batched weight gradients: fp32 atomics, the order of the last bits is free)
            assert ((u - v).norm() / v.norm().clamp_min(1e-20)).item() < 2e-6, lab
        else:
            assert torch.equal(u, v), "%s: %s differs from the two-launch form by up to %.3g" % (name, lab, (u - v).abs().max().item())
    # ... and the two-launch form is what torch computes (fp32)
    if dtype == F32:
        xr = x.clone()
        y1 = F.conv2d(xr, w1)
        a1 = F.relu(F.batch_norm(y1, None, None, gam, bet, True, 0.1, 1e-5))
        y2 = F.conv2d(a1, w2, padding=dil * (k // 2), dilation=dil)
        o = F.batch_norm(y2, None, None, None, None, True, 0.1, 1e-5) if with_bn2 else y2
        if relu2:
            o = F.relu(o)
        close("fused chain vs torch", res[0], o, dtype, 4.0)


def test_conv2d_bna_is_refused_outside_its_kernels():
    """emrt_conv2d_bna_supported answers 0 -- and the host then launches emrt_bn_apply itself (PendingBN.materialize) -- for a stride-2 consumer, for the
    thin OC <= 32 tile, and with the A/B knob; emrt_conv2d_bna itself fails loudly on such a layer instead of running something else."""
    from emrt_amd import _lib
    L = _lib.lib()
    c = init(BF16)
    g = torch.Generator().manual_seed(4)
    for (C, OC, k, stride, knob) in ((64, 64, 3, 2, None), (64, 16, 1, 1, None), (64, 64, 3, 1, "no_bna"), (256, 64, 3, 1, None)):
        conv1, bn1 = hnn.Conv2D(32, C, 1, bias=False), hnn.BatchNorm2D(C)
        conv2, bn2 = hnn.Conv2D(C, OC, k, stride, k // 2, bias=False), hnn.BatchNorm2D(OC)
        Holder(conv1=conv1, bn1=bn1, conv2=conv2, bn2=bn2).place()
        xd = dev_map(rnd(torch.randn(2, 32, 16, 16, generator=g)))
        old = L.set_tuning(knob, 1) if knob else None
        try:
            tape = Tape()
            c.tape = tape
            L.start_record()
            out = Fn.conv_bn(conv2, bn2, Fn.conv_bn(conv1, bn1, xd, relu=True, defer="conv"), relu=True)
            names = [n for n, _ in L.stop_record()]
            c.tape = None
        finally:
            if knob:
                L.set_tuning(knob, old)
        assert "emrt_conv2d_bna" not in names and names.count("emrt_bn_apply") == 2 and torch.isfinite(out.float()).all(), names


# -----------------------------------------------------------------------------------------------------------------
# BatchNorm + ReLU applied by the loads of a streaming consumer (functional.PendingBN, csrc/bn_operand.hpp)
# -----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("consumer", ["resize", "maxpool"])
@pytest.mark.parametrize("case", [(2, 16, 16, 64, 64, 3, 1), (3, 11, 13, 32, 256, 1, 0), (1, 7, 7, 64, 128, 3, 1), (2, 20, 12, 16, 8, 3, 1)])
def test_batchnorm_relu_applied_by_the_consumers_loads(dtype, consumer, case):
    """conv -> BatchNorm(train) -> ReLU -> {x2 bilinear resize | 3x3/2 max-pool} (paddle_EMRT.py:164-175, paddle_vision_resnet.py:199-201)
    with the BatchNorm deferred into the consumer: no emrt_bn_apply launch, forward identical to the three-launch path in fp32 (same
    expression on the same values), gradients against torch, running statistics updated, and the ReLU mask re-derived from the raw map in
    backward (some gammas are negative: the affine map is not monotone, the max-pool must transform every tap)."""
    from emrt_amd import _lib
    N, H, W, Cin, C, k, pad = case
    g = torch.Generator().manual_seed(77)
    x = torch.randn(N, Cin, H, W, generator=g)
    wt = torch.randn(C, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
    gam = torch.rand(C, generator=g) + 0.5
    gam[::3] *= -1.0
    bet = torch.randn(C, generator=g) * 0.3

    def run(defer):
        c = init(dtype)
        c.bn_defer = defer
        conv, bn = hnn.Conv2D(Cin, C, k, 1, pad, bias=False), hnn.BatchNorm2D(C)
        with torch.no_grad():
            conv.weight.copy_(rnd(wt))
            bn.weight.copy_(gam)
            bn.bias.copy_(bet)
        Holder(conv=conv, bn=bn).place()
        xd = dev_map(rnd(x))
        tape = Tape()
        c.tape = tape
        L = _lib.lib()
        L.start_record()
        a = Fn.conv_bn(conv, bn, xd, relu=True, defer=True)
        if consumer == "resize":
            out = Fn.resize_bilinear(a, 2 * H, 2 * W, False)
        else:
            out = Fn.maxpool(a, 3, 2, 1)
        names = [n for n, _ in L.stop_record()]
        c.tape = None
        tape.watch(xd)
        dy = rnd(torch.randn(N, C, out.shape[1], out.shape[2], generator=torch.Generator().manual_seed(5)))
        dx, = run_bwd(tape, [(out, dev_map(dy))], [xd])
        torch.cuda.synchronize()
        c.bn_defer = True
        return (names, host_map(out), host_map(dx), host(conv.weight.grad), host(bn.weight.grad), host(bn.bias.grad),
                host(bn._buffers["_mean"]), host(bn._buffers["_variance"]), dy)

    names, out, dx, dw, dgam, dbet, rm, rv, dy = run(True)
    names0, out0, dx0, dw0, dgam0, dbet0, rm0, rv0, _ = run(False)
    assert "emrt_bn_apply" not in names and names0.count("emrt_bn_apply") == 1
    assert ("emrt_bn_resize_bilinear_fwd" if consumer == "resize" else "emrt_bn_maxpool_fwd") in names
    # torch, fp32, same rounded inputs
    xr = rnd(x).clone().requires_grad_(True)
    wr = rnd(wt).clone().requires_grad_(True)
    gr, br_ = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    y = F.conv2d(xr, wr, None, padding=pad)
    act = F.relu(F.batch_norm(y, None, None, gr, br_, True, 0.1, 1e-5))
    o = F.interpolate(act, scale_factor=2, mode="bilinear", align_corners=False) if consumer == "resize" else F.max_pool2d(act, 3, 2, 1)
    o.backward(dy)
    close("deferred fwd", out, o.detach(), dtype)
    if dtype == F32:
        # same values through the same affine expression; the resize's interpolation may contract its multiply-adds differently in the
        # two kernels (an ulp), the max-pool has nothing to contract
        assert (out - out0).abs().max() <= (1e-6 * out0.abs().max() if consumer == "resize" else 0.0), (out - out0).abs().max()
        assert torch.equal(rm, rm0) and torch.equal(rv, rv0)
    else:
        close("deferred vs separate fwd", out, out0, dtype)      # (the separate path rounds the normalised map to bf16 first)
    sc = math.sqrt(N * H * W)
    ksc = math.sqrt(C * k * k)
    if dtype == F32:
        close("deferred dx", dx, xr.grad, dtype, 2.0 * ksc)
        close("deferred dgamma", dgam, gr.grad, dtype, sc)
        close("deferred dbeta", dbet, br_.grad, dtype, sc)
        close("deferred dw", dw, wr.grad, dtype, sc)
        close("deferred vs separate dx", dx, dx0, dtype, 2.0 * ksc * 0.05)
        close("deferred vs separate dgamma", dgam, dgam0, dtype, sc * 0.05)
        close("deferred vs separate dbeta", dbet, dbet0, dtype, sc * 0.05)
    else:
        # bf16: the raw conv output is rounded to bf16 before BatchNorm on BOTH HIP paths, which flips the ReLU of a few near-zero
        # activations against fp32 torch (each flip moves dbeta by one |dy| ~ 1), and dx is rounded before the weight gradient: norms
        def rel(a, b):
            return ((a - b).norm() / b.norm()).item()
        errs = {"dx": rel(dx, xr.grad), "dgamma": rel(dgam, gr.grad), "dbeta": rel(dbet, br_.grad), "dw": rel(dw, wr.grad),
                "dx/sep": rel(dx, dx0), "dgamma/sep": rel(dgam, dgam0), "dbeta/sep": rel(dbet, dbet0), "dw/sep": rel(dw, dw0)}
        print("deferred BatchNorm bf16 relative errors:", {k: "%.2e" % v for k, v in errs.items()})
        # the max-pool picks its winner among fp32 values here and among bf16-rounded ones on the separate path (torch: among unrounded
        # ones): near-ties go to different taps, which re-routes whole gradient elements -- the fp32 runs above are the exact check
        route = 1e-1 if consumer == "maxpool" else 0.0
        assert all(v < max(route if k.startswith(("dx", "dw")) else 0.0, 3e-2 if "/" not in k else 1.5e-2) for k, v in errs.items()), errs
    mean = y.detach().mean(dim=(0, 2, 3))
    var = y.detach().var(dim=(0, 2, 3), unbiased=False)
    close("deferred run_mean", rm, 0.1 * mean, F32, atol=2e-3 if dtype == BF16 else 1e-4)
    close("deferred run_var", rv, 0.9 + 0.1 * var, F32, atol=5e-3 if dtype == BF16 else 1e-3)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(2, 16, 16, 64, 256, 6, True), (3, 9, 13, 32, 128, 8, False), (1, 20, 20, 32, 64, 3, True), (2, 33, 17, 64, 256, 2, True)])
def test_batchnorm_relu_applied_by_the_classifiers_loads(dtype, case):
    """conv3x3 -> BatchNorm(train) -> ReLU -> conv1x1 to <= 8 channels (paddle_EMRT.py:176-179) with the BatchNorm deferred into the
    classifier (emrt_bn_pointwise_fwd / _bwd): no emrt_bn_apply, no normalised map, logits and every gradient against torch and against
    the separate path."""
    from emrt_amd import _lib
    N, H, W, Cin, C, OC, bias = case
    g = torch.Generator().manual_seed(91)
    x = torch.randn(N, Cin, H, W, generator=g)
    wt = torch.randn(C, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    w2 = torch.randn(OC, C, 1, 1, generator=g) / math.sqrt(C)
    b2 = torch.randn(OC, generator=g) * 0.1
    gam = torch.rand(C, generator=g) + 0.5
    gam[::4] *= -1.0
    bet = torch.randn(C, generator=g) * 0.3
    dy = torch.randn(N, OC, H, W, generator=g)

    def run(defer):
        c = init(dtype)
        c.bn_defer = defer
        conv, bn, cls = hnn.Conv2D(Cin, C, 3, 1, 1, bias=False), hnn.BatchNorm2D(C), hnn.Conv2D(C, OC, 1, bias=bias)
        with torch.no_grad():
            conv.weight.copy_(rnd(wt))
            bn.weight.copy_(gam)
            bn.bias.copy_(bet)
            cls.weight.copy_(rnd(w2))
            if bias:
                cls.bias.copy_(b2)
        Holder(conv=conv, bn=bn, cls=cls).place()
        xd = dev_map(rnd(x))
        tape = Tape()
        c.tape = tape
        L = _lib.lib()
        L.start_record()
        a = Fn.conv_bn(conv, bn, xd, relu=True, defer=Fn.pointwise_takes_pending(cls.gw, C))
        out = cls(a)
        c.tape = None
        tape.watch(xd)
        dx, = run_bwd(tape, [(out, dev_map(rnd(dy)))], [xd])
        names = [n for n, _ in L.stop_record()]
        torch.cuda.synchronize()
        c.bn_defer = True
        return (names, host_map(out), host_map(dx), host(conv.weight.grad), host(bn.weight.grad), host(bn.bias.grad), host(cls.weight.grad),
                host(cls.bias.grad) if bias else None, host(bn._buffers["_mean"]), host(bn._buffers["_variance"]))

    names, out, dx, dw, dgam, dbet, dw2, db2, rm, rv = run(True)
    names0, out0, dx0, dw0, dgam0, dbet0, dw20, db20, rm0, rv0 = run(False)
    assert "emrt_bn_apply" not in names and "emrt_bn_pointwise_fwd" in names and "emrt_bn_pointwise_bwd" in names
    assert "emrt_bn_bwd_reduce" not in names and names0.count("emrt_bn_apply") == 1
    xr, wr, w2r = rnd(x).clone().requires_grad_(True), rnd(wt).clone().requires_grad_(True), rnd(w2).clone().requires_grad_(True)
    b2r = b2.clone().requires_grad_(True)
    gr, br_ = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    y = F.conv2d(xr, wr, None, padding=1)
    o = F.conv2d(F.relu(F.batch_norm(y, None, None, gr, br_, True, 0.1, 1e-5)), w2r, b2r if bias else None)
    o.backward(rnd(dy))
    close("classifier fwd", out, o.detach(), dtype)
    assert torch.equal(rm, rm0) and torch.equal(rv, rv0)

    def rel(a, b):
        return ((a - b).norm() / b.norm()).item()
    errs = {"dx": rel(dx, xr.grad), "dw": rel(dw, wr.grad), "dgamma": rel(dgam, gr.grad), "dbeta": rel(dbet, br_.grad), "dw2": rel(dw2, w2r.grad),
            "out/sep": rel(out, out0), "dx/sep": rel(dx, dx0), "dw/sep": rel(dw, dw0), "dgamma/sep": rel(dgam, dgam0), "dbeta/sep": rel(dbet, dbet0),
            "dw2/sep": rel(dw2, dw20)}
    if bias:
        errs["db2"] = rel(db2, b2r.grad)
    print("deferred classifier relative errors:", {k: "%.2e" % v for k, v in errs.items()})
    tol = 2e-5 if dtype == F32 else 3e-2
    assert all(v < tol for v in errs.values()), errs


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(2, 16, 16, 64, 256, 1), (2, 12, 12, 128, 512, 2), (3, 9, 9, 64, 64, 1)])
def test_shortcut_batchnorm_applied_by_the_join(dtype, case):
    """The first block of a ResNet stage: out = relu(BN3(conv3(a)) + BN_d(conv_d(x))) (paddle_vision_resnet.py:132-147, 226-233) with the
    shortcut's BatchNorm applied by the join's loads (emrt_bn_apply_join).  Same bits as the two-launch path, forward and backward (the
    normalised shortcut is rounded to the storage type before the add exactly as the separate launch stores it), and against torch."""
    from emrt_amd import _lib
    N, H, W, Cin, C, stride = case
    g = torch.Generator().manual_seed(57)
    x = torch.randn(N, Cin, H, W, generator=g)
    a = torch.randn(N, Cin, (H - 1) // stride + 1, (W - 1) // stride + 1, generator=g)
    w3 = torch.randn(C, Cin, 1, 1, generator=g) / math.sqrt(Cin)
    wd = torch.randn(C, Cin, 1, 1, generator=g) / math.sqrt(Cin)
    g3, b3 = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    gd, bd = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    dy = torch.randn(N, C, a.shape[2], a.shape[3], generator=g)

    def run(defer):
        c = init(dtype)
        c.bn_defer = defer
        conv3, bn3 = hnn.Conv2D(Cin, C, 1, bias=False), hnn.BatchNorm2D(C)
        convd, bnd = hnn.Conv2D(Cin, C, 1, stride, 0, bias=False), hnn.BatchNorm2D(C)
        with torch.no_grad():
            conv3.weight.copy_(rnd(w3)); convd.weight.copy_(rnd(wd))
            bn3.weight.copy_(g3); bn3.bias.copy_(b3); bnd.weight.copy_(gd); bnd.bias.copy_(bd)
        Holder(conv3=conv3, bn3=bn3, convd=convd, bnd=bnd).place()
        xd, ad = dev_map(rnd(x)), dev_map(rnd(a))
        tape = Tape()
        c.tape = tape
        L = _lib.lib()
        L.start_record()
        identity = Fn.conv_bn(convd, bnd, xd, defer="join")
        out = Fn.conv_bn(conv3, bn3, ad, relu=True, residual=identity)
        c.tape = None
        tape.watch(xd)
        tape.watch(ad)
        dx, da = run_bwd(tape, [(out, dev_map(rnd(dy)))], [xd, ad])
        names = [n for n, _ in L.stop_record()]
        torch.cuda.synchronize()
        c.bn_defer = True
        return (names, [host_map(out), host_map(dx), host_map(da), host(conv3.weight.grad), host(convd.weight.grad), host(bn3.weight.grad),
                        host(bn3.bias.grad), host(bnd.weight.grad), host(bnd.bias.grad), host(bnd._buffers["_mean"]), host(bnd._buffers["_variance"])])

    names, got = run(True)
    names0, sep = run(False)
    assert names.count("emrt_bn_apply_join") == 1 and names.count("emrt_bn_apply") == 0 and names0.count("emrt_bn_apply") == 2
    labels = ["out", "dx", "da", "dw3", "dwd", "dgamma3", "dbeta3", "dgammad", "dbetad", "run_mean_d", "run_var_d"]
    for lab, u, v in zip(labels, got, sep):
        if lab in ("dw3", "dwd"):          # (weight gradients ride fp32 atomics: last-bit order effects)
            assert ((u - v).norm() / v.norm()).item() < 1e-5, lab
        else:
            assert torch.equal(u, v), (lab, (u - v).abs().max())
    xr, ar = rnd(x).clone().requires_grad_(True), rnd(a).clone().requires_grad_(True)
    o = F.relu(F.batch_norm(F.conv2d(ar, rnd(w3)), None, None, g3, b3, True, 0.1, 1e-5) +
               F.batch_norm(F.conv2d(xr, rnd(wd), stride=stride), None, None, gd, bd, True, 0.1, 1e-5))
    o.backward(rnd(dy))
    tol = 2e-5 if dtype == F32 else 3e-2
    for lab, u, v in (("out", got[0], o.detach()), ("dx", got[1], xr.grad), ("da", got[2], ar.grad)):
        assert ((u - v).norm() / v.norm()).item() < tol, (lab, ((u - v).norm() / v.norm()).item())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("OH,OW,C", [(32, 32, 256), (24, 40, 64), (64, 64, 128)])
def test_pyramid_maps_resized_in_one_launch(dtype, OH, OW, C):
    """The decoder's pyramid token maps (1x1 / 3x3 / 6x6 / 8x8 -> OH x OW, align_corners=True, written into channel slices of the concat
    buffer: paddle_EMRT.py:281-291) by ONE launch per direction (emrt_pyramid_resize_fwd / _bwd) against one launch per scale: same bits,
    forward and backward; and against torch."""
    from emrt_amd import _lib
    scales, B = [1, 3, 6, 8], 3
    ntok = sum(k * k for k in scales)
    g = torch.Generator().manual_seed(3)
    tok = rnd(torch.randn(B, ntok, C, generator=g))
    dcat = rnd(torch.randn(B, OH, OW, C * 5, generator=g))

    def run(grouped):
        c = init(dtype)
        c.pyramid_group = grouped
        td = dev(tok)
        cat = c.zeros((B, OH, OW, C * 5))
        outs = [Fn.narrow(cat, 3, C * (1 + i), C) for i in range(4)]
        tape = Tape()
        c.tape = tape
        L = _lib.lib()
        L.start_record()
        Fn.pyramid_tokens_to_maps(td, scales, OH, OW, outs)
        c.tape = None
        tape.watch(td)
        dd = dev(dcat)
        dtok, = run_bwd(tape, [(o, Fn.narrow(dd, 3, C * (1 + i), C)) for i, o in enumerate(outs)], [td])
        names = [n for n, _ in L.stop_record()]
        torch.cuda.synchronize()
        c.pyramid_group = True
        return names, host(cat), host(dtok)

    names, cat, dtok = run(True)
    names0, cat0, dtok0 = run(False)
    wide = OH * OW >= 16 * 64          # (the grouped backward is the block-per-source-pixel form: every map >= x4 smaller per axis)
    assert names.count("emrt_pyramid_resize_fwd") == 1 and "emrt_resize_bilinear_fwd" not in names
    assert names.count("emrt_pyramid_resize_bwd") == (1 if wide else 0) and names.count("emrt_resize_bilinear_bwd") == (0 if wide else 4)
    assert names0.count("emrt_resize_bilinear_fwd") == 4 and names0.count("emrt_resize_bilinear_bwd") == 4
    # same arithmetic, separately compiled kernels: the blend's multiply-adds may be contracted differently (an ulp in fp32; a bf16 output
    # then rounds the other way once in a while); the backward sums in the same order in both (same channel chunks per scale)
    ulp = 1e-6 if dtype == F32 else 2.0 ** -7
    for name_, u, v in (("maps", cat, cat0), ("dtokens", dtok, dtok0)):
        assert (u - v).abs().max().item() <= ulp * v.abs().max().item(), (name_, (u - v).abs().max().item())
        assert dtype == F32 or (u != v).float().mean().item() < 2e-3, name_
    tr = tok.clone().requires_grad_(True)
    s0, ref = 0, []
    for k in scales:
        m = tr[:, s0:s0 + k * k].reshape(B, k, k, C).permute(0, 3, 1, 2)
        ref.append(F.interpolate(m, size=(OH, OW), mode="bilinear", align_corners=True).permute(0, 2, 3, 1))
        s0 += k * k
    full = torch.cat(ref, 3)
    full.backward(dcat[..., C:])
    close("pyramid fwd", cat[..., C:], full.detach(), dtype)
    close("pyramid dtokens", dtok, tr.grad, dtype, math.sqrt(OH * OW))


# -----------------------------------------------------------------------------------------------------------------
# An independent chain's conv -> BatchNorm stages as guests of a bottleneck block's launches (functional.SideJobs / conv_bn_many, emrt_conv2d_dgrad_multi)
# -----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [F32, BF16], ids=["fp32", "bf16"])
def test_side_chain_rides_in_the_bottleneck_blocks_launches(dtype):
    """Two bottleneck blocks (the first with a shortcut conv) beside a two-stage branch block (maxpool -> conv3x3 -> BN -> ReLU -> conv3x3 -> BN -> ReLU,
    paddle_EMRT.py:80-97) on another input: with Context.side set, the branch's conv -> BatchNorm stages ride in the blocks' conv1 launches forward
    (emrt_conv2d_group) and their data gradients in the blocks' conv1 / shortcut data-gradient launches backward (emrt_conv2d_dgrad_multi, every fused
    epilogue kept: the join's ReLU mask, the BatchNorm sums, the addend).  Against the same layers run one after the other: fp32 to rounding, bf16 in the norm."""
    from emrt_amd import _lib
    from emrt_amd.src.models.emrt import BottleneckBlock, branch_block
    from emrt_amd import nn as hnn
    L = _lib.lib()
    g = torch.Generator().manual_seed(91)
    B = 2
    x_main = torch.randn(B, 256, 16, 16, generator=g)
    x_side = torch.randn(B, 64, 32, 32, generator=g)
    dy_main = torch.randn(B, 512, 16, 16, generator=g)
    dy_side = torch.randn(B, 128, 16, 16, generator=g)
    seeds = {}

    def run(side_on, dgrad_pair=True):
        c = init(dtype)
        c.dgrad_pair = dgrad_pair
        torch.manual_seed(5)
        b0 = BottleneckBlock(256, 128, 1, hnn.Sequential(hnn.Conv2D(256, 512, 1, 1, 0, bias=False), hnn.BatchNorm2D(512)))
        b1 = BottleneckBlock(512, 128)
        br = branch_block(64, 128)
        mods = dict(b0=b0, b1=b1, br=br)
        for name, m in mods.items():
            for pn, p_ in m.named_parameters():
                if (name, pn) not in seeds:
                    seeds[(name, pn)] = rnd(p_.detach().clone() if p_.dim() == 1 else torch.randn(p_.shape, generator=g) / math.sqrt(p_[0].numel()))
                with torch.no_grad():
                    p_.copy_(seeds[(name, pn)])
        Holder(**mods).place()
        xm, xs = dev_map(rnd(x_main)), dev_map(rnd(x_side))
        tape = Tape()
        c.tape = tape

        def jobs(x):
            x = Fn.maxpool(x, 3, 2, 1)
            x = yield (br.encode[0], br.encode[1], x, True, False, None)
            x = yield (br.encode[3], br.encode[4], x, True, False, None)
            return x
        L.start_record()
        if side_on:
            c.side = Fn.SideJobs(jobs(xs))
            try:
                ym = b1(b0(xm))
                ys = c.side.finish()
                hosted = c.side.hosted
            finally:
                c.side = None
        else:
            ym = b1(b0(xm))
            ys = Fn.maxpool(xs, 3, 2, 1)
            ys = Fn.conv_bn(br.encode[0], br.encode[1], ys, relu=True)
            ys = Fn.conv_bn(br.encode[3], br.encode[4], ys, relu=True)
            hosted = 0
        names = [n for n, _ in L.stop_record()]
        c.tape = None
        tape.watch(xm)
        tape.watch(xs)
        L.start_record()
        dxm, dxs = run_bwd(tape, [(ym, dev_map(rnd(dy_main))), (ys, dev_map(rnd(dy_side)))], [xm, xs])
        bnames = [(n, a) for n, a in L.stop_record()]
        torch.cuda.synchronize()
        c.dgrad_pair = True
        grads = [host(p_.grad) for m in mods.values() for p_ in m.parameters()]
        return names, bnames, hosted, [host_map(ym), host_map(ys), host_map(dxm), host_map(dxs)] + grads

    n1, bn1, hosted, r1 = run(True)
    n0, bn0, _, r0 = run(False)
    n2, bn2, _, r2 = run(True, dgrad_pair=False)
    assert hosted == 2 and n1.count("emrt_conv2d_group") == 2 and n0.count("emrt_conv2d_group") == 1      # (alone, the first block still pairs conv1 with its shortcut conv)
    multi = [a for n, a in bn1 if n == "emrt_conv2d_dgrad_multi"]
    assert sum(1 for a in multi if a[1] == 2) == 2 and not any(n == "emrt_conv2d_dgrad_multi" and a[1] == 2 for n, a in bn2)
    for u, v, w_ in zip(r1, r0, r2):
        assert torch.isfinite(u).all()
        if dtype == F32:
            assert (u - v).abs().max().item() <= 3e-5 * max(1.0, v.abs().max().item()), (u - v).abs().max().item()
            assert (u - w_).abs().max().item() <= 3e-5 * max(1.0, v.abs().max().item())
        else:
            assert ((u - v).norm() / v.norm().clamp_min(1e-20)).item() < 0.05 and ((u - w_).norm() / v.norm().clamp_min(1e-20)).item() < 0.05
