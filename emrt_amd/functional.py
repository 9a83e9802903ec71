"""Functional ops of the EMRT HIP path: each launches libemrt_hip.so kernels through the C-ABI and, when a tape is
recording, appends the closure that runs the matching backward kernels.  No torch compute op is used anywhere here
(torch only allocates memory and supplies the stream); if the HIP library is missing every op raises.

Tensor conventions: feature maps are [N, H, W, C] (NHWC) views with stride(-1) == 1 and dense rows
(stride(1) == W * stride(2)); token tensors are [B, L, C].  Views into larger buffers (level slabs of the token
tensor, channel slices of the concat buffer) are first-class: kernels take the pixel stride `ld` and batch stride `bs`.
"""
import ctypes
import math

import torch

from . import _lib
from .runtime import ctx, dtype_of, F32


def _L():
    return _lib.lib()


def P(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _check_map(t):
    """(N, H, W, C, ld, bs) of an NHWC map view; [B, L, C] tokens count as [B, 1, L, C], [M, C] as [1, 1, M, C]."""
    assert t.stride(-1) == 1, "inner stride must be 1: %s %s" % (tuple(t.shape), t.stride())
    if t.dim() == 4:
        assert t.stride(1) == t.shape[2] * t.stride(2), \
            "expected an NHWC view with dense rows, got shape %s strides %s" % (tuple(t.shape), t.stride())
        return t.shape[0], t.shape[1], t.shape[2], t.shape[3], t.stride(2), t.stride(0)
    if t.dim() == 3:
        return t.shape[0], 1, t.shape[1], t.shape[2], t.stride(1), t.stride(0)
    assert t.dim() == 2
    return 1, 1, t.shape[0], t.shape[1], t.stride(0), t.shape[0] * t.stride(0)


def _like_shape(t, C):
    """Dense output shape with the leading dims of t and C channels."""
    return tuple(t.shape[:-1]) + (C,)


def _rows(t):
    """[..., C] tensor whose leading dims collapse to dense rows -> (rows, C, ld)."""
    C = t.shape[-1]
    assert t.stride(-1) == 1
    ld = t.stride(-2) if t.dim() >= 2 else C
    rows = t.numel() // C
    for d in range(t.dim() - 2):
        assert t.stride(d) == t.stride(d + 1) * t.shape[d + 1], "rows are not uniformly strided: %s %s" % (tuple(t.shape), t.stride())
    return rows, C, ld


def tokens_as_map(t, h, w):
    """[B, h*w, C] token slab view -> [B, h, w, C] map view (no copy)."""
    B, n, C = t.shape
    assert n == h * w
    v = t.as_strided((B, h, w, C), (t.stride(0), w * t.stride(1), t.stride(1), 1), t.storage_offset())
    tape = ctx().tape
    if tape is not None:
        tape.register_alias(v, t, lambda g: g.as_strided((B, h, w, C), (g.stride(0), w * g.stride(1), g.stride(1), 1), g.storage_offset()))
    return v


def map_as_tokens(t):
    """[B, h, w, C] map view -> [B, h*w, C] tokens view (no copy)."""
    B, h, w, C = t.shape
    assert t.stride(1) == w * t.stride(2)
    v = t.as_strided((B, h * w, C), (t.stride(0), t.stride(2), 1), t.storage_offset())
    tape = ctx().tape
    if tape is not None:
        tape.register_alias(v, t, lambda g: g.as_strided((B, h * w, C), (g.stride(0), g.stride(2), 1), g.storage_offset()))
    return v


def view_as(t, shape):
    """Contiguous reshape view registered as an alias of t."""
    v = t.view(shape)
    tape = ctx().tape
    if tape is not None:
        tape.register_alias(v, t, lambda g: g.view(shape))
    return v


def param_input(p_data, p_grad):
    """A fp32 parameter used as an activation: returns it in the compute dtype; its gradient flows back into p_grad."""
    c = ctx()
    t = cast_from_f32(p_data)
    tape = c.tape
    if tape is not None:
        def bwd():
            g = tape.pop_grad(t)
            if g is not None:
                add_into(p_grad, cast_to_f32(g))
        tape.record(bwd)
    return t


def narrow(t, dim, start, length):
    """View t.narrow(dim, start, length), registered on the tape as an alias of t."""
    v = t.narrow(dim, start, length)
    tape = ctx().tape
    if tape is not None:
        tape.register_alias(v, t, lambda g: g.narrow(dim, start, length))
    return v


# ---------------------------------------------------------------------------------------------------
# elementwise helpers
# ---------------------------------------------------------------------------------------------------
def _geom3(*tensors):
    """Common (B, rows, cols) of equally-sized operands plus each operand's (batch_stride, row_stride).
    Dense tensors adopt the (B, rows, cols) split of the strided ones (token slabs / channel slices)."""
    n = tensors[0].numel()
    split = None
    for t in tensors:
        assert t.numel() == n
        if t.dim() >= 2 and not t.is_contiguous():
            N, H, W, C, ld, bs = _check_map(t)
            cur = (N, H * W, C)
            assert split is None or split == cur, "incompatible views %s vs %s" % (split, cur)
            split = cur
    if split is None:
        split = (1, 1, n)
    B, R, Cc = split
    strides = []
    for t in tensors:
        if t.dim() >= 2 and not t.is_contiguous():
            _, _, _, _, ld, bs = _check_map(t)
            strides.append((bs, ld))
        else:
            strides.append((R * Cc, Cc))
    return B, R, Cc, strides


def add_into(dst, src):
    """dst += src (dense tensors, token slabs or channel slices in any combination)."""
    c = ctx()
    assert dst.dtype == src.dtype, (dst.dtype, src.dtype)
    dt = dtype_of(dst)
    B, R, Cc, (sd, ss) = _geom3(dst, src)
    _L().call("emrt_acc3d", P(dst), sd[0], sd[1], P(src), ss[0], ss[1], B, R, Cc, dt, c.stream)


def add(a, b, period=None, bgrad=None):
    """out = a + b with b broadcast with period `period` elements (None: same shape).
    bgrad(g): optional callback that reduces the gradient of the broadcast operand (parameters / embeddings)."""
    c = ctx()
    assert a.is_contiguous() and b.is_contiguous()
    out = c.empty(tuple(a.shape), a.dtype)
    n = a.numel()
    per = n if period is None else period
    dt = dtype_of(a)
    _L().call("emrt_add", P(a), P(b), P(out), n, per, dt, c.stream)
    tape = c.tape
    if tape is not None:
        def bwd():
            g = tape.pop_grad(out)
            if g is None:
                return
            if bgrad is not None:
                bgrad(g)
            if period is None:
                tape.add_grad(b, g)
            tape.add_grad(a, g)
        tape.record(bwd)
    return out


def add_maps(a, b):
    """out (dense) = a + b where a, b may be strided views of equal shape (e.g. residual = level slab of `memory`)."""
    c = ctx()
    assert a.shape == b.shape and a.dtype == b.dtype
    out = c.empty(tuple(a.shape), a.dtype)
    B, R, Cc, (sa, sb, so) = _geom3(a, b, out)
    dt = dtype_of(a)
    _L().call("emrt_add3d", P(a), sa[0], sa[1], P(b), sb[0], sb[1], P(out), so[0], so[1], B, R, Cc, dt, c.stream)
    tape = c.tape
    if tape is not None:
        def bwd():
            g = tape.pop_grad(out)
            if g is None:
                return
            tape.add_grad(a, g)
            tape.add_grad(b, g)
        tape.record(bwd)
    return out


def concat_tokens(parts):
    """[B, n_i, C] dense parts -> dense [B, sum n_i, C]: one launch, and one for the backward split."""
    c = ctx()
    B, C = parts[0].shape[0], parts[0].shape[2]
    assert all(p_.is_contiguous() and p_.shape[0] == B and p_.shape[2] == C and p_.dtype == parts[0].dtype for p_ in parts) and len(parts) <= 8
    ns = [p_.shape[1] for p_ in parts]
    out = c.empty((B, sum(ns), C), parts[0].dtype)
    dt = dtype_of(parts[0])
    n_arr = (ctypes.c_int * len(parts))(*ns)
    _L().call("emrt_concat_tokens", (ctypes.c_void_p * len(parts))(*[p_.data_ptr() for p_ in parts]), n_arr, len(parts), P(out), B, C, 0, dt, c.stream)
    tape = c.tape
    if tape is not None:
        def bwd():
            g = tape.pop_grad(out)
            if g is None:
                return
            assert g.is_contiguous()
            gps = [c.empty(tuple(p_.shape), p_.dtype) for p_ in parts]
            _L().call("emrt_concat_tokens", (ctypes.c_void_p * len(parts))(*[g_.data_ptr() for g_ in gps]), n_arr, len(parts), P(g), B, C, 1, dt, c.stream)
            for p_, gp in zip(parts, gps):
                tape.add_grad(p_, gp, owned=True)
        tape.record(bwd)
    return out


def cast_from_f32(x):
    c = ctx()
    if c.dtype == F32:
        return x
    out = c.empty(tuple(x.shape))
    _L().call("emrt_cast", P(x), P(out), x.numel(), 0, c.dtype, c.stream)
    return out


def cast_to_f32(x):
    c = ctx()
    if x.dtype == torch.float32:
        return x
    out = c.empty(tuple(x.shape), torch.float32)
    _L().call("emrt_cast", P(x), P(out), x.numel(), 1, c.dtype, c.stream)
    return out


def dropout(x, p, salt, mode=0, hw=1, sole_consumer_is_linear=False):
    """Inverted dropout (mode 0) / Dropout2D on NHWC (mode 1); identity unless training.
    sole_consumer_is_linear: the caller promises that x is a fresh conv2d/linear(relu=True) or batch_norm(relu=True) output (x >= 0) and that the result
    feeds exactly one conv2d/linear (the FFN: linear2(dropout(relu(linear1(.)))); the Dropout2D in front of the two heads' classifier convs).  That consumer's data gradient then
    comes out already multiplied by this dropout's mask and the ReLU's (one test `y > 0`, scale 1/(1-p), in its dgrad
    epilogue), and neither this op nor the producer's ReLU runs a mask pass of its own."""
    c = ctx()
    if not c.training or p <= 0.0:
        return x
    assert x.is_contiguous()
    y = c.empty(tuple(x.shape), x.dtype)
    C = x.shape[-1]
    _L().call("emrt_dropout_fwd", P(x), P(y), x.numel(), float(p), c.seed_ptr, salt, mode, hw, C, c.dtype, c.stream)
    tape = c.tape
    rec = None
    if tape is not None and sole_consumer_is_linear:      # (element or channel mode alike: a stored value is positive iff it was kept and past the ReLU)
        rec = {"scale": 1.0 / (1.0 - float(p)), "dx": None}
        y._drop_rec = rec
    if tape is not None:
        def bwd():
            g, ncontrib = tape.pop_grad(y, with_count=True)
            if g is None:
                return
            if rec is not None and rec["dx"] is not None:
                assert ncontrib == 1 and rec["dx"] is g, "dropout(sole_consumer_is_linear=True) output had another consumer"
                x._premasked = g                  # the producer's ReLU backward is done too (y > 0 implies relu input > 0)
                tape.add_grad(x, g, owned=True)
                return
            assert g.is_contiguous()
            dx = c.empty(tuple(x.shape), x.dtype)
            _L().call("emrt_mask_bwd", P(g), None, P(dx), x.numel(), float(p), c.seed_ptr, salt, mode, hw, C, c.dtype, c.stream)
            tape.add_grad(x, dx, owned=True)
        tape.record(bwd)
    return y


# ---------------------------------------------------------------------------------------------------
# convolution / linear
# ---------------------------------------------------------------------------------------------------
class GemmWeight:
    """A [OC][KH][KW][C] GEMM weight living in the flat parameter store: fp32 master / grad views plus the packed
    forward and (transposed) dgrad copies in the compute dtype."""

    def __init__(self, OC, C, KH=1, KW=1):
        self.OC, self.C, self.KH, self.KW = OC, C, KH, KW
        self.fwd_ptr = None     # int device address of packed [OC][KH][KW][C]
        self.bwd_ptr = None     # int device address of packed [C][KH][KW][OC]
        self.grad = None        # fp32 tensor view [OC*KH*KW*C]
        self.bias = None        # fp32 tensor view [OC] or None
        self.bias_grad = None
        self.need_bwd = True


class _WgradDesc(ctypes.Structure):       # EmrtWgradDesc (include/emrt_hip.h)
    _fields_ = [("x", ctypes.c_void_p), ("dy", ctypes.c_void_p), ("dw", ctypes.c_void_p), ("dbias", ctypes.c_void_p),
                ("N", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int), ("C", ctypes.c_int), ("ldx", ctypes.c_int),
                ("x_bs", ctypes.c_longlong), ("OH", ctypes.c_int), ("OW", ctypes.c_int), ("OC", ctypes.c_int), ("lddy", ctypes.c_int),
                ("dy_bs", ctypes.c_longlong), ("KH", ctypes.c_int), ("KW", ctypes.c_int), ("stride", ctypes.c_int), ("pad", ctypes.c_int),
                ("dilation", ctypes.c_int), ("dw_is_zero", ctypes.c_int)]


def wgrad_deferred(w):
    """True when this layer's weight gradient is batched with other layers' (Context.wgrad_batch).  The 1x1 classifiers (OC <= 8) keep
    their one-pass streaming backward (data + weight gradient in one launch, csrc/conv.hip: thin_bwd_kernel)."""
    c = ctx()
    return c.wgrad_batch > 0 and not c.overlap and w.OC > 8 and c._in_backward


def defer_wgrad(tape, x, dy, w, geom, stride, pad, dil):
    """Queue dW += wgrad(x, dy) (+ dbias) of one layer; x and dy stay alive until the batch is launched (Tape.flush_wgrads)."""
    N, H, Wd, C, ldx, x_bs, OH, OW, lddy, dy_bs = geom
    # first contribution to this weight's gradient since ParamStore.zero_grad(): dW is still all zero, a one-slice launch may store its tiles
    fresh = int(getattr(w, "grad_is_zero", False))
    w.grad_is_zero = False
    # a weight used twice in one step (a shared layer): its earlier contribution may be queued as "dW is zero: store the tile"; that
    # launch must be on the stream before this one's adds are (emrt_conv2d_wgrad_group also drops the store for aliased problems)
    dwp = w.grad.data_ptr()
    if any(f[2] == dwp for f, _keep in tape.wgrads):
        tape.flush_wgrads()
    tape.wgrads.append(((x.data_ptr(), dy.data_ptr(), w.grad.data_ptr(), w.bias_grad.data_ptr() if w.bias is not None else None,
                         N, H, Wd, C, ldx, x_bs, OH, OW, w.OC, lddy, dy_bs, w.KH, w.KW, stride, pad, dil, fresh), (x, dy)))
    if len(tape.wgrads) >= ctx().wgrad_batch:
        tape.flush_wgrads()


def launch_wgrads(pending):
    c = ctx()
    arr = (_WgradDesc * len(pending))()
    for d, (f, _keep) in zip(arr, pending):
        (d.x, d.dy, d.dw, d.dbias, d.N, d.H, d.W, d.C, d.ldx, d.x_bs, d.OH, d.OW, d.OC, d.lddy, d.dy_bs, d.KH, d.KW, d.stride, d.pad, d.dilation, d.dw_is_zero) = f
    stream = c.wgrad_fork([t for _f, keep in pending for t in keep]) if c.wgrad_side else c.stream
    _L().call("emrt_conv2d_wgrad_group", arr, len(pending), c.dtype, stream)


class _DgradDesc(ctypes.Structure):       # EmrtConvDgradDesc (include/emrt_hip.h)
    _fields_ = [("dy", ctypes.c_void_p), ("w_bwd_packed", ctypes.c_void_p), ("dx", ctypes.c_void_p), ("lddx", ctypes.c_int), ("dx_bs", ctypes.c_longlong),
                ("accumulate", ctypes.c_int), ("N", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int), ("C", ctypes.c_int), ("OH", ctypes.c_int),
                ("OW", ctypes.c_int), ("OC", ctypes.c_int), ("lddy", ctypes.c_int), ("dy_bs", ctypes.c_longlong), ("KH", ctypes.c_int), ("KW", ctypes.c_int),
                ("stride", ctypes.c_int), ("pad", ctypes.c_int), ("dilation", ctypes.c_int), ("bn_stats", ctypes.c_void_p), ("mask_y", ctypes.c_void_p),
                ("ldy", ctypes.c_int), ("y_bs", ctypes.c_longlong), ("mask_scale", ctypes.c_float), ("stat_x", ctypes.c_void_p), ("ldsx", ctypes.c_int),
                ("sx_bs", ctypes.c_longlong), ("addend", ctypes.c_void_p), ("ldadd", ctypes.c_int), ("add_bs", ctypes.c_longlong)]


_PAIR_IDS = [0]


def conv2d(x, w, stride=1, pad=0, relu=False, residual=None, out=None, out_f32=False, need_dx=True, bn_stats=None, dilation=1,
           out_scale=None, out_shift=None, drop=None, _launched=False, _pair=None):
    """x [N,H,W,C] view -> [N,OH,OW,OC].  bias comes from w.bias.  `out` may be a strided view (concat slice).
    out_scale / out_shift (fp32 [OC], inference only): out = conv * out_scale + out_shift -- an eval-mode BatchNorm folded in.
    drop=(p, salt) with relu=True (training, 1x1): out = dropout_p(relu(linear(x))) with the mask drawn in the GEMM epilogue
    (emrt_conv2d_drop); the caller promises that the result feeds exactly one conv2d / linear, whose data gradient applies both masks.
    _launched=True: the forward has already been launched into `out` (with these arguments, as one problem of a grouped launch: conv_bn_pair); only the
    backward is recorded.  _pair=(group id, "host" | "guest") (conv_bn_many): in backward the guest's data gradient waits in the tape's stash for the
    closure that runs next -- its host's -- and goes out in THAT launch (emrt_conv2d_dgrad_multi: both keep their fused epilogues)."""
    c = ctx()
    if drop is not None and not (c.training and drop[0] > 0.0):
        drop = None
    pend = None
    if isinstance(x, PendingBN):
        if pointwise_takes_pending(w, x.C) and x.relu:
            assert stride == 1 and pad == 0 and not relu and residual is None and out is None and not out_f32 and bn_stats is None and out_scale is None
            return _pointwise_on_pending(x, w)
        # a conv -> BatchNorm (-> ReLU) -> conv chain: this convolution applies the BatchNorm with its own operand loads when its kernel has the
        # transform (emrt_conv2d_bna_supported, asked below once the geometry is known); otherwise the BatchNorm gets its own launch after all
        pend, x = x, x.raw
    N, H, W, C, ldin, in_bs = _check_map(x)
    assert C == w.C, (C, w.C)
    dil = int(dilation)
    OH = (H + 2 * pad - dil * (w.KH - 1) - 1) // stride + 1
    OW = (W + 2 * pad - dil * (w.KW - 1) - 1) // stride + 1
    if out is None:
        oshape = (N, OH, OW, w.OC) if x.dim() == 4 else _like_shape(x, w.OC)
        out = c.empty(oshape, torch.float32 if out_f32 else None)
    _, oh_, ow_, oc_, ldout, out_bs = _check_map(out)
    assert (oh_, ow_, oc_) == (OH, OW, w.OC), ((oh_, ow_, oc_), (OH, OW, w.OC))
    ldres = res_bs = 0
    if residual is not None:
        _, _, _, _, ldres, res_bs = _check_map(residual)
    assert out_scale is None or c.tape is None, "BatchNorm folding is inference only"     # (out_shift already holds w.bias * scale)
    fused_drop = (drop is not None and relu and w.KH * w.KW == 1 and stride == 1 and pad == 0 and residual is None and not out_f32 and bn_stats is None
                  and out_scale is None and w.OC % 8 == 0 and in_bs == H * W * ldin and out_bs == OH * OW * ldout and c.fuse_ffn_dropout
                  and C % 8 == 0 and ldin % 8 == 0 and ldout % 8 == 0 and x.data_ptr() % 16 == 0 and out.data_ptr() % 16 == 0 and pend is None)
    bna_done = False
    if pend is not None:
        bn = pend.bn
        a = c.empty(tuple(x.shape))          # relu(BatchNorm(raw)): written by the convolution (its first tile column) when it takes the operand form
        bna_args = (P(x), ctypes.c_void_p(w.fwd_ptr), P(out), P(w.bias), P(residual), N, H, W, C, ldin, in_bs, OH, OW, w.OC, ldout, out_bs, ldres, res_bs,
                    w.KH, w.KW, stride, pad, int(relu), int(out_f32), P(bn_stats), dil, P(pend.sums), float(pend.count), bn.eps, bn.momentum, P(pend.mean),
                    P(pend.invstd), P(bn.run_mean), P(bn.run_var), P(bn.gamma), P(bn.beta), int(pend.relu), P(a), c.dtype, c.stream)
        if c.training and c.bn_conv and drop is None and out_scale is None and _L().query("emrt_conv2d_bna_supported", *bna_args):
            _L().call("emrt_conv2d_bna", *bna_args)
            # the BatchNorm's backward, recorded BEFORE this layer's own (it runs after it): as if emrt_bn_apply had written `a`
            x = batch_norm(x, bn, relu=pend.relu, out=a, sums=pend.sums, _applied=(pend.mean, pend.invstd, pend.count))
            bna_done = True
        else:
            x = pend.materialize()
        _, _, _, _, ldin, in_bs = _check_map(x)
    if _launched:
        assert out is not None and pend is None and not fused_drop and residual is None and out_scale is None
    if bna_done or _launched:
        pass
    elif fused_drop:
        _L().call("emrt_conv2d_drop", P(x), ctypes.c_void_p(w.fwd_ptr), P(out), P(w.bias), N * H * W, C, ldin, w.OC, ldout, float(drop[0]), c.seed_ptr,
                  int(drop[1]), c.dtype, c.stream)
    else:
        _L().call("emrt_conv2d", P(x), ctypes.c_void_p(w.fwd_ptr), P(out), P(out_shift) if out_scale is not None else P(w.bias), P(residual), N, H, W, C, ldin, in_bs,
                  OH, OW, w.OC, ldout, out_bs, ldres, res_bs, w.KH, w.KW, stride, pad, 0, int(relu), int(out_f32), P(bn_stats), None, 0, 0,
                  dil, P(out_scale), c.dtype, c.stream)
    tape = c.tape
    own_drop = None
    if fused_drop and tape is not None:
        # as Fn.dropout(sole_consumer_is_linear=True): the consumer's data gradient multiplies by (out > 0) / (1 - p) and leaves its dx here
        own_drop = {"scale": 1.0 / (1.0 - float(drop[0])), "dx": None}
        out._drop_rec = own_drop
    bn_rec = getattr(x, "_bn_rec", None)      # x = relu(BatchNorm(.)) fresh from batch_norm(): see the dgrad call below
    drop_rec = getattr(x, "_drop_rec", None)  # x = dropout(relu(linear(.))) with this layer as its only consumer
    res_rec = getattr(x, "_bnres_rec", None)  # x = relu(BatchNorm(.) + residual): a residual join, several consumers
    if tape is not None:
        def bwd():
            dy = tape.pop_grad(out)
            if dy is None:
                return
            if out_f32 and dy.dtype == torch.float32:      # (a producer may hand the gradient over in the compute dtype already: Fn.msda)
                dy = cast_from_f32(dy)
            if own_drop is not None:
                # the gradient of dropout(relu(.)): already masked and scaled when the consumer's dgrad did it (the usual case); otherwise
                # (out > 0) is the mask of both -- a stored value is positive iff it was kept AND past the ReLU -- times 1 / (1 - p)
                if own_drop["dx"] is not dy:
                    assert dy.is_contiguous() and out.is_contiguous()
                    dm = c.empty(tuple(dy.shape), dy.dtype)
                    _L().call("emrt_mask_bwd", P(dy), P(out), P(dm), dy.numel(), float(drop[0]), None, 0, 0, 1, 1, c.dtype, c.stream)
                    dy = dm
            elif relu and getattr(out, "_premasked", None) is not dy:
                assert dy.is_contiguous() and out.is_contiguous()
                dm = c.empty(tuple(dy.shape), dy.dtype)
                _L().call("emrt_mask_bwd", P(dy), P(out), P(dm), dy.numel(), 0.0, None, 0, 0, 1, 1, c.dtype, c.stream)
                dy = dm
            _, _, _, _, lddy, dy_bs = _check_map(dy)
            dbias = P(w.bias_grad) if w.bias is not None else None
            deferred = wgrad_deferred(w)
            if not deferred:
                w.grad_is_zero = False           # (an immediate weight gradient accumulates into dW: it is not zero afterwards)
            if deferred:
                defer_wgrad(tape, x, dy, w, (N, H, W, C, ldin, in_bs, OH, OW, lddy, dy_bs), stride, pad, dil)
            if not need_dx:
                if not deferred:
                    _L().call("emrt_conv2d_wgrad", P(x), P(dy), P(w.grad), N, H, W, C, ldin, in_bs, OH, OW, w.OC, lddy, dy_bs,
                              w.KH, w.KW, stride, pad, dbias, dil, c.dtype, c.stream)
            else:
                slot = tape.grad_slot(x)          # accumulate straight into an existing gradient / a slice of the base buffer
                # contributions other consumers made so far that are not ours to write into: the kernel reads them as an
                # addend (dx = dgrad + addend) instead of a separate accumulate pass afterwards
                addend = tape.unowned_grad(x) if (slot is None and not c.overlap) else None
                dx = slot if slot is not None else c.empty(tuple(x.shape))
                _, _, _, _, lddx, dx_bs = _check_map(dx)
                ldadd, add_bs = _check_map(addend)[4:6] if addend is not None else (0, 0)
                ysums = ymask = stat = None
                ldsx = sx_bs = 0
                mscale = 1.0
                if drop_rec is not None and bn_rec is None and slot is None and addend is None and not c.overlap:
                    ymask, mscale = x, drop_rec["scale"]
                if bn_rec is not None and c.training and slot is None and addend is None:
                    # dgrad also applies that BatchNorm's ReLU mask and accumulates its backward sums (sum dy', sum dy'*y):
                    # the BatchNorm's own reduction pass is skipped when this turns out to be its only gradient
                    ysums = c.zeros_f64(BN_REPLICAS * 2 * C)
                    ymask = x
                elif res_rec is not None and c.training and not c.overlap and ymask is None and x.dim() == 4:
                    # residual join: this dgrad (+ what was accumulated before it) is masked by x > 0 and summed against
                    # the BatchNorm's INPUT; if it turns out to be the last contribution, the join's backward needs
                    # neither its reduction pass nor the accumulate (batch_norm checks the contribution count)
                    ysums = c.zeros_f64(BN_REPLICAS * 2 * C)
                    ymask, stat = x, res_rec["x"]
                    ldsx, sx_bs = _check_map(stat)[4:6]
                side = c.fork(x, dy)
                if side is None and deferred and _pair is not None and c.dgrad_pair and dil == 1:
                    me = _DgradDesc()
                    me.dy, me.w_bwd_packed, me.dx, me.lddx, me.dx_bs, me.accumulate = dy.data_ptr(), w.bwd_ptr, dx.data_ptr(), lddx, dx_bs, int(slot is not None)
                    me.N, me.H, me.W, me.C, me.OH, me.OW, me.OC, me.lddy, me.dy_bs = N, H, W, C, OH, OW, w.OC, lddy, dy_bs
                    me.KH, me.KW, me.stride, me.pad, me.dilation = w.KH, w.KW, stride, pad, dil
                    me.bn_stats, me.mask_y = _dp(ysums), _dp(ymask)
                    me.ldy, me.y_bs, me.mask_scale = (ldin, in_bs, float(mscale)) if ymask is not None else (0, 0, float(mscale))
                    me.stat_x, me.ldsx, me.sx_bs = _dp(stat), ldsx, sx_bs
                    me.addend, me.ldadd, me.add_bs = _dp(addend), ldadd, add_bs
                    keep = (dy, dx, ysums, ymask, stat, addend)

                    def launch(host, me=me, keep=keep):
                        arr = (_DgradDesc * (2 if host is not None else 1))()
                        if host is not None:
                            arr[0] = host
                        arr[len(arr) - 1] = me
                        _L().call("emrt_conv2d_dgrad_multi", arr, len(arr), c.dtype, c.stream)
                    if _pair[1] == "guest":
                        tape.flush_stash()
                        tape.stash = (_pair[0], launch)
                    elif tape.stash is not None and tape.stash[0] == _pair[0]:
                        guest_launch = tape.stash[1]
                        tape.stash = None
                        guest_launch(me)          # [host, guest] in one launch
                    else:
                        launch(None)
                elif side is None:
                    # one call for both gradients: small layers run their dgrad and wgrad tiles in ONE launch
                    _L().call("emrt_conv2d_bwd", P(x), P(dy), ctypes.c_void_p(w.bwd_ptr), P(dx), lddx, dx_bs, int(slot is not None),
                              None if deferred else P(w.grad), None if deferred else dbias, N, H, W, C, ldin, in_bs,
                              OH, OW, w.OC, lddy, dy_bs, w.KH, w.KW, stride, pad, P(ysums), P(ymask), ldin if ymask is not None else 0,
                              in_bs if ymask is not None else 0, float(mscale), P(stat), ldsx, sx_bs, P(addend), ldadd, add_bs,
                              dil, c.dtype, c.stream)
                else:       # two-stream experiment (Context.overlap): wgrad on the side stream next to dgrad
                    _L().call("emrt_conv2d_wgrad", P(x), P(dy), P(w.grad), N, H, W, C, ldin, in_bs, OH, OW, w.OC, lddy, dy_bs,
                              w.KH, w.KW, stride, pad, dbias, dil, c.dtype, side)
                    _L().call("emrt_conv2d", P(dy), ctypes.c_void_p(w.bwd_ptr), P(dx), None, P(dx) if slot is not None else None,
                              N, OH, OW, w.OC, lddy, dy_bs, H, W, C, lddx, dx_bs, lddx if slot is not None else 0, dx_bs if slot is not None else 0,
                              w.KH, w.KW, stride, pad, 1, 0, 0, P(ysums), P(ymask), ldin if ymask is not None else 0,
                              in_bs if ymask is not None else 0, dil, None, c.dtype, c.stream)
                    c.join()
                if addend is not None:
                    tape.replace_grad(x, dx)
                elif slot is None:
                    tape.add_grad(x, dx, owned=True)
                if stat is not None:
                    res_rec["dx"], res_rec["sums"], res_rec["n"] = dx, ysums, tape.grad_count(x)
                elif ysums is not None:
                    bn_rec["dx"], bn_rec["sums"] = dx, ysums
                elif ymask is not None:
                    drop_rec["dx"] = dx
            if residual is not None:
                tape.add_grad(residual, dy)
        if _pair is not None:
            bwd._group = _pair[0]
        tape.record(bwd)
    if drop is not None and not fused_drop:      # (a geometry the fused entry point does not take, or the A/B knob: the separate dropout launch,
        return dropout(out, drop[0], drop[1], sole_consumer_is_linear=True)      #  recorded AFTER this layer's own backward)
    return out


def linear(x, w, relu=False, out_f32=False, need_dx=True, drop=None):
    """x [B, L, C] or [M, C] (strided rows allowed) -> [..., OC]: a 1x1 convolution over the row axis."""
    return conv2d(x, w, 1, 0, relu=relu, out_f32=out_f32, need_dx=need_dx, drop=drop)


def colsum_acc(x, dst_f32):
    """dst[c] += sum over all rows of x[..., c]  (bias / embedding-row gradients)."""
    c = ctx()
    N, H, W, C, ld, bs = _check_map(x)
    assert dst_f32.dtype == torch.float32 and dst_f32.is_contiguous() and dst_f32.numel() == C
    dt = dtype_of(x)
    ws = c.workspace(_L().query("emrt_colreduce_workspace_bytes", N * H * W, C))
    _L().call("emrt_colsum_acc", P(x), ld, H * W, bs, N * H * W, C, P(dst_f32), P(ws), dt, c.stream)


# ---------------------------------------------------------------------------------------------------
# normalisation
# ---------------------------------------------------------------------------------------------------
class BNState:
    """Parameter/buffer views for one BatchNorm layer (all fp32)."""

    def __init__(self, C, eps=1e-5, momentum=0.9, sync=False):
        self.C, self.eps, self.momentum, self.sync = C, eps, momentum, sync
        self.gamma = self.beta = self.dgamma = self.dbeta = self.run_mean = self.run_var = None
        self.fold_scale = self.fold_shift = None     # eval-mode affine form, views of ParamStore.bn_fold
        self.fold_conv = None                        # the biased conv whose bias the fold absorbs (nn.BatchNorm2D(after=))


BN_REPLICAS = 8     # fp64 BatchNorm sums are [8][2C]: producers spread atomics over replicas, consumers add them


def _sync_active(bn):
    """This BatchNorm all-reduces its statistics over ranks (nn.SyncBatchNorm: paddle_EMRT.py:64, fcn_head.py:53)."""
    c = ctx()
    return bool(bn.sync and c.sync_bn and (c.world_size > 1 or c.sync_always))


def _allreduce_sums(sums, count):
    """SyncBatchNorm: sum the per-rank fp64 (sum, sumsq) vectors over ranks (RCCL all-reduce); returns the global row count.
    Inside a captured step the collective is issued between two hipGraphs (runtime.Context.collective)."""
    import torch.distributed as dist
    ctx().collective(lambda: dist.all_reduce(sums))
    return count * dist.get_world_size()


def batch_norm(x, bn, relu=False, residual=None, out=None, sums=None, _applied=None):
    """y = [relu](BN(x) [+ residual]).  Training uses batch statistics from the fp64 `sums` [2C] that the producing conv's
    epilogue accumulated (or from a statistics pass when `sums` is None), all-reduced over ranks when bn.sync.
    _applied = (mean, invstd, count): `out` already holds the result and mean / invstd are (being) saved -- the consuming convolution applied this
    BatchNorm with its own operand loads and wrote the normalised map on the way (conv2d on a PendingBN: emrt_conv2d_bna) -- so nothing is launched
    here and only the backward is recorded, exactly as for the separate launch."""
    c = ctx()
    N, H, W, C, ldx, x_bs = _check_map(x)
    assert x_bs == H * W * ldx, "batch_norm input must be a dense NHWC tensor (possibly channel-sliced)"
    M = N * H * W
    if out is None:
        out = c.empty(tuple(x.shape))
    _, _, _, _, ldy, y_bs = _check_map(out)
    assert y_bs == H * W * ldy
    ldres = 0
    res_pend = None
    if isinstance(residual, PendingBN):
        # the shortcut's conv -> BatchNorm (no ReLU), not applied yet: this join applies it as it loads the raw map (emrt_bn_apply_join)
        res_pend, residual = residual, residual.raw
        assert c.training and not res_pend.relu and res_pend.C == C and res_pend.M == M
    if residual is not None:
        _, _, _, _, ldres, r_bs = _check_map(residual)
        assert r_bs == H * W * ldres
    count = M
    mean = invstd = None
    if _applied is not None:
        assert c.training and residual is None and out is not None
        mean, invstd, count = _applied
    elif c.training:
        mean = c.empty((C,), torch.float32)
        invstd = c.empty((C,), torch.float32)
        if sums is None:
            sums = c.zeros_f64(BN_REPLICAS * 2 * C)
            _L().call("emrt_bn_stats", P(x), ldx, M, C, P(sums), c.dtype, c.stream)
        if _sync_active(bn):
            count = _allreduce_sums(sums, M)
        if res_pend is not None:
            rb = res_pend.bn
            _L().call("emrt_bn_apply_join", P(x), ldx, P(residual), ldres, P(out), ldy, P(sums), float(count), bn.eps, bn.momentum, P(mean), P(invstd),
                      P(bn.run_mean), P(bn.run_var), P(bn.gamma), P(bn.beta), P(res_pend.sums), float(res_pend.count), rb.eps, rb.momentum,
                      P(res_pend.mean), P(res_pend.invstd), P(rb.run_mean), P(rb.run_var), P(rb.gamma), P(rb.beta), M, C, int(relu), c.dtype, c.stream)
        else:
            _L().call("emrt_bn_apply", P(x), ldx, P(residual), ldres, P(out), ldy, P(sums), float(count), bn.eps, bn.momentum, P(mean), P(invstd),
                      P(bn.run_mean), P(bn.run_var), P(bn.gamma), P(bn.beta), M, C, int(relu), c.dtype, c.stream)
    else:
        _L().call("emrt_bn_apply", P(x), ldx, P(residual), ldres, P(out), ldy, None, 1.0, bn.eps, bn.momentum, None, None,
                  P(bn.run_mean), P(bn.run_var), P(bn.gamma), P(bn.beta), M, C, int(relu), c.dtype, c.stream)
    tape = c.tape
    rec = None
    if tape is not None and c.training and relu and out.dim() == 4:
        # lets a conv that consumes this tensor fold the backward reduction into its dgrad (functional.conv2d)
        if residual is None:
            rec = {"dx": None, "sums": None}
            out._bn_rec = rec
        else:
            rec = {"dx": None, "sums": None, "n": 0, "x": x}
            out._bnres_rec = rec
    if tape is not None:
        assert c.training, "backward through eval-mode BatchNorm is not supported"

        def bwd():
            dy, ncontrib = tape.pop_grad(out, with_count=True)
            if dy is None:
                return
            _, _, _, _, lddy, dy_bs = _check_map(dy)
            assert dy_bs == H * W * lddy
            yv = out if relu else None
            sync = _sync_active(bn)
            # the consumer conv's dgrad already produced the sums when its dx is the one and only gradient of `out`
            # ... or, for a residual join, when that dgrad was the LAST of its contributions (it folded the earlier ones in)
            fused = rec is not None and rec["dx"] is dy and not sync and ncontrib == (1 if residual is None else rec["n"])
            if fused:
                sums2 = rec["sums"]
                yv = None                 # dy is already masked
            else:
                sums2 = c.zeros_f64(BN_REPLICAS * 2 * C)
                _L().call("emrt_bn_bwd_reduce", P(x), ldx, P(dy), lddy, P(yv), ldy, P(mean), P(invstd), M, C, P(sums2), None, None, c.dtype, c.stream)
            # dgamma/dbeta use the LOCAL sums (the gradient all-reduce combines ranks); dx needs the GLOBAL sums
            local = None
            if sync:
                local = c.empty((BN_REPLICAS * 2 * C,), torch.float64)
                _L().call("emrt_cast", P(sums2), P(local), BN_REPLICAS * 4 * C, 0, F32, c.stream)     # raw 8-byte copy as 2 x f32
                _allreduce_sums(sums2, M)
            dx = c.empty(tuple(x.shape))
            dres = c.empty(tuple(x.shape)) if (residual is not None and relu and not fused) else None
            _L().call("emrt_bn_bwd_dx", P(x), ldx, P(dy), lddy, P(yv), ldy, P(dx), C, P(dres), C, P(mean), P(invstd), P(bn.gamma),
                      P(sums2), P(local), float(count), P(bn.dgamma), P(bn.dbeta), M, C,
                      P(bn.beta) if (fused and residual is None) else None, int(fused and residual is not None), None, c.dtype, c.stream)
            tape.add_grad(x, dx, owned=True)
            if res_pend is not None:
                # the shortcut's BatchNorm backward on (its raw map, this join's masked gradient); no ReLU of its own
                res_pend.backward(tape, dres if dres is not None else dy, masked=True)
            elif residual is not None:
                tape.add_grad(residual, dres if dres is not None else dy, owned=dres is not None)
        tape.record(bwd)
    return out


class PendingBN:
    """relu(BatchNorm_train(raw)) that has NOT been applied: the raw conv output plus its complete fp64 batch sums.  A streaming consumer
    (resize_bilinear, maxpool) applies the affine map + ReLU to every element it loads (csrc/bn_operand.hpp), so the normalised map is
    never written and the emrt_bn_apply launch disappears; the consumer's backward then runs this layer's BatchNorm backward with the
    ReLU mask re-derived from `raw`.  Only conv_bn(..., defer=True) makes one, and only resize_bilinear / maxpool accept one."""

    def __init__(self, raw, bn, sums, count, relu):
        c = ctx()
        self.raw, self.bn, self.sums, self.count, self.relu = raw, bn, sums, count, relu
        self.shape = raw.shape
        N, H, W, C, ldx, x_bs = _check_map(raw)
        assert x_bs == H * W * ldx and ldx == C, "a deferred BatchNorm needs a dense NHWC map"
        self.M, self.C = N * H * W, C
        self.mean = c.empty((C,), torch.float32)
        self.invstd = c.empty((C,), torch.float32)

    def materialize(self):
        """The ordinary path after all: relu(BatchNorm(raw)) through emrt_bn_apply (the sums are complete, and all-reduced where the layer is a
        SyncBatchNorm), with the usual backward.  For a consumer that turns out not to take the operand form (conv2d: emrt_conv2d_bna_supported)."""
        c = ctx()
        bn, C = self.bn, self.C
        out = c.empty(tuple(self.raw.shape))
        _L().call("emrt_bn_apply", P(self.raw), C, None, 0, P(out), C, P(self.sums), float(self.count), bn.eps, bn.momentum, P(self.mean), P(self.invstd),
                  P(bn.run_mean), P(bn.run_var), P(bn.gamma), P(bn.beta), self.M, C, int(self.relu), c.dtype, c.stream)
        return batch_norm(self.raw, bn, relu=self.relu, out=out, sums=self.sums, _applied=(self.mean, self.invstd, self.count))

    def operand(self):
        """the BatchNorm arguments of emrt_bn_resize_bilinear_fwd / emrt_bn_maxpool_fwd"""
        bn = self.bn
        return (P(self.sums), float(self.count), bn.eps, bn.momentum, P(self.mean), P(self.invstd), P(bn.run_mean), P(bn.run_var), P(bn.gamma),
                P(bn.beta), int(self.relu))

    def backward(self, tape, dy, masked=False):
        """dy: gradient of relu(BN(raw)), dense; unmasked unless `masked` -> BatchNorm backward (emrt_bn_bwd_reduce + emrt_bn_bwd_dx,
        the ReLU mask re-derived from raw)"""
        c = ctx()
        bn, M, C = self.bn, self.M, self.C
        _, _, _, _, lddy, dy_bs = _check_map(dy)
        assert dy_bs == (M // self.raw.shape[0]) * lddy
        mg, mb = (P(bn.gamma), P(bn.beta)) if (self.relu and not masked) else (None, None)
        sums2 = c.zeros_f64(BN_REPLICAS * 2 * C)
        _L().call("emrt_bn_bwd_reduce", P(self.raw), C, P(dy), lddy, None, 0, P(self.mean), P(self.invstd), M, C, P(sums2), mg, mb, c.dtype, c.stream)
        local = None
        if _sync_active(bn):
            local = c.empty((BN_REPLICAS * 2 * C,), torch.float64)
            _L().call("emrt_cast", P(sums2), P(local), BN_REPLICAS * 4 * C, 0, F32, c.stream)     # raw 8-byte copy as 2 x f32
            _allreduce_sums(sums2, M)
        dx = c.empty(tuple(self.raw.shape))
        _L().call("emrt_bn_bwd_dx", P(self.raw), C, P(dy), lddy, None, 0, P(dx), C, None, C, P(self.mean), P(self.invstd), P(bn.gamma),
                  P(sums2), P(local), float(self.count), P(bn.dgamma), P(bn.dbeta), M, C, None, 0, mb, c.dtype, c.stream)
        tape.add_grad(self.raw, dx, owned=True)


def pointwise_takes_pending(w, C):
    """the thin classifier kernels (emrt_bn_pointwise_fwd / _bwd): a 1x1 conv to at most 8 channels from 64 / 128 / 256"""
    return w.KH == 1 and w.KW == 1 and w.OC <= 8 and C in (64, 128, 256)


def _pointwise_on_pending(pend, w):
    """out = conv1x1(relu(BN(raw))) with the BatchNorm + ReLU applied by the classifier's own loads (paddle_EMRT.py:176-179: conv_2 ->
    SyncBatchNorm -> ReLU -> conv_3); backward: one pass gives the masked input gradient, dW, dbias and the BatchNorm's backward sums."""
    c = ctx()
    x = pend.raw
    N, H, W, C, ldin, in_bs = _check_map(x)
    assert C == w.C and pointwise_takes_pending(w, C) and pend.relu
    out = c.empty((N, H, W, w.OC))
    _L().call("emrt_bn_pointwise_fwd", P(x), ldin, in_bs, ctypes.c_void_p(w.fwd_ptr), P(w.bias), P(out), w.OC, H * W * w.OC, N, H * W, C, w.OC,
              *pend.operand(), c.dtype, c.stream)
    tape = c.tape
    if tape is not None:
        def bwd():
            dy = tape.pop_grad(out)
            if dy is None:
                return
            if dy.dtype != x.dtype:
                dy = cast_from_f32(dy)
            _, _, _, _, lddy, dy_bs = _check_map(dy)
            bn = pend.bn
            da = c.empty((N, H, W, C))
            sync = _sync_active(bn)
            sums2 = None if sync else c.zeros_f64(BN_REPLICAS * 2 * C)
            w.grad_is_zero = False
            _L().call("emrt_bn_pointwise_bwd", P(x), ldin, in_bs, P(dy), lddy, dy_bs, ctypes.c_void_p(w.bwd_ptr), P(da), C, H * W * C, P(w.grad),
                      P(w.bias_grad) if w.bias is not None else None, P(sums2), N, H * W, C, w.OC, P(pend.mean), P(pend.invstd), P(bn.gamma),
                      P(bn.beta), c.dtype, c.stream)
            if sync:
                pend.backward(tape, da, masked=True)      # SyncBatchNorm over ranks: the sums go through the reduce -> all-reduce path
                return
            dx = c.empty((N, H, W, C))
            _L().call("emrt_bn_bwd_dx", P(x), C, P(da), C, None, 0, P(dx), C, None, C, P(pend.mean), P(pend.invstd), P(bn.gamma), P(sums2), None,
                      float(pend.count), P(bn.dgamma), P(bn.dbeta), pend.M, C, P(bn.beta), 0, None, c.dtype, c.stream)
            tape.add_grad(x, dx, owned=True)
        tape.record(bwd)
    return out


def conv_bn(conv, bn, x, relu=False, residual=None, out=None, defer=False):
    """conv -> BatchNorm with the batch statistics accumulated in the conv's epilogue (training).  Inference: the BatchNorm is
    an affine map of running statistics and is folded into the conv's epilogue -- no BatchNorm launch, no intermediate tensor.
    defer=True (training, when the only consumer is resize_bilinear or maxpool): returns a PendingBN instead of launching emrt_bn_apply.
    defer="conv" (training, when the only consumer is a conv2d / conv_bn): the same; that convolution applies the BatchNorm with its operand loads
    (emrt_conv2d_bna) or, when its kernel has no such form, launches emrt_bn_apply itself (PendingBN.materialize)."""
    c = ctx()
    if c.fold_live and not c.training and c.tape is None and (conv.gw.bias is None or bn.state.fold_conv is conv):
        scale, shift = bn.state.fold_scale, bn.state.fold_shift      # ParamStore.fold_bn(): refreshed at the top of every eval forward
        return conv2d(x, conv.gw, conv.stride, conv.padding, relu=relu, residual=residual, out=out, need_dx=False,
                      dilation=getattr(conv, "dilation", 1), out_scale=scale, out_shift=shift)
    sums = c.zeros_f64(BN_REPLICAS * 2 * bn.C) if c.training else None
    y = conv2d(x, conv.gw, conv.stride, conv.padding, need_dx=conv.need_dx, bn_stats=sums, dilation=getattr(conv, "dilation", 1))
    # what the consumers' fused entry points accept (emrt_bn_resize_bilinear_fwd, emrt_bn_maxpool_fwd: C <= 1024 and C / (elements per 16 bytes)
    # a divisor of 256 -- C / 4 in fp32); a PendingBN has no fallback once created, so anything else takes the separate emrt_bn_apply here (_bn_tail)
    return _bn_tail(y, bn, sums, relu, residual, out, defer)


class _BnGroupDesc(ctypes.Structure):      # EmrtBnGroupDesc (include/emrt_hip.h)
    _fields_ = [("x", ctypes.c_void_p), ("y", ctypes.c_void_p), ("dy", ctypes.c_void_p), ("dx", ctypes.c_void_p), ("sums", ctypes.c_void_p),
                ("mean", ctypes.c_void_p), ("invstd", ctypes.c_void_p), ("run_mean", ctypes.c_void_p), ("run_var", ctypes.c_void_p),
                ("gamma", ctypes.c_void_p), ("beta", ctypes.c_void_p), ("dgamma", ctypes.c_void_p), ("dbeta", ctypes.c_void_p),
                ("count", ctypes.c_double), ("eps", ctypes.c_float), ("momentum", ctypes.c_float),
                ("M", ctypes.c_int), ("C", ctypes.c_int), ("ldx", ctypes.c_int), ("ldy", ctypes.c_int), ("lddy", ctypes.c_int), ("lddx", ctypes.c_int),
                ("relu", ctypes.c_int), ("res", ctypes.c_void_p), ("ldres", ctypes.c_int), ("res_hw", ctypes.c_int), ("res_bs", ctypes.c_longlong)]


def _small_group_ok(convs, bns, xs, post_adds=None):
    """the grouped form: 2..4 independent stride-1 "same" conv -> BatchNorm (-> ReLU) branches on token slices / NHWC map views, one rank"""
    c = ctx()
    if not (c.training and c.bn_small_group and 2 <= len(convs) <= 4 and c.tape is not None and not c.overlap):
        return False
    per16 = 4 if c.dtype == F32 else 8
    tiles = 0
    for i, (cv, b, x) in enumerate(zip(convs, bns, xs)):
        w = cv.gw
        if isinstance(x, PendingBN) or x.dim() not in (3, 4) or x.stride(-1) != 1:
            return False
        N, H, W, C, ld, bs = _check_map(x)
        pad = cv.padding
        if not (w.KH == w.KW and w.KH in (1, 3) and cv.stride == 1 and pad == w.KH // 2 and getattr(cv, "dilation", 1) == 1 and w.bias is None
                and w.OC > 32 and C == w.C and C % per16 == 0 and w.OC % per16 == 0 and ld % per16 == 0 and bs % per16 == 0 and x.data_ptr() % 16 == 0
                and w.OC % 4 == 0 and 256 % (w.OC // 4) == 0 and N * H * W <= 16384 and cv.need_dx and b.C == w.OC and not _sync_active(b.state)):
            return False
        if post_adds is not None and post_adds[i] is not None:
            r = post_adds[i]
            if tuple(r.shape[:-1]) != tuple(x.shape[:-1]) or r.shape[-1] != w.OC or r.stride(-1) != 1 or r.dtype != x.dtype:
                return False
            rN, rH, rW, rC, rld, rbs = _check_map(r)
            if rld % 4 or rbs % 4:
                return False
        tiles += ((N * H * W + 63) // 64) * ((w.OC + 63) // 64)
    return tiles <= 4096


def conv_bn_small_group(convs, bns, xs, relu=True, post_adds=None):
    """[conv_i -> BatchNorm_i (-> ReLU) (+ post_add_i)] for 2..4 INDEPENDENT small branches in ONE launch per pass: grouped convolution with the batch
    statistics in its epilogue (emrt_conv2d_group), grouped BatchNorm apply (emrt_bn_group_apply); backward: grouped BatchNorm backward (reduce + dx:
    emrt_bn_group_bwd) and grouped data gradient (emrt_conv2d_bwd_group), the weight gradients batched as everywhere.
      * the four pyramid-pooling branches (paddle_EMRT.py:61-66,70-78: conv1x1 on 8 ... 512 pooled tokens of 256 channels at batch 8): round 5 launched
        8 kernels forward and 12 backward for them, ~5 us each;
      * the three Conv2dBlocks of EFP (paddle_EMRT.py:13-48: 3x3 convs on the 32^2 / 16^2 / 8^2 level maps, "conv2(conv1(x)) + x"): the three levels side
        by side fill the machine where each alone is a latency-bound launch (as the encoder's per-level convs, level_conv_gn).
    xs: [B, n_i, C] token slices or [B, h_i, w_i, C] map views (strided views of a token slab are fine); post_adds[i] (optional, dense rows): added AFTER
    the ReLU by the BatchNorm launch; its gradient is the output's."""
    c = ctx()
    n = len(convs)
    tape = c.tape
    states = [b.state for b in bns]
    fd = (_ConvDesc * n)()
    gd = (_BnGroupDesc * n)()
    raws, outs, saved, geo, alive = [], [], [], [], []
    for i, (d, q, cv, st, x) in enumerate(zip(fd, gd, convs, states, xs)):
        w = cv.gw
        N, H, W, C, ld, bs = _check_map(x)
        oshape = _like_shape(x, w.OC)
        raw = c.empty(oshape)
        out = c.empty(oshape)
        sm = c.zeros_f64(BN_REPLICAS * 2 * w.OC)
        mean, invstd = c.empty((w.OC,), torch.float32), c.empty((w.OC,), torch.float32)
        M = N * H * W
        d.inp, d.w_packed, d.out, d.bias, d.residual, d.bn_stats = x.data_ptr(), w.fwd_ptr, raw.data_ptr(), None, None, sm.data_ptr()
        d.N, d.H, d.W, d.C, d.ldin, d.in_bs = N, H, W, C, ld, bs
        d.OH, d.OW, d.OC, d.ldout, d.out_bs = H, W, w.OC, w.OC, H * W * w.OC
        d.ldres, d.res_bs, d.KH, d.KW, d.stride, d.pad, d.relu, d.out_f32 = 0, 0, w.KH, w.KW, 1, cv.padding, 0, 0
        r = post_adds[i] if post_adds is not None else None
        q.x, q.y, q.dy, q.dx, q.sums = raw.data_ptr(), out.data_ptr(), None, None, sm.data_ptr()
        q.mean, q.invstd, q.run_mean, q.run_var = mean.data_ptr(), invstd.data_ptr(), _dp(st.run_mean), _dp(st.run_var)
        q.gamma, q.beta, q.dgamma, q.dbeta = st.gamma.data_ptr(), st.beta.data_ptr(), None, None
        q.count, q.eps, q.momentum = float(M), st.eps, st.momentum
        q.M, q.C, q.ldx, q.ldy, q.lddy, q.lddx, q.relu = M, w.OC, w.OC, w.OC, w.OC, w.OC, int(relu)
        if r is not None:
            _, rH, rW, _, rld, rbs = _check_map(r)
            q.res, q.ldres, q.res_hw, q.res_bs = r.data_ptr(), rld, rH * rW, rbs
        else:
            q.res, q.ldres, q.res_hw, q.res_bs = None, 0, 1, 0
        raws.append(raw); outs.append(out); saved.append((mean, invstd)); geo.append((N, H, W, C, ld, bs))
        alive.append(sm)          # (outside a training step the sums are buffers of their own: they must outlive the two launches below)
    _L().call("emrt_conv2d_group", fd, n, c.dtype, c.stream)
    _L().call("emrt_bn_group_apply", gd, n, c.dtype, c.stream)
    del alive

    def bwd():
        dys = [tape.pop_grad(o) for o in outs]
        live = [i for i in range(n) if dys[i] is not None]
        if not live:
            return
        m = len(live)
        bq = (_BnGroupDesc * m)()
        bd = (_ConvBwdDesc * m)()
        keep, fresh = [], []
        for q, d, i in zip(bq, bd, live):
            cv, st, x = convs[i], states[i], xs[i]
            w = cv.gw
            N, H, W, C, ld, bs = geo[i]
            M = N * H * W
            dy = dys[i]
            # (a gradient may arrive as a channel slice of a wider buffer -- the concat buffer's gradient: rows uniformly strided, which is all the kernels need)
            _, dH, dW, _, lddy, dy_bs = _check_map(dy)
            assert dy.dtype == outs[i].dtype and dy_bs == dH * dW * lddy and lddy % 4 == 0, (tuple(dy.shape), dy.stride())
            if post_adds is not None and post_adds[i] is not None:
                tape.add_grad(post_adds[i], dy)                 # d(x + f(x)) / dx, the identity part
            draw = c.empty(tuple(outs[i].shape))                # gradient of the raw conv output
            sm2 = c.zeros_f64(BN_REPLICAS * 2 * w.OC)
            q.x, q.y, q.dy, q.dx, q.sums = raws[i].data_ptr(), outs[i].data_ptr(), dy.data_ptr(), draw.data_ptr(), sm2.data_ptr()
            q.mean, q.invstd, q.run_mean, q.run_var = saved[i][0].data_ptr(), saved[i][1].data_ptr(), None, None
            q.gamma, q.beta, q.dgamma, q.dbeta = st.gamma.data_ptr(), st.beta.data_ptr(), st.dgamma.data_ptr(), st.dbeta.data_ptr()
            q.count, q.eps, q.momentum = float(M), st.eps, st.momentum
            q.M, q.C, q.ldx, q.ldy, q.lddy, q.lddx, q.relu = M, w.OC, w.OC, w.OC, lddy, w.OC, int(relu)
            q.res, q.ldres, q.res_hw, q.res_bs = None, 0, 1, 0
            deferred = wgrad_deferred(w)
            if deferred:
                defer_wgrad(tape, x, draw, w, (N, H, W, C, ld, bs, H, W, w.OC, H * W * w.OC), 1, cv.padding, 1)
            slot = tape.grad_slot(x)
            dx = slot if slot is not None else c.empty(tuple(x.shape))
            _, _, _, _, lddx, dx_bs = _check_map(dx)
            d.x, d.dy, d.w_bwd_packed, d.dx = x.data_ptr(), draw.data_ptr(), w.bwd_ptr, dx.data_ptr()
            d.lddx, d.dx_bs, d.accumulate, d.dw, d.dbias = lddx, dx_bs, int(slot is not None), None, None
            d.N, d.H, d.W, d.C, d.ldx, d.x_bs = N, H, W, C, ld, bs
            d.OH, d.OW, d.OC, d.lddy, d.dy_bs = H, W, w.OC, w.OC, H * W * w.OC
            d.KH, d.KW, d.stride, d.pad = w.KH, w.KW, 1, cv.padding
            keep.append((draw, sm2, deferred))
            if slot is None:
                fresh.append((x, dx))
        _L().call("emrt_bn_group_bwd", bq, m, c.dtype, c.stream)
        if all(k[2] for k in keep):
            _L().call("emrt_conv2d_bwd_group", bd, m, c.dtype, c.stream)
        else:           # (weight gradients not batched: EMRT_WGRAD_BATCH=0 / outside Tape.backward): the ordinary one-layer calls
            for d, i in zip(bd, live):
                w = convs[i].gw
                w.grad_is_zero = False
                _L().call("emrt_conv2d_bwd", ctypes.c_void_p(d.x), ctypes.c_void_p(d.dy), ctypes.c_void_p(w.bwd_ptr), ctypes.c_void_p(d.dx), d.lddx, d.dx_bs, d.accumulate,
                          P(w.grad), None, d.N, d.H, d.W, d.C, d.ldx, d.x_bs, d.OH, d.OW, d.OC, d.lddy, d.dy_bs, d.KH, d.KW, 1, d.pad, None, None, 0, 0, 1.0,
                          None, 0, 0, None, 0, 0, 1, c.dtype, c.stream)
        for x, dx in fresh:
            tape.add_grad(x, dx, owned=True)
    tape.record(bwd)
    return outs


class SideJobs:
    """A second, INDEPENDENT chain of layers run beside the main one (EMRT's spatial branch beside the ResNet: paddle_EMRT.py:99-113, 252-262).  `gen` is a
    generator that does its own launches and, whenever it needs a conv -> BatchNorm stage, yields (conv, bn, x, relu, defer, out) and is sent the result
    (what conv_bn would have returned).  The main chain, at a layer whose own launch leaves most of the machine idle, takes the pending request into ITS
    grouped launch (conv_bn_many) and delivers the result; finish() runs whatever is left the ordinary way and returns the generator's return value."""

    def __init__(self, gen):
        self.gen, self.req, self.done, self.result, self.hosted = gen, None, False, None, 0
        self._advance(None)

    def _advance(self, value):
        try:
            self.req = self.gen.send(value)
        except StopIteration as e:
            self.req, self.done, self.result = None, True, e.value

    def pending(self):
        return self.req

    def deliver(self, value):
        self.hosted += 1
        self._advance(value)

    def finish(self):
        while not self.done:
            cv, bn, x, relu, defer, out = self.req
            self._advance(conv_bn(cv, bn, x, relu=relu, out=out, defer=defer))
        return self.result


def _bn_tail(y, bn, sums, relu, residual, out, defer):
    """what conv_bn does with the raw convolution output y and its batch sums: a PendingBN (deferred forms) or the BatchNorm launch"""
    c = ctx()
    per16 = 4 if c.dtype == F32 else 8
    fits = bn.C % per16 == 0 and bn.C <= 1024 and 256 % (bn.C // per16) == 0
    if defer == "conv" and not c.bn_conv:
        defer = False
    if defer and c.training and c.bn_defer and residual is None and out is None and (fits or defer in ("join", "conv")):
        count = y.shape[0] * y.shape[1] * y.shape[2]
        if _sync_active(bn.state):
            count = _allreduce_sums(sums, count)
        return PendingBN(y, bn.state, sums, count, relu)
    return batch_norm(y, bn.state, relu=relu, residual=residual, out=out, sums=sums)


def conv_bn_many(items, host_tiles=128):
    """items: [(conv, bn, x, relu, defer, out)], 2..6 INDEPENDENT conv -> BatchNorm stages on materialised maps (different inputs, 1x1 / 3x3, any stride)
    whose forward convolutions go out as ONE grouped launch (emrt_conv2d_group, statistics in the epilogue).  Each keeps conv_bn's own tail (PendingBN or
    the BatchNorm launch) and its own backward.  The first items are the HOST's (a ResNet layer3 / layer4 block's conv1 [+ shortcut conv]: <= host_tiles
    tiles of 64 x 64, a launch that fills a fraction of the 256 CUs), the last one a guest from another chain (SideJobs).  Returns the conv_bn results in
    order, or None when the launch cannot be grouped (the caller then runs them one by one)."""
    c = ctx()
    n = len(items)
    if not (c.training and c.bn_defer and c.tape is not None and 2 <= n <= 6 and not c.overlap):
        return None
    per16 = 4 if c.dtype == F32 else 8
    tiles, geo = [], []
    for cv, bn, x, relu, defer, out in items:
        w = cv.gw
        if isinstance(x, PendingBN) or x.dim() != 4 or (defer == "conv" and not c.bn_conv):
            return None
        N, H, W, C, ld, bs = _check_map(x)
        if not (w.KH == w.KW and w.KH in (1, 3) and getattr(cv, "dilation", 1) == 1 and w.bias is None and w.OC > 32 and w.C == C and C % per16 == 0
                and w.OC % per16 == 0 and ld % per16 == 0 and bs % per16 == 0 and x.data_ptr() % 16 == 0 and not _sync_active(bn.state)):
            return None
        OH, OW = (H + 2 * cv.padding - w.KH) // cv.stride + 1, (W + 2 * cv.padding - w.KW) // cv.stride + 1
        tiles.append(((N * OH * OW + 63) // 64) * ((w.OC + 63) // 64))
        geo.append((N, H, W, C, ld, bs, OH, OW))
    if sum(tiles[:-1]) > host_tiles or sum(tiles) > 4096:
        return None
    fd = (_ConvDesc * n)()
    ys, sums = [], []
    for d, (cv, bn, x, relu, defer, out), (N, H, W, C, ld, bs, OH, OW) in zip(fd, items, geo):
        w = cv.gw
        y = c.empty((N, OH, OW, w.OC))
        sm = c.zeros_f64(BN_REPLICAS * 2 * bn.C)
        d.inp, d.w_packed, d.out, d.bias, d.residual, d.bn_stats = x.data_ptr(), w.fwd_ptr, y.data_ptr(), None, None, sm.data_ptr()
        d.N, d.H, d.W, d.C, d.ldin, d.in_bs = N, H, W, C, ld, bs
        d.OH, d.OW, d.OC, d.ldout, d.out_bs = OH, OW, w.OC, w.OC, OH * OW * w.OC
        d.ldres, d.res_bs, d.KH, d.KW, d.stride, d.pad, d.relu, d.out_f32 = 0, 0, w.KH, w.KW, cv.stride, cv.padding, 0, 0
        ys.append(y); sums.append(sm)
    _L().call("emrt_conv2d_group", fd, n, c.dtype, c.stream)
    _PAIR_IDS[0] += 1
    gid = _PAIR_IDS[0]
    res = []
    for i, ((cv, bn, x, relu, defer, out), y, sm) in enumerate(zip(items, ys, sums)):
        # records this layer's backward; the guest's (last item) data gradient rides in the launch of the host whose closure runs right after it
        conv2d(x, cv.gw, cv.stride, cv.padding, need_dx=cv.need_dx, bn_stats=sm, out=y, _launched=True, _pair=(gid, "guest" if i == n - 1 else "host"))
        res.append(_bn_tail(y, bn, sm, relu, None, out, defer))
    return res


def conv_bn_pair(items, x):
    """Two (or more, <= 4) conv -> BatchNorm stages that read the SAME input and are both consumed in their deferred form (PendingBN): conv1 of a ResNet
    stage's first block (-> bn1 -> relu, applied by conv2's loads) and the stage's shortcut conv (-> BatchNorm, applied by the join)
    (paddle_vision_resnet.py:108-123,129-147,226-233).  Their forward convolutions go out as ONE grouped launch (emrt_conv2d_group, statistics in the
    epilogue); each keeps its own backward.  items: [(conv, bn, relu, defer)], defer as conv_bn's.  Returns the conv_bn results in order, or None when the
    pair cannot be grouped (the caller then calls conv_bn one by one)."""
    c = ctx()
    n = len(items)
    if not (c.training and c.bn_defer and c.conv_pair and 2 <= n <= 4 and not isinstance(x, PendingBN) and x.dim() == 4):
        return None
    per16 = 4 if c.dtype == F32 else 8
    N, H, W, C, ld, bs = _check_map(x)
    tiles = 0
    for cv, bn, relu, defer in items:
        w = cv.gw
        if defer == "conv" and not c.bn_conv:
            return None
        if not (w.KH == w.KW == 1 and cv.padding == 0 and getattr(cv, "dilation", 1) == 1 and w.bias is None and w.OC > 32 and w.C == C and C % per16 == 0
                and w.OC % per16 == 0 and ld % per16 == 0 and bs % per16 == 0 and x.data_ptr() % 16 == 0 and not _sync_active(bn.state)):
            return None
        OH, OW = (H - 1) // cv.stride + 1, (W - 1) // cv.stride + 1
        tiles += ((N * OH * OW + 63) // 64) * ((w.OC + 63) // 64)
    if tiles > 4096:
        return None
    fd = (_ConvDesc * n)()
    ys, sums = [], []
    for d, (cv, bn, relu, defer) in zip(fd, items):
        w = cv.gw
        OH, OW = (H - 1) // cv.stride + 1, (W - 1) // cv.stride + 1
        y = c.empty((N, OH, OW, w.OC))
        sm = c.zeros_f64(BN_REPLICAS * 2 * bn.C)
        d.inp, d.w_packed, d.out, d.bias, d.residual, d.bn_stats = x.data_ptr(), w.fwd_ptr, y.data_ptr(), None, None, sm.data_ptr()
        d.N, d.H, d.W, d.C, d.ldin, d.in_bs = N, H, W, C, ld, bs
        d.OH, d.OW, d.OC, d.ldout, d.out_bs = OH, OW, w.OC, w.OC, OH * OW * w.OC
        d.ldres, d.res_bs, d.KH, d.KW, d.stride, d.pad, d.relu, d.out_f32 = 0, 0, 1, 1, cv.stride, 0, 0, 0
        ys.append(y); sums.append(sm)
    _L().call("emrt_conv2d_group", fd, n, c.dtype, c.stream)
    res = []
    for (cv, bn, relu, defer), y, sm in zip(items, ys, sums):
        conv2d(x, cv.gw, cv.stride, 0, need_dx=cv.need_dx, bn_stats=sm, out=y, _launched=True)      # records this layer's backward
        count = y.shape[0] * y.shape[1] * y.shape[2]
        res.append(PendingBN(y, bn.state, sm, count, relu))
    return res


def conv_bn_group(convs, bns, xs, relu=True):
    """[conv_i -> SyncBatchNorm_i (-> ReLU)] for several INDEPENDENT branches (the four pyramid-pooling branches,
    paddle_EMRT.py:61-66,70-78) with ONE cross-rank all-reduce of all their statistics per direction instead of one per
    branch: inside a captured multi-GPU step every collective is a cut between two hipGraphs (runtime.Context.collective),
    so the five SyncBatchNorm layers cost 4 cuts per step instead of 10.  Without ranks to talk to it is conv_bn per branch."""
    c = ctx()
    n = len(convs)
    states = [b.state for b in bns]
    if not (c.training and any(_sync_active(st) for st in states)):
        if _small_group_ok(convs, bns, xs):
            return conv_bn_small_group(convs, bns, xs, relu=relu)
        return [conv_bn(cv, b, x, relu=relu) for cv, b, x in zip(convs, bns, xs)]
    assert all(_sync_active(st) for st in states)
    sizes = [BN_REPLICAS * 2 * st.C for st in states]
    sums_all = c.zeros_f64(sum(sizes))
    offs = [sum(sizes[:i]) for i in range(n)]
    ys, geo = [], []
    for cv, x, o, sz in zip(convs, xs, offs, sizes):
        ys.append(conv2d(x, cv.gw, cv.stride, cv.padding, need_dx=cv.need_dx, bn_stats=sums_all[o:o + sz], dilation=getattr(cv, "dilation", 1)))
    Ms = []
    for y in ys:
        N, H, W, C, ldx, x_bs = _check_map(y)
        assert x_bs == H * W * ldx
        Ms.append(N * H * W)
        geo.append((C, ldx))
    import torch.distributed as dist
    world = dist.get_world_size()
    c.collective(lambda: dist.all_reduce(sums_all))
    outs, saved = [], []
    for y, st, o, sz, M, (C, ldx) in zip(ys, states, offs, sizes, Ms, geo):
        out = c.empty(tuple(y.shape))
        mean, invstd = c.empty((C,), torch.float32), c.empty((C,), torch.float32)
        _L().call("emrt_bn_apply", P(y), ldx, None, 0, P(out), C, P(sums_all[o:o + sz]), float(M * world), st.eps, st.momentum, P(mean), P(invstd),
                  P(st.run_mean), P(st.run_var), P(st.gamma), P(st.beta), M, C, int(relu), c.dtype, c.stream)
        outs.append(out)
        saved.append((mean, invstd))
    tape = c.tape
    if tape is not None:
        def bwd():
            dys = [tape.pop_grad(o_) for o_ in outs]
            live = [i for i in range(n) if dys[i] is not None]
            if not live:
                return
            sums2 = c.zeros_f64(sum(sizes))
            for i in live:
                C, ldx = geo[i]
                lddy = _check_map(dys[i])[4]
                _L().call("emrt_bn_bwd_reduce", P(ys[i]), ldx, P(dys[i]), lddy, P(outs[i]) if relu else None, C, P(saved[i][0]), P(saved[i][1]), Ms[i], C,
                          P(sums2[offs[i]:offs[i] + sizes[i]]), None, None, c.dtype, c.stream)
            # dgamma / dbeta from this rank's own sums (the gradient all-reduce combines ranks), dx from the rank-summed ones
            local = c.empty((sum(sizes),), torch.float64)
            _L().call("emrt_cast", P(sums2), P(local), 2 * sum(sizes), 0, F32, c.stream)       # raw 8-byte copy as 2 x f32
            c.collective(lambda: dist.all_reduce(sums2))
            for i in live:
                C, ldx = geo[i]
                st = states[i]
                lddy = _check_map(dys[i])[4]
                dx = c.empty(tuple(ys[i].shape))
                _L().call("emrt_bn_bwd_dx", P(ys[i]), ldx, P(dys[i]), lddy, P(outs[i]) if relu else None, C, P(dx), C, None, C, P(saved[i][0]), P(saved[i][1]),
                          P(st.gamma), P(sums2[offs[i]:offs[i] + sizes[i]]), P(local[offs[i]:offs[i] + sizes[i]]), float(Ms[i] * world), P(st.dgamma),
                          P(st.dbeta), Ms[i], C, None, 0, None, c.dtype, c.stream)
                tape.add_grad(ys[i], dx, owned=True)
        tape.record(bwd)
    return outs


def group_norm(x, gamma, beta, dgamma, dbeta, G=32, eps=1e-5, gelu=False, residual=None, out=None):
    """out = [gelu](GN(x)) [+ residual];  x [N,H,W,C] view."""
    c = ctx()
    N, H, W, C, ldx, x_bs = _check_map(x)
    if out is None:
        out = c.empty(tuple(x.shape))
    _, _, _, _, ldo, o_bs = _check_map(out)
    ldr = r_bs = 0
    if residual is not None:
        _, _, _, _, ldr, r_bs = _check_map(residual)
    mean = c.empty((N * G,), torch.float32)
    rstd = c.empty((N * G,), torch.float32)
    _L().call("emrt_groupnorm_fwd", P(x), ldx, x_bs, P(residual), ldr, r_bs, P(out), ldo, o_bs, P(gamma), P(beta), P(mean), P(rstd),
              P(c.zeros_f64(N * G * 2)), N, H * W, C, G, eps, int(gelu), c.dtype, c.stream)
    tape = c.tape
    if tape is not None:
        def bwd():
            dy = tape.pop_grad(out)
            if dy is None:
                return
            _, _, _, _, lddy, dy_bs = _check_map(dy)
            dx = c.empty(tuple(x.shape))
            _L().call("emrt_groupnorm_bwd", P(x), ldx, x_bs, P(dy), lddy, dy_bs, P(dx), C, H * W * C, P(gamma), P(beta), P(mean), P(rstd),
                      P(dgamma), P(dbeta), P(c.zeros_f64(N * C * 2)), N, H * W, C, G, int(gelu), c.dtype, c.stream)
            tape.add_grad(x, dx, owned=True)
            if residual is not None:
                tape.add_grad(residual, dy)
        tape.record(bwd)
    return out


class _ConvDesc(ctypes.Structure):        # EmrtConvDesc (include/emrt_hip.h)
    _fields_ = [("inp", ctypes.c_void_p), ("w_packed", ctypes.c_void_p), ("out", ctypes.c_void_p), ("bias", ctypes.c_void_p),
                ("residual", ctypes.c_void_p), ("N", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int), ("C", ctypes.c_int),
                ("ldin", ctypes.c_int), ("in_bs", ctypes.c_longlong), ("OH", ctypes.c_int), ("OW", ctypes.c_int), ("OC", ctypes.c_int),
                ("ldout", ctypes.c_int), ("out_bs", ctypes.c_longlong), ("ldres", ctypes.c_int), ("res_bs", ctypes.c_longlong),
                ("KH", ctypes.c_int), ("KW", ctypes.c_int), ("stride", ctypes.c_int), ("pad", ctypes.c_int), ("relu", ctypes.c_int),
                ("bn_stats", ctypes.c_void_p), ("out_f32", ctypes.c_int)]


class _ConvBwdDesc(ctypes.Structure):     # EmrtConvBwdDesc
    _fields_ = [("x", ctypes.c_void_p), ("dy", ctypes.c_void_p), ("w_bwd_packed", ctypes.c_void_p), ("dx", ctypes.c_void_p),
                ("lddx", ctypes.c_int), ("dx_bs", ctypes.c_longlong), ("accumulate", ctypes.c_int), ("dw", ctypes.c_void_p),
                ("dbias", ctypes.c_void_p), ("N", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int), ("C", ctypes.c_int),
                ("ldx", ctypes.c_int), ("x_bs", ctypes.c_longlong), ("OH", ctypes.c_int), ("OW", ctypes.c_int), ("OC", ctypes.c_int),
                ("lddy", ctypes.c_int), ("dy_bs", ctypes.c_longlong), ("KH", ctypes.c_int), ("KW", ctypes.c_int), ("stride", ctypes.c_int),
                ("pad", ctypes.c_int)]


def _dp(t):
    return t.data_ptr() if t is not None else None


def level_conv_gn_takes_linears(src, convs, linears):
    """True when level_conv_gn(linears=) can put the projections into the level convolutions' forward launch: the grouped kernel's limits (6 problems,
    4096 tiles of 64 x 64 in all -- beyond that emrt_conv2d_group falls back to one launch per problem)"""
    B, Lv, C = src.shape
    tiles = ((B * Lv + 63) // 64) * ((C + 63) // 64)          # (an upper bound for the level convs: their rows are the levels' pixels)
    for x, w, _f32 in linears:
        if not (x.dim() == 3 and x.is_contiguous() and w.KH == w.KW == 1 and x.shape[2] == w.C):
            return False
        tiles += ((x.shape[0] * x.shape[1] + 63) // 64) * ((w.OC + 63) // 64)
    return len(convs) + len(linears) <= 6 and tiles + len(convs) * ((C + 63) // 64) <= 4096


def level_conv_gn(src, convs, gns, spatial_shapes, level_spans, G=32, eps=1e-5, linears=None):
    """The conv branch of an encoder layer (transformer_encoder_decoder.py:125-144, 163-182) over ALL levels at once:
        out[:, level l] = GELU(GroupNorm_l(conv3x3_l(src[:, level l] as an h_l x w_l map))) + src[:, level l]
    src: dense tokens [B, Lv, C]; convs: GemmWeight per level (3x3, stride 1, pad 1, no bias); gns: (gamma, beta, dgamma,
    dbeta) per level.  Two launches forward (grouped conv, multi-level GroupNorm) and two backward instead of six each:
    the per-level problems are small, latency-bound launches on their own.
    linears (optional, as linear_group's items): independent linear layers that ride in the FORWARD grouped launch of the level convolutions -- the
    deformable attention's value_proj(src) and its offsets | logits projection of the query, which read the same tokens and depend on nothing the conv
    branch produces (transformer_encoder_decoder.py:83-92, 184-204).  Their backward is linear_group's own (a data-gradient launch of its own: value_proj's
    gradient accumulates into the same d src rows as the level convolutions', which two problems of ONE launch must not).  -> (out, [linear outputs])."""
    c = ctx()
    assert src.is_contiguous() and src.dim() == 3
    B, Lv, C = src.shape
    L = len(convs)
    assert L == len(gns) == len(spatial_shapes) == len(level_spans) and 1 <= L <= 4
    assert all(w.C == C and w.OC == C and w.KH == w.KW == 3 and w.bias is None for w in convs)
    esz = src.element_size()
    y = c.empty((B, Lv, C))
    out = c.empty((B, Lv, C))
    nl = len(linears) if linears else 0
    fd = (_ConvDesc * (L + nl))()
    lin_outs, lin_geo = [], []
    for j in range(nl):
        lin_outs.append(_linear_desc(fd[L + j], *linears[j]))
        lin_geo.append(tuple(linears[j][0].shape))
    for l, (w, (h, wd), (s0, n)) in enumerate(zip(convs, spatial_shapes, level_spans)):
        assert n == h * wd
        d = fd[l]
        d.inp, d.w_packed, d.out = src.data_ptr() + s0 * C * esz, w.fwd_ptr, y.data_ptr() + s0 * C * esz
        d.bias = d.residual = d.bn_stats = None
        d.N, d.H, d.W, d.C, d.ldin, d.in_bs = B, h, wd, C, C, Lv * C
        d.OH, d.OW, d.OC, d.ldout, d.out_bs = h, wd, C, C, Lv * C
        d.ldres, d.res_bs, d.KH, d.KW, d.stride, d.pad, d.relu = 0, 0, 3, 3, 1, 1, 0
    _L().call("emrt_conv2d_group", fd, L + nl, c.dtype, c.stream)
    if nl and c.tape is not None:
        _record_linear_group_bwd(linears, lin_outs, lin_geo)
    starts = (ctypes.c_int * L)(*[s0 for s0, _ in level_spans])
    hws = (ctypes.c_int * L)(*[n for _, n in level_spans])
    gam = (ctypes.c_void_p * L)(*[g[0].data_ptr() for g in gns])
    bet = (ctypes.c_void_p * L)(*[g[1].data_ptr() for g in gns])
    mean = c.empty((L * B * G,), torch.float32)
    rstd = c.empty((L * B * G,), torch.float32)
    _L().call("emrt_groupnorm_levels_fwd", P(y), C, Lv * C, P(src), C, Lv * C, P(out), C, Lv * C, gam, bet, P(mean), P(rstd), starts, hws, L,
              B, C, G, eps, 1, P(c.zeros_f64(L * B * G * 2)), c.dtype, c.stream)
    tape = c.tape
    if tape is not None:
        def bwd():
            dout, dout_n = tape.pop_grad(out, with_count=True)
            if dout is None:
                return
            assert dout.is_contiguous()
            dy = c.empty((B, Lv, C))
            dgam = (ctypes.c_void_p * L)(*[_dp(g[2]) for g in gns])
            dbet = (ctypes.c_void_p * L)(*[_dp(g[3]) for g in gns])
            _L().call("emrt_groupnorm_levels_bwd", P(y), C, Lv * C, P(dout), C, Lv * C, P(dy), C, Lv * C, gam, bet, P(mean), P(rstd), dgam, dbet,
                      starts, hws, L, B, C, G, 1, P(c.zeros_f64(L * B * G * 2)), c.dtype, c.stream)
            slot = tape.grad_slot(src)            # accumulate the data gradients straight into src's gradient when it has one
            dx = slot if slot is not None else c.empty((B, Lv, C))
            assert dx.is_contiguous()
            bd = (_ConvBwdDesc * L)()
            deferred = all(wgrad_deferred(w) for w in convs)
            if not deferred:
                for w in convs:
                    w.grad_is_zero = False
            for l, (w, (h, wd), (s0, n)) in enumerate(zip(convs, spatial_shapes, level_spans)):
                d = bd[l]
                off = s0 * C * esz
                d.x, d.dy, d.w_bwd_packed, d.dx = src.data_ptr() + off, dy.data_ptr() + off, w.bwd_ptr, dx.data_ptr() + off
                d.lddx, d.dx_bs, d.accumulate, d.dw, d.dbias = C, Lv * C, int(slot is not None), None if deferred else w.grad.data_ptr(), None
                if deferred:
                    xs, dys = src.narrow(1, s0, n), dy.narrow(1, s0, n)
                    defer_wgrad(tape, xs, dys, w, (B, h, wd, C, C, Lv * C, h, wd, C, Lv * C), 1, 1, 1)
                d.N, d.H, d.W, d.C, d.ldx, d.x_bs = B, h, wd, C, C, Lv * C
                d.OH, d.OW, d.OC, d.lddy, d.dy_bs = h, wd, C, C, Lv * C
                d.KH, d.KW, d.stride, d.pad = 3, 3, 1, 1
            _L().call("emrt_conv2d_bwd_group", bd, L, c.dtype, c.stream)
            if slot is None:
                tape.add_grad(src, dx, owned=True)
            if id(out) in tape.identity_done:     # a later consumer of src (the encoder layer's norm1, layer_norm(identity_from=out)) has already summed it in
                # ... and it summed THIS gradient: a contribution that arrived after it ran would have replaced the tensor (the promise of identity_from is broken)
                # (a gradient the tape owns is accumulated IN PLACE: the tensor's identity alone would not show a late contribution, the count does)
                summed, summed_n = tape.identity_done.pop(id(out))
                assert summed is dout and summed_n == dout_n, "layer_norm(identity_from=t): t received another gradient contribution after that LayerNorm's backward"
            else:
                tape.add_grad(src, dout)          # the residual path of every level: one add over the whole token tensor
        tape.record(bwd)
    return (out, lin_outs) if linears else out


def _linear_desc(d, x, w, out_f32):
    """fills one EmrtConvDesc with the linear layer y = x W^T + b over dense tokens x [B, L, C]; -> the output tensor"""
    c = ctx()
    assert x.is_contiguous() and x.dim() == 3 and w.KH == w.KW == 1 and x.shape[2] == w.C
    B, L_, C = x.shape
    out = c.empty((B, L_, w.OC), torch.float32 if out_f32 else None)
    d.inp, d.w_packed, d.out, d.bias, d.residual, d.bn_stats = x.data_ptr(), w.fwd_ptr, out.data_ptr(), _dp(w.bias), None, None
    d.N, d.H, d.W, d.C, d.ldin, d.in_bs = B, 1, L_, C, C, L_ * C
    d.OH, d.OW, d.OC, d.ldout, d.out_bs = 1, L_, w.OC, w.OC, L_ * w.OC
    d.ldres, d.res_bs, d.KH, d.KW, d.stride, d.pad, d.relu, d.out_f32 = 0, 0, 1, 1, 1, 0, 0, int(bool(out_f32))
    return out


def linear_group(items):
    """Up to 4 INDEPENDENT linear layers in one launch each way (emrt_conv2d_group / emrt_conv2d_bwd_group): items = [(x [B, L, C] dense, GemmWeight,
    out_f32)].  The deformable attention's value_proj(value) and its fused offsets | logits projection of the query (transformer_encoder_decoder.py:
    83-92) are two 1344-token linears of 9 and 13 us that do not depend on each other, forward or backward: side by side they share one launch's
    ramp and tail.  Returns the outputs.  Weight gradients are deferred to the batched launches as for conv2d."""
    c = ctx()
    n = len(items)
    assert 1 <= n <= 4
    outs, geo = [], []
    fd = (_ConvDesc * n)()
    for d, (x, w, out_f32) in zip(fd, items):
        outs.append(_linear_desc(d, x, w, out_f32))
        geo.append(tuple(x.shape))
    _L().call("emrt_conv2d_group", fd, n, c.dtype, c.stream)
    if c.tape is not None:
        _record_linear_group_bwd(items, outs, geo)
    return outs


def _record_linear_group_bwd(items, outs, geo):
    """the backward of linear_group (and of the linears that rode in level_conv_gn's forward launch): ONE grouped data-gradient launch, weight gradients batched"""
    c = ctx()
    tape = c.tape
    n = len(items)

    def bwd():
        dys = [tape.pop_grad(o) for o in outs]
        live = [i for i in range(n) if dys[i] is not None]
        if not live:
            return
        bd = (_ConvBwdDesc * len(live))()
        fresh = []
        for d, i in zip(bd, live):
            x, w, out_f32 = items[i]
            B, L_, C = geo[i]
            dy = dys[i]
            if out_f32 and dy.dtype == torch.float32:      # (a producer may hand the gradient over in the compute dtype already: Fn.msda)
                dy = cast_from_f32(dy)
            assert dy.is_contiguous()
            dys[i] = dy
            deferred = wgrad_deferred(w)
            if deferred:
                defer_wgrad(tape, x, dy, w, (B, 1, L_, C, C, L_ * C, 1, L_, w.OC, L_ * w.OC), 1, 0, 1)
            else:
                w.grad_is_zero = False
                _L().call("emrt_conv2d_wgrad", P(x), P(dy), P(w.grad), B, 1, L_, C, C, L_ * C, 1, L_, w.OC, w.OC, L_ * w.OC, 1, 1, 1, 0,
                          P(w.bias_grad) if w.bias is not None else None, 1, c.dtype, c.stream)
            slot = tape.grad_slot(x)
            dx = slot if slot is not None else c.empty((B, L_, C))
            assert dx.is_contiguous()
            d.x, d.dy, d.w_bwd_packed, d.dx = x.data_ptr(), dy.data_ptr(), w.bwd_ptr, dx.data_ptr()
            d.lddx, d.dx_bs, d.accumulate, d.dw, d.dbias = C, L_ * C, int(slot is not None), None, None
            d.N, d.H, d.W, d.C, d.ldx, d.x_bs = B, 1, L_, C, C, L_ * C
            d.OH, d.OW, d.OC, d.lddy, d.dy_bs = 1, L_, w.OC, w.OC, L_ * w.OC
            d.KH, d.KW, d.stride, d.pad = 1, 1, 1, 0
            if slot is None:
                fresh.append((x, dx))
        _L().call("emrt_conv2d_bwd_group", bd, len(live), c.dtype, c.stream)
        for x, dx in fresh:
            tape.add_grad(x, dx, owned=True)
    tape.record(bwd)


def level_proj_gn(feats, convs, gns, G=32, eps=1e-5):
    """input_proj of the encoder-decoder (transformer_encoder_decoder.py:375-379, 424-428) for all levels at once:
        src[:, level l] = GroupNorm_l(conv1x1_l(feats[l]) + bias_l)          feats[l]: dense [B, h_l, w_l, C_l]
    -> dense tokens [B, sum h_l w_l, OC].  One grouped conv launch + one multi-level GroupNorm launch each way."""
    c = ctx()
    L = len(feats)
    assert L == len(convs) == len(gns) and 1 <= L <= 4
    B = feats[0].shape[0]
    OC = convs[0].OC
    assert all(f.is_contiguous() and f.dim() == 4 and f.shape[0] == B for f in feats)
    assert all(w.OC == OC and w.KH == w.KW == 1 and w.C == f.shape[3] for w, f in zip(convs, feats))
    spans, s0 = [], 0
    for f in feats:
        spans.append((s0, f.shape[1] * f.shape[2]))
        s0 += f.shape[1] * f.shape[2]
    Lv = s0
    esz = feats[0].element_size()
    y = c.empty((B, Lv, OC))
    src = c.empty((B, Lv, OC))
    fd = (_ConvDesc * L)()
    for l, (w, f, (a, n)) in enumerate(zip(convs, feats, spans)):
        _, h, wd, Cl = f.shape
        d = fd[l]
        d.inp, d.w_packed, d.out, d.bias = f.data_ptr(), w.fwd_ptr, y.data_ptr() + a * OC * esz, _dp(w.bias)
        d.residual = d.bn_stats = None
        d.N, d.H, d.W, d.C, d.ldin, d.in_bs = B, h, wd, Cl, Cl, h * wd * Cl
        d.OH, d.OW, d.OC, d.ldout, d.out_bs = h, wd, OC, OC, Lv * OC
        d.ldres, d.res_bs, d.KH, d.KW, d.stride, d.pad, d.relu = 0, 0, 1, 1, 1, 0, 0
    _L().call("emrt_conv2d_group", fd, L, c.dtype, c.stream)
    starts = (ctypes.c_int * L)(*[a for a, _ in spans])
    hws = (ctypes.c_int * L)(*[n for _, n in spans])
    gam = (ctypes.c_void_p * L)(*[g[0].data_ptr() for g in gns])
    bet = (ctypes.c_void_p * L)(*[g[1].data_ptr() for g in gns])
    mean = c.empty((L * B * G,), torch.float32)
    rstd = c.empty((L * B * G,), torch.float32)
    _L().call("emrt_groupnorm_levels_fwd", P(y), OC, Lv * OC, None, 0, 0, P(src), OC, Lv * OC, gam, bet, P(mean), P(rstd), starts, hws, L,
              B, OC, G, eps, 0, P(c.zeros_f64(L * B * G * 2)), c.dtype, c.stream)
    tape = c.tape
    if tape is not None:
        def bwd():
            dsrc = tape.pop_grad(src)
            if dsrc is None:
                return
            assert dsrc.is_contiguous()
            dy = c.empty((B, Lv, OC))
            dgam = (ctypes.c_void_p * L)(*[_dp(g[2]) for g in gns])
            dbet = (ctypes.c_void_p * L)(*[_dp(g[3]) for g in gns])
            _L().call("emrt_groupnorm_levels_bwd", P(y), OC, Lv * OC, P(dsrc), OC, Lv * OC, P(dy), OC, Lv * OC, gam, bet, P(mean), P(rstd), dgam,
                      dbet, starts, hws, L, B, OC, G, 0, P(c.zeros_f64(L * B * G * 2)), c.dtype, c.stream)
            slots = [tape.grad_slot(f) for f in feats]
            dxs = [s_ if s_ is not None else c.empty(tuple(f.shape)) for s_, f in zip(slots, feats)]
            bd = (_ConvBwdDesc * L)()
            deferred = all(wgrad_deferred(w) for w in convs)
            if not deferred:
                for w in convs:
                    w.grad_is_zero = False
            for l, (w, f, (a, n), dx) in enumerate(zip(convs, feats, spans, dxs)):
                _, h, wd, Cl = f.shape
                _, _, _, _, lddx, dx_bs = _check_map(dx)
                d = bd[l]
                d.x, d.dy, d.w_bwd_packed, d.dx = f.data_ptr(), dy.data_ptr() + a * OC * esz, w.bwd_ptr, dx.data_ptr()
                d.lddx, d.dx_bs, d.accumulate = lddx, dx_bs, int(slots[l] is not None)
                d.dw = None if deferred else w.grad.data_ptr()
                d.dbias = None if deferred else (_dp(w.bias_grad) if w.bias is not None else None)
                if deferred:
                    defer_wgrad(tape, f, dy.narrow(1, a, n), w, (B, h, wd, Cl, Cl, h * wd * Cl, h, wd, OC, Lv * OC), 1, 0, 1)
                d.N, d.H, d.W, d.C, d.ldx, d.x_bs = B, h, wd, Cl, Cl, h * wd * Cl
                d.OH, d.OW, d.OC, d.lddy, d.dy_bs = h, wd, OC, OC, Lv * OC
                d.KH, d.KW, d.stride, d.pad = 1, 1, 1, 0
            _L().call("emrt_conv2d_bwd_group", bd, L, c.dtype, c.stream)
            for s_, f, dx in zip(slots, feats, dxs):
                if s_ is None:
                    tape.add_grad(f, dx, owned=True)
        tape.record(bwd)
    return src, spans


def layer_norm(a, b, gamma, beta, dgamma, dbeta, post=None, eps=1e-5, drop_p=0.0, drop_salt=0, identity_from=None, q_pos=None, q_bgrad=None):
    """out = LN(a + dropout(b)) * gamma + beta (+ post);  a, b, post contiguous [.., C].  The inverted dropout on the branch
    input b (every residual LayerNorm of the transformer has one in front: t_e_d.py:199,202,287,291,294) runs inside the
    LayerNorm kernels: the forward drops b on the fly, the backward emits dz for a and the masked dz for b.
    identity_from: a tensor t computed EARLIER from `a` with an identity path (t = f(a) + a: level_conv_gn's output) whose consumers all ran
    AFTER this call, so that t's gradient is complete when this backward runs: the identity's contribution d a += d t is then summed by this
    kernel (dz_addend) and t's producer is told not to add it again (Tape.identity_done) -- one accumulate launch per encoder layer less.
    q_pos ([Lp, C] contiguous, a is [B, Lp, C]): also returns q = out + q_pos (broadcast over the batch) -- the next attention's query
    with_pos_embed(out, pos) -- written by the same launch; returns (out, q).  Backward: q's gradient is summed with out's as the kernel loads
    them (dy2) and handed to q_bgrad (the embedding's gradient, as Fn.add(bgrad=)) -- no add launch forward, no accumulate launch backward."""
    c = ctx()
    assert a.is_contiguous() and (b is None or b.is_contiguous()) and (post is None or post.is_contiguous())
    C = a.shape[-1]
    rows = a.numel() // C
    out = c.empty(tuple(a.shape))
    q = None
    if q_pos is not None:
        assert q_pos.is_contiguous() and q_pos.dtype == a.dtype and q_pos.shape[-1] == C and rows % (q_pos.numel() // C) == 0
        q = c.empty(tuple(a.shape))
    keep = c.tape is not None
    p = float(drop_p) if (c.training and b is not None) else 0.0
    z = c.empty(tuple(a.shape)) if (keep and b is not None) else None
    mean = c.empty((rows,), torch.float32) if keep else None
    rstd = c.empty((rows,), torch.float32) if keep else None
    _L().call("emrt_layernorm_fwd", P(a), P(b), P(post), P(z), P(out), P(gamma), P(beta), P(mean), P(rstd), rows, C, eps, p,
              c.seed_ptr if p > 0 else None, drop_salt, P(q_pos) if q is not None else None, (q_pos.numel() // C) if q is not None else 0, P(q),
              c.dtype, c.stream)
    tape = c.tape
    if tape is not None:
        zz = z if z is not None else a

        def bwd():
            dy = tape.pop_grad(out)
            dq = tape.pop_grad(q) if q is not None else None
            if dq is not None:
                assert dq.is_contiguous() and dq.dtype == out.dtype
                if q_bgrad is not None:
                    q_bgrad(dq)
                if dy is None:
                    dy, dq = dq, None
            if dy is None:
                return
            assert dy.is_contiguous()
            dz = c.empty(tuple(a.shape))
            dzb = c.empty(tuple(a.shape)) if (b is not None and p > 0) else None
            ws = c.workspace(_L().query("emrt_layernorm_bwd_workspace_bytes", rows, C))
            dysum = c.empty(tuple(a.shape)) if (dq is not None and post is not None) else None      # (the post addend saw out AND q = out + q_pos)
            extra = tape.peek_grad(identity_from) if identity_from is not None else None
            if extra is not None and (not extra.is_contiguous() or tuple(extra.shape) != tuple(a.shape) or extra.dtype != dz.dtype or (b is not None and dzb is None)):
                extra = None                      # (without branch dropout a and b share dz: the addend must not reach b)
            _L().call("emrt_layernorm_bwd", P(zz), P(dy), P(dz), P(gamma), P(mean), P(rstd), P(dgamma), P(dbeta), rows, C, P(ws), P(dzb), p,
                      c.seed_ptr if p > 0 else None, drop_salt, P(extra), P(dq), P(dysum), c.dtype, c.stream)
            if extra is not None:
                tape.identity_done[id(identity_from)] = (extra, tape.grad_count(identity_from))
            # dz is a's alone when the branch got its own (masked) gradient: handed over, so that the next contribution to a -- the data gradient of
            # the GEMM that read a -- accumulates into it in its epilogue instead of through an add launch
            tape.add_grad(a, dz, owned=(b is None or dzb is not None))
            if b is not None:
                if dzb is not None:
                    tape.add_grad(b, dzb, owned=True)
                else:
                    tape.add_grad(b, dz)
            if post is not None:
                tape.add_grad(post, dy if dysum is None else dysum, owned=dysum is not None)
        tape.record(bwd)
    return out if q is None else (out, q)


# ---------------------------------------------------------------------------------------------------
# attention
# ---------------------------------------------------------------------------------------------------
def msda(value, offw, ref, shapes, n_heads, n_points, need_dref=False):
    """value [B,Lv,M*32] (T), offw fp32 [B,Lq,ldo], ref fp32 [B or 1, Lq, L, 2] -> [B,Lq,M*32] (T)."""
    c = ctx()
    B, Lv, CC = value.shape
    M, L, Pn = n_heads, len(shapes), n_points
    assert CC == M * 32 and value.stride(2) == 1
    Lq, ldo = offw.shape[1], offw.shape[2]
    assert offw.is_contiguous() and offw.dtype == torch.float32 and ref.dtype == torch.float32 and ref.is_contiguous()
    ref_L = ref.shape[2]
    assert ref.shape[1] == Lq and ref_L in (1, L) and ref.shape[3] == 2
    ref_bs = 0 if ref.shape[0] == 1 else Lq * ref_L * 2
    arr = (ctypes.c_int * (2 * L))(*[int(v) for hw in shapes for v in hw])
    out = c.empty((B, Lq, CC))
    _L().call("emrt_msda_fwd", P(value), value.stride(1), value.stride(0), P(offw), ldo, P(ref), ref_bs, ref_L, P(out), B, Lq, Lv, M, 32, L, Pn,
              ctypes.cast(arr, ctypes.c_void_p), c.dtype, c.stream)
    tape = c.tape
    if tape is not None:
        def bwd():
            dy = tape.pop_grad(out)
            if dy is None:
                return
            assert dy.is_contiguous()
            use_lds = bool(_L().query("emrt_msda_bwd_uses_lds", ctypes.cast(arr, ctypes.c_void_p), L))
            odt = None if c.dtype != F32 else torch.float32      # compute dtype: the projection's backward GEMM reads it directly
            doffw = c.zeros((B, Lq, ldo), odt) if ldo != M * L * Pn * 3 else c.empty((B, Lq, ldo), odt)
            dref = c.zeros((B, Lq, ref_L, 2), torch.float32) if need_dref else None      # zeroed: the LDS gradient kernel accumulates into it
            if use_lds:
                dvalue = c.empty((B, Lv, CC))          # compute dtype, fully overwritten by the LDS scatter
                ws = c.empty((_L().query("emrt_msda_bwd_workspace_bytes", B, Lq, M, L, Pn, ctypes.cast(arr, ctypes.c_void_p), c.dtype) // 4,), torch.float32)
            else:
                dvalue = c.zeros((B, Lv, CC), torch.float32)
                ws = None
            _L().call("emrt_msda_bwd", P(value), value.stride(1), value.stride(0), P(offw), ldo, P(ref), ref_bs, ref_L, P(dy), P(dvalue), P(doffw),
                      int(c.dtype != F32), P(dref), B, Lq, Lv, M, 32, L, Pn, ctypes.cast(arr, ctypes.c_void_p), P(ws), ws.numel() * 4 if ws is not None else 0,
                      c.dtype, c.stream)
            tape.add_grad(value, dvalue if use_lds else cast_from_f32(dvalue), owned=True)
            tape.add_grad(offw, doffw, owned=True)
            if need_dref:
                if ref.shape[0] == 1 and B > 1:   # reference points shared by the batch: reduce over b
                    red = c.zeros((1, Lq, ref_L, 2), torch.float32)
                    colsum_acc(dref.view(B, Lq * ref_L * 2), red.view(-1))
                    dref = red
                tape.add_grad(ref, dref)
        tape.record(bwd)
    return out


def mha(qk, v, n_heads, pdrop, salt):
    """qk [B,L,2E] (q | k), v [B,L,E] -> [B,L,E]; softmax(q k^T / sqrt(d)) v with dropout on the weights."""
    c = ctx()
    B, L, E2 = qk.shape
    E = E2 // 2
    assert qk.is_contiguous() and v.is_contiguous() and E == n_heads * 32
    out = c.empty((B, L, E))
    probs = c.empty((B, n_heads, L, L), torch.float32)
    p = float(pdrop) if c.training else 0.0
    scale = 1.0 / math.sqrt(32.0)
    q_ptr = qk.data_ptr()
    k_ptr = qk.data_ptr() + E * qk.element_size()
    path = ctypes.c_int(-1)          # which kernel filled `probs` (row statistics or L x L probabilities): the backward is told, it does not guess
    _L().call("emrt_mha_fwd", ctypes.c_void_p(q_ptr), E2, ctypes.c_void_p(k_ptr), E2, P(v), E, P(out), E, P(probs), B, n_heads, L, 32, scale, p,
              c.seed_ptr, salt, ctypes.pointer(path), c.dtype, c.stream)
    tape = c.tape
    if tape is not None:
        def bwd():
            dy = tape.pop_grad(out)
            if dy is None:
                return
            assert dy.is_contiguous()
            dqk = c.empty((B, L, E2))
            dv = c.empty((B, L, E))
            _L().call("emrt_mha_bwd", ctypes.c_void_p(q_ptr), E2, ctypes.c_void_p(k_ptr), E2, P(v), E, P(probs), P(dy), E,
                      ctypes.c_void_p(dqk.data_ptr()), E2, ctypes.c_void_p(dqk.data_ptr() + E * dqk.element_size()), E2, P(dv), E,
                      B, n_heads, L, 32, scale, p, c.seed_ptr, salt, path.value, c.dtype, c.stream)
            tape.add_grad(qk, dqk, owned=True)
            tape.add_grad(v, dv, owned=True)
        tape.record(bwd)
    return out


# ---------------------------------------------------------------------------------------------------
# spatial ops
# ---------------------------------------------------------------------------------------------------
def nchw_to_nhwc(img, c_out=None):
    """fp32 [N,C,H,W] -> compute-dtype [N,H,W,c_out] (c_out >= C, extra channels zero)."""
    c = ctx()
    N, C, H, W = img.shape
    assert img.dtype == torch.float32 and img.is_contiguous()
    co = C if c_out is None else int(c_out)
    out = c.empty((N, H, W, co))
    _L().call("emrt_nchw_to_nhwc", P(img), P(out), N, C, H, W, co, c.dtype, c.stream)
    return out


def resize_bilinear(x, OH, OW, align_corners, add_t=None, out=None, out_nchw_f32=False):
    """x [N,IH,IW,C] view -> [N,OH,OW,C] (or fp32 [N,C,OH,OW] when out_nchw_f32), optional fused '+ add_t'."""
    c = ctx()
    if isinstance(x, PendingBN):
        assert add_t is None and not out_nchw_f32
        pend, x = x, x.raw
        N, IH, IW, C, in_ld, in_bs = _check_map(x)
        if out is None:
            out = c.empty((N, OH, OW, C))
        _, _, _, _, out_ld, out_bs = _check_map(out)
        _L().call("emrt_bn_resize_bilinear_fwd", P(x), in_bs, in_ld, IH, IW, P(out), out_bs, out_ld, OH, OW, N, C, int(align_corners),
                  *pend.operand(), c.dtype, c.stream)
        tape = c.tape
        if tape is not None:
            def bwd_pending():
                dy = tape.pop_grad(out)
                if dy is None:
                    return
                da = c.empty((N, IH, IW, C))
                _, _, _, _, do_ld, do_bs = _check_map(dy)
                _L().call("emrt_resize_bilinear_bwd", P(dy), do_bs, do_ld, OH, OW, P(da), IH * IW * C, C, IH, IW, N, C, int(align_corners), 0,
                          None, c.dtype, c.stream)
                pend.backward(tape, da)
            tape.record(bwd_pending)
        return out
    N, IH, IW, C, in_ld, in_bs = _check_map(x)
    if out_nchw_f32:
        assert out is None and add_t is None
        out = c.empty((N, C, OH, OW), torch.float32)
        out_ld, out_bs = 0, 0
    else:
        if out is None:
            out = c.empty((N, OH, OW, C))
        _, _, _, _, out_ld, out_bs = _check_map(out)
    add_ld = add_bs = 0
    if add_t is not None:
        _, _, _, _, add_ld, add_bs = _check_map(add_t)
    _L().call("emrt_resize_bilinear_fwd", P(x), in_bs, in_ld, IH, IW, P(out), out_bs, out_ld, OH, OW, P(add_t), add_bs, add_ld, N, C,
              int(align_corners), int(out_nchw_f32), c.dtype, c.stream)
    tape = c.tape
    if tape is not None:
        def bwd():
            dy = tape.pop_grad(out)
            if dy is None:
                return
            dx = c.empty((N, IH, IW, C))
            if out_nchw_f32:
                assert dy.is_contiguous() and dy.dtype == torch.float32
                do_bs, do_ld = 0, 0
            else:
                _, _, _, _, do_ld, do_bs = _check_map(dy)
            ws = None
            if out_nchw_f32:
                ws = c.workspace(_L().query("emrt_resize_bwd_workspace_bytes", N, C, OH, IW, 1))
            _L().call("emrt_resize_bilinear_bwd", P(dy), do_bs, do_ld, OH, OW, P(dx), IH * IW * C, C, IH, IW, N, C, int(align_corners),
                      int(out_nchw_f32), P(ws), c.dtype, c.stream)
            tape.add_grad(x, dx, owned=True)
            if add_t is not None:
                tape.add_grad(add_t, dy)
        tape.record(bwd)
    return out


def pyramid_tokens_to_maps(tokens, scales, OH, OW, outs):
    """tokens [B, sum k^2, C] (the decoder's pyramid queries) -> for each scale k its k x k map resized (bilinear,
    align_corners=True) to OH x OW and written into outs[i] (channel slices of the concat buffer): paddle_EMRT.py:281-291.
    Forward = one resize launch per scale on token-slab views; the backward writes every scale's gradient straight into its rows
    of ONE token-gradient buffer (the slabs tile it exactly), instead of a zeroed buffer and one accumulate launch per scale."""
    c = ctx()
    B, ntok, C = tokens.shape
    assert tokens.is_contiguous() and ntok == sum(k * k for k in scales) and len(outs) == len(scales)
    esz = tokens.element_size()
    L = len(scales)
    sc_arr = (ctypes.c_int * L)(*scales)
    geo, start, lds, bss = [], 0, [], []
    for k, out in zip(scales, outs):
        _, oh_, ow_, oc_, out_ld, out_bs = _check_map(out)
        assert (oh_, ow_, oc_) == (OH, OW, C)
        geo.append((k, start, out))
        lds.append(out_ld)
        bss.append(out_bs)
        start += k * k
    # one launch for all scales when the shapes are on the vector path (each scale alone is a launch of a few blocks)
    grouped = (c.pyramid_group and L <= 4 and C % 4 == 0 and tokens.data_ptr() % 16 == 0 and
               all(o.data_ptr() % 16 == 0 and ld % 4 == 0 and bs % 4 == 0 for o, ld, bs in zip(outs, lds, bss)))
    if grouped:
        _L().call("emrt_pyramid_resize_fwd", P(tokens), ctypes.cast(sc_arr, ctypes.c_void_p), L, (ctypes.c_void_p * L)(*[o.data_ptr() for o in outs]),
                  (ctypes.c_int * L)(*lds), (ctypes.c_longlong * L)(*bss), OH, OW, B, C, 1, c.dtype, c.stream)
    else:
        for (k, s0, out), out_ld, out_bs in zip(geo, lds, bss):
            _L().call("emrt_resize_bilinear_fwd", ctypes.c_void_p(tokens.data_ptr() + s0 * C * esz), ntok * C, C, k, k, P(out), out_bs, out_ld, OH, OW,
                      None, 0, 0, B, C, 1, 0, c.dtype, c.stream)
    tape = c.tape
    if tape is not None:
        def bwd():
            dys = [tape.pop_grad(out) for _, _, out in geo]
            if all(d is None for d in dys):
                return
            if grouped and all(d is not None for d in dys) and all(OH * OW >= 16 * k * k for k in scales):
                dgeo = [_check_map(d)[4:6] for d in dys]
                if all(d.data_ptr() % 16 == 0 and ld % 4 == 0 and bs % 4 == 0 for d, (ld, bs) in zip(dys, dgeo)):
                    dtok = c.empty((B, ntok, C))
                    _L().call("emrt_pyramid_resize_bwd", (ctypes.c_void_p * L)(*[d.data_ptr() for d in dys]), (ctypes.c_int * L)(*[g_[0] for g_ in dgeo]),
                              (ctypes.c_longlong * L)(*[g_[1] for g_ in dgeo]), OH, OW, P(dtok), ctypes.cast(sc_arr, ctypes.c_void_p), L, B, C, 1,
                              c.dtype, c.stream)
                    tape.add_grad(tokens, dtok, owned=True)
                    return
            dtok = c.empty((B, ntok, C)) if all(d is not None for d in dys) else c.zeros((B, ntok, C))
            for (k, s0, _), dy in zip(geo, dys):
                if dy is None:
                    continue
                _, _, _, _, do_ld, do_bs = _check_map(dy)
                _L().call("emrt_resize_bilinear_bwd", P(dy), do_bs, do_ld, OH, OW, ctypes.c_void_p(dtok.data_ptr() + s0 * C * esz), ntok * C, C, k, k,
                          B, C, 1, 0, None, c.dtype, c.stream)
            tape.add_grad(tokens, dtok, owned=True)
        tape.record(bwd)
    return outs


def adaptive_avgpool_tokens(x, scales):
    """x [N,H,W,C] view -> tokens [N, sum k^2, C] (all pyramid scales in one launch)."""
    c = ctx()
    N, H, W, C, in_ld, in_bs = _check_map(x)
    ntok = sum(k * k for k in scales)
    out = c.empty((N, ntok, C))
    arr = (ctypes.c_int * len(scales))(*scales)
    # fp32 partial-sum slots: large maps are pooled by several blocks per bin (csrc/spatial.hip: adaptive_pool_part_kernel)
    nws = _L().query("emrt_adaptive_avgpool_workspace_bytes", H, W, N, C, ctypes.cast(arr, ctypes.c_void_p), len(scales))
    ws = c.empty((nws // 4,), torch.float32) if nws else None
    _L().call("emrt_adaptive_avgpool_fwd", P(x), in_bs, in_ld, H, W, P(out), ntok * C, C, N, C, ctypes.cast(arr, ctypes.c_void_p), len(scales),
              P(ws), nws, c.dtype, c.stream)
    tape = c.tape
    if tape is not None:
        def bwd():
            dy = tape.pop_grad(out)
            if dy is None:
                return
            assert dy.is_contiguous()
            dx = c.empty((N, H, W, C))
            _L().call("emrt_adaptive_avgpool_bwd", P(dy), ntok * C, C, P(dx), H * W * C, C, H, W, N, C, ctypes.cast(arr, ctypes.c_void_p),
                      len(scales), c.dtype, c.stream)
            tape.add_grad(x, dx, owned=True)
        tape.record(bwd)
    return out


def maxpool(x, k=3, stride=2, pad=1, need_dx=True):
    c = ctx()
    pend = None
    if isinstance(x, PendingBN):
        pend, x = x, x.raw
    assert x.is_contiguous()
    N, H, W, C = x.shape
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    out = c.empty((N, OH, OW, C))
    tape = c.tape
    keep = tape is not None and need_dx
    arg = c.empty((N, OH, OW, C), torch.uint8) if keep else None      # winning window slot per output, for the backward
    if pend is not None:
        _L().call("emrt_bn_maxpool_fwd", P(x), P(out), P(arg), N, H, W, C, k, stride, pad, *pend.operand(), c.dtype, c.stream)
    else:
        _L().call("emrt_maxpool_fwd", P(x), P(out), P(arg), N, H, W, C, k, stride, pad, c.dtype, c.stream)
    if keep:
        def bwd():
            dy = tape.pop_grad(out)
            if dy is None:
                return
            assert dy.is_contiguous()
            dx = c.empty((N, H, W, C))
            _L().call("emrt_maxpool_bwd", P(arg), P(dy), P(dx), N, H, W, C, k, stride, pad, c.dtype, c.stream)
            if pend is not None:
                pend.backward(tape, dx)
            else:
                tape.add_grad(x, dx, owned=True)
        tape.record(bwd)
    return out


def sigmoid_f32(x):
    c = ctx()
    assert x.dtype == torch.float32 and x.is_contiguous()
    y = c.empty(tuple(x.shape), torch.float32)
    _L().call("emrt_sigmoid_fwd", P(x), P(y), x.numel(), c.stream)
    tape = c.tape
    if tape is not None:
        def bwd():
            dy = tape.pop_grad(y)
            if dy is None:
                return
            dx = c.empty(tuple(x.shape), torch.float32)
            _L().call("emrt_sigmoid_bwd", P(y), P(dy), P(dx), x.numel(), c.stream)
            tape.add_grad(x, dx, owned=True)
        tape.record(bwd)
    return y


# ---------------------------------------------------------------------------------------------------
# loss
# ---------------------------------------------------------------------------------------------------
def softmax_ce_pair(logits_a, logits_b, labels, ignore_index, wa, wb):
    """CE of two heads on the same labels (MixSoftmaxCrossEntropyLoss: main + aux) in one forward pass and one backward launch.
    Returns (res_a, res_b, total): device float[2] = {loss, count} per head and float[1] = wa * loss_a + wb * loss_b."""
    c = ctx()
    N, C, H, W = logits_a.shape
    assert tuple(logits_b.shape) == (N, C, H, W) and logits_a.dtype == logits_b.dtype == torch.float32 and logits_a.is_contiguous() and logits_b.is_contiguous()
    assert labels.dtype == torch.int64 and labels.is_contiguous()
    res_a, res_b, total = c.empty((2,), torch.float32), c.empty((2,), torch.float32), c.empty((1,), torch.float32)
    ws = c.workspace(_L().query("emrt_ce_workspace_bytes"))
    _L().call("emrt_softmax_ce_pair_fwd", P(logits_a), P(logits_b), P(labels), N, C, H, W, ignore_index, float(wa), float(wb), P(res_a), P(res_b), P(total),
              P(ws), c.stream)
    tape = c.tape
    if tape is not None:
        def bwd():
            up_a, up_b = tape.pop_grad(res_a), tape.pop_grad(res_b)      # device scalars or None (== 1)
            da, db = c.empty((N, C, H, W), torch.float32), c.empty((N, C, H, W), torch.float32)
            _L().call("emrt_softmax_ce_pair_bwd", P(logits_a), P(logits_b), P(labels), P(res_a), P(up_a), P(up_b), float(wa), float(wb), N, C, H, W,
                      ignore_index, P(da), P(db), c.stream)
            tape.add_grad(logits_a, da, owned=True)
            tape.add_grad(logits_b, db, owned=True)
        tape.record(bwd)
    return res_a, res_b, total


def softmax_ce(logits, labels, ignore_index, weight=1.0):
    """Mean CE over non-ignored pixels of fp32 NCHW logits; returns a device float[2] = {loss, count}.
    Backward writes weight * upstream * (softmax - onehot)/count as the gradient of `logits`."""
    c = ctx()
    N, C, H, W = logits.shape
    assert logits.dtype == torch.float32 and logits.is_contiguous() and labels.dtype == torch.int64 and labels.is_contiguous()
    res = c.empty((2,), torch.float32)
    ws = c.workspace(_L().query("emrt_ce_workspace_bytes"))
    _L().call("emrt_softmax_ce_fwd", P(logits), P(labels), N, C, H, W, ignore_index, P(res), P(ws), c.stream)
    tape = c.tape
    if tape is not None:
        def bwd():
            up = tape.pop_grad(res)   # device scalar or None (== 1)
            dl = c.empty((N, C, H, W), torch.float32)
            _L().call("emrt_softmax_ce_bwd", P(logits), P(labels), P(res), P(up), float(weight), N, C, H, W, ignore_index, P(dl), c.stream)
            tape.add_grad(logits, dl, owned=True)
        tape.record(bwd)
    return res
