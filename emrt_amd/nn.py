"""Layer classes and the flat parameter store of the HIP path.

Layers subclass torch.nn.Module only for parameter registration / naming / state_dict (keys mirror the reference,
SURVEY.md Appendix A); their forward methods launch HIP kernels through emrt_amd.functional.

ParamStore lays every parameter out in ONE flat fp32 device buffer (master), with twin flat buffers for gradients
and momentum, so that gradient clipping, the SGD update and the RCCL gradient all-reduce each touch one contiguous
range, and keeps the packed compute-dtype copies of all GEMM weights (forward layout [OC][KH][KW][C] and the
transposed dgrad layout [C][KH][KW][OC]) that emrt_pack_weights refreshes once per optimizer step.
Conv weights keep the reference's logical shape [OC, C, KH, KW] in the state dict but live in memory as OHWI.
"""

import torch
import torch.nn as tnn

from . import _lib
from . import functional as Fn
from .runtime import ctx, F32, _TORCH_DTYPE


# element alignment of weight matrices in the flat buffers / of their transposed copies (measured 8 ... 2048: 729 -> 744 tiles/s;
# flat from 128 on = 256 bytes of the bf16 mirror; larger only pads the range the optimizer and the all-reduce cover)
_PARAM_ALIGN = 128
_BWD_ALIGN = 128


def _align(n, a):
    return (n + a - 1) // a * a


class ParamStore:
    def __init__(self, model, device, dtype, nograd_names=(), fused_groups=(), lr_mult_names=(), lr_mult=0.1):
        self.dtype = dtype
        self.device = device
        named = list(model.named_parameters())
        by_name = dict(named)
        in_group = {}
        for grp in fused_groups:
            for n in grp:
                in_group[n] = grp
        order, seen = [], set()
        for n, _ in named:
            if n in seen or n in nograd_names:
                continue
            for m in in_group.get(n, [n]):
                order.append(m)
                seen.add(m)
        self.n_trainable_names = len(order)
        order += [n for n, _ in named if n in nograd_names]
        self.offsets, off = {}, 0
        fused_follow = {m for grp in fused_groups for m in grp[1:]}
        self.n_train = 0
        # conv weights stored with MORE input channels than the layer has (Conv2D(pad_cin=)): the extra channels stay zero (zero
        # input, zero gradient, zero decay) and the parameter is the [:, :cin] slice of the stored tensor
        self.padded_cin = {}
        for mname, mod in model.named_modules():
            if getattr(mod, "pad_cin", None) and isinstance(getattr(mod, "weight", None), tnn.Parameter):
                self.padded_cin[(mname + "." if mname else "") + "weight"] = int(mod.pad_cin)
        numel = lambda n: by_name[n].numel() // by_name[n].shape[1] * self.padded_cin[n] if n in self.padded_cin else by_name[n].numel()
        for i, n in enumerate(order):
            p = by_name[n]
            if n not in fused_follow:
                # 16 bytes in the bf16 / fp16 mirror of this buffer (the GEMMs' 16-byte operand loads); weight matrices start on a
                # 256-byte boundary of the mirror: with 16-byte alignment only, every k-row of a GEMM operand straddled cache lines
                # and the weight-bound layers (8x8 / 16x16 maps) ran 10-20 % slower
                off = _align(off, _PARAM_ALIGN if p.dim() >= 2 else 8)
            else:
                assert off % 8 == 0, "fused parameter group member %s must start 16-byte aligned in the compute-dtype mirror" % n
            self.offsets[n] = off
            off += numel(n)
            if i == self.n_trainable_names - 1:
                self.n_train = _align(off, 8)     # [0, n_train) is what clip / SGD / all-reduce cover
        self.n_total = _align(off, 8)
        self.master = torch.zeros(self.n_total, dtype=torch.float32, device=device)
        self.grad = torch.zeros(self.n_total, dtype=torch.float32, device=device)
        self.velocity = torch.zeros(self.n_total, dtype=torch.float32, device=device)
        self.views, self.shapes = {}, {}
        for n in order:
            p = by_name[n]
            a, cnt = self.offsets[n], numel(n)
            flat, gflat = self.master[a:a + cnt], self.grad[a:a + cnt]
            if p.dim() == 4:    # conv weight: logical [OC,C,KH,KW], memory [OC][KH][KW][C (padded)]
                OC, C, KH, KW = p.shape
                Cp = self.padded_cin.get(n, C)
                view = flat.view(OC, KH, KW, Cp).permute(0, 3, 1, 2)[:, :C]
                gview = gflat.view(OC, KH, KW, Cp).permute(0, 3, 1, 2)[:, :C]
            else:
                view, gview = flat.view(p.shape), gflat.view(p.shape)
            view.copy_(p.data.to(device=device, dtype=torch.float32))
            p.data = view
            p.grad = gview
            p.requires_grad_(False)
            self.views[n] = (a, cnt)
            self.shapes[n] = tuple(p.shape)
        self.train_order = order[:self.n_trainable_names]
        self.lr_ranges = [(self.offsets[n], self.offsets[n] + numel(n)) for n in lr_mult_names]
        self.lr_mult = lr_mult
        # buffers (BN running statistics): one flat fp32 buffer
        bufs = [(n, b) for n, b in model.named_buffers()]
        tot = sum(_align(b.numel(), 4) for _, b in bufs)
        self.buffers = torch.zeros(max(tot, 4), dtype=torch.float32, device=device)
        boff = 0
        for n, b in bufs:
            mod_name, _, bname = n.rpartition(".")
            mod = model.get_submodule(mod_name) if mod_name else model
            view = self.buffers[boff:boff + b.numel()].view(b.shape)
            view.copy_(b.to(device=device, dtype=torch.float32))
            mod._buffers[bname] = view
            boff += _align(b.numel(), 4)
        self.gemms = []
        self.packed = None
        self.desc = None
        self.total_tiles = self.total_tiles64 = 0
        self.dirty = True
        self.bn_states, self.bn_fold, self.bn_desc = [], None, None

    def named_view(self, flat, name):
        """The logical-shape view of parameter `name` inside a flat buffer laid out like master / grad / velocity."""
        a, cnt = self.views[name]
        shape = self.shapes[name]
        if len(shape) == 4:
            OC, C, KH, KW = shape
            Cp = self.padded_cin.get(name, C)
            return flat[a:a + cnt].view(OC, KH, KW, Cp).permute(0, 3, 1, 2)[:, :C]
        return flat[a:a + cnt].view(shape)

    def segment_ranges(self, segment_prefixes):
        """Partition [0, n_train) of the flat buffers by parameter-name prefix: returns [ranges_0, ranges_1, ...] where
        ranges_i (i >= 1) holds the parameters whose name starts with one of segment_prefixes[i-1] and ranges_0 everything
        else; each a list of [start, end).  Alignment padding goes with the run before it."""
        def seg_of(n):
            for i, pre in enumerate(segment_prefixes):
                if n.startswith(tuple(pre)):
                    return i + 1
            return 0
        runs = []
        for n in self.train_order:
            k = seg_of(n)
            if not runs or runs[-1][0] != k:
                runs.append([k, self.offsets[n]])
        out = [[] for _ in range(len(segment_prefixes) + 1)]
        for i, (k, start) in enumerate(runs):
            end = runs[i + 1][1] if i + 1 < len(runs) else self.n_train
            out[k].append((start, end))
        return out

    def split_ranges(self, late_prefixes):
        """(early, late): segment_ranges with one segment."""
        early, late = self.segment_ranges([tuple(late_prefixes)])
        return early, late

    # ---- GEMM weights ---------------------------------------------------------------------------
    def make_gemm(self, offset, OC, C, KH=1, KW=1, bias_offset=None, need_bwd=True):
        g = Fn.GemmWeight(OC, C, KH, KW)
        g.offset = offset
        g.need_bwd = need_bwd
        cnt = OC * KH * KW * C
        g.grad = self.grad[offset:offset + cnt]
        if bias_offset is not None:
            g.bias = self.master[bias_offset:bias_offset + OC]
            g.bias_grad = self.grad[bias_offset:bias_offset + OC]
        self.gemms.append(g)
        return g

    def finalize(self):
        """Allocate the packed-weight buffer and the device descriptor table (after every layer registered its GEMMs)."""
        esz = 4 if self.dtype == F32 else 2
        # compute-dtype modes: packed = [ mirror of the whole master buffer (same indexing: the forward operand of GEMM g starts
        # at g.offset, and the optimizer refreshes it in its update pass) | transposed dgrad copies ]; fp32: dgrad copies only
        off = 0 if self.dtype == F32 else _align(self.n_total, _BWD_ALIGN)
        self.mirror_elems = off
        rows, tiles, tiles64 = [], 0, 0
        for g in self.gemms:
            cnt = g.OC * g.KH * g.KW * g.C
            fo = bo = -1
            if self.dtype != F32:
                fo = g.offset
            if g.need_bwd:
                bo = off
                off = _align(off + cnt, _BWD_ALIGN)      # 256-byte aligned copies (see the parameter offsets above)
            g._fo, g._bo = fo, bo
            if fo < 0 and bo < 0:
                continue
            rows.append([g.offset, fo, bo, g.OC, g.KH * g.KW, g.C, tiles, tiles64])
            tiles += g.KH * g.KW * ((g.OC + 31) // 32) * ((g.C + 31) // 32)
            tiles64 += g.KH * g.KW * ((g.OC + 63) // 64) * ((g.C + 63) // 64)
        self.packed = torch.zeros(max(off, 8), dtype=_TORCH_DTYPE[self.dtype], device=self.device)
        base = self.packed.data_ptr()
        for g in self.gemms:
            g.fwd_ptr = self.master.data_ptr() + 4 * g.offset if g._fo < 0 else base + esz * g._fo
            g.bwd_ptr = None if g._bo < 0 else base + esz * g._bo
            g._keepalive = (self.packed, self.master, self.grad)   # raw device addresses above point into these
        self.desc = torch.tensor(rows, dtype=torch.int64).to(self.device) if rows else None
        self.ndesc, self.total_tiles, self.total_tiles64 = len(rows), tiles, tiles64
        self.dirty = True

    def register_bn(self, st):
        self.bn_states.append(st)

    def fold_bn(self):
        """Inference: every BatchNorm as (scale, shift) of its running statistics, one launch at the top of each eval forward (so a
        state-dict load, an optimizer step or a direct edit of the buffers is always seen); conv_bn folds them into the conv."""
        if not self.bn_states:
            return
        if self.bn_desc is None:
            rows, off = [], 0
            m0, b0 = self.master.data_ptr(), self.buffers.data_ptr()
            for st in self.bn_states:
                assert st.eps == self.bn_states[0].eps
                cb = st.fold_conv.gw.bias if st.fold_conv is not None else None
                rows.append([(st.gamma.data_ptr() - m0) // 4, (st.beta.data_ptr() - m0) // 4, (st.run_mean.data_ptr() - b0) // 4,
                             (st.run_var.data_ptr() - b0) // 4, st.C, off, (cb.data_ptr() - m0) // 4 if cb is not None else -1])
                off += 2 * st.C
            assert all(0 <= r[0] < self.master.numel() and 0 <= r[2] < self.buffers.numel() for r in rows)
            self.bn_fold = torch.empty(off, dtype=torch.float32, device=self.device)
            self.bn_desc = torch.tensor(rows, dtype=torch.int64).to(self.device)
            for st, r in zip(self.bn_states, rows):
                st.fold_scale, st.fold_shift = self.bn_fold[r[5]:r[5] + st.C], self.bn_fold[r[5] + st.C:r[5] + 2 * st.C]
        _lib.lib().call("emrt_bn_fold", Fn.P(self.master), Fn.P(self.buffers), Fn.P(self.bn_desc), len(self.bn_states),
                        self.bn_states[0].eps, Fn.P(self.bn_fold), ctx().stream)

    def pack(self, bwd_only=False):
        """Refresh the compute-dtype weight copies from the master buffer.  bwd_only: only the transposed dgrad copies (the
        optimizer step has just written the forward mirror itself)."""
        if self.desc is not None:
            _lib.lib().call("emrt_pack_weights", Fn.P(self.master), Fn.P(self.packed), Fn.P(self.desc), self.ndesc, self.total_tiles,
                            self.total_tiles64, int(bool(bwd_only)), self.dtype, ctx().stream)
        self.dirty = False

    @property
    def mirror(self):
        """Compute-dtype copy of the master buffer (None in fp32 mode), element i <-> master[i]."""
        return None if self.dtype == F32 or self.packed is None else self.packed[:self.mirror_elems]

    def zero_grad(self):
        _lib.lib().call("emrt_memset", Fn.P(self.grad), 0, self.n_train * 4, ctx().stream)
        for g in self.gemms:          # the next weight gradient of each GEMM weight is its first contribution: it may store instead of add
            g.grad_is_zero = True


# ---------------------------------------------------------------------------------------------------
# layers
# ---------------------------------------------------------------------------------------------------
class HipLayer(tnn.Module):
    def bind(self, store, prefix):
        """Create kernel-side views (GemmWeight / BNState ...) once parameters live in the flat store."""


class Conv2D(HipLayer):
    def __init__(self, cin, cout, k, stride=1, padding=0, bias=True, need_dx=True, dilation=1, pad_cin=None):
        """pad_cin: the layer is fed maps with pad_cin >= cin channels whose extra channels are zero (the image as an 8-channel
        map, Fn.nchw_to_nhwc(c_out=)); its weights are stored with that many input channels (ParamStore.padded_cin)."""
        super().__init__()
        self.cin, self.cout, self.k, self.stride, self.padding, self.need_dx = cin, cout, k, stride, padding, need_dx
        self.dilation = dilation
        self.pad_cin = pad_cin if pad_cin and pad_cin > cin else None
        self.weight = tnn.Parameter(torch.empty(cout, cin, k, k))
        self.bias = tnn.Parameter(torch.zeros(cout)) if bias else None
        self.gw = None

    def bind(self, store, prefix):
        self.gw = store.make_gemm(store.offsets[prefix + "weight"], self.cout, store.padded_cin.get(prefix + "weight", self.cin), self.k, self.k,
                                  store.offsets[prefix + "bias"] if self.bias is not None else None, need_bwd=self.need_dx)

    def forward(self, x, relu=False, residual=None, out=None, out_f32=False):
        return Fn.conv2d(x, self.gw, self.stride, self.padding, relu=relu, residual=residual, out=out, out_f32=out_f32, need_dx=self.need_dx,
                         dilation=self.dilation)


class Linear(HipLayer):
    """weight [out, in] (torch convention; the reference's Paddle layout is [in, out])."""

    def __init__(self, cin, cout, bias=True):
        super().__init__()
        self.cin, self.cout = cin, cout
        self.weight = tnn.Parameter(torch.empty(cout, cin))
        self.bias = tnn.Parameter(torch.zeros(cout)) if bias else None
        self.gw = None
        self.standalone = True   # False when a parent fuses this weight into a wider GEMM

    def bind(self, store, prefix):
        if self.standalone:
            self.gw = store.make_gemm(store.offsets[prefix + "weight"], self.cout, self.cin, 1, 1,
                                      store.offsets[prefix + "bias"] if self.bias is not None else None)

    def forward(self, x, relu=False, out_f32=False, drop=None):
        return Fn.linear(x, self.gw, relu=relu, out_f32=out_f32, drop=drop)


class BatchNorm2D(HipLayer):
    """Paddle-semantics BN (momentum 0.9 => running = 0.9*running + 0.1*batch; biased running variance); buffers keep
    Paddle's names `_mean` / `_variance`.  sync=True marks the reference's nn.SyncBatchNorm layers."""

    def __init__(self, c, sync=False, after=None):
        """after: the Conv2D feeding this layer when that conv has a bias of its own (the eval-mode fold absorbs it)."""
        super().__init__()
        self.c = self.C = c
        self.weight = tnn.Parameter(torch.ones(c))
        self.bias = tnn.Parameter(torch.zeros(c))
        self.register_buffer("_mean", torch.zeros(c))
        self.register_buffer("_variance", torch.ones(c))
        self.state = Fn.BNState(c, 1e-5, 0.9, sync)
        self.state.fold_conv = after          # on the plain state object: not a sub-module, not in the state dict

    def bind(self, store, prefix):
        st = self.state
        st.gamma, st.beta = self.weight.data.view(-1), self.bias.data.view(-1)
        st.dgamma, st.dbeta = self.weight.grad.view(-1), self.bias.grad.view(-1)
        st.run_mean, st.run_var = self._buffers["_mean"], self._buffers["_variance"]
        store.register_bn(st)

    def forward(self, x, relu=False, residual=None, out=None, sums=None):
        return Fn.batch_norm(x, self.state, relu=relu, residual=residual, out=out, sums=sums)


class GroupNorm(HipLayer):
    def __init__(self, groups, c):
        super().__init__()
        self.groups, self.c = groups, c
        self.weight = tnn.Parameter(torch.ones(c))
        self.bias = tnn.Parameter(torch.zeros(c))

    def forward(self, x, gelu=False, residual=None, out=None):
        return Fn.group_norm(x, self.weight.data, self.bias.data, self.weight.grad, self.bias.grad, self.groups, 1e-5,
                             gelu=gelu, residual=residual, out=out)


class LayerNorm(HipLayer):
    def __init__(self, c):
        super().__init__()
        self.weight = tnn.Parameter(torch.ones(c))
        self.bias = tnn.Parameter(torch.zeros(c))

    def forward(self, a, b=None, post=None, drop_p=0.0, drop_salt=0, identity_from=None, q_pos=None, q_bgrad=None):
        """LN(a + dropout(b)) (+ post): the dropout of the residual branch runs inside the LayerNorm kernels.  q_pos: also returns
        out + q_pos (the next attention's query) from the same launch -> (out, q)."""
        return Fn.layer_norm(a, b, self.weight.data, self.bias.data, self.weight.grad, self.bias.grad, post=post, drop_p=drop_p,
                             drop_salt=drop_salt, identity_from=identity_from, q_pos=q_pos, q_bgrad=q_bgrad)


class Embedding(HipLayer):
    def __init__(self, n, c):
        super().__init__()
        self.weight = tnn.Parameter(torch.empty(n, c))


class Sequential(HipLayer):
    """Index-named container (keys '0', '1', ... as paddle.nn.Sequential); entries may be None placeholders for the
    parameter-free layers of the reference (ReLU, pooling, dropout) so that state-dict indices line up."""

    def __init__(self, *mods):
        super().__init__()
        for i, m in enumerate(mods):
            if m is not None:
                self.add_module(str(i), m)

    def __getitem__(self, i):
        return self._modules[str(i)]


def bind_all(model, store):
    for name, mod in model.named_modules():
        if isinstance(mod, HipLayer):
            mod.bind(store, name + "." if name else "")
    store.finalize()
