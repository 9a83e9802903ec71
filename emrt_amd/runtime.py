"""Execution context for the HIP path: compute dtype, stream, backward tape, scratch workspaces.

PyTorch is used here only as plumbing (device memory through its caching allocator, the current HIP stream, hipGraph
capture via torch.cuda.graphs, torch.distributed/RCCL).  All arithmetic is done by libemrt_hip.so kernels launched
on torch's current stream, so a whole training step can be captured into one hipGraph and replayed.

The backward pass is an explicit tape of closures (no torch.autograd): every functional op that runs while a tape is
active appends a closure that reads the gradient of its output(s), launches the backward kernels and hands gradients
to its inputs through Tape.add_grad.  Views (token slabs, concat-buffer channel slices) are registered as aliases of
their base buffer so gradients land in the right sub-region.
"""
import ctypes

import torch

from . import _lib

F32, BF16, F16 = 0, 1, 2          # include/emrt_hip.h: EMRT_DTYPE_*; F16 is inference-only
_TORCH_DTYPE = {F32: torch.float32, BF16: torch.bfloat16, F16: torch.float16}
_DTYPE_OF_TORCH = {v: k for k, v in _TORCH_DTYPE.items()}


def dtype_of(t):
    """C-ABI dtype code of a tensor's element type."""
    return _DTYPE_OF_TORCH[t.dtype]


class Tape:
    def __init__(self):
        self.ops = []
        self.stash = None   # (group id, launch(host_desc_or_None)): a guest data gradient waiting for its host's closure (functional.conv2d, _pair=)
        self.grads = {}
        self.keep = []      # keeps every registered tensor alive so id() keys stay unique for the step
        self.alias = {}     # id(view) -> (base, slicer)
        self.watched = {}   # id(t) -> t : tensors whose gradient survives backward() (tests / debugging)
        self.results = {}
        self.splits = []    # len(ops) at the model's gradient-exchange marks, in forward order (engine.py: backward runs in
                            # segments between them, newest first, and exchanges the gradients each segment completes)
        self.wgrads = []    # weight-gradient problems whose launch is deferred (functional.defer_wgrad): (descriptor fields, tensors kept alive)
        self.identity_done = {}      # id(t) -> the gradient that was summed, for tensors t = f(a) + a whose identity contribution d a += d t a later consumer of a has already summed in
                                     # (functional.layer_norm(identity_from=t)): t's producer skips it in its backward

    @property
    def split(self):
        """The mark backward reaches first (0 = none recorded)."""
        return self.splits[-1] if self.splits else 0

    def watch(self, t):
        self.keep.append(t)
        self.watched[id(t)] = t
        return t

    def result(self, t):
        return self.results.get(id(t))

    def record(self, fn):
        self.ops.append(fn)

    def register_alias(self, view, base, slicer):
        self.keep.append(view)
        self.keep.append(base)
        self.alias[id(view)] = (base, slicer)

    def grad(self, t):
        a = self.alias.get(id(t))
        if a is not None:
            g = self.grad(a[0])
            return None if g is None else a[1](g)
        e = self.grads.get(id(t))
        return None if e is None else e[0]

    def pop_grad(self, t, with_count=False):
        """Gradient accumulated for t (None if none).  with_count=True -> (gradient, number of contributions it sums; 0 for
        views, whose gradient lives in a slice of their base)."""
        if id(t) in self.alias:
            g = self.grad(t)
            return (g, 0) if with_count else g
        e = self.grads.pop(id(t), None)
        if with_count:
            return (None, 0) if e is None else (e[0], e[2])
        return None if e is None else e[0]

    def peek_grad(self, t):
        """The gradient accumulated for t so far, left in place (None if none, or if t is a view)."""
        if id(t) in self.alias:
            return None
        e = self.grads.get(id(t))
        return None if e is None else e[0]

    def _own(self, t):
        """Make t's gradient buffer private to the tape (it may be accumulated into in place afterwards)."""
        from . import functional as Fn
        e = self.grads[id(t)]
        if not e[1]:
            priv = ctx().zeros(tuple(e[0].shape), e[0].dtype)
            Fn.add_into(priv, e[0])
            e[0], e[1] = priv, True
        return e[0]

    def grad_slot(self, t):
        """Where a producer may ACCUMULATE its gradient for t in place instead of handing over a fresh tensor: a view
        of the (zero-initialised or already populated) buffer when t is a view into a base tensor, or t's existing
        tape-owned gradient.  None when t has no gradient yet (the producer should then add_grad(..., owned=True)).
        The caller must add into the returned tensor; the contribution is counted here."""
        a = self.alias.get(id(t))
        if a is not None:
            root = a[0]
            while id(root) in self.alias:
                root = self.alias[id(root)][0]
            if id(root) not in self.grads:
                self.keep.append(root)
                self.grads[id(root)] = [ctx().zeros(tuple(root.shape)), True, 0]
            else:
                self._own(root)
            return self.grad(t)
        e = self.grads.get(id(t))
        if e is not None and e[1]:
            e[2] += 1
            return e[0]
        return None

    def unowned_grad(self, t):
        """The gradient accumulated for t so far when it is NOT tape-owned (shared with another target, so nobody may
        write into it) and t is no view: a producer whose kernel can compute `new = its gradient + addend` reads it as the
        addend and then calls replace_grad() -- the accumulate pass disappears.  None otherwise."""
        if id(t) in self.alias:
            return None
        e = self.grads.get(id(t))
        return e[0] if e is not None and not e[1] else None

    def replace_grad(self, t, g):
        """g (fresh, handed over) already contains t's previous gradient (see unowned_grad) plus one more contribution."""
        e = self.grads[id(t)]
        e[0], e[1] = g, True
        e[2] += 1

    def grad_count(self, t):
        e = self.grads.get(id(t))
        return 0 if e is None else e[2]

    def add_grad(self, t, g, owned=False):
        """Accumulate gradient g (same shape as t) into t's gradient.
        owned=True: g is a fresh buffer that the caller hands over (nobody else reads it afterwards), so the tape may
        accumulate into it in place.  A gradient passed without it is never modified, so one tensor may safely be
        passed for several targets.  A second contribution costs ONE launch: in place when either side is owned,
        else an out-of-place sum."""
        from . import functional as Fn
        a = self.alias.get(id(t))
        if a is not None:
            base, slicer = a
            root = base
            while id(root) in self.alias:
                root = self.alias[id(root)][0]
            if id(root) not in self.grads:
                self.keep.append(root)
                self.grads[id(root)] = [ctx().zeros(tuple(root.shape), g.dtype), True, 0]
            else:
                self._own(root)
            Fn.add_into(self.grad(t), g)
            return
        e = self.grads.get(id(t))
        if tuple(g.shape) != tuple(t.shape):
            g = g.reshape(t.shape)
        if e is None:
            self.keep.append(t)
            self.grads[id(t)] = [g, bool(owned), 1]
            return
        e[2] += 1
        if e[1]:
            Fn.add_into(e[0], g)
        elif owned:
            Fn.add_into(g, e[0])
            e[0], e[1] = g, True
        else:
            e[0], e[1] = Fn.add_maps(e[0], g), True

    def flush_wgrads(self):
        """Launch the deferred weight gradients (one emrt_conv2d_wgrad_group call: grouped launches of up to 24 layers)."""
        if self.wgrads:
            from . import functional as Fn
            pending, self.wgrads = self.wgrads, []
            Fn.launch_wgrads(pending)

    def flush_stash(self):
        if self.stash is not None:
            _gid, launch = self.stash
            self.stash = None
            launch(None)

    def backward(self, stop_at=0):
        """Run the recorded closures newest-first.  stop_at > 0 stops once only the first `stop_at` ops are left (the
        engine's early gradient exchange: everything recorded after `self.split` first, the rest in a second call)."""
        _CTX.tape = None            # backward kernels must not record
        _CTX._in_backward = True
        try:
            while len(self.ops) > stop_at:
                op = self.ops.pop()
                # a data gradient waiting for the host it rode with in forward (functional.conv_bn_many): only the very next closure, and only a
                # member of the same group, may take it into its launch; anything else sends it out on its own first
                if self.stash is not None and getattr(op, "_group", None) != self.stash[0]:
                    self.flush_stash()
                op()
            self.flush_stash()
            # the weight gradients this segment deferred: the caller (optimizer, or the gradient exchange of this segment's ranges) needs them now
            self.flush_wgrads()
            _CTX.wgrad_join()
        finally:
            _CTX._in_backward = False
        if stop_at > 0:
            return
        self.results = {k: self.grad(t) for k, t in self.watched.items()}
        self.ops = []
        self.grads = {}
        self.keep = []
        self.alias = {}
        self.identity_done = {}


class Context:
    def __init__(self):
        self.dtype = F32
        self.tape = None
        self.training = False
        self.device = None
        self._ws = None
        self._seed = None
        self.step_counter = None
        self.world_size = 1
        self.sync_bn = True
        self.salt_counter = 0
        self.keepalive = None     # list: while set, every tensor handed out by empty()/zeros() is kept alive (bench replay)
        self._arena = None        # fp64 zero arena for BatchNorm sums: one memset per step instead of ~150 tiny ones
        self._arena_off = 0
        self._arena_live = False
        self.overlap = False      # run independent backward kernels (wgrad next to dgrad) on a second HIP stream
        self._side = None
        self._in_backward = False # Tape.backward() is running (step-scoped allocations allowed)
        self._main = None         # the stream every launch of this runtime goes to (torch's current stream after init_device)
        self.fold_eval_bn = True  # inference: BatchNorm folded into the producing conv's epilogue (False: separate emrt_bn_apply, A/B knob)
        self.fold_live = False    # True inside an eval forward whose ParamStore.fold_bn() has just run (the folds are fresh)
        self.capture = None       # engine.GraphSequence while a step is being captured: collective() then breaks the graph
        self.sync_always = False  # issue the SyncBatchNorm collectives even in a 1-rank group (single-GPU test of the N > 1 path)
        # Weight gradients are not on backward's dependency chain: each conv / linear launches its data gradient alone and the weight
        # gradients of up to `wgrad_batch` layers go out as ONE grouped launch (emrt_conv2d_wgrad_group).  0 = every layer on its own.
        import os
        self.wgrad_batch = int(os.environ.get("EMRT_WGRAD_BATCH", "24"))
        # training: a BatchNorm + ReLU whose only consumer streams the map once (x2 resize, 3x3 max-pool) is applied by that consumer's
        # loads instead of its own emrt_bn_apply launch (functional.PendingBN).  0 = always the separate launch (A/B knob).
        self.bn_defer = bool(int(os.environ.get("EMRT_BN_DEFER", "1")))
        # ... and a BatchNorm + ReLU between two convolutions by the consuming convolution's operand loads (emrt_conv2d_bna: the 64x64-tile kernels transform
        # the raw map between its global load and the LDS write, the first tile column writes the normalised map backward needs).  0 = A/B knob.
        self.bn_conv = bool(int(os.environ.get("EMRT_BN_CONV", "1")))
        # the four pyramid-pooling branches (conv1x1 -> BatchNorm -> ReLU on 8 ... 512 pooled tokens) as grouped launches, 4 forward + 3 backward instead of
        # 8 + 12 (functional.conv_bn_small_group); 0 = one launch per branch and pass (A/B knob)
        self.bn_small_group = bool(int(os.environ.get("EMRT_BN_SMALL_GROUP", "1")))
        # conv1 and the shortcut conv of a bottleneck stage's first block (same input) as one grouped forward launch (functional.conv_bn_pair); 0 = A/B knob
        self.conv_pair = bool(int(os.environ.get("EMRT_CONV_PAIR", "1")))
        # the spatial branch's conv -> BatchNorm stages in the forward launches of the ResNet's layer3 / layer4 blocks (functional.SideJobs, one rank); 0 = A/B knob
        self.side_branch = bool(int(os.environ.get("EMRT_SIDE_BRANCH", "1")))
        self.side = None
        # backward of the forward pairings: the guest's data gradient in the host's launch (emrt_conv2d_dgrad_multi); 0 = A/B knob
        self.dgrad_pair = bool(int(os.environ.get("EMRT_DGRAD_PAIR", "1")))
        self.side_host_tiles = int(os.environ.get("EMRT_SIDE_HOST_TILES", "128"))      # largest host launch (64 x 64 tiles) that takes a guest
        self.segment_order = False      # True: an engine exchanges gradient segments while backward runs (engine.TrainEngine, N > 1): layers keep the reference's order
        # cls_psp's second conv -> BatchNorm -> ReLU and the auxiliary head's in one grouped launch per pass (EMRT.forward, one rank); 0 = A/B knob
        self.head_pair = bool(int(os.environ.get("EMRT_HEAD_PAIR", "1")))
        # an encoder layer's value_proj | offsets-logits projections in the forward grouped launch of its per-level 3x3 convolutions (Fn.level_conv_gn(linears=)); 0 = A/B knob
        self.enc_front = bool(int(os.environ.get("EMRT_ENC_FRONT", "1")))
        self.group_attn_proj = bool(int(os.environ.get("EMRT_GROUP_ATTN_PROJ", "1")))      # A/B: value_proj and the offsets | logits projection as one grouped launch
        self.fuse_ffn_dropout = bool(int(os.environ.get("EMRT_FFN_DROPOUT_FUSED", "1")))      # A/B: dropout(relu(linear1)) drawn in the GEMM epilogue (emrt_conv2d_drop)
        # the query of the NEXT attention (out + pos) written by the LayerNorm launch that produces `out`, its gradient summed by that LayerNorm's backward
        # as it loads (functional.layer_norm(q_pos=)); 0 = the separate add / accumulate launches (A/B knob)
        self.ln_query = bool(int(os.environ.get("EMRT_LN_QUERY", "1")))
        # the decoder's pyramid maps resized by ONE launch per direction (emrt_pyramid_resize_fwd / _bwd); 0 = one launch per scale (A/B knob)
        self.pyramid_group = bool(int(os.environ.get("EMRT_PYRAMID_GROUP", "1")))
        # EMRT_WGRAD_SIDE=1 (A/B experiment): the batched weight-gradient launches go to a second stream, next to the latency-bound
        # data-gradient / BatchNorm chain of the main stream; joined at the end of each backward segment
        self.wgrad_side = bool(int(os.environ.get("EMRT_WGRAD_SIDE", "0")))
        self._wside = None
        self._wside_keep = []
        # Step prologue off the critical path (engine.TrainEngine): zeroing the 216 MB gradient buffer and transposing the data-gradient weight copies
        # (emrt_pack_weights bwd_only: 211 MB) need nothing from the forward and nothing in the forward needs them -- ~70 us of pure HBM streaming that
        # round 5 ran in front of the forward / in front of the backward.  They go to a second stream forked at the top of the step and joined before the
        # first backward launch (or the first collective that cuts a captured step): ONE fork and ONE join per step, beside a forward whose small
        # launches leave the memory system mostly idle.  MEASURED (round 6, same box, captured step): 909.8 / 909.9 tiles/s with it against 949.2 / 950.7
        # without -- one pair of cross-stream edges inside the hipGraph costs 0.36 ms per replay, five times what it hides (the same finding as round 3's
        # two-stream weight gradients).  Default OFF; EMRT_PROLOGUE_SIDE=1 for the A/B.
        self.prologue_side = bool(int(os.environ.get("EMRT_PROLOGUE_SIDE", "0")))
        self._pside = None
        self._prologue_pending = False
        self.pack_bwd_done = False      # set by the engine when this step's transposed weight copies are already (being) made on the side stream

    def collective(self, fn):
        """Run a host-issued collective (fn enqueues it on the current stream's timeline) at this point of the step.  Eager:
        just call it.  While the step is being captured into hipGraphs: end the current graph, call it -- now and on every
        replay -- between the graphs, and continue capturing in a new graph (engine.GraphSequence.interlude).  The transports
        differ in what they can capture (gloo stages through the host; RCCL kernels can be captured but then cost graph-side
        cross-stream edges), an eager call between two graph launches is right for all of them."""
        self.prologue_join()            # (a forked stream must be joined inside the graph it was forked in)
        if self.capture is not None:
            self.capture.interlude(fn)
        else:
            fn()

    # ---- second stream -------------------------------------------------------------------------------------------
    # overlap = False | "pair" | "deferred".  fork() returns the side stream (C handle) ordered after everything issued so
    # far on the current stream.  "pair": the caller join()s right after the overlapping main-stream kernel.  "deferred":
    # nothing is joined until join_all() (before the optimizer); the buffers the side-stream kernels read are parked in
    # _side_keep so that the allocator cannot hand them out again meanwhile.
    def prologue_stream(self):
        """Fork: the side stream of the step prologue, ordered after everything issued so far on the current stream (use as `with torch.cuda.stream(s)`)."""
        if self._pside is None:
            self._pside = torch.cuda.Stream(device=self.device)
        self._pside.wait_stream(torch.cuda.current_stream())
        self._prologue_pending = True
        return self._pside

    def prologue_join(self):
        """The current stream waits for the step prologue (idempotent; called before backward and before any collective that cuts a captured step)."""
        if self._prologue_pending:
            torch.cuda.current_stream().wait_stream(self._pside)
            self._prologue_pending = False

    def fork(self, *keep):
        if not self.overlap:
            return None
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
            self._side_keep = []
        self._side.wait_stream(torch.cuda.current_stream())
        if self.overlap == "deferred":
            self._side_keep.extend(keep)
        return ctypes.c_void_p(self._side.cuda_stream)

    def join(self):
        if self.overlap == "pair":
            torch.cuda.current_stream().wait_stream(self._side)

    def join_all(self):
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)
            self._side_keep = []
        self.wgrad_join()

    def wgrad_fork(self, keep):
        """Side stream for a batch of weight gradients, ordered after everything issued so far on the current stream; `keep`: the tensors its
        kernels read (they must not go back to the allocator before the join)."""
        if self._wside is None:
            self._wside = torch.cuda.Stream(device=self.device)
        self._wside.wait_stream(torch.cuda.current_stream())
        self._wside_keep.extend(keep)
        return ctypes.c_void_p(self._wside.cuda_stream)

    def wgrad_join(self):
        if self._wside is not None and self._wside_keep:
            torch.cuda.current_stream().wait_stream(self._wside)
            self._wside_keep = []

    # ---- device / dtype -------------------------------------------------------------------------
    def init_device(self, device="cuda:0", dtype=F32, seed=1234):
        if not torch.cuda.is_available():
            raise _lib.EmrtHipError("the EMRT HIP path needs a GPU (torch.cuda.is_available() is False); "
                                    "there is no CPU fallback")
        _lib.lib()
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        # One explicitly created stream carries all work and becomes torch's current stream.  On the NULL stream,
        # back-to-back hipGraph launches (graph A -> graph B -> graph A ... of the multi-GPU step) were measured to lose
        # their ordering whenever the host was not running ahead of the GPU (DESIGN.md "Multi-GPU"); a created stream
        # orders them as it should.
        if self._main is None or self._main.device != self.device:
            self._main = torch.cuda.Stream(device=self.device)
        self._main.wait_stream(torch.cuda.default_stream(self.device))
        torch.cuda.set_stream(self._main)
        self.dtype = dtype
        self._ws = torch.empty(8 << 20, dtype=torch.uint8, device=self.device)
        # (the partial-sum scratch of the 256 x 256 weight-gradient kernel is allocated by the first TRAINING step in bf16: begin_step)
        if getattr(self, "_scratch", None) is not None and self._scratch.device != self.device:
            # the library keeps the registered pointer: withdraw it before the tensor (and its memory) goes away
            _lib.lib().call("emrt_set_scratch", ctypes.c_void_p(0), ctypes.c_size_t(0), ctypes.c_void_p(0))
            self._scratch = None
            self._scratch_stream = None
        self._seed = torch.tensor([seed * 0x9E3779B97F4A7C15 % (1 << 63)], dtype=torch.int64, device=self.device)
        self.step_counter = torch.zeros(1, dtype=torch.int64, device=self.device)
        self._arena, self._arena_off, self._arena_live = None, 0, False

    @property
    def tdtype(self):
        return _TORCH_DTYPE[self.dtype]

    @property
    def stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def workspace(self, nbytes):
        if self._ws.numel() < nbytes:
            self._ws = torch.empty(int(nbytes * 1.5), dtype=torch.uint8, device=self.device)
        return self._ws

    @property
    def seed_ptr(self):
        return ctypes.c_void_p(self._seed.data_ptr())

    def empty(self, shape, dtype=None):
        t = torch.empty(shape, dtype=dtype or self.tdtype, device=self.device)
        if self.keepalive is not None:
            self.keepalive.append(t)
        return t

    def zeros(self, shape, dtype=None):
        dt = dtype or self.tdtype
        n = 1
        for d in shape:
            n *= int(d)
        nbytes = n * torch.empty((), dtype=dt).element_size()
        # (only inside a recorded forward or a backward: nothing allocated there outlives the step, so the next step's
        # clear cannot pull the rug from under a persistent buffer)
        if self._arena_live and (self.tape is not None or self._in_backward) and self.keepalive is None and 0 < nbytes <= (1 << 20):
            # small zero buffers come out of the arena that is cleared ONCE per step (a memset launch costs ~5 us in the
            # graph whatever its size, and a step asks for a dozen of these)
            n8 = (nbytes + 15) // 16 * 2
            if self._arena_off + n8 <= self._arena.numel():
                raw = self._arena[self._arena_off:self._arena_off + n8]
                self._arena_off += n8
                return raw.view(dt)[:n].view(tuple(shape))
        t = self.empty(shape, dtype)
        _lib.lib().call("emrt_memset", ctypes.c_void_p(t.data_ptr()), 0, t.numel() * t.element_size(), self.stream)
        return t

    def begin_step(self, arena_doubles=1 << 21):
        """Start of a training step: re-zero the fp64 arena that zeros_f64() hands out slices of."""
        if self._arena is None or self._arena.numel() < arena_doubles:
            self._arena = torch.empty(arena_doubles, dtype=torch.float64, device=self.device)
        _lib.lib().call("emrt_memset", ctypes.c_void_p(self._arena.data_ptr()), 0, self._arena.numel() * 8, self.stream)
        self._arena_off = 0
        self._arena_live = True
        self.ensure_scratch()

    def ensure_scratch(self):
        """Partial-sum scratch of the 256 x 256 weight-gradient kernel (one 256 KiB fp32 tile per CU = 64 MiB) and of the convolutions'
        cross-block K split, registered for THIS context's stream by the first bf16 / fp16 forward -- fp32 contexts (the parity mode) never
        pay for it and never take those kernels.  Its address is baked into captured graphs: allocated once, outside any capture (the eager
        warm-up steps come first; TrainEngine._capture and SlidingWindowEngine make sure of it), never reallocated."""
        if self.dtype == F32:
            return
        cur = torch.cuda.current_stream().cuda_stream
        if getattr(self, "_scratch", None) is not None:
            # the region belongs to ONE stream at a time (partial tiles and their reader are only ordered within a stream).  torch captures a
            # hipGraph on a side stream of its own, so a step being captured would find the region registered for another stream and silently
            # bake the fallback kernels into the graph (fp32 atomics instead of the slab + reduce pair, no cross-block K split: round 4's captured
            # step ran that way); the region follows the stream that launches: a host-side assignment, no device work, legal while capturing.
            if getattr(self, "_scratch_stream", None) != cur and not self.wgrad_side:
                _lib.lib().call("emrt_set_scratch", ctypes.c_void_p(self._scratch.data_ptr()), ctypes.c_size_t(self._scratch.numel()), ctypes.c_void_p(cur))
                self._scratch_stream = cur
            return
        if torch.cuda.is_current_stream_capturing():
            return
        # + 8 MiB + 64 KiB: the partial tiles and the arrival counters of the convolutions' cross-block K split (csrc/conv.hip, igemm_body XK),
        # carved off the tail by emrt_set_scratch (counters zeroed there); the K split does NOT share addresses with the weight-gradient slab
        self._scratch = torch.empty((64 << 20) + (8 << 20) + 65536, dtype=torch.uint8, device=self.device)
        st = self.stream
        if self.wgrad_side:          # (every weight gradient, the large layers' included, is then launched from the side stream)
            if self._wside is None:
                self._wside = torch.cuda.Stream(device=self.device)
            st = ctypes.c_void_p(self._wside.cuda_stream)
        _lib.lib().call("emrt_set_scratch", ctypes.c_void_p(self._scratch.data_ptr()), ctypes.c_size_t(self._scratch.numel()), st)
        self._scratch_stream = st.value if hasattr(st, "value") else cur

    def end_step(self):
        """Outside a training step (eval forward) the arena is not re-zeroed: zeros_f64() must hand out fresh zeroed buffers."""
        self._arena_live = False
        self.ensure_scratch()          # (fp16 / bf16 inference: the few-tile layers' cross-block K split needs its partial-tile scratch too)

    def zeros_f64(self, n):
        """Zeroed fp64 [n] buffer (BatchNorm sums).  Inside a step it is a slice of the pre-zeroed arena."""
        if self._arena_live and self._arena_off + n <= self._arena.numel():
            t = self._arena[self._arena_off:self._arena_off + n]
            self._arena_off += (n + 1) // 2 * 2
            return t
        return self.zeros((n,), torch.float64)

    def zeros_like(self, t):
        """Zero buffer with the same logical shape as t (dense)."""
        return self.zeros(tuple(t.shape), t.dtype)

    def next_salt(self):
        self.salt_counter += 1
        return self.salt_counter


_CTX = Context()


def ctx():
    return _CTX


class recording:
    """with recording() as tape: ... forward ...; tape.backward()"""

    def __enter__(self):
        self.prev = _CTX.tape
        _CTX.tape = Tape()
        return _CTX.tape

    def __exit__(self, *exc):
        _CTX.tape = self.prev
        return False
