"""One training step of the reference loop (train.py:141-159) as a replayable unit.

    clear grads -> forward -> CE + 0.4 aux CE -> backward -> [RCCL gradient all-reduce] -> clip + SGD-momentum + re-pack

Every kernel of the step is launched on torch's current stream, so after a few eager warm-up steps the whole step is
captured into a hipGraph (torch.cuda.CUDAGraph) and replayed: ~1.5k launches cost one graph launch on the host.
With world_size > 1 the step is captured as four graphs around the gradient exchange: graph A1 (zero/fwd/loss/backward
down to the ResNet layer4 input) -> launch the bucketed RCCL all-reduce(AVG) of the flat-gradient ranges that are final
by then (heads, transformer, layer4) -> graph A2 (backward of layer3) -> all-reduce of layer3's range -> graph A3
(backward of layer2..conv1; both collectives overlap it) -> all-reduce of the rest (3 % of the elements: the only exposed
one) -> graph B (clip + optimizer); early_exchange=False keeps one graph A and one exchange.  The five SyncBatchNorm
layers (paddle_EMRT.py:64, fcn_head.py:53) all-reduce their statistics in every mode, as the reference's nn.SyncBatchNorm
does: inside a captured stretch the graph is cut at each of those collectives (GraphSequence) and the all-reduce is issued
eagerly between the two pieces, forward and backward.
"""
import ctypes

import torch

from . import _lib
from . import functional as Fn
from .runtime import ctx
from .distributed import FlatGradReducer

_SEED_STRIDE = 0x2545F4914F6CDD1D


class GraphSequence:
    """A stretch of the training step captured as hipGraphs with eager interludes between them.

    capture(fn) runs fn under stream capture; whenever fn reaches a host-issued collective (runtime.Context.collective: the
    SyncBatchNorm statistics all-reduces) the current graph is ended, the collective is called eagerly -- at capture time and
    again at every replay -- and capture resumes in a new graph that shares the first one's memory pool.  replay() launches
    graphs and interludes in the captured order.  Without interludes this is exactly one torch.cuda.CUDAGraph."""

    def __init__(self, pool=None, mode=None):
        self.items = []            # torch.cuda.CUDAGraph | callable
        self._pool = pool
        self.mode = dict(mode or {})
        self._cm = self._g = None

    def pool(self):
        return self._pool

    @property
    def n_graphs(self):
        return sum(isinstance(it, torch.cuda.CUDAGraph) for it in self.items)

    def _begin(self):
        self._g = torch.cuda.CUDAGraph()
        kw = dict(self.mode)
        if self._pool is not None:
            kw["pool"] = self._pool
        self._cm = torch.cuda.graph(self._g, **kw)
        self._cm.__enter__()

    def _end(self):
        cm, self._cm = self._cm, None
        cm.__exit__(None, None, None)
        self.items.append(self._g)
        if self._pool is None:
            self._pool = self._g.pool()

    def capture(self, fn):
        c = ctx()
        assert c.capture is None, "nested step capture"
        c.capture = self
        self._begin()
        try:
            out = fn()
        finally:
            c.capture = None
            if self._cm is not None:
                self._end()
        return out

    def interlude(self, fn):
        self._end()
        fn()
        self.items.append(fn)
        self._begin()

    def replay(self):
        for it in self.items:
            if isinstance(it, torch.cuda.CUDAGraph):
                it.replay()
            else:
                it()


class TrainEngine:
    def __init__(self, model, optimizer, loss_fn, world_size=1, use_graph=True, warmup_eager=2, bucket_elems=32 * 1024 * 1024,
                 overlap=False, two_phase=None, early_exchange=True, exchange_dtype=None):
        self.model, self.opt, self.loss_fn = model, optimizer, loss_fn
        self.world = world_size
        self.use_graph = use_graph
        self.warmup_eager = warmup_eager
        self.calls = 0
        self.graph_a = self.graph_b = None
        self.images = self.labels = None
        self.loss_t = None
        c = ctx()
        c.world_size = world_size
        c.sync_bn = True
        c.overlap = overlap            # False | "pair" | "deferred": wgrad on a second stream (runtime.Context.fork)
        # two_phase=True with world_size == 1 runs the N > 1 structure (graph A / RCCL all-reduce / graph B, per-rank BN
        # statistics inside the capture) in a 1-rank process group: the single-GPU test of the multi-GPU path
        self.two_phase = (world_size > 1) if two_phase is None else bool(two_phase)
        c.sync_always = self.two_phase and world_size == 1      # 1-rank group: still issue the SyncBatchNorm collectives
        import os
        # gradient exchange precision: fp32 as the reference (paddle.DataParallel), or bf16 = half the bytes over xGMI (EMRT_GRAD_EXCHANGE=bf16)
        exchange_dtype = exchange_dtype or os.environ.get("EMRT_GRAD_EXCHANGE", "fp32")
        self.reducer = FlatGradReducer(model.store.grad, model.store.n_train, world_size, bucket_elems,
                                       always=self.two_phase, exchange_dtype=exchange_dtype) if self.two_phase else None
        # early exchange: backward runs in segments between the model's marks (ResNet.forward: before layer3 and before
        # layer4); the gradients a segment completes are all-reduced on RCCL's stream while the next segments still run:
        #   segment 1 (heads, transformer, layer4: ~84 % of the elements) -> exchanged under layer3 .. conv1's backward
        #   segment 2 (layer3: 13 %)                                      -> exchanged under layer2 .. conv1's backward
        #   segment 3 (conv1, layer1, layer2: 3 %)                        -> the only exposed collective
        segs = getattr(model, "grad_segment_prefixes", None)
        self.seg_ranges = None
        self.early_ranges = self.late_ranges = None
        if self.two_phase and early_exchange and segs:
            self.seg_ranges = model.store.segment_ranges(segs)
            self.early_ranges = self.seg_ranges[0]
            self.late_ranges = [r for seg in self.seg_ranges[1:] for r in seg]
        c.segment_order = self.seg_ranges is not None      # (EMRT.forward: no spatial-branch stages inside layer3 / layer4's launches then)
        self.graph_rest = []         # hipGraphs of backward segments 2.. (graph_a holds forward + segment 1)

    @property
    def graph_a2(self):
        return self.graph_rest[0] if self.graph_rest else None

    # -- pieces ------------------------------------------------------------------------------------
    def _fwd_bwd(self, images, labels, split=False):
        """split=False: the whole forward + backward, returns the loss tensor.  split=True: runs backward's first segment
        only and returns (loss tensor, [callable per remaining segment])."""
        c = ctx()
        _lib.lib().call("emrt_counter_add", Fn.P(c._seed), _SEED_STRIDE & 0x7FFFFFFFFFFFFFFF, c.stream)   # fresh dropout masks
        side = c.prologue_side and not c.overlap and not c.wgrad_side and getattr(self.model.store, "desc", None) is not None
        if side and self.model.store.dirty:      # master weights edited since the last step: the full refresh (forward mirror first) on this stream, now
            self.model.store.pack()
        if side:
            # gradient zeroing + the transposed data-gradient weight copies on the prologue stream, beside the forward (runtime.Context.prologue_stream)
            with torch.cuda.stream(c.prologue_stream()):
                self.model.clear_gradients()
                self.model.store.pack(bwd_only=True)
            c.pack_bwd_done = True
        else:
            self.model.clear_gradients()
        seg_prev = c.segment_order
        if split:      # backward in segments whose gradients must be final at the marks: the layers keep the reference's order (EMRT.forward: no side chain)
            c.segment_order = True
        try:
            out = self.model(images)
        finally:
            c.pack_bwd_done = False
            c.segment_order = seg_prev
        loss = self.loss_fn(out, labels)
        c.prologue_join()               # backward reads the zeroed gradients and the transposed copies
        if split:
            rest = loss.backward_until_split(segments=True)
            if rest:
                last = rest[-1]

                def finish_last():
                    last()
                    ctx().join_all()
                rest[-1] = finish_last
            return loss.tensor, rest
        loss.backward()
        ctx().join_all()               # side-stream weight gradients (if any) land before the reducer / optimizer
        return loss.tensor

    def _exchange(self, segments):
        """Gradient all-reduce around the remaining backward segments (callables or graph replays)."""
        if self.seg_ranges is None:
            self.reducer.allreduce()
            return
        self.reducer.launch(self.seg_ranges[0])      # RCCL stream picks up after everything enqueued so far
        done = 1
        for seg in segments:
            seg()
            if done < len(self.seg_ranges):
                self.reducer.launch(self.seg_ranges[done])
            done += 1
        for ranges in self.seg_ranges[done:]:        # (a model that recorded fewer marks than it declared segments)
            self.reducer.launch(ranges)
        self.reducer.wait()

    def _eager_step(self, images, labels):
        if self.seg_ranges is not None:
            loss_t, rest = self._fwd_bwd(images, labels, split=True)
            self._exchange(rest)
        else:
            loss_t = self._fwd_bwd(images, labels)
            if self.reducer is not None:
                self.reducer.allreduce()
        self.opt.step()
        return loss_t

    def _capture(self, images, labels):
        c = ctx()
        self.images = images.clone()
        self.labels = labels.clone()
        c.workspace(64 << 20)
        # the weight-gradient scratch is registered by the first bf16 training step, but never from inside a capture: when the captured
        # step IS the first one (warmup_eager=0, a re-capture after init_device) it has to exist before the capture begins, or the graph
        # is baked with the slab path of the 256 x 256 weight-gradient kernel switched off for good
        c.ensure_scratch()
        torch.cuda.synchronize()
        # other threads make HIP calls while this one captures: ProcessGroupNCCL's watchdog polls events, the TileLoader's reader threads
        # page-lock their batches (train.py --data dataset).  The default "global" capture mode turns any such foreign-thread call into a
        # capture error (seen as a watchdog abort, and as hipErrorStreamCaptureInvalidated at the first captured step of a real-data run)
        mode = {"capture_error_mode": "thread_local"}
        # every stretch is a GraphSequence: one hipGraph, or several around the SyncBatchNorm statistics all-reduces
        self.graph_a = GraphSequence(mode=mode)
        if self.seg_ranges is not None:
            self.loss_t, rest = self.graph_a.capture(lambda: self._fwd_bwd(self.images, self.labels, split=True))
            self.graph_rest = []
            for seg in rest:
                g = GraphSequence(pool=self.graph_a.pool(), mode=mode)
                g.capture(seg)
                self.graph_rest.append(g)
        else:
            def whole():
                loss_t = self._fwd_bwd(self.images, self.labels)
                if self.reducer is None:
                    self.opt.step()
                return loss_t
            self.loss_t = self.graph_a.capture(whole)
        if self.reducer is not None:
            self.graph_b = GraphSequence(pool=self.graph_a.pool(), mode=mode)
            self.graph_b.capture(self.opt.step)

    @staticmethod
    def _stage(dst, src):
        """Next batch into the buffer the captured graph reads: a device-to-device copy through the C-ABI on the step's stream."""
        src = src.contiguous()
        assert src.dtype == dst.dtype and src.numel() == dst.numel() and src.is_cuda, (src.dtype, dst.dtype, src.shape, dst.shape)
        _lib.lib().call("emrt_memcpy", ctypes.c_void_p(dst.data_ptr()), ctypes.c_void_p(src.data_ptr()), dst.numel() * dst.element_size(), ctx().stream)

    # -- public ------------------------------------------------------------------------------------
    def step(self, images, labels):
        """images fp32 [B,3,H,W], labels int64 [B,H,W] on the device.  Returns the device loss tensor (float[1])."""
        self.model.train()
        self.calls += 1
        if not self.use_graph or self.calls <= self.warmup_eager:
            loss_t = self._eager_step(images, labels)
        else:
            if self.graph_a is None:
                self._capture(images, labels)
            if images.data_ptr() != self.images.data_ptr():
                self._stage(self.images, images)
                self._stage(self.labels, labels)
            if self.model.store.dirty:     # master weights edited since the last step (load_state_dict after the capture): refresh the
                self.model.store.pack()    # compute-dtype mirror eagerly -- inside the graph only the optimizer's own update writes it
            self.graph_a.replay()
            if self.reducer is not None:
                self._exchange([g.replay for g in self.graph_rest])
                self.graph_b.replay()
            loss_t = self.loss_t
        self.opt._learning_rate.step()       # host mirror of the device step counter (train.py:156-158)
        return loss_t
