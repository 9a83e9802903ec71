"""One training step of the reference loop (train.py:141-159) as a replayable unit.

    clear grads -> forward -> CE + 0.4 aux CE -> backward -> [RCCL gradient all-reduce] -> clip + SGD-momentum + re-pack

Every kernel of the step is launched on torch's current stream, so after a few eager warm-up steps the whole step is
captured into a hipGraph (torch.cuda.CUDAGraph) and replayed: ~1.5k launches cost one graph launch on the host.
With world_size > 1 the step is captured as two graphs around the gradient all-reduce (graph A: zero/fwd/loss/bwd;
eager: bucketed all-reduce(AVG) of the flat gradient buffer over RCCL; graph B: optimizer), and the five SyncBatchNorm
layers use per-rank statistics inside the captured region (documented deviation, DESIGN.md "Multi-GPU"); in eager mode
(use_graph=False) they all-reduce their statistics as the reference's nn.SyncBatchNorm does.
"""
import torch

from . import _lib
from . import functional as Fn
from .runtime import ctx
from .distributed import FlatGradReducer

_SEED_STRIDE = 0x2545F4914F6CDD1D


class TrainEngine:
    def __init__(self, model, optimizer, loss_fn, world_size=1, use_graph=True, warmup_eager=2, bucket_elems=32 * 1024 * 1024,
                 overlap=False, two_phase=None):
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
        c.overlap = overlap            # False | "pair" | "deferred": wgrad on a second stream (runtime.Context.fork)
        # two_phase=True with world_size == 1 runs the N > 1 structure (graph A / RCCL all-reduce / graph B, per-rank BN
        # statistics inside the capture) in a 1-rank process group: the single-GPU test of the multi-GPU path
        self.two_phase = (world_size > 1) if two_phase is None else bool(two_phase)
        self.reducer = FlatGradReducer(model.store.grad, model.store.n_train, world_size, bucket_elems,
                                       always=self.two_phase) if self.two_phase else None

    # -- pieces ------------------------------------------------------------------------------------
    def _fwd_bwd(self, images, labels):
        c = ctx()
        _lib.lib().call("emrt_counter_add", Fn.P(c._seed), _SEED_STRIDE & 0x7FFFFFFFFFFFFFFF, c.stream)   # fresh dropout masks
        self.model.clear_gradients()
        out = self.model(images)
        loss = self.loss_fn(out, labels)
        loss.backward()
        ctx().join_all()               # side-stream weight gradients (if any) land before the reducer / optimizer
        return loss.tensor

    def _eager_step(self, images, labels):
        ctx().sync_bn = True
        loss_t = self._fwd_bwd(images, labels)
        if self.reducer is not None:
            self.reducer.allreduce()
        self.opt.step()
        return loss_t

    def _capture(self, images, labels):
        c = ctx()
        c.sync_bn = not self.two_phase
        self.images = images.clone()
        self.labels = labels.clone()
        c.workspace(64 << 20)
        torch.cuda.synchronize()
        self.graph_a = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_a):
            self.loss_t = self._fwd_bwd(self.images, self.labels)
            if self.reducer is None:
                self.opt.step()
        if self.reducer is not None:
            self.graph_b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_b, pool=self.graph_a.pool()):
                self.opt.step()

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
                self.images.copy_(images, non_blocking=True)
                self.labels.copy_(labels, non_blocking=True)
            self.graph_a.replay()
            if self.reducer is not None:
                self.reducer.allreduce()
                self.graph_b.replay()
            loss_t = self.loss_t
        self.opt._learning_rate.step()       # host mirror of the device step counter (train.py:156-158)
        return loss_t
