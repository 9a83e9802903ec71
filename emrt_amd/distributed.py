"""Data-parallel plumbing: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm) / gloo on CPU.

Reference: paddle.DataParallel + DistributedBatchSampler + nn.SyncBatchNorm (train.py:116-123, dataloader.py:38-41).
Because every gradient lives in ONE flat fp32 buffer (ParamStore.grad[:n_train]), the gradient exchange is a handful of
large all-reduces over contiguous slices -- sized for xGMI's point-to-point links -- instead of per-tensor buckets.
The two parameters that never receive a gradient (backbone.fc.*, model.tgt_embed) sit outside [0, n_train).
"""
import os

import torch
import torch.distributed as dist


def env_rank_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init_process_group(backend=None):
    rank, local_rank, world = env_rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # EMRT_DIST_BACKEND=gloo: test aid -- several ranks can then share ONE GPU (RCCL refuses duplicate devices), which is
        # how the multi-rank step is exercised on a single-GPU box (tests/test_gpu_dp2.py)
        backend = backend or os.environ.get("EMRT_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def bucket_slices(n, bucket_elems, start=0):
    """Contiguous [start, end) slices covering [start, n)."""
    out, s = [], start
    while s < n:
        e = min(n, s + bucket_elems)
        out.append((s, e))
        s = e
    return out


class FlatGradReducer:
    """Averages the flat gradient buffer over ranks.  `launch()` enqueues async all-reduces (one per bucket) and
    `wait()` blocks the compute stream on them; with bucket_elems >= n it is a single collective."""

    def __init__(self, flat_grad, n, world_size, bucket_elems=32 * 1024 * 1024, always=False):
        self.flat, self.n, self.world = flat_grad, n, world_size
        self.bucket = bucket_elems
        self.slices = bucket_slices(n, bucket_elems)
        self.handles = []
        self.always = always        # issue the collectives even in a 1-rank group (single-GPU test of the N > 1 path)

    def launch(self, ranges=None):
        """ranges=None: the whole buffer.  Otherwise a list of [start, end) element ranges (each bucketed); handles
        accumulate until wait(), so an early launch(early_ranges) can overlap the rest of backward."""
        if self.world <= 1 and not self.always:
            return
        use_avg = dist.get_backend() == "nccl"
        slices = self.slices if ranges is None else [b for a, e in ranges for b in bucket_slices(e, self.bucket, a)]
        for s, e in slices:
            t = self.flat[s:e]
            if use_avg:
                self.handles.append((dist.all_reduce(t, op=dist.ReduceOp.AVG, async_op=True), None))
            else:   # gloo has no AVG: sum then scale (CPU tests, single-GPU multi-rank test)
                self.handles.append((dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True), t))

    def wait(self):
        for h, t in self.handles:
            h.wait()
            if t is not None:
                t.div_(self.world)
        self.handles = []

    def allreduce(self):
        self.launch()
        self.wait()


class DistributedTileSampler:
    """Rank-strided index sharding with per-epoch reshuffle (reference: paddle.io.DistributedBatchSampler,
    dataloader.py:38-41): pads to a multiple of world*batch, rank r takes indices r, r+world, ..."""

    def __init__(self, n, batch_size, rank, world, shuffle=True, drop_last=True, seed=0):
        self.n, self.bs, self.rank, self.world, self.shuffle, self.drop_last, self.seed = n, batch_size, rank, world, shuffle, drop_last, seed
        self.epoch = 0

    def set_epoch(self, e):
        self.epoch = e

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed + self.epoch)
        idx = torch.randperm(self.n, generator=g).tolist() if self.shuffle else list(range(self.n))
        per = -(-self.n // self.world)
        idx += idx[: per * self.world - self.n]
        mine = idx[self.rank::self.world]
        for i in range(0, len(mine), self.bs):
            b = mine[i:i + self.bs]
            if len(b) == self.bs or not self.drop_last:
                yield b

    def __len__(self):
        per = -(-self.n // self.world)
        return per // self.bs if self.drop_last else -(-per // self.bs)
