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


def visible_gpu_count():
    """GPUs this process may use, counted WITHOUT a HIP call (torch.cuda.device_count() may initialise the runtime on ROCm builds that
    lack amdsmi): the kernel driver's topology nodes with SIMDs (CPUs have simd_count 0), cut down by HIP_/ROCR_/CUDA_VISIBLE_DEVICES."""
    n = 0
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:
        for node in os.listdir(root):
            try:
                with open(os.path.join(root, node, "properties")) as f:
                    props = dict(ln.split()[:2] for ln in f if len(ln.split()) >= 2)
                if int(props.get("simd_count", "0")) > 0:
                    n += 1
            except (OSError, ValueError):
                continue
    except OSError:
        n = 0
    if n == 0:      # no readable topology (containers without /sys/class/kfd): torch's count (on this image it does not initialise HIP either)
        return torch.cuda.device_count()
    # containers may list every node of the host but expose fewer render devices: a GPU without an accessible /dev/dri/renderD* node cannot be
    # opened, so it is not counted (no render nodes visible at all = nothing to cross-check against)
    try:
        rd = [d for d in os.listdir("/dev/dri") if d.startswith("renderD")]
        usable = sum(os.access(os.path.join("/dev/dri", d), os.R_OK | os.W_OK) for d in rd)
        if rd and usable:
            n = min(n, usable)
    except OSError:
        pass
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            listed = len([x for x in v.split(",") if x.strip() != ""])
            n = min(n, listed) if n else listed
    return n


def spawn_ranks(n, cmd, capture_rank0=False, log=None):
    """Launcher-less multi-GPU start: run `cmd` (an argv list) as n child processes, one rank each, with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set -- what paddle.distributed.launch gives the reference
    (train.py:116-123 reads the ranks from it).  The ranks are always FRESH interpreters (subprocess.Popen): this process is never
    forked with a live GPU runtime nor replaced by exec, and it makes no GPU call itself (the device count below comes from the
    kernel driver's sysfs nodes, not from HIP).  A rank that dies takes the job down at once (the others would otherwise wait in
    the rendezvous or a collective until a timeout), and so does anything that ends this function early (KeyboardInterrupt, SIGTERM,
    an exception in the log / reader path): no rank outlives its parent holding a GPU.
    Returns (exit codes, rank 0's stdout as str or None)."""
    import socket
    import subprocess
    import sys
    import threading
    import time
    log = log or (lambda *a: print(*a, file=sys.stderr, flush=True))
    if not os.environ.get("EMRT_ALL_RANKS_ON_GPU0"):
        have = visible_gpu_count()
        if have < n:
            log("[launch] %d ranks requested but this node exposes %d GPU(s)" % (n, have))
            return [2] * n, None
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    chunks = []
    reader = None

    def kill_all():
        for p in procs:
            if p.poll() is None:
                p.kill()                                      # exactly the processes started here
        for p in procs:
            p.wait()

    def on_term(signum, frame):
        raise KeyboardInterrupt("signal %d" % signum)
    import signal
    old_term, hooked = None, False
    if threading.current_thread() is threading.main_thread():
        old_term = signal.signal(signal.SIGTERM, on_term)     # a plain SIGTERM would end this process without running `finally`
        hooked = True
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
            out = (subprocess.PIPE if capture_rank0 else None) if r == 0 else subprocess.DEVNULL
            procs.append(subprocess.Popen(list(cmd), env=env, stdout=out))
        if capture_rank0:
            reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
            reader.start()
        while any(p.poll() is None for p in procs):
            if any(p.poll() not in (None, 0) for p in procs):
                break
            time.sleep(0.2)
    finally:
        kill_all()                                            # normal end: everything has exited already; early end: nothing survives
        if hooked:      # (signal.signal returns None for a disposition that was not installed from Python: that restores to the default)
            signal.signal(signal.SIGTERM, old_term if old_term is not None else signal.SIG_DFL)
    codes = [p.returncode for p in procs]
    if reader is not None:
        reader.join(timeout=10)
    if any(codes):
        log("[launch] rank exit codes %s" % codes)
    return codes, (b"".join(chunks).decode() if capture_rank0 else None)


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
    `wait()` blocks the compute stream on them; with bucket_elems >= n it is a single collective.

    exchange_dtype="bf16": the slices travel as bf16 (108 MB instead of 216 MB per step over xGMI: BASELINE.md section 2) -- each
    slice is cast into a bf16 staging buffer by emrt_cast on the training stream, all-reduced (AVG) there, and cast back into the
    fp32 gradient buffer in wait().  The fp32 master weights, momentum and the clip still see fp32 gradients; what is rounded is each
    rank's contribution (2^-9 relative) and the ring's partial sums.  Off by default: the reference exchanges fp32
    (paddle.DataParallel, train.py:116-123); tests/test_distributed_cpu.py and tests/test_gpu_dp2.py bound the difference."""

    def __init__(self, flat_grad, n, world_size, bucket_elems=32 * 1024 * 1024, always=False, exchange_dtype="fp32"):
        self.flat, self.n, self.world = flat_grad, n, world_size
        self.bucket = bucket_elems
        self.slices = bucket_slices(n, bucket_elems)
        self.handles = []
        self.always = always        # issue the collectives even in a 1-rank group (single-GPU test of the N > 1 path)
        self.noop = False           # never set by the product: launch() then returns without a collective (what the exchange costs = with - without)
        assert exchange_dtype in ("fp32", "bf16")
        self.exchange_dtype = exchange_dtype
        self.half = torch.empty(n, dtype=torch.bfloat16, device=flat_grad.device) if exchange_dtype == "bf16" else None

    def _cast(self, src, dst, to_bf16):
        """fp32 <-> bf16 on the current stream: the HIP cast kernel on a GPU (no torch arithmetic on the product path), torch on CPU tests."""
        if src.is_cuda:
            from . import _lib
            from .runtime import BF16, ctx
            _lib.lib().call("emrt_cast", src.data_ptr(), dst.data_ptr(), src.numel(), 0 if to_bf16 else 1, BF16, ctx().stream)
        else:
            dst.copy_(src)

    def launch(self, ranges=None):
        """ranges=None: the whole buffer.  Otherwise a list of [start, end) element ranges (each bucketed); handles
        accumulate until wait(), so an early launch(early_ranges) can overlap the rest of backward."""
        if self.world <= 1 and not self.always:
            return
        if self.noop:      # set by a measurement harness only (bench.py: exchange_exposed_ms, --exchange-noop): the step structure without the collective
            return
        use_avg = dist.get_backend() == "nccl"
        slices = self.slices if ranges is None else [b for a, e in ranges for b in bucket_slices(e, self.bucket, a)]
        for s, e in slices:
            t = self.flat[s:e]
            back = None
            if self.half is not None:
                back = t
                t = self.half[s:e]
                self._cast(back, t, True)
            if use_avg:
                self.handles.append((dist.all_reduce(t, op=dist.ReduceOp.AVG, async_op=True), None, t, back))
            else:   # gloo has no AVG: sum then scale (CPU tests, single-GPU multi-rank test)
                self.handles.append((dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True), t, t, back))

    def wait(self):
        for h, scale_t, t, back in self.handles:
            h.wait()
            if scale_t is not None:
                scale_t.div_(self.world)
            if back is not None:
                self._cast(t, back, False)
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
