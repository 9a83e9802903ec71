"""CPU: the data-parallel plumbing over gloo with world_size 2, 4 and 8 -- BASELINE configs[3]'s rank count and the driver's other
scaling points -- (the RCCL path is the same code with backend "nccl"):
flat-gradient averaging in buckets, rank-strided tile sharding, and the DP identity the reducer must deliver --
rank-averaged gradients == gradients of the concatenated batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from emrt_amd.distributed import FlatGradReducer, DistributedTileSampler, init_process_group
    torch.set_num_threads(max(1, 8 // world))          # (the ranks share this machine's cores)
    r, lr, w = init_process_group("gloo")
    assert (r, w) == (rank, world) and dist.get_backend() == "gloo"
    try:
        # 1. bucketed averaging of a flat buffer (tail beyond n is left untouched)
        n = 1000
        flat = torch.arange(n + 24, dtype=torch.float32) * (rank + 1)
        red = FlatGradReducer(flat, n, world, bucket_elems=256)
        assert len(red.slices) == 4 and red.slices[-1] == (768, 1000)
        red.allreduce()
        want = torch.arange(n + 24, dtype=torch.float32) * (sum(range(1, world + 1)) / world)
        assert torch.allclose(flat[:n], want[:n])
        assert torch.equal(flat[n:], torch.arange(n, n + 24, dtype=torch.float32) * (rank + 1))
        # 1b. early exchange: two ranged launches (the second after "more backward" has filled the late range), one wait
        flat = torch.zeros(n + 24)
        flat[:300] = rank + 1.0
        flat[700:n] = 10.0 * (rank + 1)
        red = FlatGradReducer(flat, n, world, bucket_elems=128)
        red.launch([(0, 300), (700, n)])
        red_handles = len(red.handles)
        flat[300:700] = 100.0 * (rank + 1)          # written after the first launch, exchanged by the second
        red.launch([(300, 700)])
        assert red_handles == 3 + 3 and len(red.handles) == red_handles + 4
        red.wait()
        m = sum(range(1, world + 1)) / world
        assert torch.allclose(flat[:300], torch.full((300,), m)) and torch.allclose(flat[300:700], torch.full((400,), 100 * m))
        assert torch.allclose(flat[700:n], torch.full((300,), 10 * m)) and not flat[n:].any() and not red.handles
        # 1c. bf16 exchange (half the bytes over the links): the averaged gradient is the fp32 one to bf16 rounding, ranges and tail intact
        g = torch.Generator().manual_seed(7 + rank)
        flat = torch.randn(n + 24, generator=g)
        keep = flat.clone()
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        red = FlatGradReducer(flat, n, world, bucket_elems=256, exchange_dtype="bf16")
        red.launch([(0, 300)])
        red.launch([(300, n)])
        red.wait()
        want = sum(gathered) / world
        assert torch.equal(flat[n:], keep[n:])
        err = (flat[:n] - want[:n]).abs().max().item()
        assert 0 < err < 2.0 ** -7 * want[:n].abs().max().item(), err          # rounded (not fp32-exact), within bf16 resolution
        # 2. DP identity on a BN-free piece of the oracle (decoder layer): mean of per-rank grads == full-batch grads
        from oracle.emrt_torch import TransformerDecoderLayer
        torch.manual_seed(0)
        layer = TransformerDecoderLayer(256, 8, 64, 0.0, 3, 6)
        layer.self_attn.dropout = 0.0
        shapes = [(4, 4), (2, 2), (1, 1)]
        g = torch.Generator().manual_seed(1)
        B = 2 * world
        tgt, mem = torch.randn(B, 10, 256, generator=g), torch.randn(B, 21, 256, generator=g)
        refp, qp = torch.rand(B, 10, 3, 2, generator=g), torch.randn(B, 10, 256, generator=g)
        def grads(sl):
            layer.zero_grad()
            layer(tgt[sl], refp[sl], mem[sl], shapes, None, qp[sl]).pow(2).mean().backward()
            return torch.cat([p.grad.flatten() for p in layer.parameters()])
        full = grads(slice(0, B))
        mine = grads(slice(2 * rank, 2 * rank + 2)).clone()
        FlatGradReducer(mine, mine.numel(), world, bucket_elems=50000).allreduce()
        assert (mine - full).abs().max().item() < 1e-5 * (1 + full.abs().max().item())
        # 3. sampler: ranks partition the (padded) index set, same permutation on every rank, reshuffled per epoch
        s = DistributedTileSampler(37, 4, rank, world, shuffle=True, drop_last=True, seed=3)
        s.set_epoch(2)
        mine_idx = [i for b in s for i in b]
        gathered = [None] * world
        dist.all_gather_object(gathered, mine_idx)
        allidx = [i for l in gathered for i in l]
        assert len(mine_idx) == len(s) * 4 and len(set(allidx)) >= 32 and set(allidx) <= set(range(37))
        # drop_last: every rank the same number of whole batches, no index handed to two ranks (src/utils/dataloader.py:38-41)
        assert len(allidx) == world * len(s) * 4 == len(set(allidx)) and len({len(l) for l in gathered}) == 1
        s.set_epoch(3)
        assert [i for b in s for i in b] != mine_idx
        # the reference's recipe at configs[3]'s size: 8 tiles per rank (config.py:10), every sample seen at most once per epoch
        big = DistributedTileSampler(3456, 8, rank, world, shuffle=True, drop_last=True, seed=1234)
        big.set_epoch(0)
        mine_big = [i for b in big for i in b]
        dist.all_gather_object(gathered, mine_big)
        flat_big = [i for l in gathered for i in l]
        assert len(flat_big) == len(set(flat_big)) == (3456 // (8 * world)) * 8 * world and all(len(b) == 8 for b in big)
        q.put((rank, "ok"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_gloo_world_size(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
    for p in procs:
        if p.is_alive():
            p.kill()          # (exactly the processes started here)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    got = sorted(q.get(timeout=5) for _ in range(world))
    assert got == [(r, "ok") for r in range(world)]


def test_bucket_slices():
    from emrt_amd.distributed import bucket_slices
    assert bucket_slices(10, 4) == [(0, 4), (4, 8), (8, 10)]
    assert bucket_slices(8, 8) == [(0, 8)]
    assert bucket_slices(0, 8) == []
    assert bucket_slices(10, 4, 3) == [(3, 7), (7, 10)]
