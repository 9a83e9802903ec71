"""-m gpu: BASELINE configs[3]'s PER-RANK workload through the N > 1 code.  configs[3] is 8 ranks x (ResNet-50, 8 tiles of 256 x 256, bf16) with RCCL
gradient all-reduce; the box has one GPU and RCCL refuses two ranks on one device, so two ranks run that per-rank workload side by side on GPU 0
and talk over gloo (the transport aside, this is the code that runs on a multi-GPU node: the captured multi-graph step cut at the five real
SyncBatchNorm collectives, the three-range early gradient exchange on the flat buffer, per-rank dropout streams).  tests/test_gpu_dp2.py runs
the same structure at 2 / 4 / 8 ranks on a small model; this file runs it at the size the scaling run times.
Reference: train.py:116-123,141-159 (paddle.DataParallel step), paddle_EMRT.py:64 / fcn_head.py:53 (the SyncBatchNorm layers)."""
import os

import pytest
import torch

from tests.test_gpu_dp2 import _run_workers

pytestmark = pytest.mark.gpu


def _cfg3_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      EMRT_DIST_BACKEND="gloo")
    import argparse
    import torch.distributed as dist
    from emrt_amd.config import get_config, update_config
    from emrt_amd.distributed import init_process_group
    from emrt_amd.engine import TrainEngine
    from emrt_amd.runtime import BF16
    from emrt_amd.src.models import get_model
    from emrt_amd.src.models.losses import get_loss_function
    from emrt_amd.src.models.solver import get_optimizer, get_scheduler
    here = os.path.dirname(os.path.abspath(__file__))
    torch.set_num_threads(2)
    r, _, w = init_process_group()
    assert (r, w) == (rank, world) and dist.get_backend() == "gloo"
    try:
        cfg = update_config(get_config(), argparse.Namespace(cfg=os.path.join(here, "..", "emrt_amd", "configs", "EMRT", "EMRT_256x256_160k_potsdam.yaml")))
        assert cfg.MODEL.ENCODER.TYPE == "resnet50" and cfg.DATA.BATCH_SIZE == 8 and tuple(cfg.DATA.CROP_SIZE) == (256, 256)      # configs[3]'s per-rank batch
        cfg.TRAIN.ITERS = 100
        B, S, nsteps = cfg.DATA.BATCH_SIZE, 256, 3          # (captured mode: one eager step -- it fills the per-shape constant caches, which no capture may do -- then two replays)
        g = torch.Generator().manual_seed(300 + rank)          # every rank its own tiles, as the DistributedTileSampler hands them out
        x = torch.randn(B, 3, S, S, generator=g).cuda()
        labels = torch.randint(0, 6, (B, S, S), generator=g)
        labels[torch.rand(B, S, S, generator=g) < 0.02] = 255
        labels = labels.cuda()

        def build():
            from emrt_amd.runtime import ctx
            torch.manual_seed(5)                               # identical initial weights on every rank
            ctx().salt_counter = 0                             # (dropout sites are numbered at construction: the same numbers for every build)
            model = get_model(cfg)
            model.to_hip("cuda:0", BF16, seed=9 + rank)        # (dropout stays ON, p = 0.1: the benchmark's step; every rank its own stream)
            return model, get_optimizer(model, get_scheduler(cfg), cfg)

        # the five real SyncBatchNorm layers (four pyramid-pooling branches + the auxiliary head) and nothing else synchronise
        model, opt = build()
        n_sync = sum(1 for m in model.modules() if type(m).__name__ == "BatchNorm2D" and m.state.sync)
        assert n_sync == 5, n_sync
        # 1. the reducer on the real 54 M-element device buffer: averaged flat gradient == mean of the ranks' local gradients
        eng = TrainEngine(model, opt, get_loss_function(cfg), world, use_graph=False, early_exchange=False)
        model.train()
        eng._fwd_bwd(x, labels)
        n = model.store.n_train
        assert abs(n - 54.03e6) < 0.05e6
        local = model.store.grad[:n].clone()
        gathered = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        eng.reducer.allreduce()
        torch.cuda.synchronize()
        want = sum(gathered) / world
        assert (gathered[0] - gathered[1]).abs().max() > 1e-6, "ranks should see different tiles"
        assert torch.allclose(model.store.grad[:n], want, rtol=1e-6, atol=1e-7)
        del local, gathered, want

        # 2. the eager two-rank step with the early exchange against the captured multi-graph step (what bench.py --gpus N times)
        traces, weights, info = {}, {}, {}
        for mode in ("eager_early", "graph_early"):
            model, opt = build()
            eng = TrainEngine(model, opt, get_loss_function(cfg), world, use_graph=(mode == "graph_early"), warmup_eager=1, early_exchange=True)
            assert eng.two_phase and eng.early_ranges is not None and len(eng.seg_ranges) == 3          # three exchange ranges: heads + transformer + layer4 | layer3 | the rest
            traces[mode] = [eng.step(x, labels).item() for _ in range(nsteps)]
            torch.cuda.synchronize()
            wmax, wmin = model.store.master[:n].clone(), model.store.master[:n].clone()
            dist.all_reduce(wmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(wmin, op=dist.ReduceOp.MIN)
            assert torch.equal(wmax, wmin), "ranks' weights differ after %d steps in mode %s" % (nsteps, mode)
            assert torch.isfinite(wmax).all()
            weights[mode] = model.store.master[:n].clone()
            if mode == "graph_early":
                assert eng.graph_a is not None and eng.graph_a2 is not None and eng.graph_b is not None
                info["n_graphs"] = eng.graph_a.n_graphs
                assert eng.graph_a.n_graphs >= 3, "forward + first backward segment must be cut at the SyncBatchNorm all-reduces"
        for a, b in zip(traces["eager_early"], traces["graph_early"]):
            assert a == a and abs(a - b) <= 2e-4 * max(1.0, abs(a)), traces          # same arithmetic, same dropout streams
        rel = ((weights["eager_early"] - weights["graph_early"]).norm() / weights["eager_early"].norm()).item()
        assert rel < 1e-6, rel          # (fp32 atomics in the weight gradients: the order of the last bits is free)
        print("rank %d: loss traces eager %s captured %s, weights rel diff %.2e, %d graphs in the first segment" % (
            rank, traces["eager_early"], traces["graph_early"], rel, info["n_graphs"]), flush=True)
        q.put((rank, "ok", traces["graph_early"]))
    finally:
        dist.destroy_process_group()


def test_configs3_per_rank_workload_two_ranks_on_one_gpu():
    got = _run_workers(_cfg3_worker, 2, timeout=900)
    assert [g[:2] for g in got] == [(0, "ok"), (1, "ok")]
    assert got[0][2] != got[1][2], "per-rank losses should differ (different tiles)"
