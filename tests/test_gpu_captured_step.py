"""-m gpu: the thing bench.py TIMES is the thing the parity tests RAN.

bench.py replays a captured hipGraph of the bf16 ResNet-50 train step (train.py:141-159: clear gradients -> forward -> CE + 0.4 aux CE ->
backward -> clip + SGD-momentum), and at BASELINE configs[1] / configs[2] sizes that graph holds kernels no small test selects: the
256 x 256 LDS-DMA GEMMs with the registered scratch slab, the 24-layer weight-gradient batches with their `dw_is_zero` stores, the
BatchNorm-operand consumers.  Here the SAME engine configuration as bench.py's (TrainEngine(use_graph=True), scratch registered,
Context.wgrad_batch at its default) runs three steps from identical weights next to the eager engine, and step 1 of both is compared with
the fp32 CPU oracle's train step under the bounds of tests/test_gpu_bench_shapes.py::test_full_size_bf16_model_vs_fp32_oracle.

  captured == eager : the two run the same kernels in the same order on the same data; what differs between two runs of either is the order
                      of the fp32 atomic adds of the weight gradients (tests/test_gpu_bench_shapes.py:
                      test_wgrad_fp32_atomics_run_to_run_spread_is_bounded, ~1e-7 relative per layer), which bf16 activations amplify
                      to ~1e-4 of the loss over three updates.  Bounds below are a few times what was measured on MI355X (printed).
  step 1 == oracle  : loss within 2e-3 relative, whole-gradient cosine >= 0.985, norm ratio within 2 % (dropout off on both sides: the
                      oracle cannot share the device's mask stream; weights conditioned as in the test this one borrows its bounds from).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from emrt_amd.engine import TrainEngine                                    # noqa: E402
from emrt_amd.runtime import BF16, ctx                                     # noqa: E402
from emrt_amd.src.models import get_model                                  # noqa: E402
from emrt_amd.src.models.losses import get_loss_function                   # noqa: E402
from emrt_amd.src.models.solver import get_optimizer, get_scheduler        # noqa: E402
from oracle import train_ref                                               # noqa: E402
from tests.test_gpu_bench_shapes import BF16_GRAD_COSINE, BF16_GRAD_NORM_RATIO, BF16_LOSS_REL      # noqa: E402
from tests.test_gpu_model import calibrated_oracle, make_config, perturb_sampling_offsets          # noqa: E402

STEPS = 3
TRACE_REL = 1e-3          # captured vs eager loss, every step (measured: see the printed line)
WEIGHT_REL = 2e-4         # relative L2 distance of all trainable weights after STEPS updates
GRAD_REL = 2e-2           # ... of step 1's flat gradient, or 2.5 x what two EAGER runs differ by (measured on MI355X: 5-6 %, printed)


@pytest.mark.parametrize("B,S,ncls", [(8, 256, 6)], ids=["cfg2-8x256"])      # (cfg3-4x512: tests/test_gpu_captured_step_cfg3.py)
def test_captured_bf16_step_equals_eager_and_tracks_the_oracle(B, S, ncls):
    captured_step_case(B, S, ncls)


def captured_step_case(B, S, ncls):
    g = torch.Generator().manual_seed(29)
    x = torch.randn(B, 3, S, S, generator=g)
    labels = torch.randint(0, ncls, (B, S, S), generator=g)
    labels[torch.rand(B, S, S, generator=g) < 0.02] = 255
    ref = calibrated_oracle("resnet50", x, ncls=ncls, condition=0.1)
    perturb_sampling_offsets(ref)
    state = {k: v.clone() for k, v in ref.state_dict().items()}
    cfg = make_config("resnet50", iters=1000, ncls=ncls)
    xd, ld = x.cuda(), labels.cuda()

    runs = {}
    for mode in ("eager", "eager2", "graph"):          # eager twice: the yardstick for what two runs of the SAME program differ by
        model = get_model(cfg)
        model.load_state_dict(state)
        model.to_hip("cuda:0", BF16)
        model.set_dropout(0.0)
        # the shape-keyed constants (sine position embedding, reference grid: computed on the host, emrt.py _constants) are uploaded by the first
        # forward of a shape, and a host -> device copy cannot be captured: an EVAL forward fills that cache and leaves weights, BatchNorm
        # statistics, optimizer and step counter untouched -- bench.py's eager warm-up steps do the same job there
        model.eval()
        model(xd)
        model.train()
        opt = get_optimizer(model, get_scheduler(cfg), cfg)
        # warmup_eager=0: the FIRST step is already the captured one (TrainEngine._capture registers the weight-gradient scratch before it
        # begins the capture), so all three steps of the "graph" run are replays of the graph bench.py times
        eng = TrainEngine(model, opt, get_loss_function(cfg), 1, use_graph=(mode == "graph"), warmup_eager=0)
        n = model.store.n_train
        out = {"loss": []}
        for i in range(STEPS):
            out["loss"].append(eng.step(xd, ld).item())
            if i == 0:
                torch.cuda.synchronize()
                out["grad"] = model.store.grad[:n].clone()           # raw gradient of step 1 (the optimizer scales a copy, not the buffer)
                out["named_grads"] = {k: p.grad.detach().float().cpu().clone() for k, p in model.named_parameters() if p.grad is not None}
        torch.cuda.synchronize()
        out["weights"] = model.store.master[:n].clone()
        if mode == "graph":
            assert eng.graph_a is not None and eng.graph_a.n_graphs == 1 and eng.calls == STEPS
            assert getattr(ctx(), "_scratch", None) is not None, "the captured bf16 step must have the weight-gradient scratch registered"
        runs[mode] = out
        del eng, opt, model

    def distance(u, v):
        return (max(abs(a - b) / max(1.0, abs(a)) for a, b in zip(u["loss"], v["loss"])),
                ((u["grad"] - v["grad"]).norm() / u["grad"].norm()).item(), ((u["weights"] - v["weights"]).norm() / u["weights"].norm()).item())
    e, c = runs["eager"], runs["graph"]
    trace, grel, wrel = distance(e, c)
    trace0, grel0, wrel0 = distance(e, runs["eager2"])
    print("CAPTURED vs EAGER %dx%dx%d bf16: loss traces %s vs %s (worst rel %.2e), step-1 gradient rel L2 %.2e, weights after %d steps rel L2 %.2e" % (
        B, S, S, ["%.5f" % v for v in c["loss"]], ["%.5f" % v for v in e["loss"]], trace, grel, STEPS, wrel))
    print("EAGER vs EAGER (same program twice: fp32-atomic order only): loss %.2e, step-1 gradient %.2e, weights %.2e" % (trace0, grel0, wrel0))
    assert all(v == v and v > 0 for v in c["loss"] + e["loss"])
    # the gradient of this random-init network is the sensitive quantity: bf16 activations turn the weight gradients' summation-order noise
    # into last-bit flips that grow with depth, so two eager runs already differ by percents in the gradient while loss and weights agree
    # to 1e-5; the captured step must be no further from eager than eager is from itself (x 2.5 + a floor)
    assert trace < max(TRACE_REL, 2.5 * trace0) and wrel < max(WEIGHT_REL, 2.5 * wrel0), (trace, trace0, wrel, wrel0)
    assert grel < max(GRAD_REL, 2.5 * grel0), (grel, grel0)

    # ---- step 1 of BOTH against the fp32 CPU oracle's train step -----------------------------------------------------------------
    ref.train()
    loss_r = train_ref.mix_softmax_ce_loss(ref(x), labels)
    loss_r.backward()
    refp = dict(ref.named_parameters())
    for mode, r in runs.items():
        if mode == "eager2":
            continue
        rel_loss = abs(r["loss"][0] - loss_r.item()) / abs(loss_r.item())
        dot = n_hip = n_ref = 0.0
        for k, gg in r["named_grads"].items():
            gr = refp[k].grad
            if gr is None:
                continue
            gg, gr = gg.double(), gr.double()
            dot += float((gg * gr).sum())
            n_hip += float((gg * gg).sum())
            n_ref += float((gr * gr).sum())
        cos = dot / (n_hip ** 0.5 * n_ref ** 0.5)
        ratio = (n_hip / n_ref) ** 0.5
        print("%s step 1 vs fp32 oracle: loss %.5f vs %.5f (rel %.2e), gradient cosine %.5f, norm ratio %.4f" % (mode, r["loss"][0], loss_r.item(), rel_loss, cos, ratio))
        assert rel_loss < BF16_LOSS_REL, (mode, rel_loss)
        assert cos > BF16_GRAD_COSINE and abs(ratio - 1.0) < BF16_GRAD_NORM_RATIO, (mode, cos, ratio)
