"""-m gpu parity tests, model level: the HIP EMRT (through get_model / config surface) against the torch-CPU oracle
on identical weights and inputs.  fp32: logits within 1e-3, argmax masks bit-exact (BASELINE.json north_star);
gradients, a 3-step training trace and hipGraph replay are checked too.  bf16 is checked for sanity/correlation."""
import argparse
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from emrt_amd.config import get_config, update_config                    # noqa: E402
from emrt_amd.runtime import ctx, F32, BF16                               # noqa: E402
from emrt_amd.src.models import get_model                                  # noqa: E402
from emrt_amd.src.models.losses import get_loss_function                   # noqa: E402
from emrt_amd.src.models.solver import get_optimizer, get_scheduler        # noqa: E402
from emrt_amd.engine import TrainEngine                                    # noqa: E402
from oracle.emrt_torch import EMRT as OracleEMRT, BatchNorm2D as OBN        # noqa: E402
from oracle import train_ref                                               # noqa: E402

CFG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "emrt_amd/configs/EMRT/EMRT_256x256_160k_potsdam.yaml")


def make_config(backbone="resnet18", iters=1000, ncls=6):
    cfg = update_config(get_config(), argparse.Namespace(cfg=CFG))
    cfg.MODEL.ENCODER.TYPE = backbone
    cfg.TRAIN.ITERS = iters
    cfg.DATA.NUM_CLASSES = ncls
    return cfg


def oracle_no_dropout(m):
    for mod in m.modules():
        if isinstance(mod, (torch.nn.Dropout, torch.nn.Dropout2d)):
            mod.p = 0.0
        if hasattr(mod, "dropout") and isinstance(getattr(mod, "dropout"), float):
            mod.dropout = 0.0


def condition_residual_branches(ref, scale):
    """Scale the last BatchNorm gamma of every residual branch (ResNet bottleneck bn3 / basic-block bn2, EFP Conv2dBlock
    conv2): a randomly initialised BatchNorm network is CHAOTIC -- a perturbation grows ~1.2x per conv-BN-ReLU layer, so torch's
    own CPU bf16 autocast of the fp32 oracle lands 60 % (relative L2) away from it at ResNet-50 depth -- whereas trained
    networks (and the usual zero-init-residual recipes) keep their residual branches small.  With the branches down-weighted
    the same autocast comparison gives ~5 %, which is what a bf16-vs-fp32 bound can meaningfully be stated on."""
    with torch.no_grad():
        for n, m in ref.named_modules():
            if n.startswith("backbone.layer") and (n.endswith(".bn3") or (n.endswith(".bn2") and not hasattr(ref.get_submodule(n.rsplit(".", 1)[0]), "bn3"))):
                m.weight.mul_(scale)
            if n.startswith("EFP.") and n.endswith(".conv2.1"):
                m.weight.mul_(scale)


def calibrated_oracle(backbone, x, seed=0, ncls=6, condition=None):
    """Oracle with BN running statistics set from one batch (momentum 0 => running = batch) so that eval-mode
    activations are O(1) with random-initialised weights."""
    torch.manual_seed(seed)
    ref = OracleEMRT(ncls, backbone)
    oracle_no_dropout(ref)
    if condition is not None:
        condition_residual_branches(ref, condition)
    for mod in ref.modules():
        if isinstance(mod, OBN):
            mod.momentum = 0.0
    ref.train()
    with torch.no_grad():
        ref(x)
    for mod in ref.modules():
        if isinstance(mod, OBN):
            mod.momentum = 0.9
    return ref


# Pixels whose argmax differs from the float64 oracle's in the fp32 eval forward, per parity case: ceilings frozen from the measured counts
# (rounds 4 / 5 / 6 on five boxes: 0 | 1-3 | 4 | 19-21 | 1 | 29-30 | 104-106 | 0 | 4 | 25-26 | 24; profiles/r6_parity_measured.txt) + ~20 % for the
# run-to-run spread of two fp32 summation orders.  Every counted pixel is also asserted to be a near-tie of the float64 oracle (margin < 2e-3).
ARGMAX_FLIP_CEILING = {("resnet18", 2, 64): 1, ("resnet50", 2, 128): 4, ("resnet50", 1, 256): 6, ("resnet50", 1, 512): 26, ("resnet18", 1, 256): 2,
                       ("resnet50", 8, 256): 36, ("resnet50", 4, 512): 126, ("resnet34", 1, 128): 1, ("resnet50", 2, 224): 7, ("resnet50", 2, 384): 32,
                       ("resnet50", 2, 448): 30}
# north_star's "fp32 logits within 1e-3" holds against the float64 oracle for every case but this one: at ResNet-101 depth the randomly initialised
# BatchNorm network amplifies fp32 rounding until the float32 CPU oracle ITSELF is 6.3e-3 from float64 (measured); bound there: 1.25 x that distance
LOGIT_1E3_EXCEPTIONS = {("resnet101", 1, 128)}


def assert_argmax_match(got, ref, tol=1e-3, max_flips=None):
    """argmax masks must agree everywhere the oracle's decision is not a near-tie: a pixel whose top-2 logit margin
    is below 2*tol can legitimately flip between two fp32 implementations that differ by <= tol."""
    ga, ra = got.argmax(1), ref.argmax(1)
    top2 = ref.topk(2, dim=1).values
    margin = top2[:, 0] - top2[:, 1]
    bad = (ga != ra) & (margin > 2 * tol)
    assert not bad.any(), "%d pixels differ with a decisive margin (max margin %.3g)" % (int(bad.sum()), margin[bad].max().item())
    flips = int((ga != ra).sum())
    if max_flips is None:
        # <= 0.0133 % of the pixels = 1.3 x the worst measured (0 / 3 / 4 / 19 / 106 of 8k / 33k / 65k / 262k / 1049k, i.e. <= 0.0101 %; round 4 allowed
        # 2 x); every one of them is a pixel whose float64 top-2 margin is below 2 * tol (asserted above): two fp32 summation orders cannot agree on those
        max_flips = max(8, ga.numel() // 7500)
    # near-tie flips are bounded too: at most max_flips pixels, and never more than the oracle itself has near-ties
    assert flips <= max_flips and flips <= int((margin <= 2 * tol).sum()), "%d argmax flips (allowed %d)" % (flips, max_flips)
    return flips


def perturb_sampling_offsets(ref, scale=0.05, seed=3):
    """_reset_parameters puts every sample of a query's own level EXACTLY on a pixel centre (integer coordinates),
    where bilinear sampling is not differentiable; generic offsets make gradients comparable."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if "sampling_offsets" in n or "attention_weights" in n:
                p.add_(torch.randn(p.shape, generator=g) * scale * (0.1 if p.dim() == 2 else 1.0))


def build_pair(backbone, x, dtype=F32, perturb=False, ncls=6, condition=None):
    ref = calibrated_oracle(backbone, x, ncls=ncls, condition=condition)
    if perturb:
        perturb_sampling_offsets(ref)
    model = get_model(make_config(backbone, ncls=ncls))
    model.load_state_dict(ref.state_dict())
    model.to_hip("cuda:0", dtype)
    model.set_dropout(0.0)
    return ref, model


def float64_train_forward(ref, x):
    """Train-mode logits of a float64 copy of the oracle (the exact result of the reference's arithmetic on these inputs); the
    copy's BatchNorm buffers are its own, so the fp32 oracle is left untouched."""
    import copy
    ref64 = copy.deepcopy(ref).double().train()
    with torch.no_grad():
        return [t.float() for t in ref64(x.double())]


def assert_train_logits(out, out32, out64, what):
    """north_star: fp32 logits within 1e-3.  Train-mode BatchNorm divides by batch statistics that are themselves fp32 sums over
    B*H*W values, so the float32 CPU oracle's OWN distance from float64 is measured next to the HIP path's and printed; the HIP
    path must be within 1e-3 of float64, or -- where fp32 itself cannot be -- no further from it than the fp32 oracle is."""
    for name, a, b32, b64 in (("main", out[0], out32[0], out64[0]), ("aux", out[1], out32[1], out64[1])):
        a, b32 = a.cpu(), b32.detach()
        e64, e32, o32 = (a - b64).abs().max().item(), (a - b32).abs().max().item(), (b32 - b64).abs().max().item()
        print("%s, %s logits (train mode): |hip-f64| %.3g  |hip-f32 oracle| %.3g  |f32 oracle-f64| %.3g  (|ref| max %.3g)"
              % (what, name, e64, e32, o32, b64.abs().max().item()))
        assert e64 < max(1e-3, 1.25 * o32), "%s %s: |hip - float64 oracle| = %.3g (fp32 oracle itself: %.3g)" % (what, name, e64, o32)
        assert e32 < 2e-3, "%s %s: |hip - float32 oracle| = %.3g" % (what, name, e32)


# (the three full-size cases -- resnet50 1 x 512 x 512 and configs[1] / configs[2] at their real batch sizes, 8 x 256 x 256 and 4 x 512 x 512 -- run
# the same body from tests/test_gpu_model_full.py: their float64 oracle evaluations are 4 minutes of host time, and pytest-xdist hands out whole files)
# the reference's other shipped EMRT tile sizes (configs/EMRT/EMRT_{224x224,384x384,448x448}_160k_potsdam.yaml): pyramids 28/14/7, 48/24/12, 56/28/14 --
# odd and non-power-of-two levels through the LDS-staged MSDA kernels, the adaptive-pool bins and the x2 / x4 resizes.  Run from
# tests/test_gpu_model_tiles.py (4 minutes of float64 host time: a file of their own for pytest-xdist)
FORWARD_CASES_TILES = [("resnet50", 2, 224, 6), ("resnet50", 2, 384, 6), ("resnet50", 2, 448, 6)]
FORWARD_CASES_FULL = [("resnet50", 1, 512, 7),      # BASELINE configs[2] geometry (LoveDA 512x512, 7 classes, Lv = 5376)
                      ("resnet50", 8, 256, 6),      # configs[1]: batch 8
                      ("resnet50", 4, 512, 7)]      # configs[2]: batch 4


# (the small cases run from tests/test_gpu_model_eval.py: a file of its own for pytest-xdist)
FORWARD_CASES_SMALL = [("resnet18", 2, 64, 6), ("resnet50", 2, 128, 6), ("resnet50", 1, 256, 6),
                       ("resnet18", 1, 256, 6),      # configs[0]: ResNet-18, one 256x256 tile, on the HIP path
                       # the other depths the reference's constructor accepts (paddle_EMRT.py:229-234)
                       ("resnet34", 1, 128, 6), ("resnet101", 1, 128, 6)]


def forward_logits_case(backbone, B, S, ncls):
    """fp32 logits within 1e-3 of the oracle evaluated in float64 (the exact result of the reference's arithmetic; the
    fp32 CPU oracle itself deviates from it by a few 1e-4), and within 2e-3 of the fp32 oracle; argmax masks agree
    wherever the decision is not a sub-tolerance tie."""
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(B, 3, S, S, generator=g)
    ref, model = build_pair(backbone, x, ncls=ncls)
    ref.eval()
    model.eval()
    with torch.no_grad():
        want32 = ref(x)
        ref.double()
        want64 = [t.float() for t in ref(x.double())]
    got = model(x.cuda())
    for name, a, b64, b32 in (("main", got[0], want64[0], want32[0]), ("aux", got[1], want64[1], want32[1])):
        a = a.cpu()
        assert a.shape == b64.shape == (B, ncls, S, S)
        e64, e32, o32 = (a - b64).abs().max().item(), (a - b32).abs().max().item(), (b32 - b64).abs().max().item()
        print("%s logits: |hip-f64| %.3g  |hip-f32 oracle| %.3g  |f32 oracle-f64| %.3g  (|ref| max %.3g)" % (name, e64, e32, o32, b64.abs().max().item()))
        # 1e-3 (north_star) wherever fp32 arithmetic itself can hold it: at ResNet-101 depth the randomly initialised BatchNorm network
        # amplifies rounding so much that the float32 CPU oracle is 6.3e-3 from float64 (measured); there the HIP path must be no
        # further from float64 than 1.25 x the fp32 oracle's own distance
        if (backbone, B, S) in LOGIT_1E3_EXCEPTIONS:
            assert name != "main" or o32 > 1e-3, "the fp32 oracle is within 1e-3 of float64 here: %s no longer needs its exception" % backbone
            assert e64 < max(1e-3, 1.25 * o32), "%s logits: max |diff| vs float64 oracle %.3g (fp32 oracle itself: %.3g)" % (name, e64, o32)
            assert e32 < max(2e-3, 2.0 * o32), "%s logits: max |diff| vs float32 oracle %.3g" % (name, e32)
        else:
            # the north_star bound as written: |hip - float64| < 1e-3, no escape through the fp32 oracle's own error (measured: <= 7e-4 on every case)
            assert e64 < 1e-3, "%s logits: max |diff| vs float64 oracle %.3g (north_star: 1e-3)" % (name, e64)
            assert e32 < 2e-3, "%s logits: max |diff| vs float32 oracle %.3g" % (name, e32)
    exception = (backbone, B, S) in LOGIT_1E3_EXCEPTIONS
    tol = 1.25 * (want32[0] - want64[0]).abs().max().item() if exception else 1e-3
    # (where fp32 itself is beyond 1e-3 -- ResNet-101 -- the count bound is the number of near-ties at that tolerance)
    flips = assert_argmax_match(got[0].cpu(), want64[0], tol=tol, max_flips=B * S * S if exception else ARGMAX_FLIP_CEILING[(backbone, B, S)])
    exact = int((got[0].cpu().argmax(1) != want64[0].argmax(1)).sum())
    print("ARGMAX %s %dx%dx%d fp32 eval: %d of %d pixels differ from the float64 oracle's mask (every one a near-tie: float64 top-2 margin < %.1e); "
          "vs the fp32 oracle's mask: %d" % (backbone, B, S, S, exact, B * S * S, 2 * tol, int((got[0].cpu().argmax(1) != want32[0].argmax(1)).sum())))
    assert exact == flips


def train_forward_and_gradients_case():      # run from tests/test_gpu_model_train.py (a file of its own for pytest-xdist: 1-2 minutes of oracle time)
    """Train-mode forward, loss and EVERY parameter gradient against the oracle evaluated in float64.
    This random-initialised, tiny-batch network is ill-conditioned: the float32 CPU oracle's own gradients are 1-4 %
    (norm-wise) away from float64 (measured, see tools/debug_stages.py), so the HIP gradients are held to the same
    standard: per parameter <= max(2e-2, 3x the fp32 oracle's error), and no worse than it in the median."""
    g = torch.Generator().manual_seed(7)
    B, S = 2, 128
    x = torch.randn(B, 3, S, S, generator=g)
    labels = torch.randint(0, 6, (B, S, S), generator=g)
    labels[torch.rand(B, S, S, generator=g) < 0.02] = 255
    ref, model = build_pair("resnet50", x, perturb=True)
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    ref.train()
    train_ref.mix_softmax_ce_loss(ref(x), labels).backward()
    g32 = {n: p.grad.clone() for n, p in ref.named_parameters() if p.grad is not None}
    ref.zero_grad()
    ref.load_state_dict(sd)
    ref.double()
    model.train()
    out_r = ref(x.double())
    loss_r = train_ref.mix_softmax_ce_loss(out_r, labels)
    loss_r.backward()
    loss_fn = get_loss_function(make_config("resnet50"))
    model.clear_gradients()
    out = model(x.cuda())
    loss = loss_fn(out, labels.cuda())
    loss.backward()
    torch.cuda.synchronize()
    assert (out[0].cpu() - out_r[0].detach().float()).abs().max().item() < 1e-3
    assert (out[1].cpu() - out_r[1].detach().float()).abs().max().item() < 1e-3
    assert abs(loss.item() - loss_r.item()) < 1e-4 * max(1.0, abs(loss_r.item()))
    refp = dict(ref.named_parameters())
    rows = []
    for n, p in model.named_parameters():
        gr = refp[n].grad
        if gr is None:
            assert float(p.grad.abs().max()) == 0.0, "%s should receive no gradient" % n
            continue
        gr = gr.float()
        gg = p.grad.cpu()
        if gr.norm().item() < 1e-9:      # conv biases in front of a BatchNorm: mathematically zero gradient
            assert gg.norm().item() < 1e-5, (n, gg.norm().item())
            continue
        e_hip = ((gg - gr).norm() / gr.norm()).item()
        e_o32 = ((g32[n] - gr).norm() / gr.norm()).item()
        rows.append((e_hip, e_o32, n))
    rows.sort(reverse=True)
    med_hip = sorted(r[0] for r in rows)[len(rows) // 2]
    med_o32 = sorted(r[1] for r in rows)[len(rows) // 2]
    print("gradient rel err vs float64 oracle: HIP median %.4f, fp32 CPU oracle median %.4f; worst HIP %s" % (med_hip, med_o32, rows[:4]))
    bad = [r for r in rows if r[0] > max(2e-2, 3 * r[1])]
    assert not bad, "gradient mismatch (hip err, fp32-oracle err, name): %s" % bad[:8]
    assert med_hip <= 2 * med_o32 + 1e-3, (med_hip, med_o32)
    # BN running statistics were updated identically
    refb = dict(ref.named_buffers())
    for n, b in model.named_buffers():
        assert (b.cpu() - refb[n].float()).abs().max().item() < 1e-3 * (1 + refb[n].abs().max().item()), n


def full_size_train_step_case():      # run from tests/test_gpu_model_train.py
    """The benchmark workload itself (BASELINE configs[1]: ResNet-50, batch 8, 256x256, 6 classes), one train-mode
    forward + loss + backward in fp32 against the fp32 CPU oracle: logits, loss, the whole gradient as one vector
    (norm and direction) and the per-parameter relative errors."""
    g = torch.Generator().manual_seed(17)
    B, S = 8, 256
    x = torch.randn(B, 3, S, S, generator=g)
    labels = torch.randint(0, 6, (B, S, S), generator=g)
    labels[torch.rand(B, S, S, generator=g) < 0.02] = 255
    ref, model = build_pair("resnet50", x, perturb=True)
    ref.train()
    out64 = float64_train_forward(ref, x)
    out_r = ref(x)
    loss_r = train_ref.mix_softmax_ce_loss(out_r, labels)
    loss_r.backward()
    model.train()
    model.clear_gradients()
    out = model(x.cuda())
    loss = get_loss_function(make_config("resnet50"))(out, labels.cuda())
    loss.backward()
    torch.cuda.synchronize()
    assert_train_logits(out, out_r, out64, "256x256 batch 8")
    assert abs(loss.item() - loss_r.item()) < 2e-4 * max(1.0, abs(loss_r.item()))
    refp = dict(ref.named_parameters())
    gmax = max(float(q.grad.norm()) for q in refp.values() if q.grad is not None)
    errs, dot, n_hip, n_ref = [], 0.0, 0.0, 0.0
    for n, p in model.named_parameters():
        gr = refp[n].grad
        if gr is None:
            continue
        if gr.norm().item() < 1e-5 * gmax:     # conv biases in front of a BatchNorm: mathematically zero, rounding noise on both sides
            assert p.grad.norm().item() < 1e-4 * gmax, n
            continue
        gg = p.grad.cpu().double()
        gr = gr.double()
        dot += float((gg * gr).sum())
        n_hip += float((gg * gg).sum())
        n_ref += float((gr * gr).sum())
        errs.append((((gg - gr).norm() / gr.norm()).item(), n))
    cos = dot / (n_hip ** 0.5 * n_ref ** 0.5)
    errs.sort(reverse=True)
    med = errs[len(errs) // 2][0]
    print("full-size gradient: cosine %.6f, norm ratio %.5f, per-parameter rel err median %.4f, worst %s" % (
        cos, (n_hip / n_ref) ** 0.5, med, errs[:3]))
    assert cos > 0.9995 and abs((n_hip / n_ref) ** 0.5 - 1.0) < 1e-2
    assert med < 2e-2 and errs[0][0] < 0.15, errs[:5]


def test_three_step_training_trace_matches_oracle():
    g = torch.Generator().manual_seed(11)
    B, S = 2, 64
    x = torch.randn(B, 3, S, S, generator=g)
    labels = torch.randint(0, 6, (B, S, S), generator=g)
    ref, model = build_pair("resnet18", x)
    cfg = make_config("resnet18", iters=100)
    ref.train()
    ropt = train_ref.MomentumRef(list(ref.named_parameters()), 0.9, 1e-4, 1.0)
    sch = get_scheduler(cfg)
    opt = get_optimizer(model, sch, cfg)
    eng = TrainEngine(model, opt, get_loss_function(cfg), 1, use_graph=False)
    for step in range(3):
        lr_ref_loss, lr_ref = train_ref.train_step(ref, ropt, x, labels, step, 0.01, 0.0, 100, 0.9)
        lr_host = opt.get_lr()
        loss_t = eng.step(x.cuda(), labels.cuda())
        assert abs(lr_host - lr_ref) < 1e-12
        assert abs(float(opt.lr_dev.item()) - lr_ref) < 1e-8
        assert abs(loss_t.item() - lr_ref_loss) < 2e-3 * max(1.0, lr_ref_loss), (step, loss_t.item(), lr_ref_loss)
        assert abs(opt.grad_norm() - ropt.last_grad_norm) < 2e-2 * ropt.last_grad_norm, (step, opt.grad_norm(), ropt.last_grad_norm)


def fifty_step_training_trajectory_case():      # run from tests/test_gpu_model_traj.py (a file of its own for pytest-xdist)
    """The whole loop of train.py:141-159 (forward, loss, backward, clip, SGD-momentum with decay, polynomial LR) for 50 steps on a
    ResNet-18 / 64x64 variant, fp32, dropout off, HIP path against the CPU oracle from the same weights on the same two batches: loss,
    gradient norm and a weight checksum at every step.  The two are different fp32 programs integrating a non-linear recurrence, so
    their distance grows with the step count; the bounds below are 3x what was measured (printed: loss 1.4e-3 at worst, weights 4.4e-4 after 50
    steps), not a tolerance chosen in advance.  The gradient NORM of this network is a noisy quantity (batch 2 at 64x64 leaves BatchNorm layers
    with 8 values per channel on the deepest map: a rounding-level change of a normalised activation moves the norm by tens of percent while the
    loss and the weights -- which only see the clipped update -- stay together), so it is bounded on the first ten steps and by its median after."""
    g = torch.Generator().manual_seed(23)
    B, S = 2, 64
    threads = torch.get_num_threads()
    torch.set_num_threads(min(threads, 8))          # (the oracle's tensors are tiny here: 64 threads spend their time handing work round)
    xs = [torch.randn(B, 3, S, S, generator=g) for _ in range(2)]
    labs = [torch.randint(0, 6, (B, S, S), generator=g) for _ in range(2)]
    ref, model = build_pair("resnet18", xs[0])
    cfg = make_config("resnet18", iters=100)
    ref.train()
    ropt = train_ref.MomentumRef(list(ref.named_parameters()), 0.9, 1e-4, 1.0)
    opt = get_optimizer(model, get_scheduler(cfg), cfg)
    eng = TrainEngine(model, opt, get_loss_function(cfg), 1, use_graph=False)
    refp = dict(ref.named_parameters())
    names = [n for n, p in model.named_parameters() if n in refp]
    worst = [0.0, 0.0, 0.0]
    rows = []
    for step in range(50):
        x, lab = xs[step % 2], labs[step % 2]
        loss_ref, lr_ref = train_ref.train_step(ref, ropt, x, lab, step, 0.01, 0.0, 100, 0.9)
        lr_host = opt.get_lr()
        loss_t = eng.step(x.cuda(), lab.cuda())
        assert abs(lr_host - lr_ref) < 1e-12 and abs(float(opt.lr_dev.item()) - lr_ref) < 1e-8
        dl = abs(loss_t.item() - loss_ref) / max(1.0, abs(loss_ref))
        dg = abs(opt.grad_norm() - ropt.last_grad_norm) / ropt.last_grad_norm
        dw = rows[-1][5] if rows else 0.0
        if step % 5 == 4 or step == 0:          # (every parameter to the host: not at every step)
            num = den = 0.0
            hp = dict(model.named_parameters())
            for n in names:
                a, b = hp[n].detach().cpu().double(), refp[n].detach().double()
                num += float((a - b).pow(2).sum())
                den += float(b.pow(2).sum())
            dw = (num / den) ** 0.5
        rows.append((step, loss_t.item(), loss_ref, dl, dg, dw))
        worst = [max(worst[0], dl), max(worst[1], dg), max(worst[2], dw)]
    for r in rows[::5] + [rows[-1]]:
        print("step %2d: loss %.5f (oracle %.5f, rel %.1e)  grad-norm rel %.1e  weights rel L2 %.1e" % r)
    torch.set_num_threads(threads)
    print("TRAJECTORY 50 steps: worst loss rel %.2e, worst grad-norm rel %.2e, final weight distance %.2e" % (worst[0], worst[1], rows[-1][5]))
    assert rows[-1][1] < rows[0][1], "the loss did not go down"
    med_g = sorted(r[4] for r in rows)[len(rows) // 2]
    print("gradient-norm relative difference: median %.2e, first ten steps at most %.2e" % (med_g, max(r[4] for r in rows[:10])))
    assert worst[0] < 4.5e-3 and worst[2] < 1.3e-3 and worst[1] < 0.9 and med_g < 0.15, (worst, med_g)
    assert max(r[3] for r in rows[:10]) < 2e-3 and max(r[4] for r in rows[:10]) < 3e-2      # the early steps: the 3-step test's bounds


def test_hipgraph_replay_equals_eager():
    g = torch.Generator().manual_seed(12)
    B, S = 2, 64
    x = torch.randn(B, 3, S, S, generator=g)
    labels = torch.randint(0, 6, (B, S, S), generator=g)
    losses = {}
    for mode in ("eager", "graph"):
        ref, model = build_pair("resnet18", x)
        cfg = make_config("resnet18", iters=100)
        opt = get_optimizer(model, get_scheduler(cfg), cfg)
        eng = TrainEngine(model, opt, get_loss_function(cfg), 1, use_graph=(mode == "graph"), warmup_eager=1)
        ls = []
        for _ in range(4):
            ls.append(eng.step(x.cuda(), labels.cuda()).item())
        losses[mode] = ls
    for a, b in zip(losses["eager"], losses["graph"]):
        assert abs(a - b) < 1e-3 * max(1.0, abs(a)), losses
    assert losses["graph"][-1] < losses["graph"][0], "loss should decrease on a repeated batch: %s" % losses


def test_two_phase_dp_step_over_rccl_matches_single_graph():
    """The N > 1 step structure -- hipGraph A (zero/fwd/loss/bwd) -> eager RCCL all-reduce(AVG) of the flat gradient
    buffer -> hipGraph B (clip + SGD + re-pack), per-rank BatchNorm statistics inside the capture -- exercised on one GPU
    in a 1-rank nccl process group.  Averaging over one rank is the identity, so the loss trace must equal the
    single-graph engine's."""
    import os
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        g = torch.Generator().manual_seed(21)
        B, S = 2, 64
        x = torch.randn(B, 3, S, S, generator=g)
        labels = torch.randint(0, 6, (B, S, S), generator=g)
        losses = {}
        for mode in ("single", "two_phase", "one_exchange"):
            ref, model = build_pair("resnet18", x)
            cfg = make_config("resnet18", iters=100)
            opt = get_optimizer(model, get_scheduler(cfg), cfg)
            eng = TrainEngine(model, opt, get_loss_function(cfg), 1, use_graph=True, warmup_eager=1, two_phase=(mode != "single"),
                              bucket_elems=4 * 1024 * 1024, early_exchange=(mode == "two_phase"))
            losses[mode] = [eng.step(x.cuda(), labels.cuda()).item() for _ in range(4)]
            if mode != "single":
                assert eng.graph_b is not None and eng.reducer is not None and len(eng.reducer.slices) > 1
                assert (eng.graph_a2 is not None) == (mode == "two_phase")
        for mode in ("two_phase", "one_exchange"):
            for a, b in zip(losses["single"], losses[mode]):
                assert abs(a - b) < 1e-3 * max(1.0, abs(a)), losses
    finally:
        if created:
            dist.destroy_process_group()


def test_split_graphs_keep_their_order_when_the_host_waits_each_step():
    """Regression: with the step captured as several hipGraphs (A1 / A2 / A3 / B around the gradient exchange) and the host
    synchronising every step (so each graph is launched while its predecessor is still RUNNING rather than queued),
    launches on the NULL stream lost their ordering -- the optimizer read half-written gradients and the loss blew up
    within ~10 steps at this size.  runtime.Context.init_device therefore moves all work to a created stream.  bf16
    ResNet-50 at 256x256 (the size it showed at), dropout off, multi-graph engine vs the single-graph engine."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        g = torch.Generator().manual_seed(3)
        B, S = 8, 256
        x = torch.randn(B, 3, S, S, generator=g)
        labels = torch.randint(0, 6, (B, S, S), generator=g)
        traces = {}
        for mode in ("single", "split"):
            torch.manual_seed(7)
            cfg = make_config("resnet50", iters=1000)
            model = get_model(cfg)
            model.to_hip("cuda:0", BF16, seed=11)
            model.set_dropout(0.0)
            assert torch.cuda.current_stream().cuda_stream != 0, "the runtime must not run on the NULL stream"
            opt = get_optimizer(model, get_scheduler(cfg), cfg)
            eng = TrainEngine(model, opt, get_loss_function(cfg), 1, use_graph=True, warmup_eager=1, two_phase=(mode == "split"))
            xs, ls = x.cuda(), labels.cuda()
            tr = []
            for _ in range(24):
                tr.append(eng.step(xs, ls).item())          # .item(): the host waits for the step, then launches the next
            traces[mode] = tr
            assert (eng.graph_a2 is not None) == (mode == "split")
        for a, b in zip(traces["single"], traces["split"]):
            assert b == b and abs(a - b) < 0.02 * max(1.0, abs(a)), traces
        assert traces["split"][-1] < traces["split"][0]
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("backbone", ["resnet18", "resnet50"])
def test_early_exchange_ranges_are_final_at_the_split(backbone):
    """The early gradient exchange (engine.py) all-reduces `early_ranges` of the flat gradient buffer while the second
    part of backward still runs.  That is only correct if the second part never writes them: run the two parts eagerly
    and compare the buffer before / after -- early ranges bit-identical, late ranges zero before and filled after,
    the two sets tiling [0, n_train) exactly."""
    g = torch.Generator().manual_seed(5)
    B, S = 2, 64
    x = torch.randn(B, 3, S, S, generator=g)
    labels = torch.randint(0, 6, (B, S, S), generator=g)
    ref, model = build_pair(backbone, x)
    cfg = make_config(backbone, iters=100)
    opt = get_optimizer(model, get_scheduler(cfg), cfg)
    eng = TrainEngine(model, opt, get_loss_function(cfg), 1, use_graph=False)
    early, late = model.store.split_ranges(model.late_grad_prefixes)
    cover = sorted(early + late)
    assert cover[0][0] == 0 and cover[-1][1] == model.store.n_train
    assert all(a[1] == b[0] for a, b in zip(cover, cover[1:])) and late
    model.train()
    seg_ranges = model.store.segment_ranges(model.grad_segment_prefixes)
    assert seg_ranges[0] == early and len(seg_ranges) == 3
    _, rest = eng._fwd_bwd(x.cuda(), labels.cuda(), split=True)
    assert len(rest) == 2
    torch.cuda.synchronize()
    snaps = [model.store.grad.clone()]
    for seg in rest:
        seg()
        torch.cuda.synchronize()
        snaps.append(model.store.grad.clone())
    for a, e in late:
        assert not snaps[0][a:e].any(), "a late-range gradient was written before the first mark"
    assert sum(float(snaps[0][a:e].abs().sum()) for a, e in early) > 0
    # ranges_i are final after segment i: later segments leave them bit-identical; ranges_j (j > i) are still zero
    for i, ranges in enumerate(seg_ranges):
        for a, e in ranges:
            for later in snaps[i + 1:]:
                assert torch.equal(snaps[i][a:e], later[a:e]), "segment %d's gradients changed afterwards" % i
            for earlier in snaps[:i]:
                assert not earlier[a:e].any(), "segment %d's gradients were written early" % i
    after = snaps[-1]
    names = [n for n in model.store.train_order if n.startswith(model.late_grad_prefixes) and n.endswith("weight")]
    for n in names:
        a, cnt = model.store.views[n]
        assert after[a:a + cnt].abs().sum() > 0, n


def test_bf16_path_is_sane():
    """bf16 storage / fp32 accumulate.  The random-initialised ResNet-50 amplifies bf16 rounding strongly (c4 is ~50 %
    off in relative norm for ANY bf16 implementation), so the yardstick is torch's own CPU bf16 autocast run of the
    oracle backbone: the HIP bf16 features must be no further from fp32 than 1.3x that, and training must reduce the loss."""
    from emrt_amd import functional as Fn
    g = torch.Generator().manual_seed(13)
    B, S = 2, 128
    x = torch.randn(B, 3, S, S, generator=g)
    labels = torch.randint(0, 6, (B, S, S), generator=g)
    ref, model = build_pair("resnet50", x, BF16)
    ref.train()
    model.train()
    with torch.no_grad():
        f32 = ref.backbone(x)
        with torch.autocast("cpu", dtype=torch.bfloat16):
            fbf = ref.backbone(x)
    c = ctx()
    c.training, c.tape = True, None
    from emrt_amd.src.models.emrt import IMAGE_CHANNELS
    feats = model.backbone(Fn.nchw_to_nhwc(x.cuda(), c_out=IMAGE_CHANNELS))
    for name, a, b, h in zip(("c1", "c2", "c3", "c4"), f32, fbf, feats):
        h = h.float().cpu().permute(0, 3, 1, 2)
        e_torch = ((a - b.float()).norm() / a.norm()).item()
        e_hip = ((a - h).norm() / a.norm()).item()
        print("%s: bf16 rel err vs fp32 oracle: HIP %.4f, torch CPU autocast %.4f" % (name, e_hip, e_torch))
        assert e_hip <= 1.3 * e_torch + 0.01, (name, e_hip, e_torch)
    got = model(x.cuda())[0].cpu()
    assert torch.isfinite(got).all()
    cfg = make_config("resnet50", iters=100)
    opt = get_optimizer(model, get_scheduler(cfg), cfg)
    eng = TrainEngine(model, opt, get_loss_function(cfg), 1, use_graph=False)
    l0 = eng.step(x.cuda(), labels.cuda()).item()
    for _ in range(5):
        l1 = eng.step(x.cuda(), labels.cuda()).item()
    assert l1 == l1 and l1 < l0, (l0, l1)


@pytest.mark.parametrize("ih,iw", [(96, 128), (90, 102)])      # (90, 102): window origins / width off the 4-pixel grid -> element-wise accumulate kernel
def test_sliding_window_inference_and_metrics(ih, iw):
    from emrt_amd.src.api import infer
    from emrt_amd.src.utils import metrics
    from oracle import infer_ref
    g = torch.Generator().manual_seed(14)
    x = torch.randn(1, 3, 64, 64, generator=g)
    ref, model = build_pair("resnet18", x)
    ref.eval()
    model.eval()
    img = torch.randn(3, ih, iw, generator=g)
    with torch.no_grad():
        want = infer_ref.slide_inference(ref, [img], (64, 64), (32, 32), 6)[0]
    got = infer.slide_inference(model, [img.cuda()], (64, 64), (32, 32), 6)[0].cpu()
    assert (got - want).abs().max().item() < 2e-3
    pred = infer.ss_inference(model, [img.cuda()], [(ih, iw)], True, 64, (32, 32), (64, 64), 6)[0]
    assert pred.dtype == torch.int32 and tuple(pred.shape) == (1, 1, ih, iw)
    assert_argmax_match(got, want, tol=2e-3)
    assert torch.equal(pred.cpu()[0, 0], got.argmax(1)[0].to(torch.int32))
    lab = torch.randint(0, 6, (ih, iw), generator=g)
    lab[:4] = 255
    a = metrics.calculate_area(pred, lab.cuda(), 6, 255)
    b = infer_ref.calculate_area(pred.cpu().numpy(), lab.numpy(), 6, 255)     # same predictions -> identical counts
    for u, v in zip(a, b):
        assert u.cpu().tolist() == v.tolist()


def test_ss_inference_resizes_logits_to_ori_shape_on_the_hip_path():
    """infer.py:146-150: when the network output size differs from ori_shape the logits are resized (bilinear, align_corners False)
    before the argmax; the HIP resize kernel against torch's F.interpolate on the same logits, up- and down-scaling, odd sizes."""
    from emrt_amd.src.api import infer
    g = torch.Generator().manual_seed(16)
    x = torch.randn(1, 3, 64, 64, generator=g)
    ref, model = build_pair("resnet18", x)
    model.eval()
    img = torch.randn(3, 64, 64, generator=g)
    logits = model(img.unsqueeze(0).cuda())[0]
    for shape in ((97, 75), (40, 52), (64, 64)):
        got = infer._resize_nchw_f32(logits, *shape).cpu()
        want = torch.nn.functional.interpolate(logits.cpu(), shape, mode="bilinear", align_corners=False)
        assert (got - want).abs().max().item() < 1e-5, shape
        pred = infer.ss_inference(model, [img.cuda()], [shape], False, 64, (32, 32), (64, 64), 6)[0]
        assert pred.dtype == torch.int32 and tuple(pred.shape) == (1, 1) + shape
        agree = (pred.cpu()[0, 0] == want.argmax(1)[0].to(torch.int32)).float().mean().item()
        assert agree > 0.999, (shape, agree)      # (a 1e-5 logit difference can flip an exact tie)


def test_large_tile_train_step_512_bf16():
    """BASELINE configs[2]: LoveDA 512x512, 7 classes, batch 4 -- the large-tile attention path (Lv = 5376: the MSDA
    scatter cuts level 0 into four LDS ranges, the forward takes the global-gather kernel).  bf16 fwd + bwd + SGD steps
    through the hipGraph engine: finite, and the loss decreases on a repeated batch."""
    g = torch.Generator().manual_seed(41)
    B, S, ncls = 4, 512, 7
    x = torch.randn(B, 3, S, S, generator=g).cuda()
    labels = torch.randint(0, ncls, (B, S, S), generator=g)
    labels[torch.rand(B, S, S, generator=g) < 0.02] = 255
    labels = labels.cuda()
    torch.manual_seed(7)
    cfg = make_config("resnet50", iters=100, ncls=ncls)
    model = get_model(cfg)
    model.to_hip("cuda:0", BF16)
    opt = get_optimizer(model, get_scheduler(cfg), cfg)
    eng = TrainEngine(model, opt, get_loss_function(cfg), 1, use_graph=True, warmup_eager=1)
    losses = [eng.step(x, labels).item() for _ in range(5)]
    assert all(l == l and abs(l) < 1e4 for l in losses), losses
    assert losses[-1] < losses[0], losses


def test_sliding_window_1024_resnet50():
    """BASELINE configs[4] shape: one 1024x1024 image, crop 256, stride 256 -> 16 windows evaluated as ONE batch of 16,
    accumulated / normalised / argmax-ed by the HIP glue kernels; fp32 against the oracle's slide_inference.  Stride 192
    (25 overlapping windows) exercises the count map against the same kernels' non-overlapping result on the interior."""
    from emrt_amd.src.api import infer
    from oracle import infer_ref
    g = torch.Generator().manual_seed(15)
    x = torch.randn(2, 3, 256, 256, generator=g)
    ref, model = build_pair("resnet50", x)
    ref.eval()
    model.eval()
    img = torch.randn(3, 1024, 1024, generator=g)
    with torch.no_grad():
        want = infer_ref.slide_inference(ref, [img], (256, 256), (256, 256), 6)[0]
    got = infer.slide_inference(model, [img.cuda()], (256, 256), (256, 256), 6)[0].cpu()
    assert tuple(got.shape) == (1, 6, 1024, 1024)
    assert (got - want).abs().max().item() < 2e-3
    assert_argmax_match(got, want, tol=2e-3)
    pred = infer.ss_inference(model, [img.cuda()], [(1024, 1024)], True, 1024, (256, 256), (256, 256), 6)[0]
    assert pred.dtype == torch.int32 and tuple(pred.shape) == (1, 1, 1024, 1024)
    assert torch.equal(pred.cpu()[0, 0], got.argmax(1)[0].to(torch.int32))
    got192 = infer.slide_inference(model, [img.cuda()], (256, 256), (192, 192), 6)[0].cpu()
    assert torch.isfinite(got192).all()                       # every pixel covered at least once
    assert (got192[..., :192, :192] - got[..., :192, :192]).abs().max().item() < 5e-4   # singly-covered corner: same window (other batch size => other GEMM tiling / summation order)


def test_multi_scale_flip_inference_matches_oracle():
    """infer.ms_inference: scales + horizontal flip, softmax-averaged; HIP glue kernels against the
    oracle's restatement of infer.py:160-260 (including its cumulative-resize quirk)."""
    from emrt_amd.src.api import infer
    from oracle import infer_ref
    g = torch.Generator().manual_seed(16)
    x = torch.randn(1, 3, 64, 64, generator=g)
    ref, model = build_pair("resnet18", x)
    ref.eval()
    model.eval()
    img = torch.randn(3, 96, 128, generator=g)
    scales = [0.75, 1.0, 1.25]
    with torch.no_grad():
        want_prob, want_pred = infer_ref.ms_inference(ref, img, (96, 128), (32, 32), (64, 64), 6, scales, True)
    pred = infer.ms_inference(model, [img.cuda()], (96, 128), True, 64, (32, 32), (64, 64), 6, scales=scales, flip_horizontal=True)
    assert pred.dtype == torch.int32 and tuple(pred.shape) == (1, 1, 96, 128)
    # decisions may only differ where the summed probabilities are a near-tie
    top2 = want_prob.topk(2, dim=1).values
    margin = (top2[:, 0] - top2[:, 1])[0]
    bad = (pred.cpu()[0, 0] != want_pred[0, 0]) & (margin > 2e-3)
    assert not bad.any(), int(bad.sum())


@pytest.mark.parametrize("B,H,W", [(1, 96, 160), (3, 160, 64)])
def test_non_square_tiles_match_oracle(B, H, W):
    """The reference takes any H, W that are multiples of 32 (paddle_EMRT.py:293); every tile the configs use is square, so a
    swapped height / width in a kernel's geometry would go unseen.  Ragged case: odd batch, H != W, three different level
    shapes (12x20 / 6x10 / 3x5 ...), a quarter of the labels ignored: eval logits, then train-mode loss and the whole
    gradient against the float64 oracle."""
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, 3, H, W, generator=g)
    labels = torch.randint(0, 6, (B, H, W), generator=g)
    labels[torch.rand(B, H, W, generator=g) < 0.25] = 255
    ref, model = build_pair("resnet18", x, perturb=True)
    ref.double().eval()
    model.eval()
    with torch.no_grad():
        want = ref(x.double())
    got = model(x.cuda())
    for a, b in zip(got, want):
        assert a.shape == (B, 6, H, W)
        assert (a.cpu() - b.float()).abs().max().item() < 1e-3
    ref.train()
    model.train()
    out_r = ref(x.double())
    loss_r = train_ref.mix_softmax_ce_loss(out_r, labels)
    loss_r.backward()
    model.clear_gradients()
    out = model(x.cuda())
    loss = get_loss_function(make_config("resnet18"))(out, labels.cuda())
    loss.backward()
    torch.cuda.synchronize()
    assert (out[0].cpu() - out_r[0].detach().float()).abs().max().item() < 1e-3
    assert abs(loss.item() - loss_r.item()) < 1e-4 * max(1.0, abs(loss_r.item()))
    refp = dict(ref.named_parameters())
    num = den = dot = 0.0
    for n, p in model.named_parameters():
        gr = refp[n].grad
        if gr is None:
            continue
        gr, gg = gr.double(), p.grad.cpu().double()
        num += float(((gg - gr) ** 2).sum())
        den += float((gr ** 2).sum())
        dot += float((gg * gr).sum())
    gn = sum(float((p.grad.cpu().double() ** 2).sum()) for n, p in model.named_parameters() if refp[n].grad is not None)
    cos = dot / (den ** 0.5 * gn ** 0.5)
    print("non-square %dx%dx%d: whole-gradient rel err %.3g, cosine %.6f" % (B, H, W, (num / den) ** 0.5, cos))
    assert cos > 0.999 and (num / den) ** 0.5 < 0.05


def large_tile_train_step_512_case():      # run from tests/test_gpu_model_traj.py
    """BASELINE configs[2] geometry (LoveDA: 512x512 tiles, 7 classes, Lv = 5376) in fp32 against the fp32 CPU oracle, one train-mode
    forward + loss + backward at batch 2: the large-map paths (global-gather MSDA forward and gradient kernels, the |dout| pre-pass,
    the scatter's four LDS ranges on level 0, 128x128 conv tiles everywhere) at model level, not only kernel by kernel."""
    g = torch.Generator().manual_seed(19)
    B, S, ncls = 2, 512, 7
    x = torch.randn(B, 3, S, S, generator=g)
    labels = torch.randint(0, ncls, (B, S, S), generator=g)
    labels[torch.rand(B, S, S, generator=g) < 0.02] = 255
    ref, model = build_pair("resnet50", x, perturb=True, ncls=ncls)
    ref.train()
    out64 = float64_train_forward(ref, x)
    out_r = ref(x)
    loss_r = train_ref.mix_softmax_ce_loss(out_r, labels)
    loss_r.backward()
    model.train()
    model.clear_gradients()
    out = model(x.cuda())
    loss = get_loss_function(make_config("resnet50", ncls=ncls))(out, labels.cuda())
    loss.backward()
    torch.cuda.synchronize()
    assert_train_logits(out, out_r, out64, "512x512 batch 2")
    assert abs(loss.item() - loss_r.item()) < 2e-4 * max(1.0, abs(loss_r.item()))
    refp = dict(ref.named_parameters())
    dot = n_hip = n_ref = 0.0
    for n, p in model.named_parameters():
        gr = refp[n].grad
        if gr is None:
            continue
        gg, gr = p.grad.cpu().double(), gr.double()
        dot += float((gg * gr).sum())
        n_hip += float((gg * gg).sum())
        n_ref += float((gr * gr).sum())
    cos = dot / (n_hip ** 0.5 * n_ref ** 0.5)
    print("512x512 gradient: cosine %.6f, norm ratio %.5f" % (cos, (n_hip / n_ref) ** 0.5))
    assert cos > 0.9995 and abs((n_hip / n_ref) ** 0.5 - 1.0) < 1e-2
