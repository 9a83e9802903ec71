"""Inference: BatchNorm folded into the producing convolution's epilogue (emrt_bn_fold + emrt_conv2d's out_scale).

The folded forward must (a) agree with the unfolded HIP forward and with the oracle's eval forward, (b) launch no emrt_bn_apply for
conv -> BatchNorm pairs, (c) see every change of the running statistics / affine parameters (the fold is recomputed at the top of
each eval forward, never cached across calls).  Reference semantics: paddle BatchNorm2D in eval mode (paddle_EMRT.py:252-304 run
under model.eval(), val.py:134).
"""
import pytest
import torch

from emrt_amd import _lib
from emrt_amd.runtime import ctx, F32, BF16, F16

from tests.test_gpu_model import build_pair

pytestmark = pytest.mark.gpu


def _eval_logits(model, x, fold):
    c = ctx()
    c.fold_eval_bn = fold
    try:
        L = _lib.lib()
        L.start_record()
        out = model(x)
        rec = L.stop_record()
    finally:
        c.fold_eval_bn = True
    names = [n for n, _ in rec]
    return out[0].float().cpu(), out[1].float().cpu(), names


@pytest.mark.parametrize("backbone,S", [("resnet18", 64), ("resnet50", 128)])
def test_folded_eval_forward_matches_unfolded_and_oracle(backbone, S):
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 3, S, S, generator=g)
    ref, model = build_pair(backbone, x)
    # running statistics that are not the identity map: one training-mode oracle pass moves them
    ref.train()
    with torch.no_grad():
        ref(x)
    model.load_state_dict(ref.state_dict())
    ref.eval()
    model.eval()
    with torch.no_grad():
        want = ref.double()(x.double())
    main_f, aux_f, names_f = _eval_logits(model, x.cuda(), True)
    main_u, aux_u, names_u = _eval_logits(model, x.cuda(), False)
    n_fold, n_unfold = names_f.count("emrt_bn_apply"), names_u.count("emrt_bn_apply")
    print("emrt_bn_apply launches: folded %d, unfolded %d; launches %d vs %d" % (n_fold, n_unfold, len(names_f), len(names_u)))
    assert names_f.count("emrt_bn_fold") == 1 and "emrt_bn_fold" not in names_u
    assert n_fold == 0 and n_unfold > 10
    for a, u, w in ((main_f, main_u, want[0].float()), (aux_f, aux_u, want[1].float())):
        assert (a - u).abs().max().item() < 2e-4          # same fp32 arithmetic up to one fused multiply-add per element
        assert (a - w).abs().max().item() < 1e-3          # north_star tolerance against the float64 oracle


def test_fold_follows_the_running_statistics():
    g = torch.Generator().manual_seed(8)
    x = torch.randn(1, 3, 64, 64, generator=g)
    ref, model = build_pair("resnet18", x)
    model.eval()
    a0, _, _ = _eval_logits(model, x.cuda(), True)
    bn = model.backbone.bn1
    bn._buffers["_mean"].add_(0.5)                         # direct edit of the device buffer, no state-dict load, no pack
    bn._buffers["_variance"].mul_(1.7)
    a1, _, _ = _eval_logits(model, x.cuda(), True)
    u1, _, _ = _eval_logits(model, x.cuda(), False)
    assert (a1 - a0).abs().max().item() > 1e-3            # the edit is visible ...
    assert (a1 - u1).abs().max().item() < 2e-4            # ... and equals what the unfolded BatchNorm computes


@pytest.mark.parametrize("dtype,tol", [(BF16, 0.06), (F16, 0.01)])
def test_folded_low_precision_forward_is_closer_to_fp32_than_unfolded(dtype, tol):
    """Folding removes one rounding (the pre-BatchNorm tensor is never written), so it must not be further from fp32."""
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 3, 128, 128, generator=g)
    ref, model = build_pair("resnet50", x, dtype=dtype, condition=0.1)
    ref.eval()
    with torch.no_grad():
        want = ref(x)[0]
    model.eval()
    a, _, _ = _eval_logits(model, x.cuda(), True)
    u, _, _ = _eval_logits(model, x.cuda(), False)
    ef = ((a - want).norm() / want.norm()).item()
    eu = ((u - want).norm() / want.norm()).item()
    print("rel L2 vs fp32 oracle: folded %.4g, unfolded %.4g" % (ef, eu))
    assert ef < tol and ef < 1.15 * eu
