"""-m gpu: float16 (EMRT_DTYPE_F16 = 2), the arithmetic type of BASELINE configs[4] (Vaihingen 1024x1024 sliding-window
inference in fp16; caller: src/api/infer.py:22-80,130-155).  fp16 storage, fp32 accumulation, forward entry points only."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from emrt_amd import _lib                       # noqa: E402
from emrt_amd import functional as Fn          # noqa: E402
from emrt_amd import nn as hnn                  # noqa: E402
from emrt_amd.runtime import ctx, F32, F16, Tape   # noqa: E402
from tests.hip_utils import init, dev_map, host_map, dev, host, rnd, Holder   # noqa: E402
from tests.test_gpu_kernels import _msda_ref   # noqa: E402

HALF_EPS = 2.0 ** -11          # fp16 unit round-off: one rounding of an O(1) result is <= 4.9e-4 relative


@pytest.mark.parametrize("case", [(2, 32, 32, 256, 256, 3, 1, 1, True), (2, 16, 16, 1024, 256, 1, 1, 0, False), (2, 64, 64, 3, 64, 7, 2, 3, False),
                                  (16, 32, 32, 1536, 512, 3, 1, 1, False)],       # last: cls_psp at the cfg5 batch (128x128 tile, K = 13 824)
                         ids=["3x3", "1x1", "stem-7x7", "cls_psp-b16"])
def test_conv_fwd_fp16(case):
    N, H, W, Cin, Cout, k, stride, pad, bias = case
    init(F16)
    g = torch.Generator().manual_seed(51)
    x = rnd(torch.randn(N, Cin, H, W, generator=g))
    conv = hnn.Conv2D(Cin, Cout, k, stride, pad, bias=bias)
    with torch.no_grad():
        conv.weight.copy_(rnd(torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)))
        if bias:
            conv.bias.copy_(torch.randn(Cout, generator=g))
    w, b = conv.weight.detach().clone(), (conv.bias.detach().clone() if bias else None)
    Holder(conv=conv).place()
    want = F.conv2d(x, w, b, stride=stride, padding=pad)
    got = host_map(conv(dev_map(x)))
    rel = ((got - want).norm() / want.norm()).item()
    assert rel < 1.5 * HALF_EPS, rel                      # a uniform rounding error is eps / sqrt(3) relative in L2
    assert (got - want).abs().max().item() < 4e-3 + 2 * HALF_EPS * want.abs().max().item()


@pytest.mark.parametrize("Lq", [None, 110], ids=["encoder-b16", "decoder-b16"])
def test_msda_fwd_fp16_cfg5_shape(Lq):
    """B = 16 windows of 256x256: Lv = 1344 per window.  The encoder call takes the LDS-staged kernel; both kernels must
    agree bit for bit and match the oracle's core function (utils.py:64-97) on fp16-rounded values."""
    init(F16)
    L_ = _lib.lib()
    g = torch.Generator().manual_seed(52)
    M, L, Pn, B = 8, 3, 6, 16
    shapes = [(32, 32), (16, 16), (8, 8)]
    Lv = sum(h * w for h, w in shapes)
    nq = Lq or Lv
    tp = M * L * Pn
    value = rnd(torch.randn(B, Lv, M * 32, generator=g))
    offw = torch.cat([torch.randn(B, nq, 2 * tp, generator=g) * 2.5, torch.randn(B, nq, tp, generator=g)], -1)
    ref = torch.rand(1, nq, 1, 2, generator=g)
    want = _msda_ref(value, offw, ref, shapes, M, L, Pn)
    vd, od, rd = dev(value), dev(offw, torch.float32), dev(ref, torch.float32)
    y = Fn.msda(vd, od, rd, shapes, M, Pn)
    rel = ((host(y) - want).norm() / want.norm()).item()
    assert rel < 1.5 * HALF_EPS, rel
    old = L_.set_tuning("msda_fwd_global", 1)
    try:
        y2 = Fn.msda(vd, od, rd, shapes, M, Pn)
    finally:
        L_.set_tuning("msda_fwd_global", old)
    assert torch.equal(y, y2)


def test_fp16_is_inference_only():
    """Backward / training entry points refuse dtype 2 with an error instead of running an untested path."""
    c = init(F16)
    x = dev(torch.randn(2, 8, 8, 64))
    conv = hnn.Conv2D(64, 64, 3, 1, 1, bias=False)
    Holder(conv=conv).place()
    tape = Tape()
    c.tape = tape
    y = conv(x)
    c.tape = None
    tape.add_grad(y, dev(torch.randn(2, 8, 8, 64)))
    with pytest.raises(_lib.EmrtHipError, match="inference-only"):
        tape.backward()


def test_cfg5_sliding_window_1024_fp16_vs_fp32_oracle():
    """BASELINE configs[4]: one 1024x1024 image, CROP_SIZE 256, STRIDE_SIZE 256 -> 16 windows as ONE batch of 16 through the
    fp16 model; logits against the fp32 CPU oracle's slide_inference (infer.py:22-80) and against the same model in fp32 on
    the device.  Weights: residual branches down-weighted (tests/test_gpu_model.py: condition_residual_branches) -- at
    random init the network is chaotic and ANY reduced precision decorrelates; stated bounds need a conditioned network."""
    from emrt_amd.src.api import infer
    from oracle import infer_ref
    from tests.test_gpu_model import build_pair
    g = torch.Generator().manual_seed(15)
    x = torch.randn(4, 3, 256, 256, generator=g)
    ref, model = build_pair("resnet50", x, dtype=F16, condition=0.1)
    ref.eval()
    model.eval()
    model.compute_aux_in_eval = False        # the aux head is discarded by every inference caller (infer.py:66)
    img = torch.randn(3, 1024, 1024, generator=g)
    with torch.no_grad():
        want = infer_ref.slide_inference(ref, [img], (256, 256), (256, 256), 6)[0]
    got = infer.slide_inference(model, [img.cuda()], (256, 256), (256, 256), 6)[0].cpu()
    assert tuple(got.shape) == (1, 6, 1024, 1024) and torch.isfinite(got).all()
    rel = ((got - want).norm() / want.norm()).item()
    agree = (got.argmax(1) == want.argmax(1)).float().mean().item()
    top2 = want.topk(2, dim=1).values
    decisive = (top2[:, 0] - top2[:, 1]) > 0.05 * (want.max() - want.min())
    agree_dec = (got.argmax(1) == want.argmax(1))[decisive].float().mean().item()
    print("fp16 1024x1024 (16 windows, one batch): logits rel L2 %.5f, max |diff| %.4f (|ref| max %.3f), argmax agreement %.5f, among %d decisive pixels %.6f" % (
        rel, (got - want).abs().max().item(), want.abs().max().item(), agree, int(decisive.sum()), agree_dec))
    assert rel < 2e-2, rel                   # fp16 keeps 11 significant bits: ~8x below the bf16 bound of the training path
    assert agree > 0.985 and agree_dec > 0.9995, (agree, agree_dec)
    pred = infer.ss_inference(model, [img.cuda()], [(1024, 1024)], True, 1024, (256, 256), (256, 256), 6)[0]
    assert pred.dtype == torch.int32 and tuple(pred.shape) == (1, 1, 1024, 1024)
    assert torch.equal(pred.cpu()[0, 0], got.argmax(1)[0].to(torch.int32))
