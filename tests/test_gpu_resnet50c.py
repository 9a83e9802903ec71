"""-m gpu: the resnet50c backbone (deep 3x3 stem, dilated stages: reference src/models/backbones/resnet.py:61-234, selected by
MODEL.ENCODER.TYPE "resnet50c", paddle_EMRT.py:227-228) -- dilated convolution kernels and whole-model parity."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from emrt_amd import nn as hnn                  # noqa: E402
from emrt_amd.runtime import ctx, F32, BF16, Tape   # noqa: E402
from tests.hip_utils import init, dev_map, host_map, host, rnd, Holder, close   # noqa: E402
from tests.test_gpu_kernels import run_bwd   # noqa: E402


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("case", [(2, 16, 16, 64, 64, 3, 1, 2, 2), (2, 16, 16, 128, 64, 3, 1, 4, 4), (2, 17, 13, 64, 128, 3, 2, 2, 2), (1, 12, 12, 3, 32, 3, 1, 2, 2)],
                         ids=["d2", "d4", "d2-stride2-ragged", "d2-cin3"])
def test_dilated_conv_fwd_dgrad_wgrad(dtype, case):
    N, H, W, Cin, Cout, k, stride, pad, dil = case
    c = init(dtype)
    g = torch.Generator().manual_seed(61)
    x = rnd(torch.randn(N, Cin, H, W, generator=g))
    conv = hnn.Conv2D(Cin, Cout, k, stride, pad, bias=False, dilation=dil)
    with torch.no_grad():
        conv.weight.copy_(rnd(torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)))
    w_ref = conv.weight.detach().clone()
    Holder(conv=conv).place()
    xr, wr = x.clone().requires_grad_(True), w_ref.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, stride=stride, padding=pad, dilation=dil)
    dy = rnd(torch.randn(yr.shape, generator=g))
    yr.backward(dy)
    xd = dev_map(x)
    tape = Tape()
    c.tape = tape
    y = conv(xd)
    c.tape = None
    tape.watch(xd)
    close("dilated conv fwd", host_map(y), yr.detach(), dtype)
    dx, = run_bwd(tape, [(y, dev_map(dy))], [xd])
    kscale = math.sqrt(Cout * k * k / (stride * stride))
    close("dilated conv dgrad", host_map(dx), xr.grad, dtype, kscale * (1.0 if dtype == F32 else 0.3))
    wscale = math.sqrt(N * yr.shape[2] * yr.shape[3])
    close("dilated conv wgrad", host(conv.weight.grad), wr.grad, dtype, wscale * (1.0 if dtype == F32 else 0.3))


@pytest.mark.parametrize("output_stride", [32, 16])
def test_resnet50c_model_matches_oracle(output_stride):
    """Whole EMRT with the resnet50c backbone through get_model(config): eval logits within 1e-3 of the float64 oracle, argmax
    masks equal up to sub-tolerance ties, and one train-mode forward + backward (loss, whole-gradient cosine).  OUTPUT_STRIDE
    16 puts dilation 2 in layer4 and gives the transformer the level shapes (S, S/2, S/2)."""
    import argparse
    from emrt_amd.config import get_config, update_config
    from emrt_amd.src.models import get_model
    from emrt_amd.src.models.losses import get_loss_function
    from oracle.emrt_torch import EMRT as OracleEMRT, BatchNorm2D as OBN
    from oracle import train_ref
    from tests.test_gpu_model import CFG, oracle_no_dropout, perturb_sampling_offsets, assert_argmax_match
    g = torch.Generator().manual_seed(71)
    B, S = 2, 128
    x = torch.randn(B, 3, S, S, generator=g)
    labels = torch.randint(0, 6, (B, S, S), generator=g)
    torch.manual_seed(0)
    ref = OracleEMRT(6, "resnet50c", output_stride=output_stride)
    oracle_no_dropout(ref)
    for mod in ref.modules():
        if isinstance(mod, OBN):
            mod.momentum = 0.0
    ref.train()
    with torch.no_grad():
        ref(x)
    for mod in ref.modules():
        if isinstance(mod, OBN):
            mod.momentum = 0.9
    perturb_sampling_offsets(ref)
    cfg = update_config(get_config(), argparse.Namespace(cfg=CFG))
    cfg.MODEL.ENCODER.TYPE = "resnet50c"
    cfg.MODEL.OUTPUT_STRIDE = output_stride
    model = get_model(cfg)
    model.load_state_dict(ref.state_dict())
    model.to_hip("cuda:0", F32)
    model.set_dropout(0.0)
    ref.eval()
    model.eval()
    got = model(x.cuda())
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    with torch.no_grad():
        ref.double()
        want = [t.float() for t in ref(x.double())]
    for name, a, b in (("main", got[0].cpu(), want[0]), ("aux", got[1].cpu(), want[1])):
        err = (a - b).abs().max().item()
        print("resnet50c OS%d %s logits: max |diff| vs float64 oracle %.3g" % (output_stride, name, err))
        assert err < 1e-3, (name, err)
    assert_argmax_match(got[0].cpu(), want[0])
    ref.float()
    ref.load_state_dict(sd)
    ref.train()
    loss_r = train_ref.mix_softmax_ce_loss(ref(x), labels)
    loss_r.backward()
    model.train()
    model.clear_gradients()
    loss = get_loss_function(cfg)(model(x.cuda()), labels.cuda())
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - loss_r.item()) < 2e-4 * max(1.0, abs(loss_r.item()))
    refp = dict(ref.named_parameters())
    dot = n1 = n2 = 0.0
    for n, p in model.named_parameters():
        gr = refp[n].grad
        if gr is None:
            continue
        gg, gr = p.grad.cpu().double(), gr.double()
        dot += float((gg * gr).sum()); n1 += float((gg * gg).sum()); n2 += float((gr * gr).sum())
    cos = dot / (n1 ** 0.5 * n2 ** 0.5)
    print("resnet50c OS%d gradient cosine %.6f, norm ratio %.5f" % (output_stride, cos, (n1 / n2) ** 0.5))
    assert cos > 0.999 and abs((n1 / n2) ** 0.5 - 1.0) < 2e-2
