"""Generates the committed golden vectors from the ORACLE (seeded, torch-CPU fp32 / numpy f64).

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz

The reference (PaddlePaddle) cannot run here and ships no fixtures, so these vectors pin the oracle
against accidental change and give the GPU tests size-independent, machine-independent expectations.  Fixtures are
data only: inputs and expected outputs."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle.emrt_torch import EMRT, MSDeformableAttention, TransformerEncoder, sine_position_embedding  # noqa: E402
from oracle.msda_numpy import msda_core_f64  # noqa: E402
from oracle import infer_ref, train_ref  # noqa: E402


def main():
    g = torch.Generator().manual_seed(20261002)
    shapes = [(8, 8), (4, 4), (2, 2)]
    Lv = sum(h * w for h, w in shapes)
    B, Lq, M, D, L, P = 2, 19, 8, 32, 3, 6
    value = torch.randn(B, Lv, M, D, generator=g)
    loc = torch.rand(B, Lq, M, L, P, 2, generator=g) * 1.3 - 0.15
    aw = torch.softmax(torch.randn(B, Lq, M, L * P, generator=g), -1).reshape(B, Lq, M, L, P)
    out = msda_core_f64(value.numpy(), shapes, loc.numpy(), aw.numpy())
    np.savez_compressed(os.path.join(HERE, "msda_core.npz"), value=value.numpy(), loc=loc.numpy(), aw=aw.numpy(),
                        shapes=np.array(shapes), out=out.astype(np.float32))

    pos = torch.cat([sine_position_embedding(torch.ones(1, h, w, dtype=torch.bool), 128).flatten(2).transpose(1, 2)[0] for h, w in shapes], 0)
    ref = TransformerEncoder.get_reference_points(shapes, torch.ones(1, 3, 2))[0]
    np.savez_compressed(os.path.join(HERE, "pos_ref.npz"), shapes=np.array(shapes), pos=pos.numpy().astype(np.float32), ref=ref.numpy())

    torch.manual_seed(7)
    m = MSDeformableAttention(256, 8, 3, 6)
    np.savez_compressed(os.path.join(HERE, "msda_init.npz"), offsets_bias=m.sampling_offsets.bias.detach().numpy())

    # window grid / metrics integer fixtures
    wins = infer_ref.window_grid(1024, 1024, (256, 256), (192, 192))
    lab = torch.randint(0, 6, (64, 64), generator=g).numpy()
    pred = torch.randint(0, 6, (64, 64), generator=g).numpy()
    lab[:3] = 255
    inter, pa, la = infer_ref.calculate_area(pred, lab, 6, 255)
    np.savez_compressed(os.path.join(HERE, "infer_metrics.npz"), wins=np.array(wins), pred=pred, lab=lab, inter=inter, pa=pa, la=la,
                        miou=infer_ref.mean_iou(inter, pa, la)[1], acc=infer_ref.accuracy(inter, pa, la)[0], kappa=infer_ref.kappa(inter, pa, la))

    # whole-model plumbing check (BASELINE config 1: ResNet-18 variant, 1x3x256x256, eval) -- strided logits sample + checksum
    torch.manual_seed(1234)
    model = EMRT(6, "resnet18").eval()
    x = torch.randn(1, 3, 256, 256, generator=torch.Generator().manual_seed(1234))
    with torch.no_grad():
        y, aux = model(x)
    np.savez_compressed(os.path.join(HERE, "emrt_r18_256.npz"), sample=y[0, :, ::32, ::32].numpy(), aux_sample=aux[0, :, ::32, ::32].numpy(),
                        mean=np.float32(y.mean().item()), std=np.float32(y.std().item()), n_params=np.int64(sum(p.numel() for p in model.parameters())))

    lrs = np.array([train_ref.poly_lr(t, 0.01, 0.0, 160000, 0.9) for t in (0, 1, 1000, 80000, 159999, 160000, 200000)])
    np.savez_compressed(os.path.join(HERE, "poly_lr.npz"), steps=np.array([0, 1, 1000, 80000, 159999, 160000, 200000]), lr=lrs)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
