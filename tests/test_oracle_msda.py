"""Pins the oracle's deformable-attention core (the reference has no tests: SURVEY.md section 8c).

torch restatement of utils.py:64-97  ==  numpy-f64 direct bilinear  ==  python scalar loops
==  transformers' MultiScaleDeformableAttention (independent third-party code, same algebra)
+ the analytic known-answer of MSDeformableAttention._reset_parameters (t_e_d.py:46-63).
"""
import math

import numpy as np
import pytest
import torch

from oracle.emrt_torch import MSDeformableAttention, TransformerEncoder, deformable_attention_core_func
from oracle.msda_numpy import msda_core_f64, msda_core_loops

SHAPES = [(8, 8), (4, 4), (2, 2)]


def _rand_inputs(B=2, Lq=37, M=8, D=32, L=3, P=6, seed=0, oob=True):
    g = torch.Generator().manual_seed(seed)
    Lv = sum(h * w for h, w in SHAPES)
    value = torch.randn(B, Lv, M, D, generator=g)
    loc = torch.rand(B, Lq, M, L, P, 2, generator=g)
    if oob:  # ~20% out of bounds, as in SURVEY 3.3
        loc = loc * 1.4 - 0.2
    aw = torch.softmax(torch.randn(B, Lq, M, L * P, generator=g), -1).reshape(B, Lq, M, L, P)
    return value, loc, aw


def test_torch_vs_numpy_f64():
    value, loc, aw = _rand_inputs()
    a = deformable_attention_core_func(value, SHAPES, loc, aw).numpy()
    b = msda_core_f64(value.numpy(), SHAPES, loc.numpy(), aw.numpy())
    assert np.abs(a - b).max() < 2e-6


def test_numpy_vectorised_vs_loops():
    value, loc, aw = _rand_inputs(B=1, Lq=5, M=2, D=4, seed=3)
    b = msda_core_f64(value.numpy(), SHAPES, loc.numpy(), aw.numpy())
    c = msda_core_loops(value.numpy(), SHAPES, loc.numpy(), aw.numpy())
    assert np.abs(b - c).max() < 1e-12


def test_vs_transformers_msda():
    tr = pytest.importorskip("transformers.models.deformable_detr.modeling_deformable_detr")
    value, loc, aw = _rand_inputs(seed=5)
    ref = tr.MultiScaleDeformableAttention().forward(value, torch.tensor(SHAPES), SHAPES, None, loc, aw, 64)
    a = deformable_attention_core_func(value, SHAPES, loc, aw)
    assert (a - ref).abs().max() < 2e-6


def test_zero_padding_constant_value():
    """constant value map => output = constant * in-bounds corner mass."""
    B, Lq, M, D, L, P = 1, 3, 1, 2, 3, 6
    Lv = sum(h * w for h, w in SHAPES)
    value = torch.full((B, Lv, M, D), 3.0)
    loc = torch.full((B, Lq, M, L, P, 2), 0.5)     # well inside every level
    aw = torch.full((B, Lq, M, L, P), 1.0 / (L * P))
    out = deformable_attention_core_func(value, SHAPES, loc, aw)
    assert torch.allclose(out, torch.full_like(out, 3.0), atol=1e-6)
    loc[...] = -1.0                                   # entirely outside => 0
    assert deformable_attention_core_func(value, SHAPES, loc, aw).abs().max() == 0


def test_reset_parameters_known_answer():
    """At init: offsets weight 0, bias = compass direction * point index, attention uniform 1/18."""
    torch.manual_seed(0)
    m = MSDeformableAttention(256, 8, 3, 6)
    assert m.sampling_offsets.weight.abs().max() == 0
    b = m.sampling_offsets.bias.reshape(8, 3, 6, 2)
    th = torch.arange(8, dtype=torch.float32) * (2 * math.pi / 8)
    d = torch.stack([th.cos(), th.sin()], -1)
    d = d / d.abs().max(-1, keepdim=True)[0]
    for k in range(6):
        assert torch.equal(b[:, :, k], (d * (k + 1))[:, None, :].expand(8, 3, 2))
    assert float(b[7, 0, 0, 1]) != -1.0 and abs(float(b[7, 0, 0, 1]) + 1.0) < 2e-6   # head 7 is not exactly (1,-1)
    # in the query's own level the samples land on pixel centres: value = arange map => exact integer pick
    B = 1
    ref = TransformerEncoder.get_reference_points(SHAPES, torch.ones(B, 3, 2))
    Lv = ref.shape[1]
    with torch.no_grad():
        m.value_proj.weight.copy_(torch.eye(256)); m.output_proj.weight.copy_(torch.eye(256))
        src = torch.zeros(B, Lv, 256)
        src[:, :64, :] = torch.arange(64, dtype=torch.float32).reshape(1, 64, 1)      # level 0 only (8x8)
        out = m(src, ref, src, SHAPES)
    # query at level-0 pixel (y=3,x=3), head 0 samples (3, 3+k) for k=1..6 in level 0: x in-range for k<=4
    q = 3 * 8 + 3
    lvl0 = sum(float(3 * 8 + 3 + k) for k in range(1, 5)) / 18.0
    # levels 1,2 hold zeros => only level 0 contributes
    assert abs(float(out[0, q, 0]) - lvl0) < 1e-5
