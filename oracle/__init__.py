"""CPU oracle for the EMRT per-tile hot path -- TEST INFRASTRUCTURE ONLY.

This package restates, in plain torch-CPU fp32 (and numpy float64 for the
deformable-attention core), the algorithm of the reference's hot path
(peach-xiao/EMRT, `semantic_segmentation/src/models/paddle_EMRT.py` and the
files it calls).  Every function cites the reference file:line it follows.

PARITY UNPINNED: the reference is PaddlePaddle dygraph code; Paddle is not
installed in the build container or on the GPU box and the reference ships no
tests, golden vectors or fixtures.  The
restatement is therefore pinned only by
  (1) a numpy-float64 direct four-corner bilinear implementation of
      `deformable_attention_core_func` (oracle/msda_numpy.py),
  (2) the independent `transformers` MultiScaleDeformableAttention (same
      algebra as reference utils.py:64-97), used in tests only,
  (3) the analytic known-answer state of `_reset_parameters`
      (transformer_encoder_decoder.py:46-63),
  (4) algebraic identities (see tests/test_oracle_*.py).
The Paddle->torch semantic assumptions are listed in DESIGN.md.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg
may import this package.  The product (`emrt_amd/`) never does.
"""
