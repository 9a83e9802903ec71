"""The `.pdparams` importer (emrt_amd/src/utils/checkpoint.py; reference: src/utils/checkpoint.py:21-93,
backbones/paddle_vision_resnet.py:276-287) against files it did NOT write: tests/golden/make_pdparams_fixture.py builds them
in paddle.save's layout from the oracle, deciding the transposed tensors by module type, without importing emrt_amd."""
import os
import pickle
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_pdparams_fixture as fx                                          # noqa: E402

from emrt_amd.src.utils import checkpoint as ck                            # noqa: E402
from emrt_amd.src.models.emrt import EMRT                                  # noqa: E402


def test_whole_model_file_written_by_the_independent_writer(tmp_path):
    ref = fx.seeded_oracle("resnet18")
    path = str(tmp_path / "emrt.pdparams")
    raw = fx.write_fixture(path, ref)
    with open(path, "rb") as f:
        disk = pickle.load(f)
    # the file really is in Paddle's layout: protocol-2 pickle, Linear [in, out], attention in-projection [E, 3E], the name table
    assert "StructuredToParameterName@@" in disk and isinstance(disk["StructuredToParameterName@@"], dict)
    assert disk["model.encoder.layers.0.linear1.weight"].shape == (256, 1024)
    assert disk["model.decoder.layers.0.self_attn.in_proj_weight"].shape == (256, 768)
    assert disk["model.reference_points.weight"].shape == (256, 2) and disk["backbone.fc.weight"].shape[0] == 512
    assert "backbone.bn1._mean" in disk and "backbone.bn1._variance" in disk
    state = ck.paddle_to_torch_state(ck.load_pdparams(path))
    want = ref.state_dict()
    assert set(state) == set(want)                                           # the name table is skipped, nothing else is
    for k, v in want.items():
        assert tuple(state[k].shape) == tuple(v.shape), k
        assert torch.equal(state[k].to(v.dtype), v), k
    # ... and through the public entry point into the product model (CPU side: parameters only, no kernels)
    model = EMRT(num_classes=6, backbone="resnet18")
    n = ck.load_entire_model(model, path)
    assert n == len(model.state_dict()) == len(raw) - 1
    got = model.state_dict()
    for k, v in want.items():
        assert torch.equal(got[k], v.to(got[k].dtype)), k
    # the writer of this repo produces the same bytes per tensor as the independent one
    out = str(tmp_path / "ours.pdparams")
    ck.save_pdparams(model.state_dict(), out)
    with open(out, "rb") as f:
        ours = pickle.load(f)
    for k, v in raw.items():
        if k != "StructuredToParameterName@@":
            assert ours[k].shape == v.shape and np.array_equal(ours[k], v), k


def test_backbone_only_file_goes_under_the_backbone_prefix(tmp_path):
    """paddle.vision's ImageNet ResNet file: keys `conv1.weight`, `layer1.0.bn1._mean`, `fc.weight` [in, out] with no prefix."""
    ref = fx.seeded_oracle("resnet18", seed=5)
    path = str(tmp_path / "resnet18.pdparams")
    raw = fx.write_fixture(path, ref, keys_prefix_strip="backbone.")
    assert "conv1.weight" in raw and raw["fc.weight"].shape == (512, 1000) and not any(k.startswith("backbone.") for k in raw)
    model = EMRT(num_classes=6, backbone="resnet18")
    before = {k: v.clone() for k, v in model.state_dict().items()}
    n = ck.load_pretrained_model(model, path, prefix="backbone.")
    want = ref.state_dict()
    nb = sum(k.startswith("backbone.") for k in want)
    assert n == nb == len(raw) - 1
    for k, v in model.state_dict().items():
        if k.startswith("backbone."):
            assert torch.equal(v, want[k].to(v.dtype)), k
        else:
            assert torch.equal(v, before[k]), k                              # everything else untouched
