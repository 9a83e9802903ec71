"""`.pdparams` round trip: Paddle layouts <-> this repo's state-dict layouts, no Paddle needed."""
import pickle

import numpy as np
import torch

from emrt_amd.src.utils import checkpoint as ck
from oracle.emrt_torch import EMRT as OracleEMRT


def test_pdparams_round_trip_and_layouts(tmp_path):
    torch.manual_seed(3)
    ref = OracleEMRT(6, "resnet18")
    state = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    path = str(tmp_path / "m.pdparams")
    ck.save_pdparams(state, path)
    with open(path, "rb") as f:
        raw = pickle.load(f)
    assert set(raw) == set(state) and all(isinstance(v, np.ndarray) for v in raw.values())
    # Paddle conventions on disk: Linear [in, out], packed MHA projection [E, 3E], conv / norm / embedding unchanged
    assert raw["model.encoder.layers.0.linear1.weight"].shape == (256, 1024)
    assert raw["model.encoder.layers.0.self_attn.sampling_offsets.weight"].shape == (256, 288)
    assert raw["model.decoder.layers.0.self_attn.in_proj_weight"].shape == (256, 768)
    assert raw["model.decoder.layers.0.self_attn.out_proj.weight"].shape == (256, 256)
    assert raw["model.level_embed.weight"].shape == tuple(state["model.level_embed.weight"].shape)        # nn.Embedding: not transposed
    assert raw["backbone.conv1.weight"].shape == tuple(state["backbone.conv1.weight"].shape)
    assert "backbone.bn1._mean" in raw and "backbone.bn1._variance" in raw
    np.testing.assert_array_equal(raw["model.encoder.layers.1.linear2.weight"], state["model.encoder.layers.1.linear2.weight"].numpy().T)
    # a Paddle-style extra entry and (name, array) packing are tolerated
    raw["StructuredToParameterName@@"] = {"x": "y"}
    k0 = "model.reference_points.weight"
    raw[k0] = (k0, raw[k0])
    with open(path, "wb") as f:
        pickle.dump(raw, f, protocol=2)
    back = ck.paddle_to_torch_state(ck.load_pdparams(path))
    assert set(back) == set(state)
    for k, v in state.items():
        assert back[k].shape == v.shape and torch.equal(back[k], v), k


def test_load_pretrained_model_matches_by_name_and_shape(tmp_path):
    torch.manual_seed(4)
    src = OracleEMRT(6, "resnet18")
    dst = OracleEMRT(7, "resnet18")          # different class count: the two classifier heads must be skipped, not crash
    dst.set_state_dict = dst.load_state_dict
    path = str(tmp_path / "src.pdparams")
    ck.save_pdparams(src.state_dict(), path)
    n = ck.load_pretrained_model(dst, path)
    own = dst.state_dict()
    mismatched = [k for k, v in src.state_dict().items() if tuple(v.shape) != tuple(own[k].shape)]
    assert 0 < len(mismatched) <= 4 and n == len(own) - len(mismatched)
    for k, v in src.state_dict().items():
        if k not in mismatched:
            assert torch.equal(own[k], v), k
    # backbone-only file (paddle.vision resnet keys without the "backbone." prefix)
    bb = {k[len("backbone."):]: v for k, v in src.state_dict().items() if k.startswith("backbone.")}
    p2 = str(tmp_path / "bb.pdparams")
    ck.save_pdparams({"backbone." + k: v for k, v in bb.items()}, p2)     # write with prefix so Linear rules apply, then strip
    raw = ck.load_pdparams(p2)
    with open(p2, "wb") as f:
        pickle.dump({k[len("backbone."):]: v for k, v in raw.items()}, f, protocol=2)
    fresh = OracleEMRT(6, "resnet18")
    m = ck.load_pretrained_model(fresh, p2, prefix="backbone.")
    assert m == len(bb)
    assert torch.equal(fresh.state_dict()["backbone.layer3.0.conv1.weight"], src.state_dict()["backbone.layer3.0.conv1.weight"])
    assert torch.equal(fresh.state_dict()["backbone.fc.weight"], src.state_dict()["backbone.fc.weight"])
