"""An INDEPENDENT writer / reader of `.pdparams` files in the layout `paddle.save(model.state_dict())` produces (Paddle 2.1+), used
to pin emrt_amd/src/utils/checkpoint.py against something it did not write itself (reference: semantic_segmentation/src/utils/
checkpoint.py:21-93 loads such files with paddle.load; paddle_vision_resnet.py:276-287 the ImageNet backbone file).

Test infrastructure: imports the oracle (torch restatement of the reference's module tree) and NOTHING from emrt_amd.  What a
Paddle state dict looks like on disk, restated from the Paddle conventions SURVEY.md Appendix A lists:
  * a pickle (protocol 2) of a plain dict  {structured parameter name: numpy.ndarray}
  * one extra entry  "StructuredToParameterName@@": {structured name: framework-internal name such as "conv2d_3.w_0"}
  * nn.Linear.weight is stored [in_features, out_features]; MultiHeadAttention.in_proj_weight [E, 3E]
  * BatchNorm running statistics are the buffers `_mean` / `_variance`; conv weights [out, in, kh, kw]
Which tensors are transposed is decided here from the owning MODULE'S TYPE in the oracle (nn.Linear, the attention module) --
deliberately a different mechanism from the importer's name table, so that a wrong name in that table shows up as a mismatch.
The full model is ~100 MB, far too large to commit: tests call write_fixture() with a seed and compare against the oracle."""
import os
import pickle
import sys

import numpy as np
import torch


def _owner_types(model):
    """{parameter / buffer name: type name of the module that owns it}."""
    own = {}
    for mod_name, mod in model.named_modules():
        for n, _ in list(mod.named_parameters(recurse=False)) + list(mod.named_buffers(recurse=False)):
            own[(mod_name + "." if mod_name else "") + n] = type(mod)
    return own


def _stored_transposed(name, tensor, owner):
    if tensor.dim() != 2:
        return False
    if issubclass(owner, torch.nn.Linear) and name.endswith(".weight"):
        return True
    return name.endswith("in_proj_weight")          # raw [3E, E] parameter of the oracle's MultiHeadAttention


def paddle_style_state(model):
    own = _owner_types(model)
    # framework-internal names ("conv2d_3.w_0"): one running index per layer kind, shared by the tensors of one layer
    internal, counters = {}, {}
    for mod_name, mod in model.named_modules():
        direct = [n for n, _ in list(mod.named_parameters(recurse=False)) + list(mod.named_buffers(recurse=False))]
        if not direct:
            continue
        kind = type(mod).__name__.lower()
        k = counters.get(kind, 0)
        counters[kind] = k + 1
        seen = {}
        for leaf in direct:
            tag = "w" if "weight" in leaf else "b" if "bias" in leaf else leaf.strip("_")
            j = seen.get(tag, 0)
            seen[tag] = j + 1
            internal[(mod_name + "." if mod_name else "") + leaf] = "%s_%d.%s_%d" % (kind, k, tag, j)
    out = {}
    for name, t in model.state_dict().items():
        a = t.detach().cpu().numpy()
        if _stored_transposed(name, t, own[name]):
            a = a.T
        out[name] = np.ascontiguousarray(a)
    out["StructuredToParameterName@@"] = {k: internal[k] for k in out}
    return out


def write_fixture(path, model, keys_prefix_strip=None):
    """model: an oracle module (EMRT or a bare ResNet).  keys_prefix_strip: write only the keys under this prefix, with the prefix
    removed -- the shape of the ImageNet backbone file the reference downloads (paddle_vision_resnet.py:276-287)."""
    state = paddle_style_state(model)
    if keys_prefix_strip:
        names = state.pop("StructuredToParameterName@@")
        state = {k[len(keys_prefix_strip):]: v for k, v in state.items() if k.startswith(keys_prefix_strip)}
        state["StructuredToParameterName@@"] = {k[len(keys_prefix_strip):]: v for k, v in names.items() if k.startswith(keys_prefix_strip)}
    with open(path, "wb") as f:
        pickle.dump(state, f, protocol=2)
    return state


def read_into_oracle(path, model):
    """The inverse, again by module type: loads a Paddle-layout file into an oracle module."""
    with open(path, "rb") as f:
        raw = pickle.load(f)
    raw.pop("StructuredToParameterName@@", None)
    own = _owner_types(model)
    sd = {}
    for name, cur in model.state_dict().items():
        a = torch.from_numpy(np.asarray(raw[name]))
        if _stored_transposed(name, cur, own[name]):
            a = a.t()
        assert tuple(a.shape) == tuple(cur.shape), (name, tuple(a.shape), tuple(cur.shape))
        sd[name] = a.contiguous().to(cur.dtype)
    model.load_state_dict(sd)
    return model


def seeded_oracle(backbone="resnet18", ncls=6, seed=77):
    """Oracle EMRT with every tensor (BatchNorm statistics included) drawn away from its constructor value, so that a tensor the
    importer skipped or mis-routed cannot hide behind an initial 0 / 1."""
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle.emrt_torch import EMRT
    torch.manual_seed(seed)
    m = EMRT(ncls, backbone)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n, b in m.named_buffers():
            if n.endswith("_mean"):
                b.copy_(torch.randn(b.shape, generator=g) * 0.1)
            elif n.endswith("_variance"):
                b.copy_(torch.rand(b.shape, generator=g) + 0.5)
        for n, p in m.named_parameters():
            if p.dim() == 1:
                p.add_(torch.randn(p.shape, generator=g) * 0.05)
    return m


if __name__ == "__main__":
    st = write_fixture(sys.argv[1] if len(sys.argv) > 1 else "/tmp/emrt_r18.pdparams", seeded_oracle())
    print("%d tensors, %.1f MB" % (len(st) - 1, sum(v.nbytes for k, v in st.items() if k != "StructuredToParameterName@@") / 1e6))
