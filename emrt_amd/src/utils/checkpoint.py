"""Checkpoint / pretrained-weight I/O in the reference's on-disk format (reference: src/utils/checkpoint.py:21-93,
train.py:197-220).

A `.pdparams` file written by `paddle.save(state_dict)` (Paddle 2.1+) is a plain pickle of `{name: numpy.ndarray}`
(sometimes with one extra `StructuredToParameterName@@` entry and, for large tensors, `(name, ndarray)` tuples), so it
can be read and written without Paddle.  UNVERIFIED against a real file: none is available in this environment
(no network, no Paddle); the layout rules below are the documented Paddle conventions:

  * `nn.Linear.weight` is `[in, out]`      -> transposed to this repo's `[out, in]`
  * MHA `in_proj_weight` is `[E, 3E]`      -> transposed to `[3E, E]`
  * BatchNorm buffers are `_mean`/`_variance` -> same names here
  * conv weights `[out, in, kh, kw]`, embeddings `[n, C]`, norms: unchanged

Which 2-D tensors are Linear weights is decided by NAME (a 256x256 matrix cannot be told from its shape).
"""
import logging
import os
import pickle

import numpy as np
import torch

logger = logging.getLogger("emrt_amd.checkpoint")

# last path component before ".weight" of every nn.Linear on the EMRT path (transformer_encoder_decoder.py, layers.py,
# paddle_vision_resnet.py:fc)
_LINEAR_LEAVES = ("sampling_offsets", "attention_weights", "value_proj", "output_proj", "out_proj", "linear1", "linear2",
                  "reference_points", "fc")
_SKIP_KEYS = ("StructuredToParameterName@@",)


def is_transposed_in_paddle(key, ndim):
    """True when Paddle stores this 2-D tensor as the transpose of this repo's layout."""
    if ndim != 2:
        return False
    if key.endswith("in_proj_weight"):
        return True
    if key.endswith(".weight"):
        return key[:-len(".weight")].rsplit(".", 1)[-1] in _LINEAR_LEAVES
    return False


def _as_array(v):
    if isinstance(v, tuple) and len(v) == 2 and isinstance(v[1], np.ndarray):      # (name, ndarray) packing
        v = v[1]
    if isinstance(v, torch.Tensor):
        v = v.detach().cpu().numpy()
    return np.asarray(v)


def load_pdparams(path):
    """-> {key: numpy array} exactly as stored (Paddle layouts)."""
    with open(path, "rb") as f:
        raw = pickle.load(f, encoding="latin1")
    if not isinstance(raw, dict):
        raise ValueError("%s: expected a pickled dict of arrays, got %s" % (path, type(raw).__name__))
    return {k: _as_array(v) for k, v in raw.items() if k not in _SKIP_KEYS}


def paddle_to_torch_state(pd_state):
    """Paddle-layout arrays -> this repo's state-dict tensors (fp32 / int64 as stored)."""
    out = {}
    for k, v in pd_state.items():
        a = _as_array(v)
        if is_transposed_in_paddle(k, a.ndim):
            a = a.T
        out[k] = torch.from_numpy(np.ascontiguousarray(a))
    return out


def torch_to_paddle_state(state):
    """This repo's state dict -> Paddle-layout numpy arrays (what paddle.load() would hand to set_state_dict)."""
    out = {}
    for k, v in state.items():
        a = _as_array(v)
        if is_transposed_in_paddle(k, a.ndim):
            a = a.T
        out[k] = np.ascontiguousarray(a)
    return out


def save_pdparams(state, path):
    """Write a state dict as a Paddle-readable `.pdparams` pickle (protocol 2, as paddle.save uses)."""
    with open(path, "wb") as f:
        pickle.dump(torch_to_paddle_state(state), f, protocol=2)


def load_pretrained_model(model, pretrained_model, prefix=""):
    """reference: checkpoint.py:38-93.  Loads every key whose name and shape match; warns about the rest.
    `prefix` maps a backbone-only file (paddle.vision resnet50 keys `conv1.weight`, ...) under `backbone.`.
    Accepts `.pdparams` pickles and torch checkpoints written by emrt_amd/train.py."""
    if pretrained_model is None:
        return 0
    if not os.path.exists(pretrained_model):
        raise ValueError("The pretrained model directory is not Found: {}".format(pretrained_model))
    if pretrained_model.endswith(".pdparams"):
        src = paddle_to_torch_state({prefix + k: v for k, v in load_pdparams(pretrained_model).items()})
    else:
        ck = torch.load(pretrained_model, map_location="cpu")
        src = {prefix + k: v for k, v in ck.get("model", ck).items()}
    own = model.state_dict()
    loaded, todo = 0, {}
    for k, cur in own.items():
        if k not in src:
            logger.warning("%s is not in pretrained model", k)
        elif tuple(src[k].shape) != tuple(cur.shape):
            logger.warning("[SKIP] shape of pretrained params %s doesn't match (pretrained %s, actual %s)", k, tuple(src[k].shape),
                           tuple(cur.shape))
        else:
            todo[k] = src[k].to(cur.dtype)
            loaded += 1
    model.load_state_dict({**{k: v for k, v in own.items()}, **todo})
    logger.info("There are %d/%d variables loaded into %s.", loaded, len(own), model.__class__.__name__)
    return loaded


def load_entire_model(model, pretrained):
    """reference: checkpoint.py:21-35."""
    if pretrained is not None:
        return load_pretrained_model(model, pretrained)
    logger.warning("Not all pretrained params of %s are loaded, training from scratch or a pretrained backbone.", model.__class__.__name__)
    return 0
