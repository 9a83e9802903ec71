"""`get_model(config)` -- the reference's model factory (semantic_segmentation/src/models/__init__.py:14-40) for the
EMRT hot path.  Only the plain EMRT branch (:37-38) is in scope; the other names the reference dispatches on
(SETR, UperNet, DPT, Segmenter, Trans2Seg, Segformer, FCN, EMRT_CSwin/ViT, EMRT_HRNet) raise NotImplementedError
instead of the reference's fall-through UnboundLocalError (:40)."""
from .emrt import EMRT

_OUT_OF_SCOPE = ("SETR", "UperNet", "DPT", "Segmenter", "Trans2Seg", "Segformer", "FCN", "EMRT_CSwin", "EMRT_ViT", "EMRT_HRNet")


def get_model(config):
    name = config.MODEL.NAME
    for other in _OUT_OF_SCOPE:
        if other in name:      # same test order as the reference: these are matched before the plain "EMRT"
            raise NotImplementedError("model %r is outside the EMRT ResNet hot path rebuilt here" % name)
    if "EMRT" in name:
        return EMRT(config)
    raise NotImplementedError("unknown MODEL.NAME %r" % name)
