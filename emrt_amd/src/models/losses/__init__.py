"""Loss factory (reference: src/models/losses/__init__.py:6-12, mix_softmax_cross_entropy_loss.py:20-51)."""
from .... import functional as Fn
from ....runtime import ctx
from .... import _lib


class LossValue:
    """Device-resident scalar loss with the reference's call surface: .backward(), .numpy(), float()."""

    def __init__(self, total, parts, tape):
        self.tensor, self.parts, self.tape = total, parts, tape

    def backward(self):
        if self.tape is None:
            raise RuntimeError("loss was computed in eval mode; nothing to differentiate")
        self.tape.backward()
        self.tape = None

    def backward_until_split(self, segments=False):
        """First part of backward(): every op recorded after the model's last exchange mark.  Returns a callable that
        runs the rest -- or, with segments=True, one callable per remaining segment (between the marks, newest first)."""
        if self.tape is None:
            raise RuntimeError("loss was computed in eval mode; nothing to differentiate")
        tape, self.tape = self.tape, None
        marks = list(tape.splits)
        if not marks:
            tape.backward()
            return [] if segments else (lambda: None)
        tape.backward(stop_at=marks[-1])
        if not segments:
            return tape.backward
        stops = list(reversed(marks[:-1])) + [0]
        return [(lambda s_=s_: tape.backward(s_)) for s_ in stops]

    def item(self):
        return float(self.tensor[0].item())

    __float__ = item

    def numpy(self):
        import numpy as np
        return np.array([self.item()], dtype=np.float32)   # shape [1], as Paddle 2.1-2.4 (train.py:160)

    def __iter__(self):     # `sum(loss_list)` in the reference iterates the shape-[1] loss tensor (train.py:151-152)
        yield self

    def __radd__(self, other):
        return self if other == 0 else NotImplemented


class MixSoftmaxCrossEntropyLoss:
    """CE(main) + AUX_WEIGHT * CE(aux), each the mean over pixels with label != IGNORE_INDEX."""

    def __init__(self, config=None, ignore_index=255, aux=True, aux_weight=0.4):
        if config is not None:
            ignore_index, aux, aux_weight = config.TRAIN.IGNORE_INDEX, config.MODEL.AUX.LOSS, config.MODEL.AUX.AUX_WEIGHT
        self.ignore_index, self.aux, self.aux_weight = ignore_index, aux, aux_weight

    def __call__(self, preds, target):
        c = ctx()
        tape = getattr(preds, "tape", None)
        c.tape = tape
        try:
            target = target.contiguous()
            weights = [1.0] + [self.aux_weight if self.aux else 1.0] * (len(preds) - 1)
            live = [(p, w) for p, w in zip(preds, weights) if p is not None]
            if len(live) == 2 and tuple(live[0][0].shape) == tuple(live[1][0].shape):
                # the recipe's case (main + aux head at the input size): both heads in one pass, the weighted total formed by the finalize launch
                ra, rb, total = Fn.softmax_ce_pair(live[0][0], live[1][0], target, self.ignore_index, live[0][1], live[1][1])
                return LossValue(total, [ra, rb], tape)
            parts = [Fn.softmax_ce(p, target, self.ignore_index, w) for p, w in live]
        finally:
            c.tape = None
        total = c.empty((1,), parts[0].dtype)
        L = _lib.lib()
        L.call("emrt_scalar_axpby", Fn.P(total), Fn.P(parts[0]), 1.0, Fn.P(parts[1]) if len(parts) > 1 else None,
               weights[1] if len(parts) > 1 else 0.0, c.stream)
        return LossValue(total, parts, tape)


def get_loss_function(config):
    if config.TRAIN.LOSS == "MixSoftmaxCrossEntropyLoss":
        return MixSoftmaxCrossEntropyLoss(config)
    raise NotImplementedError("only MixSoftmaxCrossEntropyLoss is on the EMRT path (every EMRT yaml uses it)")
