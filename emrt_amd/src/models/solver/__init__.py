"""Optimizer / LR schedule of the EMRT recipe (reference: src/models/solver/optimizer.py:29-40, lr_scheduler.py:244-248).

Momentum = ClipGradByGlobalNorm -> L2 decay (g += wd * p) -> v = mu * v + g -> p -= lr * lr_mult * v, run by two HIP
kernels over the model's flat parameter buffer; PolynomialDecay is evaluated on the device from the step counter (the
host mirrors it for logging), so one optimizer step is capturable in a hipGraph."""
import ctypes

import torch

from .... import _lib
from .... import functional as Fn
from ....runtime import ctx


class PolynomialDecay:
    def __init__(self, learning_rate, decay_steps, end_lr=0.0, power=0.9):
        self.base_lr, self.decay_steps, self.end_lr, self.power = learning_rate, decay_steps, end_lr, power
        self.last_epoch = 0

    def get_lr(self):
        t = min(self.last_epoch, self.decay_steps)
        return (self.base_lr - self.end_lr) * (1.0 - t / self.decay_steps) ** self.power + self.end_lr

    def step(self):
        self.last_epoch += 1


class Momentum:
    def __init__(self, model, lr_scheduler, momentum=0.9, weight_decay=0.0, grad_clip=None, use_nesterov=False):
        if use_nesterov:
            raise NotImplementedError("Nesterov momentum is not used by any EMRT config")
        self.model, self._learning_rate = model, lr_scheduler
        self.momentum, self.weight_decay, self.grad_clip = momentum, float(weight_decay), grad_clip
        st = model.store
        c = ctx()
        self.clip_state = c.zeros((2,), torch.float32)
        self.lr_dev = c.zeros((1,), torch.float32)
        self.ranges = (ctypes.c_longlong * (2 * len(st.lr_ranges)))(*[v for r in st.lr_ranges for v in r])

    def get_lr(self):
        return self._learning_rate.get_lr()

    def grad_norm(self):
        return float(self.clip_state[1].item())

    def step(self):
        """Enqueues clip + update (master and compute-dtype mirror) on the current stream; advances the device step counter."""
        c, st, L = ctx(), self.model.store, _lib.lib()
        sch = self._learning_rate
        ws = c.workspace(L.query("emrt_gradnorm_workspace_bytes"))
        L.call("emrt_grad_clip_scale", Fn.P(st.grad), st.n_train, float(self.grad_clip or 0.0), Fn.P(self.clip_state), Fn.P(ws), c.stream)
        L.call("emrt_sgd_momentum_step", Fn.P(st.master), Fn.P(st.grad), Fn.P(st.velocity), st.n_train, Fn.P(self.clip_state),
               Fn.P(c.step_counter), sch.base_lr, sch.end_lr, sch.power, sch.decay_steps, self.momentum, self.weight_decay,
               ctypes.cast(self.ranges, ctypes.c_void_p), len(st.lr_ranges), st.lr_mult, Fn.P(self.lr_dev), Fn.P(st.mirror), st.dtype, c.stream)
        L.call("emrt_counter_add", Fn.P(c.step_counter), 1, c.stream)
        # the forward operands (fp32 master / its compute-dtype mirror) are current; the transposed dgrad copies are refreshed
        # where they are next needed, at the start of the next backward (EMRT.__call__ records it): the step then ends with
        # the mirror as the last thing written, which is what the next forward reads first
        if st.dirty:
            st.pack()

    def state_dict(self):
        """Momentum per parameter NAME in the parameter's logical shape (as the reference's .pdopt is keyed, train.py:204-206): the flat
        buffer's layout (alignment, padded stem channels) is an internal detail that has changed before and must not be the file format."""
        st = self.model.store
        return {"velocity": {n: st.named_view(st.velocity, n).detach().clone() for n in st.train_order},
                "step": int(ctx().step_counter.item()), "format": "per-parameter"}

    def set_state_dict(self, sd):
        st = self.model.store
        vel = sd["velocity"]
        if torch.is_tensor(vel):        # a checkpoint of an earlier version: the raw flat buffer, only valid for the identical layout
            if vel.numel() != st.velocity.numel():
                raise ValueError("optimizer checkpoint holds a flat velocity buffer of %d elements but this build lays the parameters out in %d: "
                                 "the flat format is layout-dependent and cannot be converted; resume from the model weights instead"
                                 % (vel.numel(), st.velocity.numel()))
            st.velocity.copy_(vel)
        else:
            missing = [n for n in st.train_order if n not in vel]
            if missing:
                raise KeyError("optimizer checkpoint lacks momentum for %d parameters, e.g. %s" % (len(missing), missing[:3]))
            st.velocity.zero_()
            for n in st.train_order:
                view = st.named_view(st.velocity, n)
                if tuple(vel[n].shape) != tuple(view.shape):
                    raise ValueError("optimizer checkpoint: momentum of %s has shape %s, the parameter %s" % (n, tuple(vel[n].shape), tuple(view.shape)))
                view.copy_(vel[n].to(view.device))
        ctx().step_counter.fill_(int(sd["step"]))
        self._learning_rate.last_epoch = int(sd["step"])


def get_scheduler(config):
    if config.TRAIN.LR_SCHEDULER.NAME == "PolynomialDecay":
        return PolynomialDecay(config.TRAIN.BASE_LR, config.TRAIN.ITERS, config.TRAIN.END_LR, config.TRAIN.POWER)
    raise NotImplementedError("only PolynomialDecay is on the EMRT path (every EMRT yaml uses it)")


def get_optimizer(model, lr_scheduler, config):
    if config.TRAIN.OPTIMIZER.NAME.lower() != "sgd":
        raise NotImplementedError("only SGD-momentum is on the EMRT path (every EMRT yaml uses it)")
    return Momentum(model, lr_scheduler, momentum=config.TRAIN.OPTIMIZER.MOMENTUM, weight_decay=float(config.TRAIN.OPTIMIZER.WEIGHT_DECAY),
                    grad_clip=config.TRAIN.OPTIMIZER.GRAD_CLIP, use_nesterov=config.TRAIN.OPTIMIZER.NESTEROV)
