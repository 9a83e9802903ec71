"""Torch-CPU restatement of the reference's loss / optimizer / LR schedule / train-step body
(ORACLE -- test infrastructure only; PARITY UNPINNED, see oracle/__init__.py).

Reference (under /root/reference/semantic_segmentation/):
  src/models/losses/mix_softmax_cross_entropy_loss.py:20-51   MixSoftmaxCrossEntropyLoss
  src/models/solver/optimizer.py:29-40                        Momentum + ClipGradByGlobalNorm + L2 decay
  src/models/solver/lr_scheduler.py:244-248                   PolynomialDecay
  train.py:141-159                                            step body
Paddle semantics assumed: CE 'mean' divides by the number of
non-ignored pixels; clip scale = clip/max(||g||,clip) over parameters that HAVE a gradient;
L2 decay is added after clipping (g += wd*p); v = mu*v + g; p -= lr*lr_mult*v.
"""
import torch
import torch.nn.functional as F

from .emrt_torch import lr_mult_of


def mix_softmax_ce_loss(preds, target, ignore_index=255, aux=True, aux_weight=0.4):
    """mix_softmax_cross_entropy_loss.py:29-51.  preds = (main, aux...) [B,C,H,W]; target int64 [B,H,W]."""
    loss = F.cross_entropy(preds[0], target, ignore_index=ignore_index, reduction="mean")
    for i in range(1, len(preds)):
        l = F.cross_entropy(preds[i], target, ignore_index=ignore_index, reduction="mean")
        loss = loss + (aux_weight * l if aux else l)
    return loss


def poly_lr(step, base_lr=0.01, end_lr=0.0, decay_steps=160000, power=0.9):
    """paddle.optimizer.lr.PolynomialDecay (cycle=False): lr_scheduler.py:244-248."""
    t = min(step, decay_steps)
    return (base_lr - end_lr) * (1.0 - t / decay_steps) ** power + end_lr


class MomentumRef:
    """optimizer.py:29-40 -- Momentum(momentum, weight_decay=L2, grad_clip=ClipGradByGlobalNorm)."""

    def __init__(self, named_params, momentum=0.9, weight_decay=1e-4, grad_clip=1.0):
        self.named = list(named_params)
        self.momentum, self.wd, self.clip = momentum, weight_decay, grad_clip
        self.velocity = {n: torch.zeros_like(p) for n, p in self.named}
        self.last_grad_norm = None

    @torch.no_grad()
    def step(self, lr):
        live = [(n, p) for n, p in self.named if p.grad is not None]
        if self.clip:
            gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for _, p in live)).float()
            self.last_grad_norm = float(gn)
            scale = self.clip / max(float(gn), self.clip)
        else:
            scale = 1.0
        for n, p in live:
            g = p.grad * scale + self.wd * p
            v = self.velocity[n]
            v.mul_(self.momentum).add_(g)
            p.add_(v, alpha=-lr * lr_mult_of(n))

    def clear_grad(self):
        for _, p in self.named:
            p.grad = None


def train_step(model, opt, images, labels, step, base_lr=0.01, end_lr=0.0, iters=160000, power=0.9,
               ignore_index=255, aux_weight=0.4):
    """train.py:146-159: fwd -> loss -> bwd -> optimizer.step (lr of this step) -> scheduler.step -> clear."""
    logits = model(images)
    loss = mix_softmax_ce_loss(logits, labels, ignore_index, True, aux_weight)
    loss.backward()
    lr = poly_lr(step, base_lr, end_lr, iters, power)
    opt.step(lr)
    opt.clear_grad()
    return float(loss), lr
