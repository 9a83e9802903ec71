"""Sliding-window / single-scale inference (reference: src/api/infer.py:22-80 slide_inference, :82-157 ss_inference).

Windows of one image are gathered into ONE batch per model call (the reference issues one call per window position);
logit accumulation / count normalisation / final resize + argmax follow the reference exactly.  Host-side control only:
the per-window arithmetic is the model's HIP forward; the final bilinear resize runs through the HIP resize kernel.
"""
import torch

from ... import functional as Fn
from ...runtime import ctx


def window_grid(h, w, crop_size, stride_size):
    """infer.py:39-58 for one image -> [(h1, w1, h2, w2)].  crop/stride are (w, h) as in the reference."""
    w_crop, h_crop = crop_size
    w_stride, h_stride = stride_size
    rows = max(h - h_crop + h_stride - 1, 0) // h_stride + 1
    cols = max(w - w_crop + w_stride - 1, 0) // w_stride + 1
    wins = []
    for r in range(rows):
        for c in range(cols):
            h1, w1 = r * h_stride, c * w_stride
            if h1 >= h or w1 >= w:
                continue
            h2, w2 = min(h1 + h_crop, h), min(w1 + w_crop, w)
            h1, w1 = max(h2 - h_crop, 0), max(w2 - w_crop, 0)
            wins.append((h1, w1, h2, w2))
    return wins


def slide_inference(model, imgs, crop_size, stride_size, num_classes, max_batch=32):
    """imgs: list of fp32 [3, h, w] device tensors -> list of [1, ncls, h, w] fp32 logits (infer.py:22-80)."""
    outs = []
    for img in imgs:
        h, w = img.shape[-2:]
        wins = window_grid(h, w, crop_size, stride_size)
        final = torch.zeros(1, num_classes, h, w, device=img.device)
        count = torch.zeros(1, 1, h, w, device=img.device)
        for i in range(0, len(wins), max_batch):
            chunk = wins[i:i + max_batch]
            batch = torch.stack([img[:, a:c_, b:d] for (a, b, c_, d) in chunk], 0).contiguous()
            logits = model(batch)[0]
            for j, (a, b, c_, d) in enumerate(chunk):
                final[0, :, a:c_, b:d] += logits[j]
                count[0, :, a:c_, b:d] += 1
        outs.append(final / count)      # uncovered pixels give 0/0 = NaN exactly as the reference (:79)
    return outs


def ss_inference(model, img, ori_shape, is_slide, base_size, stride_size, crop_size, num_classes, rescale_from_ori=False):
    """infer.py:82-157 (single scale).  img: list of [3,h,w]; ori_shape: list of (H, W) or None."""
    if not is_slide:
        if len(img) != 1:
            raise ValueError("batch_size should be set to 1 while is_slide is False")
        logit_list = [model(img[0].unsqueeze(0) if img[0].dim() == 3 else img[0])[0]]
    else:
        if rescale_from_ori:
            raise NotImplementedError("rescale_from_ori is off in every EMRT config (config.py:203)")
        logit_list = slide_inference(model, img, crop_size, stride_size, num_classes)
    if ori_shape is None:
        return logit_list
    preds = []
    for logit, shape in zip(logit_list, ori_shape):
        shape = tuple(int(s) for s in shape)
        if tuple(logit.shape[-2:]) != shape:
            logit = torch.nn.functional.interpolate(logit, shape, mode="bilinear", align_corners=False)
        # softmax is monotonic: argmax(softmax(x)) == argmax(x)  (infer.py:152-153)
        preds.append(torch.argmax(logit, dim=1, keepdim=True).to(torch.int32))
    return preds
