"""Torch/numpy-CPU restatement of sliding-window inference and the metrics (ORACLE -- test infra only;
PARITY UNPINNED, see oracle/__init__.py).

Reference (under /root/reference/semantic_segmentation/):
  src/api/infer.py:22-80     slide_inference  (window grid, accumulate, count, divide)
  src/api/infer.py:130-155   ss_inference tail (resize AC=False -> softmax -> argmax int32)
  src/utils/metrics.py:20-161 calculate_area, mean_iou, accuracy, kappa
"""
import numpy as np
import torch
import torch.nn.functional as F


def window_grid(h, w, crop, stride):
    """infer.py:39-58 for ONE image: list of (h1,w1,h2,w2).  crop/stride are (w,h) as in the reference."""
    w_crop, h_crop = crop
    w_stride, h_stride = stride
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


def slide_inference(model, imgs, crop_size, stride_size, num_classes):
    """infer.py:22-80.  imgs: list of [3,h,w] tensors -> list of [1,ncls,h,w] logits."""
    outs = []
    for img in imgs:  # the reference batches same-index windows of several images; result is identical
        h, w = img.shape[-2:]
        final = torch.zeros(1, num_classes, h, w)
        count = torch.zeros(1, 1, h, w)
        for (h1, w1, h2, w2) in window_grid(h, w, crop_size, stride_size):
            logit = model(img[:, h1:h2, w1:w2].unsqueeze(0))[0]
            final[:, :, h1:h2, w1:w2] += logit
            count[:, :, h1:h2, w1:w2] += 1
        outs.append(final / count)
    return outs


def logits_to_pred(logit, shape):
    """infer.py:149-154."""
    logit = F.interpolate(logit, shape, mode="bilinear", align_corners=False)
    logit = F.softmax(logit, dim=1)
    return torch.argmax(logit, dim=1, keepdim=True).to(torch.int32)


def calculate_area(pred, label, num_classes, ignore_index=255):
    """metrics.py:20-69 -> (intersect, pred_area, label_area) int64 [ncls]."""
    pred = np.asarray(pred).reshape(-1).astype(np.int64)
    label = np.asarray(label).reshape(-1).astype(np.int64)
    mask = label != ignore_index
    p, l = pred[mask], label[mask]
    pred_area = np.bincount(p[(p >= 0) & (p < num_classes)], minlength=num_classes)
    label_area = np.bincount(l[(l >= 0) & (l < num_classes)], minlength=num_classes)
    inter = np.bincount(p[p == l], minlength=num_classes)[:num_classes]
    return inter.astype(np.int64), pred_area.astype(np.int64), label_area.astype(np.int64)


def mean_iou(inter, pred_area, label_area):  # metrics.py:71-98
    union = pred_area + label_area - inter
    iou = np.array([0.0 if u == 0 else i / u for i, u in zip(inter, union)])
    return iou, float(np.mean(iou))


def accuracy(inter, pred_area, label_area):  # metrics.py:100-136 ("Acc" = micro precision)
    mean_acc = float(np.sum(inter) / np.sum(pred_area))
    prec = np.array([0.0 if p == 0 else i / p for i, p in zip(inter, pred_area)])
    rec = np.array([0.0 if l == 0 else i / l for i, l in zip(inter, label_area)])
    return mean_acc, prec, rec


def kappa(inter, pred_area, label_area):  # metrics.py:140-161
    total = np.sum(label_area)
    po = np.sum(inter) / total
    pe = np.sum(pred_area.astype(np.float64) * label_area) / (float(total) * float(total))
    return float((po - pe) / (1 - pe))


def ms_inference(model, img, ori_shape, stride_size, crop_size, num_classes, scales=(1.0,), flip_horizontal=True):
    """infer.py:160-260, sliding-window branch (rescale_from_ori is off in every EMRT config).  img: [3,h,w].
    Reference quirk kept: `img` is REASSIGNED by each scale's resize (:240), so scale k resizes the image already
    resized for scale k-1, while the target size is always derived from the ORIGINAL input size (:209-212)."""
    img = img.unsqueeze(0)
    h_input, w_input = img.shape[-2], img.shape[-1]
    final = 0
    for scale in scales:
        h, w = int(h_input * scale + 0.5), int(w_input * scale + 0.5)
        if min(h, w) < crop_size[0]:
            new_short = crop_size[0]
            h, w = (int(new_short * h / w), new_short) if h > w else (new_short, int(new_short * w / h))
        img = F.interpolate(img, (h, w), mode="bilinear", align_corners=False)
        logit = slide_inference(model, [img[0]], crop_size, stride_size, num_classes)[0]
        logit = F.interpolate(logit, tuple(ori_shape), mode="bilinear", align_corners=False)
        final = final + F.softmax(logit, dim=1)
        if flip_horizontal:
            lf = slide_inference(model, [img[0].flip(-1)], crop_size, stride_size, num_classes)[0].flip(-1)
            lf = F.interpolate(lf, tuple(ori_shape), mode="bilinear", align_corners=False)
            final = final + F.softmax(lf, dim=1)
    return final, torch.argmax(final, dim=1, keepdim=True).to(torch.int32)
