"""Segmentation metrics (reference: src/utils/metrics.py:20-161): per-class intersect / prediction / label areas with
ignore_index, mIoU, 'Acc' (= micro precision, as the reference defines it), per-class precision/recall, kappa."""
import numpy as np
import torch


def calculate_area(pred, label, num_classes, ignore_index=255):
    """pred/label integer tensors of equal shape -> (intersect, pred_area, label_area) int64 [ncls] tensors.
    Device tensors in the dtypes the inference engines and loaders produce (int32 predictions, int64 / int32 labels) are counted by one HIP kernel through the
    C-ABI (emrt_segmentation_areas); anything else (CPU tensors: the oracle-side tests) takes the torch expression below."""
    if (pred.is_cuda and label.is_cuda and pred.dtype == torch.int32 and label.dtype in (torch.int64, torch.int32) and pred.numel() == label.numel()
            and pred.is_contiguous() and label.is_contiguous() and num_classes <= 256):
        import ctypes
        from ... import _lib
        from ...runtime import ctx
        out = torch.zeros(3, num_classes, dtype=torch.int64, device=pred.device)
        _lib.lib().call("emrt_segmentation_areas", ctypes.c_void_p(pred.data_ptr()), ctypes.c_void_p(label.data_ptr()), int(label.dtype == torch.int64),
                        pred.numel(), int(num_classes), int(ignore_index), ctypes.c_void_p(out.data_ptr()), ctx().stream)
        return out[0], out[1], out[2]
    pred = pred.reshape(-1).to(torch.int64)
    label = label.reshape(-1).to(torch.int64)
    if pred.shape != label.shape:
        raise ValueError("Shape of `pred` and `label should be equal, but there are pred{} and label{}.".format(pred.shape, label.shape))
    mask = label != ignore_index
    p, l = pred[mask], label[mask]
    pv = (p >= 0) & (p < num_classes)
    lv = (l >= 0) & (l < num_classes)
    pred_area = torch.bincount(p[pv], minlength=num_classes)
    label_area = torch.bincount(l[lv], minlength=num_classes)
    inter = torch.bincount(p[(p == l) & pv], minlength=num_classes)
    return inter, pred_area, label_area


def _np(x):
    return x.detach().cpu().numpy().astype(np.float64) if isinstance(x, torch.Tensor) else np.asarray(x, dtype=np.float64)


def mean_iou(intersect_area, pred_area, label_area):
    i, p, l = _np(intersect_area), _np(pred_area), _np(label_area)
    union = p + l - i
    iou = np.array([0.0 if u == 0 else a / u for a, u in zip(i, union)])
    return iou, float(np.mean(iou))


def accuracy(intersect_area, pred_area, label_area):
    i, p, l = _np(intersect_area), _np(pred_area), _np(label_area)
    mean_acc = float(np.sum(i) / np.sum(p))
    prec = np.array([0.0 if b == 0 else a / b for a, b in zip(i, p)])
    rec = np.array([0.0 if b == 0 else a / b for a, b in zip(i, l)])
    return mean_acc, prec, rec


def kappa(intersect_area, pred_area, label_area):
    i, p, l = _np(intersect_area), _np(pred_area), _np(label_area)
    total = np.sum(l)
    po = np.sum(i) / total
    pe = np.sum(p * l) / (total * total)
    return float((po - pe) / (1 - pe))
