"""Sliding-window / single-scale inference (reference: src/api/infer.py:22-80 slide_inference, :82-157 ss_inference).

Windows of one image are gathered into ONE batch per model call (the reference issues one call per window position);
logit accumulation / count normalisation / argmax follow the reference exactly and run as HIP kernels, and so does the
bilinear resize of the logits when the network output size differs from `ori_shape` (infer.py:146-150; never the case for
the EMRT configs, whose tiles are evaluated at their own size).
"""
import ctypes

import torch

from ... import _lib, functional as Fn
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
    """imgs: list of fp32 [3, h, w] device tensors -> list of [1, ncls, h, w] fp32 logits (infer.py:22-80).
    Windows are cropped into one batch per model call and their logits accumulated / counted / normalised by HIP kernels
    (emrt_crop_windows, emrt_window_accumulate, emrt_window_normalise): no torch arithmetic on the path."""
    import ctypes
    L, c = Fn._L(), ctx()
    P = Fn.P
    outs = []
    for img in imgs:
        img = img.contiguous().float()
        C, h, w = img.shape[-3:]
        wins = window_grid(h, w, crop_size, stride_size)
        final = c.zeros((1, num_classes, h, w), torch.float32)
        count = c.zeros((1, 1, h, w), torch.float32)
        for i in range(0, len(wins), max_batch):
            chunk = wins[i:i + max_batch]
            ch, cw = chunk[0][2] - chunk[0][0], chunk[0][3] - chunk[0][1]
            assert all((c_ - a, d - b) == (ch, cw) for (a, b, c_, d) in chunk)
            org = (ctypes.c_int * (2 * len(chunk)))(*[v for (a, b, _, _) in chunk for v in (a, b)])
            orgp = ctypes.cast(org, ctypes.c_void_p)
            batch = c.empty((len(chunk), C, ch, cw), torch.float32)
            L.call("emrt_crop_windows", P(img), P(batch), orgp, len(chunk), C, h, w, ch, cw, c.stream)
            logits = model(batch)[0]
            assert logits.dtype == torch.float32 and logits.is_contiguous()
            L.call("emrt_window_accumulate", P(logits), P(final), P(count), orgp, len(chunk), num_classes, h, w, ch, cw, c.stream)
        out = c.empty((1, num_classes, h, w), torch.float32)
        L.call("emrt_window_normalise", P(final), P(count), P(out), num_classes, h, w, c.stream)
        outs.append(out)
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
            logit = _resize_nchw_f32(logit, shape[0], shape[1])      # F.interpolate(mode='bilinear'): infer.py:146-150
        # softmax is monotonic: argmax(softmax(x)) == argmax(x)  (infer.py:152-153)
        logit = logit.contiguous()
        n, ncls, hh, ww = logit.shape
        pred = ctx().empty((n, 1, hh, ww), torch.int32)
        Fn._L().call("emrt_argmax_nchw", Fn.P(logit), Fn.P(pred), n, ncls, hh, ww, ctx().stream)
        preds.append(pred)
    return preds


def _resize_nchw_f32(x, h, w):
    """fp32 [N,C,H,W] -> fp32 [N,C,h,w], bilinear, align_corners=False (F.interpolate(mode='bilinear') default), through the
    HIP resize kernel: NCHW -> NHWC ingest, then the kernel's fp32-NCHW output mode."""
    from ...runtime import F32
    L, c = Fn._L(), ctx()
    N, C, H, W = x.shape
    if (H, W) == (h, w):
        return x
    x = x.contiguous()
    nhwc = c.empty((N, H, W, C), torch.float32)
    L.call("emrt_nchw_to_nhwc", Fn.P(x), Fn.P(nhwc), N, C, H, W, C, F32, c.stream)
    out = c.empty((N, C, h, w), torch.float32)
    L.call("emrt_resize_bilinear_fwd", Fn.P(nhwc), H * W * C, C, H, W, Fn.P(out), 0, 0, h, w, None, 0, 0, N, C, 0, 1, F32, c.stream)
    return out


def _flip_w(x):
    out = ctx().empty(tuple(x.shape), torch.float32)
    x = x.contiguous()
    Fn._L().call("emrt_flip_w", Fn.P(x), Fn.P(out), x.numel() // x.shape[-1], x.shape[-1], ctx().stream)
    return out


def ms_inference(model, img, ori_shape, is_slide, base_size, stride_size, crop_size, num_classes, scales=(1.0,),
                 flip_horizontal=True, flip_vertical=False, rescale_from_ori=False):
    """Multi-scale + horizontal-flip inference (infer.py:160-260): per scale, sliding-window logits -> resize to
    `ori_shape` -> softmax, summed over scales and flips, then argmax.  img: one fp32 [3,h,w] device tensor (or a list of
    one); returns int32 [1,1,H,W].  The reference's quirk is kept: each scale resizes the image produced for the previous
    scale (its `img` variable is reassigned, :240) to a size derived from the ORIGINAL input size.  Every array operation
    is a HIP kernel (resize, flip, softmax-accumulate, argmax)."""
    if not isinstance(scales, (tuple, list)):
        raise TypeError("`scales` expects tuple/list, but received {}".format(type(scales)))
    if rescale_from_ori or not is_slide:
        raise NotImplementedError("whole-image rescale_from_ori testing is off in every EMRT config (config.py:203)")
    if flip_vertical:
        raise NotImplementedError("flip_vertical is a TODO in the reference as well (infer.py:258)")
    if isinstance(img, (list, tuple)):
        assert len(img) == 1
        img = img[0]
    L, c = Fn._L(), ctx()
    cur = img.float().unsqueeze(0) if img.dim() == 3 else img.float()
    h_input, w_input = cur.shape[-2], cur.shape[-1]
    H, W = (int(v) for v in ori_shape)
    final = c.zeros((1, num_classes, H, W), torch.float32)
    for scale in scales:
        h, w = int(h_input * scale + 0.5), int(w_input * scale + 0.5)
        if min(h, w) < crop_size[0]:
            new_short = crop_size[0]
            h, w = (int(new_short * h / w), new_short) if h > w else (new_short, int(new_short * w / h))
        cur = _resize_nchw_f32(cur, h, w)
        variants = [(cur, False)] + ([(_flip_w(cur), True)] if flip_horizontal else [])
        for im, flipped in variants:
            logit = slide_inference(model, [im[0]], crop_size, stride_size, num_classes)[0]
            if flipped:
                logit = _flip_w(logit)
            logit = _resize_nchw_f32(logit, H, W)
            L.call("emrt_softmax_nchw_acc", Fn.P(logit), Fn.P(final), 1, num_classes, H, W, c.stream)
    pred = c.empty((1, 1, H, W), torch.int32)
    L.call("emrt_argmax_nchw", Fn.P(final), Fn.P(pred), 1, num_classes, H, W, c.stream)
    return pred


class SlidingWindowEngine:
    """ss_inference(is_slide=True) for ONE fixed image shape as a replayable hipGraph (the inference counterpart of
    engine.TrainEngine): crop the windows -> model -> accumulate / count -> normalise -> argmax, ~700 launches per image,
    captured once and replayed, so the host costs one graph launch per image instead of ~5 ms of Python.

        eng = SlidingWindowEngine(model, (3, 1024, 1024), crop_size=(256, 256), stride_size=(256, 256), num_classes=6)
        pred = eng(img)          # img fp32 [3, H, W] on the device -> int32 [1, 1, H, W]; eng.logits holds [1, ncls, H, W]

    The arithmetic is exactly slide_inference + the argmax of ss_inference (reference: src/api/infer.py:22-80,149-154)."""

    def __init__(self, model, image_shape, crop_size, stride_size, num_classes, warmup=1):
        self.model, self.shape = model, tuple(int(v) for v in image_shape)
        self.crop, self.stride, self.ncls, self.warmup = tuple(crop_size), tuple(stride_size), num_classes, warmup
        self.calls, self.graph = 0, None
        self.image = self.logits = self.pred = None

    def _run(self, img):
        logits = slide_inference(self.model, [img], self.crop, self.stride, self.ncls)[0]
        n, ncls, hh, ww = logits.shape
        pred = ctx().empty((n, 1, hh, ww), torch.int32)
        Fn._L().call("emrt_argmax_nchw", Fn.P(logits), Fn.P(pred), n, ncls, hh, ww, ctx().stream)
        return logits, pred

    def __call__(self, img):
        assert tuple(img.shape) == self.shape and img.dtype == torch.float32
        self.model.eval()
        self.calls += 1
        if self.calls <= self.warmup:
            self.logits, self.pred = self._run(img.contiguous())
            return self.pred
        if self.graph is None:
            self.image = img.contiguous().clone()
            ctx().ensure_scratch()      # (registered outside the capture: the captured kernels bake its address in)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):      # (reader threads of a validation loader may make HIP calls meanwhile)
                self.logits, self.pred = self._run(self.image)
        if img.data_ptr() != self.image.data_ptr():
            src = img.contiguous()
            assert src.dtype == self.image.dtype and src.numel() == self.image.numel()
            _lib.lib().call("emrt_memcpy", ctypes.c_void_p(self.image.data_ptr()), ctypes.c_void_p(src.data_ptr()), src.numel() * src.element_size(), ctx().stream)
        if self.model.store.dirty:       # weights edited after the capture (load_state_dict of the next checkpoint): the replayed
            self.model.store.pack()      # kernels read the compute-dtype mirror, which only EMRT.__call__ -- not replay() -- refreshes
        self.graph.replay()
        return self.pred
