"""Image / label transforms of the EMRT data pipeline, cv2-free (reference: src/transforms/transforms.py:24-88 Compose,
:91-110 RandomHorizontalFlip, :136-206 Resize, :209-270 ResizeStepScaling, :273-318 Normalize, :391-478
RandomPaddingCrop; src/transforms/__init__.py:4-57 get_transforms).

Arrays are HWC float32 RGB (the reference reads BGR with cv2 and converts to RGB; here PIL reads RGB directly), labels
HW uint8.  Bilinear / nearest resizing are restated in numpy with cv2's conventions (INTER_LINEAR: half-pixel centres,
no antialiasing; INTER_NEAREST: floor(dst * scale)).  Random draws use the same `np.random` / `random` calls as the
reference, so a seeded run makes the same decisions.
"""
import random

import numpy as np
from PIL import Image


def read_image(path):
    """-> HWC float32 RGB, 0..255 (cv2.imread + BGR2RGB in the reference, transforms.py:55,71)."""
    return np.asarray(Image.open(path).convert("RGB"), dtype=np.float32)


def read_label(path):
    """-> HW uint8 class indices; palette PNGs give their indices (transforms.py:62)."""
    return np.asarray(Image.open(path).convert("P"), dtype=np.uint8)


def resize_bilinear(img, w, h):
    """cv2.resize(img, (w, h), INTER_LINEAR) for HWC float arrays."""
    H, W = img.shape[:2]
    if (H, W) == (h, w):
        return img
    ys = np.clip((np.arange(h, dtype=np.float64) + 0.5) * (H / h) - 0.5, 0, None)
    xs = np.clip((np.arange(w, dtype=np.float64) + 0.5) * (W / w) - 0.5, 0, None)
    y0 = np.minimum(ys.astype(np.int64), H - 1)
    x0 = np.minimum(xs.astype(np.int64), W - 1)
    y1 = np.minimum(y0 + 1, H - 1)
    x1 = np.minimum(x0 + 1, W - 1)
    wy = (ys - y0).astype(np.float32)[:, None, None]
    wx = (xs - x0).astype(np.float32)[None, :, None]
    a = img.reshape(H, W, -1)
    top = a[y0][:, x0] * (1 - wx) + a[y0][:, x1] * wx
    bot = a[y1][:, x0] * (1 - wx) + a[y1][:, x1] * wx
    out = top * (1 - wy) + bot * wy
    return out.reshape((h, w) + img.shape[2:]).astype(img.dtype)


def resize_nearest(lab, w, h):
    """cv2.resize(label, (w, h), INTER_NEAREST)."""
    H, W = lab.shape[:2]
    if (H, W) == (h, w):
        return lab
    ys = np.minimum((np.arange(h) * (H / h)).astype(np.int64), H - 1)
    xs = np.minimum((np.arange(w) * (W / w)).astype(np.int64), W - 1)
    return lab[ys][:, xs]


class Compose:
    def __init__(self, transforms, to_rgb=True):
        if not isinstance(transforms, list):
            raise TypeError("The transforms must be a list!")
        self.transforms = transforms
        self.to_rgb = to_rgb

    def __call__(self, img, label=None):
        if isinstance(img, str):
            img = read_image(img)
        if isinstance(label, str):
            label = read_label(label)
        if img is None:
            raise ValueError("Can't read The image file {}!".format(img))
        for op in self.transforms:
            outputs = op(img, label)
            img = outputs[0]
            if len(outputs) == 2:
                label = outputs[1]
        return np.ascontiguousarray(np.transpose(img, (2, 0, 1))), label


class RandomHorizontalFlip:
    def __init__(self, prob=0.5):
        self.prob = prob

    def __call__(self, img, label=None):
        if random.random() < self.prob:
            img = img[:, ::-1, :]
            if label is not None:
                label = label[:, ::-1]
        return (img,) if label is None else (img, label)


class Resize:
    """transforms.py:136-206: target_size (w, h) or int; keep_ori_size returns the input untouched."""

    def __init__(self, target_size=520, interp="LINEAR", keep_ori_size=False):
        self.target_size, self.interp, self.keep_ori_size = target_size, interp, keep_ori_size

    def __call__(self, img, label=None):
        if not self.keep_ori_size and self.target_size is not None:
            ts = self.target_size
            w, h = (ts, ts) if isinstance(ts, int) else (ts[0], ts[1])
            img = resize_bilinear(img, w, h)
            if label is not None:
                label = resize_nearest(label, w, h)
        return (img,) if label is None else (img, label)


class ResizeStepScaling:
    def __init__(self, min_scale_factor=0.75, max_scale_factor=1.25, scale_step_size=0.25):
        if min_scale_factor > max_scale_factor:
            raise ValueError("min_scale_factor must be less than max_scale_factor, but they are {} and {}.".format(min_scale_factor, max_scale_factor))
        self.min_scale_factor, self.max_scale_factor, self.scale_step_size = min_scale_factor, max_scale_factor, scale_step_size

    def __call__(self, img, label=None):
        if self.min_scale_factor == self.max_scale_factor:
            scale_factor = self.min_scale_factor
        elif self.scale_step_size == 0:
            scale_factor = np.random.uniform(self.min_scale_factor, self.max_scale_factor)
        else:
            num_steps = int((self.max_scale_factor - self.min_scale_factor) / self.scale_step_size + 1)
            scale_factors = np.linspace(self.min_scale_factor, self.max_scale_factor, num_steps).tolist()
            np.random.shuffle(scale_factors)
            scale_factor = scale_factors[0]
        w = int(round(scale_factor * img.shape[1]))
        h = int(round(scale_factor * img.shape[0]))
        img = resize_bilinear(img, w, h)
        if label is not None:
            label = resize_nearest(label, w, h)
        return (img,) if label is None else (img, label)


class Normalize:
    def __init__(self, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)):
        if not (isinstance(mean, (list, tuple)) and isinstance(std, (list, tuple))):
            raise ValueError("{}: input type is invalid. It should be list or tuple".format(self))
        if any(s == 0 for s in std):
            raise ValueError("{}: std is invalid!".format(self))
        self.mean, self.std = mean, std

    def __call__(self, img, label=None):
        mean = np.asarray(self.mean, dtype=np.float64).reshape(1, 1, -1)
        stdinv = 1.0 / np.asarray(self.std, dtype=np.float64).reshape(1, 1, -1)
        img = ((img.astype(np.float32) - mean) * stdinv).astype(np.float32)      # functional.imnormalize_
        return (img,) if label is None else (img, label)


class RandomPaddingCrop:
    def __init__(self, crop_size=(512, 512), img_padding_value=(123.675, 116.28, 103.53), label_padding_value=255):
        if isinstance(crop_size, (list, tuple)):
            if len(crop_size) != 2:
                raise ValueError("Type of `crop_size` is list or tuple. It should include 2 elements, but it is {}".format(crop_size))
        elif not isinstance(crop_size, int):
            raise TypeError("The type of `crop_size` is invalid. It should be list or tuple, but it is {}".format(type(crop_size)))
        self.crop_size, self.img_padding_value, self.label_padding_value = crop_size, img_padding_value, label_padding_value

    def __call__(self, img, label=None):
        cw, chh = (self.crop_size, self.crop_size) if isinstance(self.crop_size, int) else (self.crop_size[0], self.crop_size[1])
        ih, iw = img.shape[0], img.shape[1]
        if not (ih == chh and iw == cw):
            ph, pw = max(chh - ih, 0), max(cw - iw, 0)
            if ph > 0 or pw > 0:            # bottom / right padding (cv2.copyMakeBorder BORDER_CONSTANT)
                padded = np.empty((ih + ph, iw + pw, img.shape[2]), dtype=img.dtype)
                padded[...] = np.asarray(self.img_padding_value, dtype=img.dtype)
                padded[:ih, :iw] = img
                img = padded
                if label is not None:
                    pl = np.full((ih + ph, iw + pw), self.label_padding_value, dtype=label.dtype)
                    pl[:ih, :iw] = label
                    label = pl
                ih, iw = img.shape[0], img.shape[1]
            if chh > 0 and cw > 0:
                h_off = np.random.randint(ih - chh + 1)
                w_off = np.random.randint(iw - cw + 1)
                img = img[h_off:chh + h_off, w_off:w_off + cw, :]
                if label is not None:
                    label = label[h_off:chh + h_off, w_off:w_off + cw]
        return (img,) if label is None else (img, label)


_MEAN, _STD = [123.675, 116.28, 103.53], [58.395, 57.12, 57.375]


def get_transforms(config):
    """Training transforms per dataset (src/transforms/__init__.py:4-57; only the datasets the EMRT yamls use)."""
    name = config.DATA.DATASET
    if name in ("Potsdam", "Vaihingen"):
        return [ResizeStepScaling(0.5, 2.0, 0.25),
                RandomPaddingCrop(crop_size=tuple(config.DATA.CROP_SIZE), img_padding_value=(0, 0, 0), label_padding_value=255),
                RandomHorizontalFlip(prob=0.5),
                Normalize(mean=_MEAN, std=_STD)]
    if name == "LoveDA":
        return [Normalize(mean=_MEAN, std=_STD)]
    raise NotImplementedError("{} dataset is not supported".format(name))


def get_val_transforms(config):
    """train.py:89-91 / val.py:95-97."""
    return [Resize(target_size=config.VAL.IMAGE_BASE_SIZE, keep_ori_size=config.VAL.KEEP_ORI_SIZE),
            Normalize(mean=list(config.VAL.MEAN), std=list(config.VAL.STD))]
