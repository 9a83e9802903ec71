"""Evaluation entry point (reference: semantic_segmentation/val.py:34-231, val_in_train.py:19-125).

    python -m emrt_amd.val --config <yaml> --model_path <iter_N_state.pt> [--data tiles.npz]

Single-scale sliding-window inference (src/api/infer.py) over a set of tiles, per-image area accumulation, one
all-reduce of the [3, ncls] int64 areas at the end when WORLD_SIZE > 1 (the reference all-gathers three tensors per
image, val.py:164-170), then mIoU / Acc / Kappa / per-class F1 exactly as val.py:197-209 prints them.
"""
import argparse
import os
import time

import numpy as np
import torch

from .config import get_config, update_config
from .distributed import init_process_group
from .runtime import BF16, F16, F32
from .src.api import infer
from .src.models import get_model
from .src.utils import metrics


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="EMRT (MI355X HIP path) evaluation")
    p.add_argument("--config", dest="cfg", type=str,
                   default=os.path.join(os.path.dirname(__file__), "configs/EMRT/EMRT_256x256_160k_potsdam.yaml"))
    p.add_argument("--model_path", default=None, type=str)
    p.add_argument("--multi_scales", action="store_true", help="multi-scale (VAL.SCALE_RATIOS) + horizontal-flip inference, infer.py:160-260")
    p.add_argument("--data", default="synthetic", help="'synthetic', 'dataset' (DATA.DATASET under DATA.DATA_PATH) or a .npz")
    p.add_argument("--data_path", default=None, help="override DATA.DATA_PATH of the yaml")
    p.add_argument("--dtype", default="fp32", choices=["bf16", "fp16", "fp32"])
    return p.parse_args(argv)


class _OnDevice:
    """Sequence view: element i moved to the device when it is indexed (no-op for device tensors)."""

    def __init__(self, seq, dev):
        self.seq, self.dev = seq, dev

    def __len__(self):
        return len(self.seq)

    def __getitem__(self, i):
        return self.seq[i].to(self.dev, non_blocking=True)


class ValTiles:
    """Lazy validation set over a dataset object: item i is decoded when indexed (and kept on the HOST, pinned when possible, for
    the next evaluation), so a rank only ever decodes its own shard.  `which` = 0 -> image fp32 [3,h,w], 1 -> label int64 [h,w]."""

    def __init__(self, dataset, which, cache=None):
        self.ds, self.which = dataset, which
        self.cache = {} if cache is None else cache

    def __len__(self):
        return len(self.ds)

    def __getitem__(self, i):
        if i not in self.cache:
            a, b = self.ds[i]
            img = torch.from_numpy(np.ascontiguousarray(a)).float()
            lab = torch.from_numpy(np.ascontiguousarray(b[0])).long()
            try:
                img, lab = img.pin_memory(), lab.pin_memory()
            except RuntimeError:        # no pinned allocator (CPU-only host): pageable memory works too
                pass
            self.cache[i] = (img, lab)
        return self.cache[i][self.which]


def evaluate(model, images, labels, config, rank=0, nranks=1, multi_scales=False):
    """images: sequence of fp32 [3,h,w] tensors, labels: sequence of int64 [h,w], on the host or on the device (a lazy sequence
    such as ValTiles decodes on access).  Only this rank's shard (rank, rank + nranks, ...) is ever touched, and a host image is
    moved to the device when its turn comes -- as the reference streams them through a DataLoader (val.py:95-104) -- instead of
    the whole validation set living in HBM next to the training graph's pools.  Returns the reference's metric tuple."""
    from .runtime import ctx
    model.eval()
    ncls = config.DATA.NUM_CLASSES
    dev = ctx().device
    tot = torch.zeros(3, ncls, dtype=torch.int64, device=dev)
    t0 = time.time()
    images, labels = _OnDevice(images, dev), _OnDevice(labels, dev)
    for i in range(rank, len(images), nranks):
        img, lab = images[i], labels[i]          # ONE host-to-device copy each per evaluation (indexing _OnDevice copies)
        if multi_scales:            # val.py:168-181: VAL.SCALE_RATIOS + horizontal flip
            pred = infer.ms_inference(model, [img], lab.shape[-2:], True, config.VAL.IMAGE_BASE_SIZE, config.VAL.STRIDE_SIZE,
                                      config.VAL.CROP_SIZE, ncls, scales=list(config.VAL.SCALE_RATIOS), flip_horizontal=True,
                                      flip_vertical=False, rescale_from_ori=config.VAL.RESCALE_FROM_ORI)
        else:
            pred = infer.ss_inference(model, [img], [lab.shape[-2:]], True, config.VAL.IMAGE_BASE_SIZE, config.VAL.STRIDE_SIZE,
                                      config.VAL.CROP_SIZE, ncls, config.VAL.RESCALE_FROM_ORI)[0]
        inter, pa, la = metrics.calculate_area(pred, lab, ncls, config.TRAIN.IGNORE_INDEX)
        tot[0] += inter
        tot[1] += pa
        tot[2] += la
    if nranks > 1:
        torch.distributed.all_reduce(tot)
    class_iou, miou = metrics.mean_iou(tot[0], tot[1], tot[2])
    acc, class_acc, class_rec = metrics.accuracy(tot[0], tot[1], tot[2])
    kap = metrics.kappa(tot[0], tot[1], tot[2])
    with np.errstate(divide="ignore", invalid="ignore"):
        class_f1 = np.nan_to_num(2 * class_acc * class_rec / (class_acc + class_rec))
    return time.time() - t0, miou, acc, kap, class_iou, class_acc, class_f1, float(np.mean(class_f1))


def main(argv=None):
    args = parse_args(argv)
    config = update_config(get_config(), args)
    rank, local_rank, nranks = init_process_group()
    model = get_model(config)
    if args.model_path:                 # a .pdparams written by the reference / by train.py, or a torch checkpoint
        from .src.utils.checkpoint import load_entire_model
        load_entire_model(model, args.model_path)
    model.to_hip("cuda:%d" % local_rank, {"bf16": BF16, "fp16": F16, "fp32": F32}[args.dtype])
    if args.dtype == "fp16":
        model.compute_aux_in_eval = False      # the auxiliary head is computed and thrown away in eval (paddle_EMRT.py:300-302, infer.py:66)
    dev = torch.device("cuda", local_rank)
    if args.data == "synthetic":
        g = torch.Generator().manual_seed(0)
        h = w = config.VAL.IMAGE_BASE_SIZE or config.DATA.CROP_SIZE[0]
        images = [torch.randn(3, h, w, generator=g).to(dev) for _ in range(8)]
        labels = [torch.randint(0, config.DATA.NUM_CLASSES, (h, w), generator=g).to(dev) for _ in range(8)]
    elif args.data == "dataset":        # the reference's pipeline (val.py:95-104): DATA.DATASET under DATA.DATA_PATH, mode 'val'
        from .src.datasets import get_dataset
        from .src.transforms import get_val_transforms
        if getattr(args, "data_path", None):
            config.DATA.DATA_PATH = args.data_path
        ds = get_dataset(config, data_transform=get_val_transforms(config), mode="val")
        cache = {}
        images, labels = ValTiles(ds, 0, cache), ValTiles(ds, 1, cache)
    else:
        z = np.load(args.data)
        images = [torch.from_numpy(a).float().to(dev) for a in z["images"]]
        labels = [torch.from_numpy(a).long().to(dev) for a in z["labels"]]
    if list(config.VAL.STRIDE_SIZE) == [320, 320] and list(config.VAL.CROP_SIZE)[0] < 320:
        config.VAL.STRIDE_SIZE = list(config.VAL.CROP_SIZE)   # default stride > crop leaves NaN stripes
    cost, miou, acc, kap, ciou, cacc, cf1, mf1 = evaluate(model, images, labels, config, rank, nranks, multi_scales=args.multi_scales or config.VAL.MULTI_SCALES_VAL)
    if rank == 0:
        print("[EVAL] Images: {}  mIoU: {:.4f}  Acc: {:.4f}  Kappa: {:.4f}  mF1: {:.4f}".format(len(images), miou, acc, kap, mf1))
        print("[EVAL] Class IoU: " + str(np.round(ciou, 4)))
        print("[EVAL] Class Acc: " + str(np.round(cacc, 4)))
        print("[EVAL] Class F1-score: " + str(np.round(cf1, 4)))
    return miou


if __name__ == "__main__":
    main()
