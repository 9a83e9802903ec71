"""Tile datasets of the EMRT configs (reference: src/datasets/{dataset,potsdam,vaihingen,loveda}.py, __init__.py:10-69
get_dataset) plus a small prefetching batch loader that stages batches on the GPU.

Directory layouts are the reference's:
  Potsdam / Vaihingen (both map to the Potsdam class in the reference's factory, __init__.py:50-58):
      <root>/train/<n>.tif  + <root>/train_convert_labels/<n>.png ;  <root>/test/... + <root>/test_convert_labels/...
      labels are class indices 0..5, 255 = ignore
  LoveDA:  <root>/Train/images_png/<n>.png + <root>/Train/masks_png/<n>.png ; <root>/Val/...
      masks are 1..7 with 0 = ignore: shifted by -1, ignore -> 255 (loveda.py:58-70)
"""
import os
import queue
import threading

import numpy as np
import torch
from PIL import Image

from ..transforms import Compose


class Dataset:
    """dataset.py: file_list of [image_path, label_path]; train -> (CHW float32, HW), val -> (CHW float32, 1HW)."""
    label_shift = 0

    def __init__(self, transforms, dataset_root, mode, num_classes, img_dir, label_dir, label_name):
        mode = mode.lower()
        if mode not in ("train", "val"):
            raise ValueError("`mode` should be one of ('train', 'val'), but got {}.".format(mode))
        if transforms is None:
            raise ValueError("`transforms` is necessary, but it is None.")
        self.transforms, self.mode, self.num_classes, self.ignore_index = Compose(transforms), mode, num_classes, 255
        self.dataset_root = dataset_root
        files = sorted(os.listdir(img_dir), key=lambda x: int(os.path.splitext(x)[0]))
        self.file_list = [[os.path.join(img_dir, f), os.path.join(label_dir, label_name(f))] for f in files]

    def __len__(self):
        return len(self.file_list)

    def __getitem__(self, idx):
        image_path, label_path = self.file_list[idx]
        if self.mode == "val":
            img, _ = self.transforms(img=image_path)
            label = np.asarray(Image.open(label_path))
            if self.label_shift:
                label = label - np.uint8(1)          # uint8 wrap: class 0 (ignore) -> 255
            return img, label[np.newaxis, :, :]
        img, label = self.transforms(img=image_path, label=label_path)
        if self.label_shift:
            label = label - np.uint8(1)
            label[label == 254] = 255                # padding (255) shifted to 254: restore
        return img, label


class Potsdam(Dataset):
    def __init__(self, transforms, dataset_root=None, mode="train", num_classes=6):
        sub = "train" if mode.lower() == "train" else "test"
        super().__init__(transforms, dataset_root, mode, num_classes, os.path.join(dataset_root, sub),
                         os.path.join(dataset_root, sub + "_convert_labels"), lambda f: os.path.splitext(f)[0] + ".png")


class LoveDA(Dataset):
    label_shift = 1

    def __init__(self, transforms, dataset_root=None, mode="train", num_classes=7):
        sub = "Train" if mode.lower() == "train" else "Val"
        super().__init__(transforms, dataset_root, mode, num_classes, os.path.join(dataset_root, sub, "images_png"),
                         os.path.join(dataset_root, sub, "masks_png"), lambda f: f)


def get_dataset(config, data_transform, mode="train"):
    name = config.DATA.DATASET
    mode = "val" if mode in ("val", "test") else "train"
    if name in ("Potsdam", "Vaihingen"):
        return Potsdam(transforms=data_transform, dataset_root=config.DATA.DATA_PATH, num_classes=config.DATA.NUM_CLASSES, mode=mode)
    if name == "LoveDA":
        return LoveDA(transforms=data_transform, dataset_root=config.DATA.DATA_PATH, num_classes=config.DATA.NUM_CLASSES, mode=mode)
    raise NotImplementedError("{} dataset is not supported".format(name))


class TileLoader:
    """Iteration-based training loader (utils/dataloader.py:22-49): `sampler` yields index lists (DistributedTileSampler);
    `workers` threads decode + augment tiles (PIL and numpy release the GIL for the heavy parts) into PINNED host batches;
    at most `prefetch` finished batches exist at any time (a worker takes a slot before it starts a batch and the consumer
    returns it when it pops one).  The host->device copy is issued by the CONSUMER thread on the training stream (torch's
    current stream there): every later kernel on that stream is ordered after the copy, and the caching allocator ties the
    device block to that stream, so a batch can never be recycled under a queued kernel (a reader thread's `.to(device)`
    would run on that thread's own NULL stream, with nothing ordering it against the training stream)."""

    def __init__(self, dataset, sampler, device, workers=4, prefetch=4):
        self.dataset, self.sampler, self.device, self.workers, self.prefetch = dataset, sampler, device, max(1, workers), max(1, prefetch)
        self.pin = torch.cuda.is_available() and torch.device(device).type == "cuda"

    def _batch(self, idx):
        """Decoded, augmented host batch (fp32 [B,3,H,W], int64 [B,H,W]), page-locked when it is going to a GPU."""
        items = [self.dataset[i] for i in idx]
        imgs = torch.from_numpy(np.stack([it[0] for it in items]))
        labs = torch.from_numpy(np.stack([it[1] for it in items]).astype(np.int64))
        if self.pin:
            imgs, labs = imgs.pin_memory(), labs.pin_memory()
        return imgs, labs

    def epochs(self, start_epoch=0):
        """Endless generator of device batches, reshuffling per epoch."""
        todo, done = queue.Queue(maxsize=self.prefetch * 2), {}
        cv, stop = threading.Condition(), threading.Event()
        slots = threading.Semaphore(self.prefetch)       # finished-but-unconsumed batches (+ the ones being built)
        turn = [0]                                        # batches are built in order: slot n is taken before slot n + 1

        def feed():
            ep, n = start_epoch, 0
            while not stop.is_set():
                self.sampler.set_epoch(ep)
                for idx in self.sampler:
                    while not stop.is_set():
                        try:
                            todo.put((n, idx), timeout=0.2)
                            break
                        except queue.Full:
                            continue
                    if stop.is_set():
                        return
                    n += 1
                ep += 1

        def work():
            while not stop.is_set():
                try:
                    n, idx = todo.get(timeout=0.2)
                except queue.Empty:
                    continue
                # wait for this batch's turn, then for a free slot: the batch the consumer is waiting for always gets one
                with cv:
                    while turn[0] != n and not stop.is_set():
                        cv.wait(timeout=0.2)
                while not stop.is_set() and not slots.acquire(timeout=0.2):
                    pass
                with cv:
                    turn[0] = n + 1
                    cv.notify_all()
                if stop.is_set():
                    return
                b = self._batch(idx)
                with cv:
                    done[n] = b
                    cv.notify_all()

        threads = [threading.Thread(target=feed, daemon=True)] + [threading.Thread(target=work, daemon=True) for _ in range(self.workers)]
        for t in threads:
            t.start()
        try:
            n = 0
            while True:
                with cv:
                    while n not in done:
                        cv.wait(timeout=1.0)
                    imgs, labs = done.pop(n)
                slots.release()
                n += 1
                # consumer thread, training stream: see the class docstring
                yield imgs.to(self.device, non_blocking=True), labs.to(self.device, non_blocking=True)
        finally:
            stop.set()
