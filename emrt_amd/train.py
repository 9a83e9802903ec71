"""Training entry point with the reference's CLI and step order (semantic_segmentation/train.py:26-266).

    python -m emrt_amd.train --config emrt_amd/configs/EMRT/EMRT_256x256_160k_potsdam.yaml [--seed 1234]
    python -m emrt_amd.train --gpus 8 --config ...          # starts the 8 rank processes itself (one per GPU, RCCL)
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m emrt_amd.train --config ...

Per iteration (train.py:141-159): forward -> MixSoftmaxCrossEntropyLoss -> backward (+ RCCL gradient all-reduce when
WORLD_SIZE > 1) -> Momentum step with global-norm clip -> poly LR step -> clear grads; every LOGGING_INFO_FREQ
iterations rank 0 prints the reference's log line (:177-181); every SAVE_FREQ_CHECKPOINT iterations a checkpoint
(model + optimizer state + iteration, so training can actually be resumed -- the reference never wired that, :103).
The dataset pipeline (cv2 / Potsdam readers) is outside this path: tiles come from --data synthetic (default: seeded
random tiles of DATA.CROP_SIZE, resident on the device) or from an .npz of pre-cut tiles (--data file.npz with arrays
`images` [N,3,H,W] float32 normalised and `labels` [N,H,W] int64).
"""
import argparse
import os
import time
from collections import deque

import numpy as np
import torch

from .config import get_config, update_config
from .distributed import DistributedTileSampler, init_process_group
from .engine import TrainEngine
from .runtime import BF16, F32
from .src.models import get_model
from .src.models.losses import get_loss_function
from .src.models.solver import get_optimizer, get_scheduler


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="EMRT (MI355X HIP path) training")
    p.add_argument("--config", dest="cfg", type=str,
                   default=os.path.join(os.path.dirname(__file__), "configs/EMRT/EMRT_256x256_160k_potsdam.yaml"), help="The config file.")
    p.add_argument("--seed", dest="seed", default=1234, type=int, help="Set the random seed during training.")
    p.add_argument("--data", default="synthetic", help="'synthetic', 'dataset' (DATA.DATASET under DATA.DATA_PATH, the reference's "
                   "directory layout) or a .npz of pre-cut tiles")
    p.add_argument("--data_path", default=None, help="override DATA.DATA_PATH of the yaml")
    p.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    p.add_argument("--iters", type=int, default=None, help="override TRAIN.ITERS")
    p.add_argument("--no-graph", action="store_true", help="launch kernels eagerly instead of replaying a captured hipGraph")
    p.add_argument("--resume", default=None, help="checkpoint written by this script")
    p.add_argument("--save_dir", default=None, help="override SAVE_DIR of the yaml (the reference's yamls point at the authors' disks)")
    p.add_argument("--pretrained_backbone", default=None, help="weights to start from (.pdparams or torch): whole model or ResNet backbone")
    p.add_argument("--no-eval", action="store_true", help="skip the periodic evaluation (train.py:187-195) and best_model.pdparams")
    p.add_argument("--gpus", type=int, default=0, help="without a launcher (no RANK / WORLD_SIZE in the environment): start this many rank "
                   "processes, one per GPU (the reference gets its ranks from paddle.distributed.launch, train.py:116-123)")
    p.add_argument("--val_tiles", type=int, default=16, help="--data synthetic / .npz without val arrays: how many held-out tiles to evaluate on")
    return p.parse_args(argv)


class TimeAverager:  # utils/timer.py:17-40
    def __init__(self):
        self.reset()

    def reset(self):
        self._cnt, self._total_time, self._total_samples = 0, 0.0, 0

    def record(self, usetime, num_samples=None):
        self._cnt += 1
        self._total_time += usetime
        if num_samples:
            self._total_samples += num_samples

    def get_average(self):
        return 0 if self._cnt == 0 else self._total_time / float(self._cnt)

    def get_ips_average(self):
        return 0 if not self._total_samples or self._cnt == 0 else float(self._total_samples) / self._total_time


def calculate_eta(remaining_step, speed):  # utils/timer.py:43-51
    remaining_time = int(max(remaining_step, 0) * speed)
    h, r = divmod(remaining_time, 3600)
    m, s = divmod(r, 60)
    return "{:0>2}:{:0>2}:{:0>2}".format(h, m, s)


def synthetic_tiles(n, crop, ncls, seed, device):
    g = torch.Generator().manual_seed(seed)
    images = torch.randn(n, 3, crop[1], crop[0], generator=g)
    labels = torch.randint(0, ncls, (n, crop[1], crop[0]), generator=g)
    labels[torch.rand(n, crop[1], crop[0], generator=g) < 0.02] = 255
    return images.to(device), labels.to(device)


def main(argv=None):
    args = parse_args(argv)
    launched = "WORLD_SIZE" in os.environ or "RANK" in os.environ
    if args.gpus > 1 and not launched:          # pure parent: no GPU call in this process, N fresh rank processes
        import sys
        from .distributed import spawn_ranks
        codes, _ = spawn_ranks(args.gpus, [sys.executable, "-m", "emrt_amd.train"] + list(sys.argv[1:] if argv is None else argv))
        raise SystemExit(1 if any(codes) else 0)
    if args.gpus and launched and int(os.environ.get("WORLD_SIZE", 1)) != args.gpus:
        raise SystemExit("[train] WORLD_SIZE=%s but --gpus %d" % (os.environ.get("WORLD_SIZE"), args.gpus))
    if os.environ.get("EMRT_ALL_RANKS_ON_GPU0"):      # test aid (with EMRT_DIST_BACKEND=gloo): every rank on device 0
        os.environ["LOCAL_RANK"] = "0"
    config = update_config(get_config(), args)
    rank, local_rank, nranks = init_process_group()
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    if args.save_dir:
        config.SAVE_DIR = args.save_dir
    if args.data_path:
        config.DATA.DATA_PATH = args.data_path
    model = get_model(config)
    if config.MODEL.PRETRAINED:
        # MODEL.PRETRAINED / --pretrained_backbone (config.py:245-246): a whole-model file (.pdparams or torch) or an ImageNet
        # ResNet file as paddle.vision saves it (keys without the "backbone." prefix; detected from the key names)
        from .src.utils.checkpoint import load_pretrained_model, load_pdparams
        path = config.MODEL.PRETRAINED
        if path.endswith(".pdparams"):
            keys = load_pdparams(path).keys()
        else:       # a checkpoint of this script ({"model": state, ...}) or a bare torch state dict
            ck = torch.load(path, map_location="cpu")
            keys = ck.get("model", ck).keys()
        prefix = "" if any(k.startswith("backbone.") or k.startswith("model.") for k in keys) else "backbone."
        load_pretrained_model(model, path, prefix=prefix)
    model.to_hip("cuda:%d" % local_rank, BF16 if args.dtype == "bf16" else F32, seed=args.seed + rank)   # per-rank dropout streams
    model.train()
    iters = args.iters or config.TRAIN.ITERS
    if args.iters:
        config.TRAIN.ITERS = iters
    lr_scheduler = get_scheduler(config)
    optimizer = get_optimizer(model, lr_scheduler, config)
    loss_func = get_loss_function(config)
    bs = config.DATA.BATCH_SIZE
    dev = torch.device("cuda", local_rank)
    loader = None
    if args.data == "synthetic":
        images, labels = synthetic_tiles(max(4 * bs * nranks, 64), config.DATA.CROP_SIZE, config.DATA.NUM_CLASSES, args.seed, dev)
        n_tiles = images.shape[0]
    elif args.data == "dataset":        # the reference's pipeline: DATA.DATASET under DATA.DATA_PATH (train.py:79-86)
        from .src.datasets import get_dataset, TileLoader
        from .src.transforms import get_transforms
        dataset_train = get_dataset(config, data_transform=get_transforms(config), mode="train")
        n_tiles = len(dataset_train)
    else:
        z = np.load(args.data)
        images, labels = torch.from_numpy(z["images"]).float().to(dev), torch.from_numpy(z["labels"]).long().to(dev)
        n_tiles = images.shape[0]
    sampler = DistributedTileSampler(n_tiles, bs, rank, nranks, shuffle=True, drop_last=True, seed=args.seed)
    if args.data == "dataset":
        import copy
        loader = TileLoader(dataset_train, copy.copy(sampler), dev, workers=max(1, config.DATA.NUM_WORKERS), prefetch=4).epochs()
    start_iter = 0
    if args.resume:
        ck = torch.load(args.resume, map_location="cpu")
        model.load_state_dict(ck["model"])
        optimizer.set_state_dict(ck["optimizer"])
        start_iter = ck["iter"]
    engine = TrainEngine(model, optimizer, loss_func, nranks, use_graph=not args.no_graph)
    if rank == 0:
        # the shipped yamls carry the reference authors' own disks as SAVE_DIR: a run that cannot checkpoint must say so
        # at once, not after 160k iterations (train.py:197-220 writes there unconditionally)
        try:
            os.makedirs(config.SAVE_DIR, exist_ok=True)
            if not os.access(config.SAVE_DIR, os.W_OK):
                raise PermissionError(config.SAVE_DIR)
        except OSError as e:
            fallback = os.path.abspath(os.path.join("output", os.path.basename(os.path.normpath(config.SAVE_DIR)) or "emrt"))
            print("[WARNING] SAVE_DIR {!r} cannot be created or written ({}); checkpoints go to {!r} instead "
                  "(pass --save_dir to choose)".format(config.SAVE_DIR, e, fallback), flush=True)
            os.makedirs(fallback, exist_ok=True)        # failing here is fatal: no silent run without checkpoints
            config.SAVE_DIR = fallback
        print("train_cfg: {}\ntrain_model_name: {}\ntrain_datatset: {}".format(args.cfg, config.MODEL.NAME, config.DATA.DATASET))
    # ---- validation set for the periodic evaluation (train.py:88-100, 187-195; val_in_train.py:19-125) ---------------------
    val_images = val_labels = None
    if not args.no_eval:
        if args.data == "dataset":
            from .src.transforms import get_val_transforms
            ds_val = get_dataset(config, data_transform=get_val_transforms(config), mode="val")
            from .val import ValTiles       # decoded lazily, this rank's shard only, kept on the host between evaluations
            cache = {}
            val_images, val_labels = ValTiles(ds_val, 0, cache), ValTiles(ds_val, 1, cache)
        elif args.data != "synthetic" and "val_images" in z.files:
            val_images = [torch.from_numpy(a).float().to(dev) for a in z["val_images"]]
            val_labels = [torch.from_numpy(a).long().to(dev) for a in z["val_labels"]]
        else:       # held-out seeded tiles (synthetic) / the first tiles of the file
            vi, vl = (synthetic_tiles(args.val_tiles, config.DATA.CROP_SIZE, config.DATA.NUM_CLASSES, args.seed + 7919, dev)
                      if args.data == "synthetic" else (images[:args.val_tiles], labels[:args.val_tiles]))
            val_images, val_labels = list(vi), list(vl)
        if list(config.VAL.STRIDE_SIZE) == [320, 320] and list(config.VAL.CROP_SIZE)[0] < 320:
            config.VAL.STRIDE_SIZE = list(config.VAL.CROP_SIZE)   # the default stride > crop leaves NaN stripes
    best_mean_iou, best_acc, best_model_iter = -1.0, -1.0, -1
    iters_per_epoch = max(len(sampler), 1)
    total_epoch = iters // iters_per_epoch
    reader_cost, batch_cost = TimeAverager(), TimeAverager()
    save_models = deque()
    avg_loss, cur_iter, epoch = 0.0, start_iter, 0
    batch_start = time.time()
    pending = []
    while cur_iter < iters:
        sampler.set_epoch(epoch)
        epoch += 1
        for idx in sampler:
            if cur_iter >= iters:
                break
            cur_iter += 1
            reader_cost.record(time.time() - batch_start)
            lr = optimizer.get_lr()
            if loader is not None:
                bx, by = next(loader)       # decoded / augmented by the reader threads, already on the device
            else:
                ib = torch.as_tensor(idx, device=dev)
                bx, by = images[ib], labels[ib]
            loss_t = engine.step(bx, by)
            pending.append(loss_t.clone())          # no device->host sync per step (the reference syncs here, :160)
            batch_cost.record(time.time() - batch_start, num_samples=bs)
            if cur_iter % config.LOGGING_INFO_FREQ == 0:
                avg_loss = float(torch.stack(pending).mean().item())
                pending = []
                if rank == 0:
                    print("[TRAIN] Epochs: {}/{}, iter: {}/{}, loss: {:.4f}, lr: {:.8f}, batch_cost: {:.4f}, reader_cost: {:.5f}, ips: {:.4f} samples/sec | ETA {}".format(
                        (cur_iter - 1) // iters_per_epoch + 1, total_epoch + 1, cur_iter, iters, avg_loss, lr, batch_cost.get_average(),
                        reader_cost.get_average(), batch_cost.get_ips_average() * nranks, calculate_eta(iters - cur_iter, batch_cost.get_average())), flush=True)
                reader_cost.reset()
                batch_cost.reset()
            checkpoint_now = cur_iter % config.SAVE_FREQ_CHECKPOINT == 0 or cur_iter == iters
            mean_iou = None
            if checkpoint_now and val_images is not None:       # every rank takes part (train.py:187-195)
                from .val import evaluate
                val_time_cost, mean_iou, acc, kap, class_iou, class_acc, class_f1, mean_f1 = evaluate(model, val_images, val_labels, config, rank, nranks)
                if rank == 0:
                    print("Val_time_cost:   {}".format(val_time_cost))
                    print("In this val: mIoU {:.4f},  Acc: {:.4f}, F1-Score:{:.4f}".format(mean_iou, acc, mean_f1))
                    print("Current best_mIoU: {:.4f},  Acc: {:.4f}, iter: {}".format(best_mean_iou, best_acc, best_model_iter), flush=True)
                model.train()
            if checkpoint_now and rank == 0:
                path = os.path.join(config.SAVE_DIR, "iter_{}_state.pt".format(cur_iter))
                torch.save({"model": {k: v.detach().cpu().contiguous() for k, v in model.state_dict().items()},
                            "optimizer": {k: ({n: t.cpu() for n, t in v.items()} if isinstance(v, dict) else v) for k, v in optimizer.state_dict().items()},
                            "iter": cur_iter}, path)
                # and the weights alone in the reference's own format (train.py:200-203: iter_{n}_model_state.pdparams)
                from .src.utils.checkpoint import save_pdparams
                pd_path = os.path.join(config.SAVE_DIR, "iter_{}_model_state.pdparams".format(cur_iter))
                save_pdparams(model.state_dict(), pd_path)
                save_models.append((path, pd_path))
                print("saving the weights of model to {}".format(pd_path))
                if len(save_models) > config.KEEP_CHECKPOINT_MAX > 0:      # both files of the oldest checkpoint (train.py:210-213)
                    for old in save_models.popleft():
                        os.remove(old)
                if mean_iou is not None and mean_iou > best_mean_iou:       # train.py:215-229
                    best_mean_iou, best_acc, best_model_iter = mean_iou, acc, cur_iter
                    save_pdparams(model.state_dict(), os.path.join(config.SAVE_DIR, "best_model.pdparams"))
                    print("\n[EVAL] The model with the best validation mIoU ({:.4f}) was saved at iter {}.".format(best_mean_iou, best_model_iter))
                    print("[EVAL] Images: {}  mIoU: {:.4f}  Acc: {:.4f}  Kappa: {:.4f}  mean_f1: {:.4f}".format(len(val_images), mean_iou, acc, kap, mean_f1))
                    print("[EVAL] Class IoU: " + str(np.round(class_iou, 4)))
                    print("[EVAL] Class Acc: " + str(np.round(class_acc, 4)))
                    print("[EVAL] Class F1-score: " + str(np.round(class_f1, 4)) + "\n", flush=True)
            batch_start = time.time()
    torch.cuda.synchronize()
    total = sum(int(np.prod(p.shape)) for p in model.parameters())
    if rank == 0:
        print("Total params: {}".format(total))
    return model


if __name__ == "__main__":
    main()
