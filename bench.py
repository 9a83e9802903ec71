#!/usr/bin/env python3
"""bench.py -- training tiles/sec of the EMRT hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One step = forward + CE/aux-CE loss + backward + (RCCL gradient all-reduce when N > 1) + global-norm clip + SGD-momentum
+ weight re-pack, on a synthetic batch of 8 normalised 256x256x3 tiles per GPU that is already resident in HBM
(workload = BASELINE.json configs[1]: EMRT ResNet-50, Potsdam 256x256, 6 classes, batch 8, bf16 storage / fp32
accumulate).  Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline       -- the dominant kernel of the step (the MFMA implicit-GEMM convolution), algorithmic FLOPs / HIP-event time
  roofline_msda  -- the deformable-attention gather kernel against the HBM roofline (BASELINE north_star's 85 % target)
  cpu_baseline   -- the oracle (torch-CPU fp32 restatement of the reference) timed on the host cores, rank 0, N = 1 only
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

FLOP_PER_TILE_FWD_BWD = 235.0e9   # 78.34 GFLOP forward, x3 for forward + backward, 256x256, 6 classes
PEAK_BF16_TFLOPS = 2500.0         # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_HBM_GBPS = 8000.0            # HBM3E spec (6.3 TB/s achievable)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def conv_flops(name, a):
    """Algorithmic FLOPs of one emrt_conv2d / emrt_conv2d_wgrad call from its C-ABI arguments."""
    if name == "emrt_conv2d":
        N, H, W, C = a[5:9]
        OH, OW, OC = a[11:14]
        KH, KW, stride, pad, mode = a[18:23]
        if mode == 0:
            return 2.0 * N * OH * OW * OC * KH * KW * C
        return 2.0 * N * H * W * C * KH * KW * OC          # dgrad: useful MACs = those of the forward conv
    if name in ("emrt_conv2d_group", "emrt_conv2d_bwd_group"):      # a[0]: ctypes array of descriptors, a[1]: how many
        f = 2.0 if name == "emrt_conv2d_group" else 4.0
        return sum(f * d.N * d.OH * d.OW * d.OC * d.KH * d.KW * d.C for d in list(a[0])[:a[1]])
    if name == "emrt_conv2d_bwd":           # data gradient + weight gradient of one layer
        N, H, W, C = a[9:13]
        OH, OW, OC = a[15:18]
        KH, KW = a[20:22]
        return 4.0 * N * OH * OW * OC * KH * KW * C
    N, H, W, C = a[3:7]
    OH, OW, OC = a[9:12]
    KH, KW = a[14:16]
    return 2.0 * N * OH * OW * OC * KH * KW * C


def msda_bytes(a, esz):
    """B*[Lv*256*e_v + Lq*288*4 (offsets f32) + Lq*144*4 (logits f32) + Lq*256*e_o] (+ reference points)."""
    B, Lq, Lv, M, D, L, P = a[9:16]
    tp = M * L * P
    return B * (Lv * M * D * esz + Lq * tp * 2 * 4 + Lq * tp * 4 + Lq * M * D * esz) + Lq * 2 * 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=8, help="tiles per GPU (BASELINE config: 8)")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--overlap", default="", choices=["", "pair", "deferred"], help="wgrad on a second stream (A/B experiment)")
    ap.add_argument("--two-phase", action="store_true", help="run the N>1 step structure (graphs around RCCL all-reduce) in a 1-rank group")
    ap.add_argument("--no-early-exchange", action="store_true", help="N>1: one all-reduce after the whole backward (A/B experiment)")
    ap.add_argument("--dump-calls", default=None, help="write the per-launch HIP-event timings of the profiled step to this file")
    args = ap.parse_args()

    from emrt_amd import _lib
    from emrt_amd.distributed import init_process_group
    from emrt_amd.engine import TrainEngine
    from emrt_amd.runtime import BF16, F32, ctx
    from emrt_amd.src.models.emrt import EMRT
    from emrt_amd.src.models.losses import MixSoftmaxCrossEntropyLoss
    from emrt_amd.src.models.solver import Momentum, PolynomialDecay

    rank, local_rank, world = init_process_group()
    if world != args.gpus:
        log("[bench] WORLD_SIZE=%d but --gpus %d; using WORLD_SIZE" % (world, args.gpus))
    dtype = BF16 if args.dtype == "bf16" else F32
    if os.environ.get("EMRT_ALL_RANKS_ON_GPU0"):      # test aid (with EMRT_DIST_BACKEND=gloo): every rank on device 0
        local_rank = 0
    dev = torch.device("cuda", local_rank)
    torch.manual_seed(1234)
    model = EMRT(num_classes=6, backbone="resnet50")
    model.to_hip(str(dev), dtype, seed=1234 + rank)
    opt = Momentum(model, PolynomialDecay(0.01, 160000, 0.0, 0.9), momentum=0.9, weight_decay=1e-4, grad_clip=1.0)
    loss_fn = MixSoftmaxCrossEntropyLoss(ignore_index=255, aux=True, aux_weight=0.4)
    if args.two_phase and world == 1 and not torch.distributed.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    eng = TrainEngine(model, opt, loss_fn, world, use_graph=not args.no_graph, overlap=args.overlap or False,
                      two_phase=True if args.two_phase else None, early_exchange=not args.no_early_exchange)
    g = torch.Generator().manual_seed(1234 + rank)
    B, S = args.batch, args.size
    images = torch.randn(B, 3, S, S, generator=g).to(dev)
    labels = torch.randint(0, 6, (B, S, S), generator=g)
    labels[torch.rand(B, S, S, generator=g) < 0.02] = 255
    labels = labels.to(dev)

    # setup (untimed, not part of the W warm-up steps): eager steps + hipGraph capture
    for _ in range(eng.warmup_eager + 1):
        eng.step(images, labels)
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        eng.step(images, labels)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss_t = eng.step(images, labels)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_val = float(loss_t.item())
    tiles_per_s = world * B * args.steps / elapsed

    result = None
    if rank == 0:
        # ---- live per-kernel timing: one extra eager step with a HIP event pair around every C-ABI launch --------------
        L = _lib.lib()
        eng_prof = TrainEngine(model, opt, loss_fn, 1, use_graph=False, overlap=False)   # serialised: per-kernel durations
        eng_prof.reducer = None
        c = ctx()
        c.keepalive = []                       # nothing allocated during the recorded step is freed until the replay is done
        L.start_record()
        eng_prof._eager_step(images, labels) if world == 1 else eng_prof._fwd_bwd(images, labels)
        rec = L.stop_record()
        torch.cuda.synchronize()
        L.replay(rec)                          # backlog: the host gets ~20 ms ahead of the GPU
        calls = L.replay(rec, timed=True)      # HIP events on the launch stream around every launch
        c.keepalive = None
        if args.dump_calls:
            with open(args.dump_calls, "w") as f:
                for name, a, ms in calls:
                    vals = [x.value if hasattr(x, "value") else x for x in a]
                    extra = ""
                    if name == "emrt_conv2d":
                        extra = "mode%d N%d in%dx%dx%d out%dx%dx%d k%d s%d gflop %.2f" % (vals[22], vals[5], vals[6], vals[7], vals[8], vals[11], vals[12], vals[13], vals[18], vals[20], conv_flops(name, vals) / 1e9)
                    elif name == "emrt_conv2d_bwd":
                        extra = "N%d in%dx%dx%d out%dx%dx%d k%d s%d gflop %.2f" % (vals[9], vals[10], vals[11], vals[12], vals[15], vals[16], vals[17], vals[20], vals[22], conv_flops(name, vals) / 1e9)
                    elif name == "emrt_conv2d_wgrad":
                        extra = "N%d in%dx%dx%d out%dx%dx%d k%d s%d gflop %.2f" % (vals[3], vals[4], vals[5], vals[6], vals[9], vals[10], vals[11], vals[14], vals[16], conv_flops(name, vals) / 1e9)
                    elif name in ("emrt_conv2d_group", "emrt_conv2d_bwd_group"):
                        extra = "n=%d " % vals[1] + " ".join("%dx%dx%d->%d k%d" % (d.H, d.W, d.C, d.OC, d.KH) for d in list(vals[0])[:vals[1]]) + " gflop %.2f" % (conv_flops(name, vals) / 1e9)
                    else:       # integer arguments only: enough to recognise the layer
                        extra = " ".join(str(v) for v in vals if isinstance(v, int) and not isinstance(v, bool) and abs(v) < (1 << 31))
                    f.write("%-26s %9.4f ms  %s\n" % (name, ms, extra))
        fam = {}
        for name, a, ms in calls:
            f = fam.setdefault(name, [0, 0.0, 0.0])
            f[0] += 1
            f[1] += ms
            if name in ("emrt_conv2d", "emrt_conv2d_wgrad", "emrt_conv2d_bwd", "emrt_conv2d_group", "emrt_conv2d_bwd_group"):
                f[2] += conv_flops(name, [x.value if hasattr(x, "value") else x for x in a])
        total_ms = sum(v[1] for v in fam.values())
        top = sorted(fam.items(), key=lambda kv: -kv[1][1])
        log("[bench] per-launch HIP-event time of one replayed step: %.2f ms over %d launches" % (total_ms, len(calls)))
        for name, (cnt, ms, fl) in top[:12]:
            log("    %-28s %5d calls %9.3f ms %5.1f%%%s" % (name, cnt, ms, 100 * ms / total_ms, "  %.1f TFLOP/s" % (fl / ms / 1e9) if fl else ""))
        peak = PEAK_BF16_TFLOPS if dtype == BF16 else PEAK_F32_MFMA_TFLOPS
        gemm_fams = {"emrt_conv2d_group": "igemm_group_kernel (emrt_conv2d_group: per-level encoder convs)",
                     "emrt_conv2d_bwd_group": "bwd_group_kernel (emrt_conv2d_bwd_group)",
                     "emrt_conv2d": "igemm_kernel (emrt_conv2d: forward convs / linears)",
                     "emrt_conv2d_bwd": "igemm_kernel + wgrad_kernel (emrt_conv2d_bwd: data + weight gradients, paired launch for small layers)",
                     "emrt_conv2d_wgrad": "wgrad_kernel (emrt_conv2d_wgrad)"}
        dom_name = max(gemm_fams, key=lambda k: fam.get(k, [0, 0.0, 0.0])[1])
        cnt, ms, fl = fam[dom_name]
        ach = fl / ms / 1e9
        all_ms = sum(fam.get(k, [0, 0.0, 0.0])[1] for k in gemm_fams)
        all_fl = sum(fam.get(k, [0, 0.0, 0.0])[2] for k in gemm_fams)
        roofline = {"kernel": gemm_fams[dom_name], "bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "traffic": None, "launches_per_step": cnt, "avg_launch_us": round(1e3 * ms / cnt, 2),
                    "algorithmic_gflop_per_step": round(fl / 1e9, 1), "share_of_step_kernel_time": round(ms / total_ms, 3),
                    "all_gemm_tflops": round(all_fl / all_ms / 1e9, 2), "all_gemm_share_of_step_kernel_time": round(all_ms / total_ms, 3),
                    "method": "recorded launches of one step replayed back-to-back behind a backlog, HIP-event pair on the launch stream around each launch"}
        # HBM traffic per launch from the committed rocprofv3 PMC passes (bench.py cannot profile itself); bf16 B=8 256^2 only
        pmc, pmc_src = None, os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r1k_pmc_traffic.json")
        if os.path.exists(pmc_src) and dtype == BF16 and B == 8 and S == 256:
            with open(pmc_src) as f:
                pmc = json.load(f)
            key = {"emrt_conv2d": "igemm_kernel", "emrt_conv2d_bwd": "bwd_pair_kernel", "emrt_conv2d_wgrad": "wgrad_kernel"}.get(dom_name, "igemm_kernel")
            if key not in pmc:
                key = "igemm_kernel"
            roofline["traffic"] = int(pmc[key]["traffic_mb_corrected"] * 1e6)
            roofline["traffic_kernel"] = key
            roofline["traffic_unit"] = "bytes per launch (2*FETCH_SIZE + WRITE_SIZE, average over the step's launches)"
            roofline["traffic_source"] = "profiles/r1k_pmc_traffic.json: " + pmc["method"]
        esz = 2 if dtype == BF16 else 4
        enc = [(a, ms) for name, a, ms in calls if name == "emrt_msda_fwd" and (a[9].value if hasattr(a[9], "value") else a[9]) > 0]
        enc = [(a, ms) for a, ms in enc if a[10] == a[11]]     # Lq == Lv: encoder self-attention calls
        roofline_msda = None
        if enc:
            vals = [x.value if hasattr(x, "value") else x for x in enc[0][0]]
            by = msda_bytes(vals, esz)
            avg_ms = sum(ms for _, ms in enc) / len(enc)
            roofline_msda = {"kernel": "msda_fwd_kernel (encoder call, B=%d Lq=Lv=%d)" % (vals[9], vals[10]), "bound": "hbm",
                             "achieved": round(by / avg_ms / 1e6, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                             "frac": round(by / avg_ms / 1e6 / PEAK_HBM_GBPS, 4), "traffic": None,
                             "algorithmic_mbytes_per_launch": round(by / 1e6, 2), "avg_launch_us": round(1e3 * avg_ms, 2)}
            if pmc is not None:
                m = pmc["msda_fwd_kernel_encoder"]
                roofline_msda["traffic"] = int((m["fetch_mb_raw"] + m["write_mb"]) * 1e6)
                roofline_msda["traffic_unit"] = ("bytes per launch, FETCH_SIZE + WRITE_SIZE as counted; the gfx950 x2 read correction "
                                                 "(valid for 16-B/lane streams) gives the upper bound %d" % int(m["traffic_mb_corrected"] * 1e6))
                roofline_msda["traffic_source"] = "profiles/r1k_pmc_traffic.json"
        cpu_baseline = None
        if world == 1 and not args.no_cpu_baseline:
            cpu_baseline = run_cpu_baseline(B, S, args.cpu_threads)
        result = {
            "metric": "training tiles/sec at 256x256", "value": round(tiles_per_s, 2), "unit": "tiles/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "EMRT ResNet50, Potsdam 256x256, 6 classes, batch %d per GPU %s, fwd+bwd+SGD step (BASELINE configs[1])" % (B, args.dtype),
                       "global_batch": world * B, "tile": [S, S, 3], "parallelism": "dp%d" % world, "hipgraph": not args.no_graph},
            "end_to_end_tflops": round(tiles_per_s * FLOP_PER_TILE_FWD_BWD / 1e12, 2), "final_loss": round(loss_val, 4),
            "roofline": roofline, "roofline_msda": roofline_msda, "cpu_baseline": cpu_baseline,
        }
    if world > 1:
        torch.distributed.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


def run_cpu_baseline(B, S, threads):
    """The oracle's train step (fwd + loss + bwd + clip + SGD-momentum) on the host cores: 1 warm-up + 2 timed steps."""
    from oracle.emrt_torch import EMRT as OracleEMRT
    from oracle import train_ref
    n = threads or min(os.cpu_count() or 1, 64)
    torch.set_num_threads(n)
    torch.manual_seed(1234)
    ref = OracleEMRT(6, "resnet50").train()
    opt = train_ref.MomentumRef(list(ref.named_parameters()), 0.9, 1e-4, 1.0)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(B, 3, S, S, generator=g)
    lab = torch.randint(0, 6, (B, S, S), generator=g)
    train_ref.train_step(ref, opt, x, lab, 0)
    t0 = time.perf_counter()
    steps = 2
    for i in range(steps):
        train_ref.train_step(ref, opt, x, lab, i + 1)
    dt = time.perf_counter() - t0
    return {"value": round(B * steps / dt, 3), "unit": "tiles/s", "cores": n, "kind": "port",
            "sample": "%d full train steps (fwd+bwd+clip+SGD) of the torch-CPU fp32 oracle at batch %d, %dx%d, after 1 warm-up step; %.1f s" % (steps, B, S, S, dt)}


if __name__ == "__main__":
    main()
