#!/usr/bin/env python3
"""bench.py -- tiles/sec of the EMRT hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus 1] [--steps 50] [--warmup 10]                     # BASELINE configs[1]: the headline line
    python bench.py --config cfg3                                             # configs[2]: LoveDA 512x512, 7 classes, batch 4
    python bench.py --config cfg5                                             # configs[4]: 1024x1024 sliding-window inference, fp16
    python bench.py --dtype fp32                                              # cfg2 in the reference's own precision
    python bench.py --gpus N                                                  # N > 1 without a launcher: starts N rank processes itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Multi-GPU (reference: paddle.DataParallel with ranks from the launcher, semantic_segmentation/train.py:116-123): one process per
GPU over RCCL.  Under a launcher (RANK / WORLD_SIZE in the environment) this process IS one rank and WORLD_SIZE must equal --gpus.
Without one, `--gpus N` makes this process a pure parent: it touches no GPU, starts N fresh children of itself with RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_* set, relays rank 0's JSON line and exits non-zero if any rank fails.

Training configs: one step = forward + CE/aux-CE loss + backward + (RCCL gradient all-reduce when N > 1) + global-norm clip +
SGD-momentum + weight re-pack on a synthetic batch that is already resident in HBM.  cfg5: one step = one 1024x1024 image =
16 windows of 256x256 evaluated as ONE batch through the fp16 model + the window glue kernels, replayed from a hipGraph.
Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline       -- the dominant kernel family of the step (MFMA implicit-GEMM convolutions), algorithmic FLOPs / HIP-event time
  roofline_msda  -- the deformable-attention gather kernel (encoder call of THIS config) against the HBM roofline
  cpu_baseline   -- the oracle (torch-CPU fp32 restatement of the reference) timed on the host cores, rank 0, N = 1 only
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

# forward GFLOP per tile; training = 3x (forward + backward)
CONFIGS = {
    "cfg2": dict(batch=8, size=256, ncls=6, mode="train", dtype="bf16", fwd_gflop=78.34,
                 name="EMRT ResNet50, Potsdam 256x256, 6 classes, batch 8 per GPU, fwd+bwd+SGD step (BASELINE configs[1])"),
    "cfg3": dict(batch=4, size=512, ncls=7, mode="train", dtype="bf16", fwd_gflop=312.0,
                 name="EMRT ResNet50, LoveDA 512x512, 7 classes, batch 4 per GPU, fwd+bwd+SGD step (BASELINE configs[2])"),
    "cfg5": dict(batch=16, size=256, ncls=6, mode="infer", dtype="fp16", fwd_gflop=78.34 - 1.21, image=1024,
                 name="EMRT ResNet50, 1024x1024 image, sliding window crop 256 stride 256 = 16 windows as one batch, eval (BASELINE configs[4])"),
}
PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3}   # MI355X dense MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBPS = 8000.0            # HBM3E spec (6.3 TB/s achievable)
PEAK_LDS_GBPS = 128 * 256 * 2.4   # LDS gather: 128 B/clk/CU (MI355X_MICROARCH.md, LDS: ds_read_b32 rate; random 64-B pixel reads get no more) x 256 CUs x 2.4 GHz
GEMM_FAMILIES = {"emrt_conv2d_group": "igemm_group_kernel (emrt_conv2d_group: per-level encoder convs)",
                 "emrt_conv2d_bwd_group": "igemm_group_kernel<mode 1> (emrt_conv2d_bwd_group: data gradients of the per-level convs)",
                 "emrt_conv2d": "igemm_kernel / igemm8p_kernel / igemm_xk_kernel (emrt_conv2d: forward convs / linears)",
                 "emrt_conv2d_drop": "igemm_drop_kernel (emrt_conv2d_drop: dropout(relu(linear1)) of the FFN, the mask drawn in the epilogue)",
                 "emrt_conv2d_bna": "igemm_bna_kernel / igemm_xk_bna_kernel (emrt_conv2d_bna: forward convs that apply their input's BatchNorm + ReLU on load)",
                 "emrt_conv2d_bwd": "igemm_kernel / igemm8p_kernel / igemm_xk_kernel mode 1 (emrt_conv2d_bwd: data gradients; thin_bwd_kernel for the classifiers)",
                 "emrt_conv2d_wgrad": "wgrad_kernel (emrt_conv2d_wgrad)",
                 "emrt_bn_pointwise_fwd": "thin_fwd_bn_kernel (emrt_bn_pointwise_fwd: the classifier with its BatchNorm operand)",
                 "emrt_bn_pointwise_bwd": "thin_bwd_kernel (emrt_bn_pointwise_bwd: the classifier's data + weight gradient in one pass)",
                 "emrt_conv2d_wgrad_group": "wgrad_group_kernel / wgrad8p_kernel (emrt_conv2d_wgrad_group: the weight gradients of up to 24 layers per launch)"}
# the roofline's kernel family: every launch that computes a convolution / linear layer's BACKWARD (data gradient + weight gradient) --
# the same population of work whether a layer's two gradients share a launch (round 3's pair kernel) or not (round 4: batched dW)
CONV_BWD = ("emrt_conv2d_bwd", "emrt_conv2d_bwd_group", "emrt_conv2d_wgrad", "emrt_conv2d_wgrad_group", "emrt_bn_pointwise_bwd")
CONV_FWD = ("emrt_conv2d", "emrt_conv2d_drop", "emrt_conv2d_bna", "emrt_conv2d_group", "emrt_bn_pointwise_fwd")


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def conv_flops(name, a):
    """Algorithmic FLOPs of one emrt_conv2d / emrt_conv2d_wgrad call from its C-ABI arguments."""
    if name == "emrt_bn_pointwise_fwd":        # the classifier with its BatchNorm operand: N, HW, C, OC = a[8:12]
        return 2.0 * a[8] * a[9] * a[10] * a[11]
    if name == "emrt_bn_pointwise_bwd":        # data + weight gradient in one pass: N, HW, C, OC = a[13:17]
        return 4.0 * a[13] * a[14] * a[15] * a[16]
    if name == "emrt_conv2d_drop":             # (in, w, out, bias, M, C, ldin, OC, ldout, ...)
        return 2.0 * a[4] * a[5] * a[7]
    if name == "emrt_conv2d_bna":              # emrt_conv2d's leading arguments without `mode` (always forward)
        return 2.0 * a[5] * a[11] * a[12] * a[13] * a[18] * a[19] * a[8]
    if name == "emrt_conv2d":
        N, H, W, C = a[5:9]
        OH, OW, OC = a[11:14]
        KH, KW, stride, pad, mode = a[18:23]
        if mode == 0:
            return 2.0 * N * OH * OW * OC * KH * KW * C
        return 2.0 * N * H * W * C * KH * KW * OC          # dgrad: useful MACs = those of the forward conv
    if name in ("emrt_conv2d_group", "emrt_conv2d_bwd_group", "emrt_conv2d_wgrad_group"):      # a[0]: ctypes array of descriptors, a[1]: how many
        ds = list(a[0])[:a[1]]
        f = 4.0 if (name == "emrt_conv2d_bwd_group" and ds[0].dw) else 2.0       # (a backward group without dw: the data gradients only)
        return sum(f * d.N * d.OH * d.OW * d.OC * d.KH * d.KW * d.C for d in ds)
    if name == "emrt_conv2d_bwd":           # data gradient (+ weight gradient unless dw == NULL: batched elsewhere) of one layer
        N, H, W, C = a[9:13]
        OH, OW, OC = a[15:18]
        KH, KW = a[20:22]
        return (4.0 if a[7] else 2.0) * N * OH * OW * OC * KH * KW * C
    N, H, W, C = a[3:7]
    OH, OW, OC = a[9:12]
    KH, KW = a[14:16]
    return 2.0 * N * OH * OW * OC * KH * KW * C


def conv_bytes(name, a, esz):
    """Algorithmic (compulsory) HBM bytes of one conv C-ABI call: every operand read once, every result written once (dW: fp32
    read-modify-write; masks / residuals / BatchNorm inputs read by a fused epilogue count too)."""
    def one(N, H, W, C, OH, OW, OC, KH, KW, fwd, dgrad, wgrad, extra_in=0, extra_out=0):
        x, y, w = N * H * W * C * esz, N * OH * OW * OC * esz, OC * KH * KW * C
        b = 0
        if fwd:
            b += x + w * esz + y + extra_in * y
        if dgrad:
            b += y + w * esz + x + extra_out * x
        if wgrad:
            b += x + y + 2 * 4 * w
        return b
    if name == "emrt_bn_pointwise_fwd":        # raw map read once, weight, logits written
        N, HW, C, OC = a[8:12]
        return N * HW * C * esz + OC * C * esz + N * HW * OC * esz
    if name == "emrt_bn_pointwise_bwd":        # raw map + dy read, masked input gradient written, dW read-modify-write
        N, HW, C, OC = a[13:17]
        return 2 * N * HW * C * esz + N * HW * OC * esz + OC * C * esz + 2 * 4 * OC * C
    if name == "emrt_conv2d_drop":
        return a[4] * a[5] * esz + a[7] * a[5] * esz + a[4] * a[7] * esz
    if name == "emrt_conv2d_bna":              # the raw map read, the normalised map and the output written, the weight read (+ a residual)
        N, H, W, C = a[5:9]
        OH, OW, OC = a[11:14]
        return one(N, H, W, C, OH, OW, OC, a[18], a[19], True, False, False, extra_in=1 if a[4] else 0) + N * H * W * C * esz
    if name == "emrt_conv2d":
        N, H, W, C = a[5:9]
        OH, OW, OC = a[11:14]
        KH, KW, stride, pad, mode = a[18:23]
        extra = (1 if a[4] else 0) + (1 if a[26] else 0)          # residual, mask
        if mode == 0:
            return one(N, H, W, C, OH, OW, OC, KH, KW, True, False, False, extra_in=extra)
        return N * H * W * C * esz + C * KH * KW * OC * esz + N * OH * OW * OC * esz * (1 + extra)      # dgrad call: `in` is dY, `out` is dX
    if name in ("emrt_conv2d_group", "emrt_conv2d_bwd_group", "emrt_conv2d_wgrad_group"):
        ds = list(a[0])[:a[1]]
        if name == "emrt_conv2d_group":
            return sum(one(d.N, d.H, d.W, d.C, d.OH, d.OW, d.OC, d.KH, d.KW, True, False, False, extra_in=1 if d.residual else 0) for d in ds)
        if name == "emrt_conv2d_wgrad_group":
            return sum(one(d.N, d.H, d.W, d.C, d.OH, d.OW, d.OC, d.KH, d.KW, False, False, True) for d in ds)
        return sum(one(d.N, d.H, d.W, d.C, d.OH, d.OW, d.OC, d.KH, d.KW, False, True, bool(d.dw), extra_out=1 if d.accumulate else 0) for d in ds)
    if name == "emrt_conv2d_bwd":
        N, H, W, C = a[9:13]
        OH, OW, OC = a[15:18]
        KH, KW = a[20:22]
        extra = (1 if a[6] else 0) + (1 if a[25] else 0) + (1 if a[29] else 0) + (1 if a[32] else 0)     # accumulate, mask, stat_x, addend
        return one(N, H, W, C, OH, OW, OC, KH, KW, False, True, bool(a[7]), extra_out=extra)
    N, H, W, C = a[3:7]
    OH, OW, OC = a[9:12]
    KH, KW = a[14:16]
    return one(N, H, W, C, OH, OW, OC, KH, KW, False, False, True)


def msda_bytes(a, esz):
    """B*[Lv*256*e_v + Lq*288*4 (offsets f32) + Lq*144*4 (logits f32) + Lq*256*e_o] (+ reference points)."""
    B, Lq, Lv, M, D, L, P = a[9:16]
    tp = M * L * P
    return B * (Lv * M * D * esz + Lq * tp * 2 * 4 + Lq * tp * 4 + Lq * M * D * esz) + Lq * 2 * 4


def msda_bwd_bytes(a, esz):
    """Compulsory HBM bytes of one emrt_msda_bwd call (same accounting as SURVEY.md 8(d) uses for the forward): read value, the fp32
    offsets | logits rows and dout; write dvalue and the offset / logit gradients (compute dtype when the projection's GEMM reads that)."""
    B, Lq, Lv, M, D, L, P = a[13:20]
    tp = M * L * P
    doffw_esz = esz if a[11] else 4
    return B * (Lv * M * D * esz + Lq * tp * 3 * 4 + Lq * M * D * esz + Lv * M * D * esz + Lq * tp * 3 * doffw_esz)


def vals_of(a):
    return [x.value if hasattr(x, "value") else x for x in a]


def family_table(calls):
    fam = {}
    for name, a, ms in calls:
        f = fam.setdefault(name, [0, 0.0, 0.0])
        f[0] += 1
        f[1] += ms
        if name in GEMM_FAMILIES:
            f[2] += conv_flops(name, vals_of(a))
    return fam


def dump_calls(path, calls, esz=2):
    with open(path, "w") as f:
        for name, a, ms in calls:
            vals = vals_of(a)
            if name == "emrt_conv2d":
                extra = "mode%d N%d in%dx%dx%d out%dx%dx%d k%d s%d gflop %.2f" % (vals[22], vals[5], vals[6], vals[7], vals[8], vals[11], vals[12], vals[13], vals[18], vals[20], conv_flops(name, vals) / 1e9)
            elif name == "emrt_conv2d_drop":
                extra = "linear+relu+dropout M%d %d->%d gflop %.2f" % (vals[4], vals[5], vals[7], conv_flops(name, vals) / 1e9)
            elif name == "emrt_conv2d_bna":
                extra = "bn+relu on load N%d in%dx%dx%d out%dx%dx%d k%d gflop %.2f" % (vals[5], vals[6], vals[7], vals[8], vals[11], vals[12], vals[13], vals[18], conv_flops(name, vals) / 1e9)
            elif name == "emrt_conv2d_bwd":
                extra = "N%d in%dx%dx%d out%dx%dx%d k%d s%d gflop %.2f" % (vals[9], vals[10], vals[11], vals[12], vals[15], vals[16], vals[17], vals[20], vals[22], conv_flops(name, vals) / 1e9)
            elif name == "emrt_conv2d_wgrad":
                extra = "N%d in%dx%dx%d out%dx%dx%d k%d s%d gflop %.2f" % (vals[3], vals[4], vals[5], vals[6], vals[9], vals[10], vals[11], vals[14], vals[16], conv_flops(name, vals) / 1e9)
            elif name in ("emrt_conv2d_group", "emrt_conv2d_bwd_group", "emrt_conv2d_wgrad_group"):
                extra = "n=%d " % vals[1] + " ".join("%dx%dx%d->%d k%d" % (d.H, d.W, d.C, d.OC, d.KH) for d in list(vals[0])[:vals[1]]) + " gflop %.2f" % (conv_flops(name, vals) / 1e9)
            elif name == "emrt_bn_pointwise_fwd":
                extra = "classifier+BN N%d px%d %d->%d gflop %.2f" % (vals[8], vals[9], vals[10], vals[11], conv_flops(name, vals) / 1e9)
            elif name == "emrt_bn_pointwise_bwd":
                extra = "classifier+BN N%d px%d %d->%d gflop %.2f" % (vals[13], vals[14], vals[15], vals[16], conv_flops(name, vals) / 1e9)
            else:       # integer arguments only: enough to recognise the layer
                extra = " ".join(str(v) for v in vals if isinstance(v, int) and not isinstance(v, bool) and abs(v) < (1 << 31))
            if name in GEMM_FAMILIES:
                extra += " bytes %d" % conv_bytes(name, vals, esz)
            f.write("%-26s %9.4f ms  %s\n" % (name, ms, extra))


def rooflines(calls, dtype_name, cfg_key, train):
    """-> (roofline of the dominant GEMM family, roofline of the MSDA encoder call, per-family log lines)."""
    fam = family_table(calls)
    esz = 4 if dtype_name == "fp32" else 2
    total_ms = sum(v[1] for v in fam.values())
    lines = ["[bench] per-launch HIP-event time of one replayed %s: %.2f ms over %d launches" % ("step" if train else "image", total_ms, len(calls))]
    for name, (cnt, ms, fl) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:12]:
        lines.append("    %-28s %5d calls %9.3f ms %5.1f%%%s" % (name, cnt, ms, 100 * ms / total_ms, "  %.1f TFLOP/s" % (fl / ms / 1e9) if fl else ""))
    peak = PEAK_TFLOPS[dtype_name]
    # training: the convolution / linear BACKWARD (data + weight gradients) is the dominant family; inference: the forward convolutions
    names = CONV_BWD if train else CONV_FWD
    cnt = sum(fam.get(k, [0, 0.0, 0.0])[0] for k in names)
    ms = sum(fam.get(k, [0, 0.0, 0.0])[1] for k in names)
    fl = sum(fam.get(k, [0, 0.0, 0.0])[2] for k in names)
    by = sum(conv_bytes(n_, vals_of(a_), esz) for n_, a_, _ in calls if n_ in names)
    ach = fl / ms / 1e9
    all_ms = sum(fam.get(k, [0, 0.0, 0.0])[1] for k in GEMM_FAMILIES)
    all_fl = sum(fam.get(k, [0, 0.0, 0.0])[2] for k in GEMM_FAMILIES)
    # what a HIP-event pair costs around a launch that does nothing: the per-launch figures above all contain it
    trivial = [ms_ for name, a, ms_ in calls if name in ("emrt_counter_add", "emrt_scalar_axpby")]
    triv_ms = min(trivial) if trivial else 0.0
    # ONE clock for `achieved` / `frac`: the kernel-duration clock (what rocprofv3 --kernel-trace reports per dispatch).  A HIP-event pair reads the
    # kernel's duration PLUS what the pair itself costs; that cost is measured in the same replay (the pair around nothing, 64 times: EMPTY_PAIR_MS)
    # and subtracted once per event pair -- a C-ABI call that makes several dispatches (a batched weight gradient: kernel + reduce) still has ONE pair.
    # per-launch cost of the clock: (sum of the per-launch readings - the same launches back to back inside ONE pair) / launches.  With it the corrected
    # times of ALL launches add up to the measured back-to-back time of the step -- the property rocprofv3's per-dispatch durations have (its gaps are ~0).
    # (The pair around NOTHING reads more, ~4.7 us: two adjacent markers do not overlap anything; it is reported, not used.)
    pair_ms = max(0.0, (total_ms - REPLAY_TOTAL_MS[0]) / max(1, len(calls))) if REPLAY_TOTAL_MS[0] > 0 else EMPTY_PAIR_MS[0]
    kern_ms = max(ms - cnt * pair_ms, 1e-6)
    all_cnt = sum(fam.get(k, [0, 0.0, 0.0])[0] for k in GEMM_FAMILIES)
    all_kern_ms = max(all_ms - all_cnt * pair_ms, 1e-6)
    total_kern_ms = max(total_ms - len(calls) * pair_ms, 1e-6)
    ach_k = fl / kern_ms / 1e9
    roofline = {"kernel": "; ".join(GEMM_FAMILIES[k] for k in names if k in fam), "bound": "mfma", "achieved": round(ach_k, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(ach_k / peak, 4), "traffic": None, "launches_per_step": cnt, "avg_launch_us": round(1e3 * kern_ms / cnt, 2),
                "clock": "kernel durations: HIP-event pair per launch minus event_pair_us, the per-launch share of (sum of the readings - the same launches "
                         "back to back inside one pair)",
                "event_pair_us": round(1e3 * pair_ms, 2), "empty_event_pair_us": round(1e3 * EMPTY_PAIR_MS[0], 2),
                "replay_back_to_back_ms": round(REPLAY_TOTAL_MS[0], 3),
                "algorithmic_gflop_per_step": round(fl / 1e9, 1), "algorithmic_bytes_per_launch": int(by / cnt),
                "achieved_event_clock": round(ach, 2), "frac_event_clock": round(ach / peak, 4),
                "share_of_step_kernel_time": round(kern_ms / total_kern_ms, 3),
                "all_gemm_tflops": round(all_fl / all_kern_ms / 1e9, 2), "all_gemm_share_of_step_kernel_time": round(all_kern_ms / total_kern_ms, 3),
                "step_kernel_time_ms": round(total_kern_ms, 3),
                "event_timed_trivial_launch_us": round(1e3 * triv_ms, 2) if trivial else None,
                "method": "recorded launches of one step replayed back-to-back behind a backlog, HIP-event pair on the launch stream around each launch; "
                          "kernel time of a launch = its pair's reading minus event_pair_us = (sum of all readings - replay_back_to_back_ms) / launches, "
                          "so that the corrected times of the whole step add up to its measured back-to-back time; achieved = algorithmic FLOPs / sum of "
                          "those kernel times -- the clock rocprofv3 --kernel-trace uses "
                          "(profiles/r6*_timeline_summary_cfg2.txt holds the same families from rocprofv3).  *_event_clock: the uncorrected readings; "
                          "event_timed_trivial_launch_us: what the pair reads around a one-thread kernel"}
    # HBM traffic per launch from the committed rocprofv3 PMC passes of this config (bench.py cannot profile itself): newest round
    pmc = None
    import glob
    suffix = "" if cfg_key == "cfg2" else "_" + cfg_key
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic%s.json" % suffix)))
    if cands and cfg_key in CONFIGS and dtype_name == CONFIGS[cfg_key]["dtype"]:
        with open(cands[-1]) as f:
            pmc = json.load(f)
        pmc_src = os.path.relpath(cands[-1], ROOT)
    if pmc is not None and "binding_roof" in pmc:
        # per layer max(FLOPs / MFMA peak, algorithmic bytes / HBM peak) over the measured kernel time, summed over every conv / linear call of a step
        # (tools/pmc_summary.py layer table of the committed profile: a profile figure, not a live one)
        roofline["binding_roof_frac"] = pmc["binding_roof"]["frac"]
        roofline["binding_roof"] = dict(pmc["binding_roof"], source=pmc_src)
    if pmc is not None:
        key = "conv_backward_family" if train else "conv_forward_family"
        if key in pmc:
            # the committed profile is only this run's traffic if it was taken on the same launch list: the profile records how many C-ABI
            # calls of the family its step made (tools/pmc_summary.py --calls), this run knows its own count
            prof_calls = pmc[key].get("calls_per_step")
            if prof_calls is not None and prof_calls != cnt:
                roofline["traffic_note"] = ("%s was taken on a different launch list (%d %s calls per step there, %d in this run): traffic left out -- "
                                            "re-run tools/r5/profile_round.sh" % (pmc_src, prof_calls, key, cnt))
                pmc = None
            else:
                roofline["traffic"] = int(pmc[key]["traffic_mb_corrected"] * 1e6)
                roofline["traffic_over_algorithmic"] = round(roofline["traffic"] * pmc[key]["dispatches_per_step"] / by, 3)
                roofline["traffic_kernels"] = pmc[key].get("kernels")
                roofline["traffic_unit"] = "bytes per kernel dispatch (2*FETCH_SIZE + WRITE_SIZE, average over the family's dispatches of a step)"
                roofline["traffic_source"] = pmc_src + ": " + pmc.get("method", "")
                roofline["traffic_calls_checked"] = prof_calls is not None
    esz = 4 if dtype_name == "fp32" else 2
    enc = [(vals_of(a), ms) for name, a, ms in calls if name == "emrt_msda_fwd"]
    enc = [(v, ms) for v, ms in enc if v[10] == v[11]]     # Lq == Lv: encoder self-attention calls
    roofline_msda = None
    if enc:
        v = enc[0][0]
        by = msda_bytes(v, esz)
        lds_by = v[9] * v[10] * v[12] * v[14] * v[15] * 4 * v[13] * esz       # B * Lq * M * L * P * 4 corners * D channels
        avg_ms = max(sum(ms for _, ms in enc) / len(enc) - pair_ms, 1e-6)      # (kernel-duration clock: the empty event pair's reading taken off)
        ideal_us = by / PEAK_HBM_GBPS / 1e3
        roofline_msda = {"kernel": "msda_fwd_lds_kernel / msda_fwd_kernel (encoder call, B=%d Lq=Lv=%d, %s)" % (v[9], v[10], dtype_name), "bound": "hbm",
                         "achieved": round(by / avg_ms / 1e6, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                         "frac": round(by / avg_ms / 1e6 / PEAK_HBM_GBPS, 4), "traffic": None,
                         "algorithmic_mbytes_per_launch": round(by / 1e6, 2), "avg_launch_us": round(1e3 * avg_ms, 2),
                         "ideal_us_at_peak": round(ideal_us, 2),
                         # the gather itself: every (query, head, level, point) reads 4 corners x 32 channels from the LDS-staged slab
                         # (SURVEY.md 8d "secondary figure"); the LDS serves 128 B/clk/CU for this access width on 256 CUs
                         "lds_bytes": int(lds_by), "lds_peak": PEAK_LDS_GBPS, "lds_achieved": round(lds_by / avg_ms / 1e6, 1),
                         "lds_frac": round(lds_by / avg_ms / 1e6 / PEAK_LDS_GBPS, 4), "lds_ideal_us_at_peak": round(lds_by / PEAK_LDS_GBPS / 1e3, 2),
                         # the same bytes against the conflict-free ds_read_b128 rate (256 B/clk/CU): what a layout whose 16-lane groups never collide would allow
                         "lds_peak_b128": 2 * PEAK_LDS_GBPS, "lds_frac_b128": round(lds_by / avg_ms / 1e6 / (2 * PEAK_LDS_GBPS), 4),
                         "binding_roof": "lds" if lds_by / PEAK_LDS_GBPS > by / PEAK_HBM_GBPS else "hbm",
                         "note": "kernel-duration clock (roofline.clock); at %.1f MB the HBM "
                                 "time is %.1f us and the LDS gather floor %.1f us, so the HBM fraction is bounded below %.2f by the gather alone and lower "
                                 "still by the launch's fixed latency; see DESIGN.md 5 for the per-shape table (bench.py --config cfg5 --batch 64 / 128 gives the large-batch rows)"
                                 % (by / 1e6, ideal_us, lds_by / PEAK_LDS_GBPS / 1e3, min(1.0, ideal_us / (lds_by / PEAK_LDS_GBPS / 1e3)))}
        encb = [(vals_of(a), ms) for name, a, ms in calls if name == "emrt_msda_bwd"]
        encb = [(vb, ms) for vb, ms in encb if vb[14] == vb[15]]     # Lq == Lv: encoder self-attention calls
        if encb:
            vb = encb[0][0]
            byb = msda_bwd_bytes(vb, esz)
            avg_b = max(sum(ms for _, ms in encb) / len(encb) - pair_ms, 1e-6)
            roofline_msda["backward"] = {
                "kernel": "emrt_msda_bwd: gradient kernel + value-gradient scatter (+ finalize), encoder call", "bound": "hbm",
                "achieved": round(byb / avg_b / 1e6, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": round(byb / avg_b / 1e6 / PEAK_HBM_GBPS, 4),
                "traffic": None, "algorithmic_mbytes_per_call": round(byb / 1e6, 2), "avg_call_us": round(1e3 * avg_b, 2),
                "note": "two or three dispatches per call inside ONE event pair (kernel-duration clock: roofline.clock); the value gradient is a matrix product at "
                        "cfg2 (msda_bwd_value_mfma_kernel: bound by the per-sample geometry on the VALU) and the LDS atomic scatter at cfg3 (7 cycles per "
                        "wave instruction); neither is bound by HBM: DESIGN.md 5.000"}
        if pmc is not None and "msda_bwd_encoder_call" in pmc and "backward" in roofline_msda:
            mb = pmc["msda_bwd_encoder_call"]
            roofline_msda["backward"]["traffic"] = int(mb["traffic_mb_corrected"] * 1e6)
            roofline_msda["backward"]["traffic_unit"] = ("bytes per encoder call (2*FETCH_SIZE + WRITE_SIZE summed over the call's kernels: %s)"
                                                         % ", ".join(k_["kernel"].split("<")[0] for k_ in mb["kernels"]))
        if pmc is not None and "msda_fwd_kernel_encoder" in pmc:
            m = pmc["msda_fwd_kernel_encoder"]
            roofline_msda["traffic"] = int((m["fetch_mb_raw"] + m["write_mb"]) * 1e6)
            roofline_msda["traffic_unit"] = ("bytes per launch, FETCH_SIZE + WRITE_SIZE as counted; the gfx950 x2 read correction "
                                             "(valid for 16-B/lane streams) gives the upper bound %d" % int(m["traffic_mb_corrected"] * 1e6))
            roofline_msda["traffic_source"] = pmc_src
    return roofline, roofline_msda, lines


EMPTY_PAIR_MS = [0.0]      # what a HIP-event pair reads around nothing, from the last timed_replay()
REPLAY_TOTAL_MS = [0.0]    # the same launch list back to back inside ONE event pair


def timed_replay(record_fn, world=1):
    """Record the C-ABI launches of record_fn(), replay them behind a backlog with a HIP-event pair around each."""
    from emrt_amd import _lib
    from emrt_amd.runtime import ctx
    L = _lib.lib()
    c = ctx()
    c.keepalive = []                       # nothing allocated during the recorded step is freed until the replay is done
    L.start_record()
    record_fn()
    rec = L.stop_record()
    torch.cuda.synchronize()
    L.replay(rec)                          # backlog: the host gets ~20 ms ahead of the GPU
    calls = L.replay(rec, timed=True)      # HIP events on the launch stream around every launch
    EMPTY_PAIR_MS[0] = L.empty_pair_ms
    REPLAY_TOTAL_MS[0] = L.replay_total_ms
    c.keepalive = None
    return calls


def spawn_ranks(n, argv):
    """Parent of a launcher-less `--gpus N` run: N children of this script, one rank each (emrt_amd.distributed.spawn_ranks).
    The parent makes no GPU call, relays rank 0's JSON line and fails loudly when a rank fails or when rank 0 reports a group
    of the wrong size."""
    from emrt_amd.distributed import spawn_ranks as launch
    codes, out0 = launch(n, [sys.executable, os.path.abspath(__file__)] + argv, capture_rank0=True, log=log)
    if any(codes):
        return codes[0] if all(c == codes[0] for c in codes) and 0 < codes[0] < 128 else 1
    line = None
    for ln in out0.splitlines():
        if ln.startswith("{"):
            line = ln
    if line is None:
        log("[bench] rank 0 printed no JSON line")
        return 1
    got = json.loads(line).get("n_gpus")
    if got != n:
        log("[bench] rank 0 reports n_gpus=%r, expected %d" % (got, n))
        return 1
    print(line, flush=True)
    return 0


def check_world(args):
    """Under a launcher WORLD_SIZE must be the --gpus the caller asked for: a silently smaller job would be a wrong number.  --gpus left out:
    the launcher's size is taken (as emrt_amd.train does).  Checked before the rendezvous and before any GPU call."""
    world = int(os.environ.get("WORLD_SIZE", 1))
    if args.gpus is None:
        args.gpus = world
    if world != args.gpus:
        raise SystemExit("[bench] WORLD_SIZE=%d but --gpus %d: start %d ranks (python bench.py --gpus %d starts them itself)"
                         % (world, args.gpus, args.gpus, args.gpus))


def _ride_along(args, key, steps, warm, extra=()):
    """One ride-along config of the default command in a child process (a fresh interpreter on the same GPU): returns its JSON line as a dict, or None when
    the child could not be run -- the caller then measures in-process.  The launcher's rank environment is not passed on (the child is a plain one-process run)."""
    import subprocess
    # (CPU baseline of a ride-along: 3 timed steps, median -- BASELINE.md 3's protocol; round 5 took one)
    cmd = [sys.executable, os.path.abspath(__file__), "--config", key, "--steps", str(steps), "--warmup", str(warm), "--no-other-configs", "--cpu-steps", "3"] + list(extra)
    if args.no_cpu_baseline and "--no-cpu-baseline" not in cmd:
        cmd.append("--no-cpu-baseline")
    if args.cpu_threads:
        cmd += ["--cpu-threads", str(args.cpu_threads)]
    if args.keep_gc:
        cmd.append("--keep-gc")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE", "GROUP_RANK",
                                                              "ROLE_RANK", "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID", "GROUP_WORLD_SIZE", "ROLE_NAME")}
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env)
    except (OSError, subprocess.TimeoutExpired) as e:
        sys.stderr.write("[bench] ride-along %s: child process not usable (%s): measuring in-process\n" % (key, e))
        return None
    sys.stderr.write(r.stderr)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    if r.returncode != 0 or not lines:
        sys.stderr.write("[bench] ride-along %s: child exited %d without a JSON line: measuring in-process\n" % (key, r.returncode))
        return None
    try:
        return json.loads(lines[-1])
    except ValueError:
        return None


def _pause_gc(args):
    """Host hygiene for the warm-up + timed region: collect now, then keep Python's collector off until the K steps are through (a generation-2
    collection of this process's heap is tens of ms of host time; at 12 ms per step that would be one stalled step).  No device work is skipped.
    It is NOT what stalled cfg3's first timed step in the default command (that needed the ride-along configs in child processes: _ride_along);
    `slowest_step` in the JSON line says which step was the slowest.  --keep-gc leaves the interpreter alone (A/B)."""
    import gc
    if getattr(args, "keep_gc", False) or not gc.isenabled():
        return False
    gc.collect()
    gc.disable()
    return True


def _resume_gc(paused):
    if paused:
        import gc
        gc.enable()


def dry_run(args):
    """`bench.py --gpus N --dry-run` (also under a launcher): every rank checks its environment, joins a gloo group on the CPU, builds the
    model's parameter layout on the host and derives what the N > 1 step would exchange -- flat-gradient ranges of the three backward
    segments, bucket slices, the SyncBatchNorm statistics group, its share of a tile set -- and the ranks compare their plans.  No GPU, no
    libemrt_hip.so: usable on a login node before the job is submitted.  Reference: train.py:116-123, dataloader.py:38-41."""
    import hashlib
    import torch.distributed as dist
    from emrt_amd import nn as hnn
    from emrt_amd.distributed import DistributedTileSampler, bucket_slices, env_rank_world
    from emrt_amd.runtime import BF16
    from emrt_amd.src.models import emrt as M
    rank, local_rank, world = env_rank_world()
    problems = []
    if world != args.gpus:
        problems.append("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    if not (0 <= rank < world) or not (0 <= local_rank < world):
        problems.append("RANK=%d LOCAL_RANK=%d outside [0, %d)" % (rank, local_rank, world))
    if world > 1:
        for k in ("MASTER_ADDR", "MASTER_PORT"):
            if not os.environ.get(k):
                problems.append("%s is not set" % k)
        if os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0") != "0":
            problems.append("HSA_ENABLE_IPC_MODE_LEGACY must be 0 (dmabuf IPC) for RCCL across processes on this driver")
    if world > 1 and not problems:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = CONFIGS[args.config]
    torch.manual_seed(1234)
    model = M.EMRT(num_classes=cfg["ncls"], backbone="resnet50")
    store = hnn.ParamStore(model, torch.device("cpu"), BF16, nograd_names=M.NOGRAD_PARAMS, fused_groups=model.fused_groups(),
                           lr_mult_names=model.lr_mult_names(), lr_mult=0.1)
    segs = store.segment_ranges(M.GRAD_SEGMENT_PREFIXES)
    covered = sorted(r for seg in segs for r in seg)
    if covered[0][0] != 0 or covered[-1][1] != store.n_train or any(a[1] != b[0] for a, b in zip(covered, covered[1:])):
        problems.append("the backward segments' ranges do not tile [0, n_train)")
    bucket = 32 * 1024 * 1024
    slices = [[b for a, e in seg for b in bucket_slices(e, bucket, a)] for seg in segs]
    sync = [n for n, m in model.named_modules() if isinstance(m, hnn.BatchNorm2D) and m.state.sync]
    n_tiles = 3456
    smp = DistributedTileSampler(n_tiles, cfg["batch"], rank, world, shuffle=True, drop_last=True, seed=1234)
    mine = [i for b in smp for i in b]
    plan = {"n_train": store.n_train, "segments": segs, "slices": slices, "sync_bn": sync, "batches_per_rank": len(smp)}
    digest = hashlib.sha256(json.dumps(plan, sort_keys=True).encode()).hexdigest()
    if world > 1 and dist.is_initialized():
        every = [None] * world
        dist.all_gather_object(every, (rank, digest, mine))
        if len({d for _, d, _ in every}) != 1:
            problems.append("ranks derived different exchange plans: %r" % [(r, d[:8]) for r, d, _ in every])
        flat = [i for _, _, m in every for i in m]
        if len(flat) != len(set(flat)) or len({len(m) for _, _, m in every}) != 1:
            problems.append("tile shards overlap or differ in size")
        if sorted(r for r, _, _ in every) != list(range(world)):
            problems.append("rank set %r" % sorted(r for r, _, _ in every))
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        out = {"dry_run": True, "n_gpus": world, "config": args.config, "ok": not problems, "problems": problems,
               "gradient_elements": store.n_train, "gradient_mbytes_fp32": round(4e-6 * store.n_train, 1),
               "exchange_ranges": [{"segment": name, "elements": sum(e - a for a, e in seg), "mbytes_fp32": round(4e-6 * sum(e - a for a, e in seg), 2),
                                    "ranges": len(seg), "all_reduce_launches": len(sl)}
                                   for name, seg, sl in zip(("heads + transformer + layer4", "layer3", "conv1 + layer1 + layer2"), segs, slices)],
               "sync_batchnorm_layers": sync, "sync_batchnorm_collectives_per_step": {"forward": 1, "backward": 1,
                    "note": "the five layers share one statistics all-reduce per direction (Fn.conv_bn_group + the auxiliary head joining the group)"},
               "tiles_per_rank_per_step": cfg["batch"], "global_batch": world * cfg["batch"],
               "sampler": {"tiles": n_tiles, "batches_per_rank_per_epoch": len(smp)}, "plan_sha256": digest}
        print(json.dumps(out), flush=True)
    if problems:
        log("[bench] dry run, rank %d: %s" % (rank, "; ".join(problems)))
        return 1
    return 0


def describe_group(rank, world, dev):
    """Rank 0: what the process group really is (a SCALE run is then self-evidencing)."""
    if rank != 0 or not torch.distributed.is_initialized():
        return
    backend = torch.distributed.get_backend()
    ver = ""
    if backend == "nccl":
        try:
            ver = " RCCL %s" % ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:      # version query only; never fatal
            ver = " (RCCL version unavailable: %s)" % e
    log("[bench] process group: backend %s%s, world size %d, rank 0 on %s, ranks %s" %
        (backend, ver, torch.distributed.get_world_size(), dev, "share GPU 0 (test aid)" if os.environ.get("EMRT_ALL_RANKS_ON_GPU0") else "one GPU each"))


def collective_probe(dev, world, n_elems, reps=5):
    """N > 1, rank 0's view: a bare all-reduce of the step's gradient payload (fp32, n_elems elements) timed on its own, so that a poor
    scaling number can be attributed (link time vs everything else) without a second run."""
    buf = torch.zeros(n_elems, dtype=torch.float32, device=dev)
    for _ in range(2):
        torch.distributed.all_reduce(buf)
    torch.cuda.synchronize()
    torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        torch.distributed.all_reduce(buf)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    nbytes = n_elems * 4
    return {"payload_mbytes": round(nbytes / 1e6, 1), "reps": reps, "ms": round(1e3 * dt, 3), "algbw_GBps": round(nbytes / dt / 1e9, 1),
            "busbw_GBps": round(nbytes / dt / 1e9 * 2 * (world - 1) / world, 1),
            "note": "bare torch.distributed.all_reduce (RCCL) of the whole flat gradient, nothing overlapping it; busbw = algbw * 2(N-1)/N"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks (default: WORLD_SIZE under a launcher, else 1)")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS), help="BASELINE.json config: cfg2 = configs[1] (headline), cfg3 = configs[2], cfg5 = configs[4]")
    ap.add_argument("--batch", type=int, default=0, help="tiles per GPU (default: the config's); cfg5: windows per image (the image becomes rows x cols crops)")
    ap.add_argument("--size", type=int, default=0)
    ap.add_argument("--dtype", default="", choices=["", "bf16", "fp16", "fp32"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="the default command also measures cfg3 and cfg5 (short runs) into other_configs; this skips them")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--overlap", default="", choices=["", "pair", "deferred"], help="wgrad on a second stream (A/B experiment)")
    ap.add_argument("--two-phase", action="store_true", help="run the N>1 step structure (graphs around RCCL all-reduce) in a 1-rank group")
    ap.add_argument("--no-early-exchange", action="store_true", help="N>1: one all-reduce after the whole backward (A/B experiment)")
    ap.add_argument("--two-phase-no-syncbn", action="store_true", help="--two-phase without the SyncBatchNorm collectives (A/B: what the graph cuts cost)")
    ap.add_argument("--dump-calls", default=None, help="write the per-launch HIP-event timings of the profiled step to this file")
    ap.add_argument("--exchange-noop", action="store_true", help="A/B: the N > 1 step structure with the gradient all-reduce switched off (NOT a valid throughput)")
    ap.add_argument("--inprocess-others", action="store_true", help="A/B: measure the ride-along configs in this process instead of in child processes")
    ap.add_argument("--cpu-steps", type=int, default=0, help="timed steps of the CPU baseline (0 = the config's default)")
    ap.add_argument("--keep-gc", action="store_true", help="A/B: leave Python's garbage collector enabled during the timed steps (default: collected before, paused during)")
    ap.add_argument("--dry-run", action="store_true", help="--gpus N without a GPU: start the N ranks, rendezvous over gloo, and check rank environment, "
                    "gradient-exchange ranges, SyncBatchNorm group membership and tile sharding; prints a JSON plan, launches no kernel")
    args = ap.parse_args()
    if (args.gpus or 1) > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        if args.dry_run:
            os.environ["EMRT_ALL_RANKS_ON_GPU0"] = "1"      # (the parent's "enough GPUs?" check does not apply: a dry run opens none)
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    check_world(args)
    if args.dry_run:
        sys.exit(dry_run(args))
    from emrt_amd.distributed import init_process_group
    if os.environ.get("EMRT_ALL_RANKS_ON_GPU0"):      # test aid (with EMRT_DIST_BACKEND=gloo): every rank on device 0
        os.environ["LOCAL_RANK"] = "0"
    rank, local_rank, world = init_process_group()
    dev = torch.device("cuda", local_rank)
    describe_group(rank, world, dev)
    env = dict(rank=rank, world=world, dev=dev)

    cfg = dict(CONFIGS[args.config])
    if args.batch:
        cfg["batch"] = args.batch
    if args.size:
        cfg["size"] = args.size
    dtype_name = args.dtype or cfg["dtype"]
    run = run_infer if cfg["mode"] == "infer" else run_train
    result = run(args, args.config, cfg, dtype_name, env, args.steps, args.warmup, cpu=not args.no_cpu_baseline, dump=args.dump_calls, cpu_steps=args.cpu_steps or None)
    # The driver only ever runs the default command: the other two single-GPU configurations of BASELINE.json ride along as short runs
    # (metric / config / value above stay configs[1]'s).  Not at N > 1, not for experiments that changed the workload.
    default_workload = (args.config == "cfg2" and not args.batch and not args.size and not args.dtype and not args.no_graph and not args.overlap
                        and not args.two_phase)
    if world == 1 and default_workload and not args.no_other_configs:
        others = {}
        # (30 timed steps after 8 warm-up steps each: round 4's 12-step mean moved 18 % on ONE stalled step in the driver's run)
        for key, steps, warm in (("cfg3", 30, 8), ("cfg5", 30, 8)):
            c2 = dict(CONFIGS[key])
            # each in a FRESH interpreter (a child process; this one keeps running and keeps its GPU context): run in-process after cfg2, cfg3's first timed
            # step stalled for 54 / 136 ms in three default runs of seven, never in eight stand-alone runs (DESIGN.md 5.000 "Bench hygiene"); --inprocess-others
            # restores the old behaviour, and a child that fails falls back to it
            r2 = None if args.inprocess_others else _ride_along(args, key, steps, warm)
            if r2 is None:
                r2 = (run_infer if c2["mode"] == "infer" else run_train)(args, key, c2, c2["dtype"], env, steps, warm, cpu=not args.no_cpu_baseline, dump=None, cpu_steps=3)
            others[key] = {k: r2[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "ms_per_step_median", "ms_per_step_max", "slowest_step", "value_at_median",
                                               "dtype", "config", "end_to_end_tflops", "loss_check", "roofline", "roofline_msda", "cpu_baseline") if k in r2}
        # the reference computes in fp32 throughout (paddle_EMRT.py has no AMP); BASELINE configs[1] names bf16, so the headline is bf16 -- the same
        # workload in the reference's own precision rides along every run (child process; no CPU leg: it is the same oracle step as cfg2's)
        r32 = _ride_along(args, "cfg2", 20, 5, extra=["--dtype", "fp32", "--no-cpu-baseline"])
        if r32 is not None:
            others["cfg2_fp32"] = {k: r32[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "ms_per_step_median", "ms_per_step_max", "slowest_step",
                                                        "value_at_median", "dtype", "config", "end_to_end_tflops", "loss_check", "roofline", "roofline_msda") if k in r32}
        result["other_configs"] = others
    if world > 1:
        torch.distributed.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    if rank == 0:
        # the JSON line goes out LAST: RCCL writes its version banner through C stdio, which is block-buffered on a pipe and would otherwise
        # be flushed after this line, at exit
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        print(json.dumps(result), flush=True)


def run_train(args, cfg_key, cfg, dtype_name, env, steps, warmup, cpu=True, dump=None, cpu_steps=None):
    """One training configuration: W warm-up + K timed steps (barrier + synchronize on both sides, max over ranks) -> rank 0's result dict."""
    if dtype_name == "fp16":
        raise SystemExit("fp16 is inference-only (include/emrt_hip.h); training configs run in bf16 or fp32")
    from emrt_amd.engine import TrainEngine
    from emrt_amd.runtime import BF16, F32
    from emrt_amd.src.models.emrt import EMRT
    from emrt_amd.src.models.losses import MixSoftmaxCrossEntropyLoss
    from emrt_amd.src.models.solver import Momentum, PolynomialDecay
    rank, world, dev = env["rank"], env["world"], env["dev"]
    dtype = BF16 if dtype_name == "bf16" else F32
    torch.manual_seed(1234)
    B, S, ncls = cfg["batch"], cfg["size"], cfg["ncls"]
    model = EMRT(num_classes=ncls, backbone="resnet50")
    model.to_hip(str(dev), dtype, seed=1234 + rank)
    opt = Momentum(model, PolynomialDecay(0.01, 160000, 0.0, 0.9), momentum=0.9, weight_decay=1e-4, grad_clip=1.0)
    loss_fn = MixSoftmaxCrossEntropyLoss(ignore_index=255, aux=True, aux_weight=0.4)
    if args.two_phase and world == 1 and not torch.distributed.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    eng = TrainEngine(model, opt, loss_fn, world, use_graph=not args.no_graph, overlap=args.overlap or False,
                      two_phase=True if args.two_phase else None, early_exchange=not args.no_early_exchange)
    if args.two_phase_no_syncbn:
        from emrt_amd.runtime import ctx as _ctx
        _ctx().sync_always = False
    if args.exchange_noop and eng.reducer is not None:
        eng.reducer.noop = True
    g = torch.Generator().manual_seed(1234 + rank)
    images = torch.randn(B, 3, S, S, generator=g).to(dev)
    labels = torch.randint(0, ncls, (B, S, S), generator=g)
    labels[torch.rand(B, S, S, generator=g) < 0.02] = 255
    labels = labels.to(dev)

    # setup (untimed, not part of the W warm-up steps): eager steps + hipGraph capture
    first_loss = None
    for _ in range(eng.warmup_eager + 1):
        lt = eng.step(images, labels)
        if first_loss is None:
            first_loss = float(lt.item())
    torch.cuda.synchronize()
    gc_paused = _pause_gc(args)          # (before the warm-up steps: the collection itself is host time during which the launch queue would drain)
    for _ in range(warmup):
        eng.step(images, labels)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
    torch.cuda.synchronize()
    barrier()
    # a device timestamp after every step (an event record on the step's stream: no host wait, nothing added to the timed region but the
    # records themselves): the mean below is the contract's number, median and max make a single stalled step visible next to it
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    # EMRT_BENCH_NO_STAGE=1 (A/B experiment, not a valid throughput): the steps read the captured buffers in place -- no per-step staging copies
    feed = (eng.images, eng.labels) if (os.environ.get("EMRT_BENCH_NO_STAGE") == "1" and getattr(eng, "images", None) is not None) else (images, labels)
    for i in range(steps):
        loss_t = eng.step(*feed)
        marks[i + 1].record()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    _resume_gc(gc_paused)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    per_step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
    loss_val = float(loss_t.item())
    tiles_per_s = world * B * steps / elapsed
    # the timed steps did real work on real weights: the loss of a repeated batch must be finite, positive, below the first step's, and inside
    # the band recorded for this workload (bf16, synthetic tiles, seed 1234: 2.95 at step 1, 2.87 after 16 steps, 2.72 after 43; ln(ncls) * 1.4
    # is the untrained value of CE + 0.4 aux CE).  A NaN, a frozen model or a diverging one fails the run instead of printing a number.
    import math
    untrained = 1.4 * math.log(ncls)
    band = (0.3 * untrained, 1.35 * untrained)
    # "below the first step's" compares two noisy losses (dropout is on; the recorded slope is ~0.005 per step): it is enforced once enough steps
    # were taken for the trend to clear the noise (>= 16), short profiling runs (--steps 2 --warmup 0) keep the NaN and band checks only
    decreasing_checked = eng.calls >= 16
    loss_ok = (loss_val == loss_val and first_loss == first_loss and band[0] < loss_val < band[1]
               and (not decreasing_checked or loss_val < first_loss * (1.0 + 1e-3)))
    if world > 1:      # every rank leaves together: a rank that raised alone would leave the others waiting in the next collective
        okt = torch.tensor([1.0 if loss_ok else 0.0], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(okt, op=torch.distributed.ReduceOp.MIN)
        loss_ok = bool(okt.item() > 0.5)
    loss_check = {"first_step": round(first_loss, 4), "final": round(loss_val, 4), "band": [round(band[0], 3), round(band[1], 3)],
                  "steps_taken": eng.calls, "decreasing_checked": decreasing_checked, "ok": bool(loss_ok)}
    if not loss_ok:
        raise SystemExit("[bench] loss check failed (on this or another rank): %r" % (loss_check,))
    probe = exposed = None
    if world > 1 and eng.reducer is not None and not args.exchange_noop:
        # what the gradient exchange costs the step: the same structure (graphs, SyncBatchNorm cuts, optimizer graph) with the all-reduce
        # switched off, 3 extra steps after the timed region (the ranks' weights drift apart meanwhile: nothing is measured after this)
        eng.reducer.noop = True
        eng.step(images, labels)
        torch.cuda.synchronize()
        barrier()
        t1 = time.perf_counter()
        for _ in range(3):
            eng.step(images, labels)
        torch.cuda.synchronize()
        barrier()
        noop_ms = 1e3 * (time.perf_counter() - t1) / 3
        eng.reducer.noop = False
        t = torch.tensor([noop_ms], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        noop_ms = float(t.item())
        ranges = eng.seg_ranges if eng.seg_ranges is not None else [[(0, model.store.n_train)]]
        exposed = {"ms_per_step_without_exchange": round(noop_ms, 3), "exchange_exposed_ms": round(1e3 * elapsed / steps - noop_ms, 3),
                   "range_payload_mbytes": [round(4e-6 * sum(e - a for a, e in seg), 2) for seg in ranges],
                   "exchange_dtype": eng.reducer.exchange_dtype,
                   "note": "3 extra steps with the all-reduce launches skipped (FlatGradReducer.noop, set here only); ranges in launch order: "
                           "heads + transformer + layer4 | layer3 | conv1 + layer1 + layer2 (the last one is the only exposed collective)"}
    if world > 1 and torch.distributed.get_backend() == "nccl":      # every rank takes part; rank 0 reports
        probe = collective_probe(dev, world, model.store.n_train)

    result = None
    if rank == 0:
        # ---- live per-kernel timing: one extra eager step with a HIP event pair around every C-ABI launch --------------
        eng_prof = TrainEngine(model, opt, loss_fn, 1, use_graph=False, overlap=False)   # serialised: per-kernel durations
        eng_prof.reducer = None
        # (the profiled step is serialised on ONE stream: the recorded launches are replayed without the fork / join events of the step prologue's
        # side stream, so that prologue runs on the main stream here)
        from emrt_amd.runtime import ctx as _ctx
        _side_was, _ctx().prologue_side = _ctx().prologue_side, False
        try:
            calls = timed_replay(lambda: eng_prof._eager_step(images, labels) if world == 1 else eng_prof._fwd_bwd(images, labels))
        finally:
            _ctx().prologue_side = _side_was
        if dump:
            dump_calls(dump, calls, 4 if dtype_name == "fp32" else 2)
        roofline, roofline_msda, lines = rooflines(calls, dtype_name, cfg_key if (B, S) == (CONFIGS[cfg_key]["batch"], CONFIGS[cfg_key]["size"]) else "custom", True)
        for ln in lines:
            log(ln)
        cpu_baseline = None
        if world == 1 and cpu:
            cpu_baseline = run_cpu_baseline_train(B, S, ncls, args.cpu_threads, timed_steps=cpu_steps or (3 if cfg_key == "cfg2" else 2))
        flop_tile = 3.0 * cfg["fwd_gflop"] * 1e9 * (S * S) / (CONFIGS[cfg_key]["size"] ** 2)
        result = {
            "metric": "training tiles/sec at %dx%d" % (S, S), "value": round(tiles_per_s, 2), "unit": "tiles/s", "n_gpus": world,
            "steps": steps, "warmup": warmup, "ms_per_step": round(1e3 * elapsed / steps, 3),
            "ms_per_step_median": round(statistics.median(per_step_ms), 3), "ms_per_step_max": round(max(per_step_ms), 3),
            "slowest_step": int(max(range(len(per_step_ms)), key=per_step_ms.__getitem__)),
            "value_at_median": round(world * B * 1e3 / statistics.median(per_step_ms), 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": dtype_name, "data": "synthetic",
            "config": {"workload": cfg["name"].replace("batch %d" % CONFIGS[cfg_key]["batch"], "batch %d" % B) + ", " + dtype_name,
                       "global_batch": world * B, "tile": [S, S, 3], "parallelism": "dp%d" % world, "hipgraph": not args.no_graph},
            "end_to_end_tflops": round(tiles_per_s * flop_tile / 1e12, 2), "final_loss": round(loss_val, 4), "loss_check": loss_check,
            "roofline": roofline, "roofline_msda": roofline_msda, "cpu_baseline": cpu_baseline,
        }
        if probe is not None:
            # what the exchange would cost if nothing hid it, next to what the step really paid over the 1-rank step structure
            result["collective"] = probe
        if exposed is not None:
            result["exchange"] = exposed
        if args.exchange_noop:
            result["invalid"] = "--exchange-noop: gradient all-reduce switched off (A/B of the step structure, not a throughput)"
    del eng
    return result


def _window_grid(n):
    """rows x cols = n windows, as square as possible."""
    r = int(n ** 0.5)
    while n % r:
        r -= 1
    return r, n // r


def run_infer(args, cfg_key, cfg, dtype_name, env, steps, warmup, cpu=True, dump=None, cpu_steps=None):
    """cfg5: sliding-window inference of 1024x1024 images (replicas only: --gpus N runs N independent replicas, no collective)."""
    from emrt_amd.runtime import BF16, F16, F32
    from emrt_amd.src.api.infer import SlidingWindowEngine, slide_inference
    from emrt_amd.src.models.emrt import EMRT
    rank, world, dev = env["rank"], env["world"], env["dev"]
    torch.manual_seed(1234)
    ncls, crop = cfg["ncls"], cfg["size"]
    rows, cols = (cfg["image"] // crop,) * 2 if cfg["batch"] == CONFIGS[cfg_key]["batch"] else _window_grid(cfg["batch"])
    img_h, img_w = rows * crop, cols * crop
    model = EMRT(num_classes=ncls, backbone="resnet50")
    model.to_hip(str(dev), {"bf16": BF16, "fp16": F16, "fp32": F32}[dtype_name], seed=1234 + rank)
    model.eval()
    model.compute_aux_in_eval = False        # every inference caller discards the auxiliary logits (infer.py:66)
    g = torch.Generator().manual_seed(1234 + rank)
    img = torch.randn(3, img_h, img_w, generator=g).to(dev)
    # BatchNorm running statistics from one pass in train-mode arithmetic would need fp16 backward entry points; the
    # throughput does not depend on their values: they stay at their initial (0, 1)
    eng = SlidingWindowEngine(model, (3, img_h, img_w), (crop, crop), (crop, crop), ncls, warmup=0 if args.no_graph else 1)
    if args.no_graph:
        eng.warmup = 1 << 30
    nwin = rows * cols
    for _ in range(3):
        eng(img)
    torch.cuda.synchronize()
    gc_paused = _pause_gc(args)
    for _ in range(warmup):
        eng(img)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
    torch.cuda.synchronize()
    barrier()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]      # device timestamps per image: see run_train
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(steps):
        pred = eng(img)
        marks[i + 1].record()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    _resume_gc(gc_paused)
    per_step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    tiles_per_s = world * nwin * steps / elapsed
    result = None
    if rank == 0:
        assert tuple(pred.shape) == (1, 1, img_h, img_w) and pred.dtype == torch.int32
        calls = timed_replay(lambda: slide_inference(model, [img], (crop, crop), (crop, crop), ncls))
        if dump:
            dump_calls(dump, calls, 4 if dtype_name == "fp32" else 2)
        roofline, roofline_msda, lines = rooflines(calls, dtype_name, cfg_key if nwin == CONFIGS[cfg_key]["batch"] else "custom", False)
        for ln in lines:
            log(ln)
        cpu_baseline = None
        if world == 1 and cpu:
            cpu_baseline = run_cpu_baseline_infer(ncls, crop, nwin, args.cpu_threads, timed=cpu_steps or 3)
        result = {
            "metric": "inference tiles/sec at %dx%d (sliding window over %dx%d images)" % (crop, crop, img_h, img_w),
            "value": round(tiles_per_s, 2), "unit": "tiles/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(1e3 * elapsed / steps, 3), "ms_per_step_median": round(statistics.median(per_step_ms), 3),
            "ms_per_step_max": round(max(per_step_ms), 3), "slowest_step": int(max(range(len(per_step_ms)), key=per_step_ms.__getitem__)),
            "value_at_median": round(world * nwin * 1e3 / statistics.median(per_step_ms), 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dtype_name, "data": "synthetic",
            "config": {"workload": cfg["name"].replace("1024x1024", "%dx%d" % (img_h, img_w)).replace("16 windows", "%d windows" % nwin) + ", " + dtype_name,
                       "windows_per_step": nwin, "image": [img_h, img_w, 3], "tile": [crop, crop, 3],
                       "parallelism": "replicas%d" % world, "hipgraph": not args.no_graph},
            "end_to_end_tflops": round(tiles_per_s * cfg["fwd_gflop"] * 1e9 / 1e12, 2),
            "roofline": roofline, "roofline_msda": roofline_msda, "cpu_baseline": cpu_baseline,
        }
    del eng
    return result


def run_cpu_baseline_train(B, S, ncls, threads, timed_steps=3):
    """The oracle's train step (fwd + loss + bwd + clip + SGD-momentum) on the host cores: 1 warm-up + `timed_steps` timed
    steps, the MEDIAN step time is reported (>= 3 timed steps for the headline config)."""
    from oracle.emrt_torch import EMRT as OracleEMRT
    from oracle import train_ref
    n = threads or min(os.cpu_count() or 1, 64)
    torch.set_num_threads(n)
    torch.manual_seed(1234)
    ref = OracleEMRT(ncls, "resnet50").train()
    opt = train_ref.MomentumRef(list(ref.named_parameters()), 0.9, 1e-4, 1.0)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(B, 3, S, S, generator=g)
    lab = torch.randint(0, ncls, (B, S, S), generator=g)
    train_ref.train_step(ref, opt, x[:2, :, :64, :64].contiguous(), lab[:2, :64, :64].contiguous(), 0)      # warm-up (thread pool, allocator, oneDNN primitives): 2 tiles of 64x64
    times = []
    for i in range(timed_steps):
        t0 = time.perf_counter()
        train_ref.train_step(ref, opt, x, lab, i + 1)
        times.append(time.perf_counter() - t0)
    med = statistics.median(times)
    return {"value": round(B / med, 3), "unit": "tiles/s", "cores": n, "kind": "port",
            "sample": "median of %d full train step(s) (fwd+bwd+clip+SGD) of the torch-CPU fp32 oracle at batch %d, %dx%d, after a small warm-up step "
                      "(2 tiles of 64x64); step times %s s" % (timed_steps, B, S, S, ["%.2f" % t for t in times])}


def run_cpu_baseline_infer(ncls, crop, nwin, threads, timed=3):
    """The oracle's eval forward over the windows of one image (fp32, the reference's precision), median of `timed` after 1 warm-up; a BOUNDED
    sample: at most 32 windows are evaluated (the rate per window does not depend on how many more there are)."""
    nwin = min(nwin, 32)
    from oracle.emrt_torch import EMRT as OracleEMRT
    n = threads or min(os.cpu_count() or 1, 64)
    torch.set_num_threads(n)
    torch.manual_seed(1234)
    ref = OracleEMRT(ncls, "resnet50").eval()
    x = torch.randn(nwin, 3, crop, crop, generator=torch.Generator().manual_seed(1234))
    times = []
    with torch.no_grad():
        ref(x[:2])                  # warm-up: thread pool, allocator, oneDNN primitives
        for _ in range(timed):
            t0 = time.perf_counter()
            ref(x)
            times.append(time.perf_counter() - t0)
    med = statistics.median(times)
    return {"value": round(nwin / med, 3), "unit": "tiles/s", "cores": n, "kind": "port",
            "sample": "median of %d eval forward(s) of the torch-CPU fp32 oracle over %d windows (one batch of %d), after a 2-window warm-up; "
                      "times %s s" % (timed, nwin, nwin, ["%.2f" % t for t in times])}


if __name__ == "__main__":
    main()
