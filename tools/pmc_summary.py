"""Summarise the rocprofv3 passes of one eager bench run per config into profiles/ (tools/profile_round.sh):

    python tools/pmc_summary.py <fetch dir> <write dir> <dst prefix> [<sq dir> <trace dir> <calls.txt>]
    python tools/pmc_summary.py <fetch dir> <write dir> <dst prefix> --calls <calls.txt>        (no per-layer table: the call counts only)

  <dst>.csv   per kernel: dispatches, FETCH_SIZE / WRITE_SIZE KB per dispatch, corrected traffic
  <dst>.json  per kernel FAMILY: bytes per dispatch and dispatches per step -- what bench.py attaches to `roofline.traffic`
              (conv_backward_family: every kernel that computes a conv / linear data or weight gradient; conv_forward_family)
  <dst>_layers.csv (with the three optional inputs) the small-launch table: ONE ROW PER C-ABI CONV CALL of a step, in step order --
              layer shape, rocprof duration, GFLOP, TFLOP/s, algorithmic bytes, PMC bytes, MFMA-busy fraction.

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 128-B read requests as 64 B, i.e. reports half of the bytes of wide
coalesced reads; WRITE_SIZE is exact for streaming stores and float atomics.  Raw and corrected (2 x FETCH + WRITE) figures are written.
The passes are separate runs of the same deterministic launch sequence, so dispatch i of one pass is dispatch i of the others."""
import collections
import csv
import glob
import json
import re
import sys


def rows_of(d, pattern):
    f = glob.glob(d + "/*/*" + pattern) or glob.glob(d + "/*" + pattern)
    return list(csv.DictReader(open(f[0]))) if f else []


def short(name):
    return name.split("(")[0].replace("void ", "")


def last_step(names):
    """Index range of the last complete step: between the last two optimizer kernels (a training run); an inference run has none and is
    cut at the kernel that ingests an image (the NCHW -> NHWC layout change of the window batch, once per image): the dispatches of ONE
    image, from the last-but-one ingest to the last one.  (Round 4 took the whole list there -- all 8 images of the eager run as one
    "step" -- and bench.py multiplied the per-dispatch traffic by 8 x 117 dispatches: traffic_over_algorithmic 12.3 instead of ~1.5.)"""
    idx = [i for i, n in enumerate(names) if "sgd_momentum" in n]
    if len(idx) >= 2:
        return idx[-2] + 1, idx[-1] + 1
    idx = [i for i, n in enumerate(names) if "nchw_to_nhwc" in n]
    if len(idx) >= 2:
        return idx[-2], idx[-1]
    return 0, len(names)


def per_dispatch(d, counter=None):
    """-> [(kernel, value, grid)] in dispatch order for one counter of a --pmc pass (rows carry Dispatch_Id)."""
    rows = rows_of(d, "counter_collection.csv")
    out = {}
    for n, r in enumerate(rows):
        if counter and r["Counter_Name"] != counter:
            continue
        key = int(r["Dispatch_Id"]) if r.get("Dispatch_Id") else n
        prev = out.get(key)
        val = float(r["Counter_Value"]) + (prev[1] if prev else 0.0)       # (a counter reported per XCD / per dimension: summed)
        out[key] = (short(r["Kernel_Name"]), val, int(r["Grid_Size"]))
    return [out[k] for k in sorted(out)]


CONV_BWD = lambda k: ("igemm_kernel" in k and re.search(r"igemm_kernel<[^,]+, \d+, \d+, \d+, \d+, 1,", k)) or ("igemm8p_kernel" in k and re.search(r"igemm8p_kernel<[^,]+, 1,", k)) \
    or "wgrad" in k or "bwd_pair" in k or "bwd_group" in k or "thin_bwd" in k or "igemm_s2" in k or re.search(r"igemm_group_kernel<[^,]+, 1>", k) \
    or re.search(r"igemm_xk_kernel<[^,]+, 1>", k)
CONV_FWD = lambda k: ("igemm" in k or "thin_fwd" in k) and not CONV_BWD(k)


CONV_BWD_CALLS = ("emrt_conv2d_bwd", "emrt_conv2d_bwd_group", "emrt_conv2d_wgrad", "emrt_conv2d_wgrad_group", "emrt_bn_pointwise_bwd")
CONV_FWD_CALLS = ("emrt_conv2d", "emrt_conv2d_drop", "emrt_conv2d_bna", "emrt_conv2d_group", "emrt_bn_pointwise_fwd")


def call_counts(calls_path):
    """C-ABI calls per family in bench.py's --dump-calls file of the SAME command (one replayed step / image): what bench.py compares
    with its own launch list before it attaches this profile's traffic to a line."""
    names = [ln.split()[0] for ln in open(calls_path) if ln.strip()]
    # (a dgrad is an emrt_conv2d call with mode 1 only on the two-stream experiment path: not in the default command)
    return {"conv_backward_family": sum(n in CONV_BWD_CALLS for n in names), "conv_forward_family": sum(n in CONV_FWD_CALLS for n in names),
            "all": len(names)}


def main():
    argv = sys.argv[:]
    calls_only = None
    if "--calls" in argv:
        i = argv.index("--calls")
        calls_only = argv[i + 1]
        del argv[i:i + 2]
    sys.argv = argv
    fetch_d, write_d, dst = sys.argv[1:4]
    fetch, write = per_dispatch(fetch_d), per_dispatch(write_d)
    names = [k for k, _, _ in fetch]
    assert len(fetch) == len(write) and names == [k for k, _, _ in write], "the two passes launched different sequences"
    by = collections.defaultdict(list)
    for (k, f, g), (_, w, _) in zip(fetch, write):
        by[k].append((f, w, g))
    rows = []
    for k in sorted(by, key=lambda k: -sum(2 * f + w for f, w, _ in by[k])):
        v = by[k]
        n = len(v)
        rows.append((k, n, sum(x[0] for x in v) / n, sum(x[1] for x in v) / n, sum(2 * x[0] + x[1] for x in v) / n))
    with open(dst + ".csv", "w") as fo:
        fo.write("kernel,dispatches,FETCH_SIZE_KB_per_dispatch_raw,WRITE_SIZE_KB_per_dispatch,traffic_KB_per_dispatch_corrected(2*FETCH+WRITE)\n")
        for r in rows:
            fo.write('"%s",%d,%.2f,%.2f,%.2f\n' % r)
    a, b = last_step(names)
    step = [(k, f, w, g) for (k, f, g), (_, w, _) in zip(fetch[a:b], write[a:b])]

    def family(pred, grid=None):
        sel = [(f, w) for k, f, w, g in step if pred(k) and (grid is None or g == grid)]
        if not sel:
            return None
        n = len(sel)
        return {"dispatches_per_step": n, "fetch_mb_raw": round(sum(f for f, _ in sel) / n / 1e3, 3), "write_mb": round(sum(w for _, w in sel) / n / 1e3, 3),
                "traffic_mb_corrected": round(sum(2 * f + w for f, w in sel) / n / 1e3, 3), "kernels": sorted({k.split("<")[0] for k, f, w, g in step if pred(k)})}
    enc = [(k, g) for k, f, w, g in step if "msda_fwd" in k]
    js = {"method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `bench.py --steps 2 --warmup 0 --no-graph`; the dispatches of the "
                    "last complete step; per-dispatch averages; corrected = 2*FETCH_SIZE + WRITE_SIZE (gfx950 tallies 128-B reads as 64 B)",
          "conv_backward_family": family(CONV_BWD), "conv_forward_family": family(CONV_FWD)}
    if enc:
        enc_grid = max(g for _, g in enc)
        enc_name = [k for k, g in enc if g == enc_grid][0]
        js["msda_fwd_kernel_encoder"] = family(lambda k: k == enc_name, enc_grid)
        js["msda_fwd_kernel_encoder"]["kernel"] = enc_name
    # the deformable attention's backward, encoder calls: per kernel name the dispatches with that name's LARGEST grid (the decoder's 110-query calls run the
    # same kernels on smaller grids), summed over the kernels of one call (gradient kernel + value-gradient kernel [+ finalize])
    bwd_names = sorted({k for k, f, w, g in step if "msda_bwd" in k})
    if bwd_names:
        parts, tot = [], {"fetch_mb_raw": 0.0, "write_mb": 0.0, "traffic_mb_corrected": 0.0}
        for nm in bwd_names:
            gmax = max(g for k, f, w, g in step if k == nm)
            fam = family(lambda k, nm=nm: k == nm, gmax)
            parts.append({"kernel": nm, "dispatches_per_step": fam["dispatches_per_step"], "traffic_mb_corrected": fam["traffic_mb_corrected"]})
            for key in tot:
                tot[key] += fam[key]
        js["msda_bwd_encoder_call"] = dict({k_: round(v_, 3) for k_, v_ in tot.items()}, kernels=parts)
    js = {k: v for k, v in js.items() if v is not None}
    js["dispatches_in_step"] = b - a
    calls_path = calls_only or (sys.argv[6] if len(sys.argv) >= 7 else None)
    if calls_path:
        cc = call_counts(calls_path)
        for fam in ("conv_backward_family", "conv_forward_family"):
            if fam in js:
                js[fam]["calls_per_step"] = cc[fam]
        js["calls_in_step"] = cc["all"]
    json.dump(js, open(dst + ".json", "w"), indent=1)
    print(json.dumps(js, indent=1))
    if len(sys.argv) >= 7:
        layer_table(dst, step, a, b, names, sys.argv[4], sys.argv[5], sys.argv[6])


def layer_table(dst, step, a, b, names, sq_d, trace_d, calls_path):
    mfma = per_dispatch(sq_d, "SQ_VALU_MFMA_BUSY_CYCLES")
    gui = per_dispatch(sq_d, "GRBM_GUI_ACTIVE")
    tr = rows_of(trace_d, "kernel_trace.csv")
    tr.sort(key=lambda r: int(r["Start_Timestamp"]))
    tnames = [short(r["Kernel_Name"]) for r in tr]
    ta, tb = last_step(tnames)
    dur = [(short(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in tr[ta:tb]]
    ok_sq = len(mfma) == len(names) and [k for k, _, _ in mfma] == names
    ok_tr = [k for k, _ in dur] == [k for k, _, _, _ in step]
    if not ok_tr:
        print("[layers] kernel-trace pass launched a different sequence (%d vs %d dispatches): durations left out" % (len(dur), len(step)))
    calls = []
    for ln in open(calls_path):
        m = re.match(r"(\S+)\s+([\d.]+) ms\s+(.*)", ln)
        if m:
            calls.append((m.group(1), float(m.group(2)) * 1e3, m.group(3)))
    conv_names = ("emrt_conv2d", "emrt_conv2d_drop", "emrt_conv2d_bna", "emrt_conv2d_bwd", "emrt_conv2d_group", "emrt_conv2d_bwd_group", "emrt_conv2d_wgrad", "emrt_conv2d_wgrad_group",
                  "emrt_bn_pointwise_fwd", "emrt_bn_pointwise_bwd")
    is_conv_kernel = lambda k: "igemm" in k or "wgrad" in k or "thin_bwd" in k or "thin_fwd" in k or "bwd_pair" in k or "bwd_group" in k
    wg_kernels = lambda k: "wgrad" in k
    i = 0
    out = []
    for cname, ev_us, extra in calls:
        if cname not in conv_names:
            continue
        while i < len(step) and not is_conv_kernel(step[i][0]):
            i += 1
        if i >= len(step):
            break
        j = i + 1
        if cname == "emrt_conv2d_wgrad_group" or (cname == "emrt_conv2d_bwd" and wg_kernels(step[i][0])):     # several dispatches: every weight-gradient kernel in a row
            while j < len(step) and wg_kernels(step[j][0]):
                j += 1
            if cname == "emrt_conv2d_bwd":      # wgrad dispatch(es) then the data gradient
                j += 1
        ks = step[i:j]
        us = sum(d for _, d in dur[i:j]) if ok_tr else float("nan")
        traffic = sum(2 * f + w for _, f, w, _ in ks) * 1e3
        busy = float("nan")
        if ok_sq:
            mb = sum(v for _, v, _ in mfma[a + i:a + j])
            cyc = sum(v for _, v, _ in gui[a + i:a + j]) / 8.0
            busy = mb / (1024.0 * cyc) if cyc > 0 else float("nan")
        gf = float(re.search(r"gflop ([\d.]+)", extra).group(1)) if "gflop" in extra else 0.0
        ab = int(re.search(r"bytes (\d+)", extra).group(1)) if "bytes" in extra else 0
        shape = re.sub(r"\s*gflop.*", "", extra)[:110]
        out.append((cname, shape, len(ks), ks[0][0].split("<")[0], us, gf, gf / us * 1e3 if us == us and us > 0 else 0.0, ab, traffic, traffic / ab if ab else 0.0, busy, ev_us))
        i = j
    with open(dst + "_layers.csv", "w") as fo:
        fo.write("call,layer,dispatches,first_kernel,rocprof_us,gflop,tflops,algorithmic_bytes,pmc_bytes(2*FETCH+WRITE),pmc_over_algorithmic,mfma_busy_frac,event_clock_us\n")
        for r in out:
            fo.write('%s,"%s",%d,%s,%.2f,%.3f,%.1f,%d,%d,%.2f,%.3f,%.2f\n' % r)
    tot_us = sum(r[4] for r in out if r[4] == r[4])
    print("[layers] %d conv calls, %.3f ms of kernel time, %.1f GFLOP -> %.1f TFLOP/s; table in %s_layers.csv" % (len(out), tot_us / 1e3, sum(r[5] for r in out), sum(r[5] for r in out) / tot_us * 1e3 if tot_us else 0, dst))
    # every layer against ITS binding roof: ideal = max(FLOPs at the dense bf16 MFMA peak, algorithmic bytes at the HBM peak); the fraction of the family
    # is sum(ideal) / sum(measured kernel time) -- what a per-kernel "fraction of roofline" averages to when each kernel is priced against the roof that binds it
    PEAK_TF, PEAK_GBS = 2500.0, 8000.0
    rows = [r for r in out if r[4] == r[4] and r[4] > 0]
    if rows:
        ideal = [max(r[5] / PEAK_TF * 1e3, r[7] / PEAK_GBS / 1e3) for r in rows]      # us: GFLOP / (TFLOP/s) * 1e3; bytes / (GB/s) / 1e3
        mfma_bound = [i for i, r in enumerate(rows) if r[5] / PEAK_TF * 1e3 >= r[7] / PEAK_GBS / 1e3]
        hbm_bound = [i for i in range(len(rows)) if i not in set(mfma_bound)]
        br = {"ideal_us": round(sum(ideal), 1), "measured_us": round(sum(r[4] for r in rows), 1), "frac": round(sum(ideal) / sum(r[4] for r in rows), 4),
              "calls": len(rows),
              "mfma_bound": {"calls": len(mfma_bound), "ideal_us": round(sum(ideal[i] for i in mfma_bound), 1), "measured_us": round(sum(rows[i][4] for i in mfma_bound), 1)},
              "hbm_bound": {"calls": len(hbm_bound), "ideal_us": round(sum(ideal[i] for i in hbm_bound), 1), "measured_us": round(sum(rows[i][4] for i in hbm_bound), 1)},
              "peaks": {"mfma_tflops": PEAK_TF, "hbm_gbs": PEAK_GBS},
              "note": "every convolution / linear C-ABI call of one eager step: max(algorithmic FLOPs / MFMA peak, algorithmic bytes / HBM peak) summed, over the summed rocprofv3 kernel durations"}
        try:
            js = json.load(open(dst + ".json"))
            js["binding_roof"] = br
            json.dump(js, open(dst + ".json", "w"), indent=1)
        except (OSError, ValueError):
            pass
        print("[layers] binding-roof fraction %.4f (ideal %.1f us / measured %.1f us; MFMA-bound %d calls, HBM-bound %d)" % (
            br["frac"], br["ideal_us"], br["measured_us"], len(mfma_bound), len(hbm_bound)))


if __name__ == "__main__":
    main()
