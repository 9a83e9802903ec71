"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KB per dispatch) per kernel into profiles/.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 2 --warmup 0 --no-graph --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 2 --warmup 0 --no-graph --no-cpu-baseline
    python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r1d_pmc_traffic

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 128-B read requests as 64 B, i.e. reports half
of the bytes of wide coalesced reads; WRITE_SIZE is exact for streaming stores and float atomics.  Both raw and
corrected (2 x FETCH + WRITE) figures are written.
"""
import collections
import csv
import glob
import json
import sys


def load(d):
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(glob.glob(d + "/*/*counter_collection.csv")[0])):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        out[name].append((float(r["Counter_Value"]), int(r["Grid_Size"])))
    return out


def main():
    fetch, write, dst = load(sys.argv[1]), load(sys.argv[2]), sys.argv[3]
    rows = []
    for name in sorted(fetch, key=lambda k: -sum(x[0] for x in fetch[k])):
        f, w = fetch[name], write.get(name, [])
        if not w or "emrt" not in name and "kernel" not in name:
            continue
        n = len(f)
        fk, wk = sum(x[0] for x in f) / n, sum(x[0] for x in w) / max(1, len(w))
        rows.append((name, n, fk, wk, 2 * fk + wk))
    with open(dst + ".csv", "w") as fo:
        fo.write("kernel,dispatches,FETCH_SIZE_KB_per_dispatch_raw,WRITE_SIZE_KB_per_dispatch,traffic_KB_per_dispatch_corrected(2*FETCH+WRITE)\n")
        for r in rows:
            fo.write('"%s",%d,%.2f,%.2f,%.2f\n' % r)

    def family(pred, grid=None):
        fs = [x[0] for k, v in fetch.items() if pred(k) for x in v if grid is None or x[1] == grid]
        ws = [x[0] for k, v in write.items() if pred(k) for x in v if grid is None or x[1] == grid]
        if not fs or not ws:
            return None
        return {"dispatches": len(fs), "fetch_mb_raw": round(sum(fs) / len(fs) / 1e3, 3), "write_mb": round(sum(ws) / len(ws) / 1e3, 3),
                "traffic_mb_corrected": round((2 * sum(fs) / len(fs) + sum(ws) / len(ws)) / 1e3, 3)}

    enc_name = max((k for k in fetch if "msda_fwd" in k), key=lambda k: max(x[1] for x in fetch[k]))    # the encoder call: largest grid
    enc_grid = max(x[1] for x in fetch[enc_name])
    js = {"method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `bench.py --steps 2 --warmup 0 --no-graph`; "
                    "per-dispatch averages; corrected = 2*FETCH_SIZE + WRITE_SIZE (gfx950 tallies 128-B reads as 64 B)",
          "igemm_kernel": family(lambda k: "igemm_kernel" in k),
          "wgrad_kernel": family(lambda k: "wgrad_kernel" in k),
          "msda_fwd_kernel_encoder": family(lambda k: k == enc_name, enc_grid),
          "bwd_pair_kernel": family(lambda k: "bwd_pair_kernel" in k)}
    js = {k: v for k, v in js.items() if v is not None}        # (inference profiles have no backward kernels)
    js["msda_fwd_kernel_encoder"]["kernel"] = enc_name
    json.dump(js, open(dst + ".json", "w"), indent=1)
    print(json.dumps(js, indent=1))


if __name__ == "__main__":
    main()
