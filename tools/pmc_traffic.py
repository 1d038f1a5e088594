#!/usr/bin/env python3
"""HBM traffic per kernel from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass).

  python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> [--steps S] [--json OUT]

Corrections (MI355X_MICROARCH.md, "HBM [CDNA4]"): counters are in KiB; on gfx950 FETCH_SIZE reports half the bytes
of a wide streaming read -> doubled.  WRITE_SIZE is calibrated on `nhwc_to_pnc_kernel`, whose byte count is known
(it writes exactly what it reads: one fp32 copy of the batch) -- the factor is printed and applied.
"""
import argparse
import collections
import csv
import json
import re


def short(name):
    m = re.search(r"seg_gemm_kernel<([^>]*)>", name)
    if m:
        return "seg_gemm<" + m.group(1).replace(" ", "") + ">"
    m = re.search(r"namespace\)::(\w+)", name)
    if m:
        return m.group(1)
    m = re.search(r"(\w+)\(", name)
    return m.group(1) if m else name[:40]


def load(path, counter):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            per[short(r["Kernel_Name"])].append(float(r["Counter_Value"]) * 1024.0)
    return per


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch")
    ap.add_argument("write")
    ap.add_argument("--known-bytes", type=float, default=1024 * 49 * 145 * 4, help="bytes nhwc_to_pnc moves one way")
    ap.add_argument("--json")
    ap.add_argument("--workload", default="hypelcnn")
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--source", default="")
    args = ap.parse_args()
    f = load(args.fetch, "FETCH_SIZE")
    w = load(args.write, "WRITE_SIZE")
    cal_w = 1.0
    cal_f = 2.0
    if "nhwc_to_pnc_kernel" in w:
        mean_w = sum(w["nhwc_to_pnc_kernel"]) / len(w["nhwc_to_pnc_kernel"])
        cal_w = args.known_bytes / mean_w
        mean_f = sum(f["nhwc_to_pnc_kernel"]) / len(f["nhwc_to_pnc_kernel"])
        print(f"calibration on nhwc_to_pnc ({args.known_bytes/1e6:.2f} MB each way): raw WRITE_SIZE {mean_w/1e6:.2f} MB "
              f"-> x{cal_w:.3f};  raw FETCH_SIZE {mean_f/1e6:.2f} MB (guide factor 2.0 -> {2*mean_f/1e6:.2f} MB)")
    out = {}
    print(f"{'kernel':34s} {'launches':>8s} {'read MB/launch':>15s} {'write MB/launch':>16s} {'total MB':>10s}")
    tot = 0.0
    for k in sorted(set(f) | set(w), key=lambda k: -(cal_f * sum(f.get(k, [0])) + cal_w * sum(w.get(k, [0])))):
        # the two passes are separate runs of a bench whose pre-warm is time based: they need not hold the same number of
        # steps, so each counter is averaged over ITS pass's launches (dividing both by the larger count under-reported
        # the reads by 1/13 whenever the write pass had one step more: the 160 vs 170 MB of rounds 3-4)
        nf, nw = len(f.get(k, [])), len(w.get(k, []))
        n = max(nf, nw)
        rd = cal_f * sum(f.get(k, [0])) / max(nf, 1)
        wr = cal_w * sum(w.get(k, [0])) / max(nw, 1)
        tot += (rd + wr) * n
        out[k] = {"launches": n, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr}
        if (rd + wr) * n > 1e6:
            print(f"{k:34s} {n:8d} {rd/1e6:15.2f} {wr/1e6:16.2f} {(rd+wr)*n/1e6:10.1f}")
    print(f"sum over the run: {tot/1e9:.2f} GB")
    if args.json:
        # summary of the dominant kernel family for bench.py's roofline.traffic: one nhwc_to_pnc launch per step
        steps = out.get("nhwc_to_pnc_kernel", {}).get("launches", 0)
        gemm = {k: v for k, v in out.items() if k.startswith("seg_gemm")}
        g_launches = sum(v["launches"] for v in gemm.values())
        g_bytes = sum(v["launches"] * (v["read_bytes_per_launch"] + v["write_bytes_per_launch"]) for v in gemm.values())
        summary = {"workload": args.workload, "batch": args.batch, "source": args.source, "fetch_factor": cal_f,
                   "write_factor": cal_w, "steps_profiled": steps, "seg_gemm_launches": g_launches,
                   "seg_gemm_launches_per_step": (g_launches // steps) if steps else None,
                   "seg_gemm_bytes_per_launch": g_bytes / max(1, g_launches),
                   "seg_gemm_bytes_per_step": (g_bytes / steps) if steps else None,
                   "all_kernels_bytes_per_step": (tot / steps) if steps else None, "kernels": out}
        json.dump(summary, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
