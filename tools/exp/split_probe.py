#!/usr/bin/env python3
"""One plain product C[M x n] = A[M x K] B[K x n] (or its data- / filter-gradient layout) on the fp32 MFMA kernel and on the
split-operand kernel: microseconds, TFLOP/s; under rocprofv3 --pmc the dispatches are easy to tell apart (one warm-up +
REPS launches per variant, in the order printed).
  python tools/exp/split_probe.py [--m 50176] [--k 480] [--n 480] [--layout nn|nt|tn] [--variants 0,3] [--reps 5]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from hypelcnn_amd.backend import HipBackend, Ref  # noqa: E402
from hypelcnn_amd.plan import GemmTables  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=50176)
    ap.add_argument("--k", type=int, default=480)
    ap.add_argument("--n", type=int, default=480)
    ap.add_argument("--layout", default="nn")
    ap.add_argument("--variants", default="0,1,2,3")  # 0 = fp32 kernel (library heuristic), 1/2/3 = split widths
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--clk", action="store_true", help="library built with -DHYPEL_GEMM_CLK=1: the shader clock each block ran at")
    args = ap.parse_args()
    be = HipBackend()
    m, k, n = args.m, args.k, args.n
    ta, tb = {"nn": (0, 0), "nt": (0, 1), "tn": (1, 0)}[args.layout]
    rng = np.random.default_rng(0)
    a = torch.from_numpy(rng.standard_normal((k, m) if ta else (m, k)).astype(np.float32)).cuda()
    b = torch.from_numpy(rng.standard_normal((n, k) if tb else (k, n)).astype(np.float32)).cuda()
    c = torch.zeros(m * n, device="cuda")
    tbs = GemmTables()
    if ta:  # filter gradient: rows = K_out (m) is small, reduction over many rows: cut into 48 slices like the planner
        S = 48
        cuts = [k * s // S // 32 * 32 for s in range(S)] + [k]
        c = torch.zeros(S * m * n, device="cuda")
        for s in range(S):
            tbs.add_group(s * m * n, [(cuts[s] * m, cuts[s] * n, cuts[s + 1] - cuts[s])], m, key=s)
    else:
        tbs.add_group(0, [(0, 0, k)], m)
    g, sg, t, macs = tbs.finalize(n)
    gt, st, tt = be.upload(g), be.upload(sg), be.upload(t)
    lda, ldb = a.shape[1], b.shape[1]
    for v in [int(x) for x in args.variants.split(",")]:
        flags = 0 if v == 0 else (0x8000 | (v << 8))
        dbg = torch.zeros(1 << 17, dtype=torch.int64, device="cuda") if args.clk else None
        f = be.bind("seg_gemm_f32", (Ref(a), lda, ta, Ref(b), ldb, tb, Ref(c), n, n, Ref(gt), Ref(st), Ref(tt), len(t),
                                     Ref(dbg.view(torch.float32)) if args.clk else None, flags))
        f()
        torch.cuda.synchronize()
        ts = []
        for _ in range(args.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            f()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        med = float(np.median(ts))
        if args.clk:
            d = dbg.cpu().numpy().reshape(-1, 2)
            d = d[d[:, 1] > 0]
            ghz = d[:, 0] / (d[:, 1] * 10.0)  # shader cycles per ns (the wall counter ticks at 100 MHz)
            print(f"    blocks {len(d)}: shader clock over each block's k loop  median {np.median(ghz):.3f} GHz  "
                  f"p10 {np.percentile(ghz, 10):.3f}  p90 {np.percentile(ghz, 90):.3f};  block life median "
                  f"{np.median(d[:, 1]) / 100:.1f} us  max {d[:, 1].max() / 100:.1f} us")
        print(f"layout {args.layout} M={m} K={k} n={n} variant {v}: med {med:8.1f} us  min {min(ts):8.1f} us  "
              f"{2 * macs / med / 1e6:6.1f} TF/s")


if __name__ == "__main__":
    main()
