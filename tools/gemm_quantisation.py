#!/usr/bin/env python3
"""How much of a hypel_seg_gemm_f32 launch is grid quantisation / tail?  One plain GEMM
(C[M x n] = A[M x k] B[k x n], one group, one segment) timed while the number of 128-row tiles grows: a launch whose
time is a staircase in the number of blocks loses its last partial round, a linear one does not.

  python tools/gemm_quantisation.py [--k 480] [--n 480] [--hint 0|1|2]
  HYPEL_LIB_PATH=<build with -DHYPEL_GEMM_CLK=1> python tools/gemm_quantisation.py --clk [--burst N]
      additionally reads the shader clock each launch actually ran at (clock64 / wall_clock64 inside the kernel);
      a -DHYPEL_GEMM_CLK=2 build with --timeline prints how many blocks are inside their k loop over the launch.
Results: profiles/r1_gemm_loop_breakdown.txt."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from hypelcnn_amd.backend import GEMM_BM, HipBackend, Ref  # noqa: E402
from hypelcnn_amd.plan import GemmTables  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, default=480)
    ap.add_argument("--n", type=int, default=480)
    ap.add_argument("--hint", type=int, default=2)
    ap.add_argument("--tiles", type=str, default="64,128,192,256,320,384,392,448,512,640,768,1024,1536,2048")
    ap.add_argument("--rounds", type=int, default=9)
    ap.add_argument("--timeline", action="store_true", help="library built with -DHYPEL_GEMM_CLK=2: blocks alive over time")
    ap.add_argument("--burst", type=int, default=0, help="with --clk: N back-to-back launches, clock of each")
    ap.add_argument("--clk", action="store_true", help="library built with -DHYPEL_GEMM_CLK=1: report the shader clock")
    args = ap.parse_args()
    be = HipBackend()
    k, n = args.k, args.n
    max_tiles = max(int(t) for t in args.tiles.split(","))
    a = torch.rand(max_tiles * GEMM_BM * k, device="cuda")
    b = torch.rand(k * n, device="cuda")
    c = torch.empty(max_tiles * GEMM_BM * n, device="cuda")
    width = 32 if args.hint == 1 else 64
    print(f"# k={k} n={n} tile 128x{width}; blocks = tiles * ceil(n/{width}); resident blocks ~ 256 CUs x (4 | 3)")
    for tiles in [int(t) for t in args.tiles.split(",")]:
        rows = tiles * GEMM_BM
        tb = GemmTables()
        tb.add_group(0, [(0, 0, k)], rows)
        garr, sarr, tarr, macs = tb.finalize(n)
        g_t, s_t, t_t = be.upload(garr), be.upload(sarr), be.upload(tarr)
        dbg = torch.zeros(4 * tiles * ((n + width - 1) // width), device="cuda") if (args.clk or args.timeline) else None
        f = be.bind("seg_gemm_f32", (Ref(a), k, 0, Ref(b), n, 0, Ref(c), n, n, Ref(g_t), Ref(s_t), Ref(t_t),
                                     len(tarr), Ref(dbg) if dbg is not None else None, args.hint << 8))
        if args.burst:
            nblk = tiles * ((n + width - 1) // width)
            bufs = [torch.zeros(4 * nblk, device="cuda") for _ in range(args.burst)]
            fs = [be.bind("seg_gemm_f32", (Ref(a), k, 0, Ref(b), n, 0, Ref(c), n, n, Ref(g_t), Ref(s_t), Ref(t_t),
                                           len(tarr), Ref(d_), args.hint << 8)) for d_ in bufs]
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for f_ in fs:
                f_()
            e1.record()
            torch.cuda.synchronize()
            tot = e0.elapsed_time(e1) * 1e3
            clks = []
            for d_ in bufs:
                d = d_.cpu().numpy().view(np.int64).reshape(-1, 2)
                d = d[d[:, 1] > 0]
                clks.append(np.median(d[:, 0] / d[:, 1]) * 100)
            print(f"tiles {tiles:5d} burst of {args.burst}: {tot / args.burst:8.1f} us/launch "
                  f"{2 * macs * args.burst / tot / 1e6:6.1f} TF/s  clocks MHz: " + " ".join(f"{c:.0f}" for c in clks))
            continue
        ts = []
        for _ in range(args.rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            f()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        med = float(np.median(ts[2:]))
        blocks = tiles * ((n + width - 1) // width)
        clk = ""
        if args.timeline:
            d = dbg.cpu().numpy().view(np.int64).reshape(-1, 2)
            d = d[d[:, 1] > 0].astype(np.float64) / 100.0  # us
            t0, t1 = d[:, 0].min(), d[:, 1].max()
            edges = np.linspace(t0, t1, 41)
            alive = [int(((d[:, 0] <= t) & (d[:, 1] > t)).sum()) for t in 0.5 * (edges[1:] + edges[:-1])]
            life = d[:, 1] - d[:, 0]
            order = np.argsort(d[:, 0])
            print(f"  timeline over {t1 - t0:.1f} us (k-loop only): blocks alive per 1/40 slice: {alive}")
            print(f"  block life us: first 10% started {np.median(life[order[:len(order) // 10]]):.1f}, "
                  f"middle {np.median(life[order[len(order) * 4 // 10: len(order) * 6 // 10]]):.1f}, "
                  f"last 10% started {np.median(life[order[-len(order) // 10:]]):.1f}; "
                  f"last start at {d[:, 0].max() - t0:.1f} us")
        if args.clk:
            d = dbg.cpu().numpy().view(np.int64).reshape(-1, 2)
            d = d[d[:, 1] > 0]
            clk = f"  shader clock {np.median(d[:, 0] / d[:, 1]) * 100:7.1f} MHz (block life {np.median(d[:, 1]) / 100:6.1f} us)"
        print(f"tiles {tiles:5d} blocks {blocks:6d} ({blocks / 1024:5.2f} x1024)  {med:8.1f} us  "
              f"{2 * macs / med / 1e6:6.1f} TF/s  {med / blocks * 1e3:7.1f} ns/block{clk}")


if __name__ == "__main__":
    main()
