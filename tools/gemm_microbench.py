#!/usr/bin/env python3
"""A/B micro-benchmark of hypel_seg_gemm_f32 on the shapes of the GRSS2013 HYPELCNN step (batch 1024).
Interleaves variants (selected through environment variables read at launch time by the library is not
possible, so variants are separate processes; within a process rounds are interleaved over shapes).

  python tools/gemm_microbench.py [--rounds 20]
prints per shape: median / min microseconds and algorithmic TFLOP/s."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=20)
    ap.add_argument("--filter", type=str, default="")
    ap.add_argument("--with-reduce", action="store_true", help="also time the reduce_splits launches (partial copies of a level)")
    ap.add_argument("--workload", type=str, default="hypelcnn", choices=list(bench.CLASSIFIER_WORKLOADS))
    ap.add_argument("--timeline", action="store_true",
                    help="HYPEL_LIB_PATH build with -DHYPEL_GEMM_CLK=2: blocks inside their k loop over each launch "
                         "(launches without a bias only; the debug buffer travels in the bias argument)")
    args = ap.parse_args()
    from hypelcnn_amd.backend import HipBackend
    be = HipBackend()
    _, _, patch, chans, classes, nb, _ = bench.CLASSIFIER_WORKLOADS[args.workload]
    ctx, train_step, lr, alg = bench.build_model(nb, be, args.workload)
    ctx.capture_graphs = False
    sess = ctx.session()
    ct = train_step.compiled(nb)
    x = torch.rand((nb, patch, patch, chans)).cuda()
    ct.set_input("x", x)
    ct.set_input("labels", torch.nn.functional.one_hot(torch.randint(0, classes, (nb,)), classes).float().cuda())
    ct.forward_backward()  # fill every buffer with realistic data
    torch.cuda.synchronize()
    items = [(l, f) for l, f in ct.serial_launches()
             if (l.name.startswith("seg_gemm") and args.filter in l.tag) or (args.with_reduce and l.name.startswith("reduce_splits"))]
    for i, (l, _) in enumerate(items):  # reduce launches share their tags: make them unique
        if not l.name.startswith("seg_gemm"):
            l.tag = f"{l.tag}#{i}"
    times = {l.tag: [] for l, _ in items}
    for r in range(args.rounds):
        for l, f in items:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            f()
            b.record()
            torch.cuda.synchronize()
            times[l.tag].append(a.elapsed_time(b) * 1e3)
    if args.timeline:
        from hypelcnn_amd.backend import Ref
        for l, _ in items:
            a = list(l.args)
            if a[13] is not None or l.flops < 2e9:
                continue
            width = 32 if ((a[14] >> 8) & 3) == 1 or a[8] <= 32 else 64
            nblk = a[12] * ((a[8] + width - 1) // width)
            dbg = torch.zeros(4 * nblk, device="cuda")
            a[13] = Ref(dbg)
            f = be.bind(l.name, tuple(a))
            f()
            torch.cuda.synchronize()
            f()
            torch.cuda.synchronize()
            d = dbg.cpu().numpy().view(np.int64).reshape(-1, 2)
            d = d[d[:, 1] > 0].astype(np.float64) / 100.0
            t0, t1 = d[:, 0].min(), d[:, 1].max()
            edges = np.linspace(t0, t1, 31)
            alive = [int(((d[:, 0] <= t) & (d[:, 1] > t)).sum()) for t in 0.5 * (edges[1:] + edges[:-1])]
            print(f"{l.tag:34s} {nblk:6d} blocks {t1 - t0:7.1f} us  last start {d[:, 0].max() - t0:6.1f}  alive {alive}")
        return
    tot = 0.0
    for l, _ in items:
        t = np.array(times[l.tag][2:])
        med = float(np.median(t))
        tot += med
        if med > 10:
            a = l.args
            if not l.name.startswith("seg_gemm"):
                print(f"{l.tag:34s} {'':7s}     med {med:8.1f} us  min {t.min():8.1f} us")
                continue
            shape = f"  n={a[8]} tiles={a[12]} ta={a[2]} tb={a[5]}" if len(a) > 12 else f"  blocks={a[6]}"
            print(f"{l.tag:34s} {l.flops / 1e9:7.2f} GF  med {med:8.1f} us  min {t.min():8.1f} us  {l.flops / med / 1e6:6.1f} TF/s"
                  + shape + ("  SPLIT6" if (len(a) > 12 and a[14] & 0x8000) or (len(a) <= 12 and a[3] & 0x100) else ""))
    flops = sum(l.flops for l, _ in items)
    print(f"TOTAL {tot / 1e3:.3f} ms  {flops / tot / 1e6:.1f} TF/s")


if __name__ == "__main__":
    main()
