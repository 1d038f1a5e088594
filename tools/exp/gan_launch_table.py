#!/usr/bin/env python3
"""Round-5 verdict item 8, by measurement: every launch of a GAN train step (default: CUT, N = 4096 x 360 bands) timed on its
own (eager replay, HIP event pair around each launch, median of --rounds), with the planner's FLOP count where the launch is a
GEMM -- and, next to it, the step as the HIP graphs replay it.  Separates matrix work from launch floor:

  * launches whose isolated time is at the floor (<= 2 x the empty-launch time measured in the same process) are 'floor';
  * for the GEMMs of the wide critic / feature-discriminator stacks: time at their own rate vs what they would cost at the
    best rate any launch of the stack reaches (what a fused stack could at most recover besides the floor).

  python tools/exp/gan_launch_table.py [--workload cut|cyclegan] [--rounds 20]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cut", choices=["cut", "cyclegan"])
    ap.add_argument("--rounds", type=int, default=20)
    args = ap.parse_args()
    from hypelcnn_amd.backend import HipBackend
    be = HipBackend()
    a = argparse.Namespace(workload=args.workload, batch=0, no_graph=False)
    nb, bands, kind, one_step, loss_fn, ops = bench.run_gan_workload(a, be, 1, 0)
    for _ in range(10):
        one_step()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(51)]
    ev[0].record()
    for i in range(50):
        one_step()
        ev[i + 1].record()
    torch.cuda.synchronize()
    step_us = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(50))[25] * 1e3
    sess = ops.ctx.session()
    phases = [(k, ct) for k, ct in sess._compiled.items()]
    # the empty-launch floor of this process: a 1-element fill, back to back
    t = be.zeros(16)
    from hypelcnn_amd.backend import Ref
    fl = []
    for _ in range(200):
        x, y = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        x.record()
        be.call("fill_f32", Ref(t), 1, 0.0)
        y.record()
        torch.cuda.synchronize()
        fl.append(x.elapsed_time(y) * 1e3)
    floor = float(np.median(fl))
    rows = []
    for key, ct in phases:
        for l, f in ct.serial_launches():
            ts = []
            for _ in range(args.rounds):
                x, y = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                x.record()
                f()
                y.record()
                torch.cuda.synchronize()
                ts.append(x.elapsed_time(y) * 1e3)
            rows.append((str(key), l.name, l.tag, float(np.median(ts[2:])), l.flops))
    tot = sum(r[3] for r in rows)
    print(f"{kind} N = {nb} x {bands} bands: graph-replayed step {step_us:.1f} us; {len(rows)} launches, sum of isolated times "
          f"{tot:.1f} us; isolated empty launch (event pair around a 1-element fill) {floor:.1f} us")
    print(f"{'phase':28s} {'kernel':34s} {'tag':40s} {'us':>8s} {'GFLOP':>8s} {'TF/s':>7s}")
    for key, name, tag, us, flops in rows:
        print(f"{key[:28]:28s} {name[:34]:34s} {tag[:40]:40s} {us:8.1f} {flops / 1e9:8.2f} {flops / us / 1e6 if flops else 0:7.1f}")
    gem = [r for r in rows if r[1].startswith("seg_gemm")]
    gen = [r for r in rows if r[1].startswith("gan_generator")]
    small = [r for r in rows if r[3] <= 2 * floor]
    gt = sum(r[3] for r in gem)
    gf = sum(r[4] for r in gem)
    best = max((r[4] / r[3] for r in gem if r[4] > 1e9), default=0.0)
    print(f"generator launches: {len(gen)}, {sum(r[3] for r in gen):.1f} us")
    print(f"seg_gemm launches (wide critic / feature-discriminator stacks): {len(gem)}, {gt:.1f} us, {gf / 1e9:.2f} GFLOP = "
          f"{gf / gt / 1e6:.1f} TFLOP/s; at the best rate of any of them ({best / 1e6:.1f} TFLOP/s) the same FLOP take "
          f"{gf / best:.1f} us: a fused stack could recover at most {gt - gf / best:.1f} us of matrix time")
    print(f"launches at the floor (isolated time <= 2 x {floor:.1f} us): {len(small)}, {sum(r[3] for r in small):.1f} us in isolation; "
          f"inside the graph a dependent boundary costs ~1.7 us: {len(rows)} x 1.7 = {len(rows) * 1.7:.0f} us of the {step_us:.0f} us step")
    other = [r for r in rows if not r[1].startswith(("seg_gemm", "gan_generator"))]
    by = {}
    for r in other:
        by.setdefault(r[1], [0, 0.0])
        by[r[1]][0] += 1
        by[r[1]][1] += r[3]
    print("elementwise / loss / optimiser launches: " + ", ".join(f"{k} x{v[0]} {v[1]:.0f} us" for k, v in
                                                                  sorted(by.items(), key=lambda kv: -kv[1][1])))


if __name__ == "__main__":
    main()
