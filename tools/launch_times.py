#!/usr/bin/env python3
"""Every launch of one train step timed on its own (HIP events, eager replay), slowest first.

  python tools/launch_times.py [--workload hypelcnn|dualcnn] [--top 25]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="hypelcnn", choices=list(bench.CLASSIFIER_WORKLOADS))
    ap.add_argument("--top", type=int, default=25)
    ap.add_argument("--rounds", type=int, default=4)
    args = ap.parse_args()
    from hypelcnn_amd.backend import HipBackend
    be = HipBackend()
    _, _, patch, chans, classes, nb, _ = bench.CLASSIFIER_WORKLOADS[args.workload]
    ctx, train_step, lr, alg = bench.build_model(nb, be, args.workload)
    ctx.capture_graphs = False
    ctx.session()
    ct = train_step.compiled(nb)
    ct.set_input("x", torch.rand((nb, patch, patch, chans)).cuda())
    ct.set_input("labels", torch.nn.functional.one_hot(torch.randint(0, classes, (nb,)), classes).float().cuda())
    ct.forward_backward()
    torch.cuda.synchronize()
    items = list(ct.serial_launches())
    times = [[] for _ in items]
    for _ in range(args.rounds):
        for i, (l, f) in enumerate(items):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            f()
            b.record()
            torch.cuda.synchronize()
            times[i].append(a.elapsed_time(b) * 1e3)
    med = [float(np.median(t[1:])) for t in times]
    order = np.argsort(med)[::-1]
    print(f"{len(items)} launches, sum of medians {sum(med) / 1e3:.3f} ms")
    for i in order[:args.top]:
        l = items[i][0]
        gbs = f"{l.bytes / med[i] / 1e3:7.1f} GB/s alg" if getattr(l, "bytes", 0) else ""
        print(f"{l.name:22s} {l.tag:34s} {med[i]:10.1f} us  {gbs}")


if __name__ == "__main__":
    main()
