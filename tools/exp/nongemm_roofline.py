#!/usr/bin/env python3
"""Verdict item 8 (the non-GEMM share of the headline step): every non-GEMM launch of the step in an eager replay with a HIP event
pair around it, its algorithmic bytes (the planner's own count: reads + writes) / time, against a copy of matching size measured
on the same box with the library's copy kernel (2 x size bytes moved).  Launches without a byte count (finalisers, losses, the small
one-launch batch norms) are the launch-floor population: reported as count x time, no bandwidth is read into them -- and an event
pair around a ~5 us kernel includes the gaps around it, so those times are upper bounds.
  tools/gpu.sh --timeout 600 -- 'python tools/exp/nongemm_roofline.py'"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from hypelcnn_amd.backend import HipBackend, Ref  # noqa: E402


def timed(f, reps):
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return float(np.median(ts))


def main():
    be = HipBackend()
    nb = 1024
    ctx, train_step, lr, alg = bench.build_model(nb, be, "hypelcnn")
    ct = train_step.compiled(nb)
    sess = ctx.session()
    g = torch.Generator(device="cpu"); g.manual_seed(1)
    ct.set_input("x", torch.rand((nb, 7, 7, 145), generator=g).cuda())
    ct.set_input("labels", torch.eye(15)[torch.randint(0, 15, (nb,), generator=g)].cuda())
    launches = ct.serial_launches()
    for _ in range(3):
        for l, f in launches:
            f()
    torch.cuda.synchronize()
    # copy ceiling by size (bytes MOVED = 2 x buffer)
    ceil = {}
    for mb in (2, 8, 24, 48, 96, 192):
        n = mb * (1 << 20) // 4
        s, d = torch.empty(n, device="cuda"), torch.empty(n, device="cuda")
        f = be.bind("copy_pair_f32", (Ref(d), Ref(s), n, Ref(d), Ref(s), 0))
        f(); torch.cuda.synchronize()
        us = timed(f, 9)
        ceil[2 * n * 4] = 2 * n * 4 / us / 1e3  # GB/s
        print(f"copy ceiling  {2 * mb:4d} MB moved  {us:7.1f} us  {ceil[2 * n * 4]:7.0f} GB/s")
    sizes = sorted(ceil)

    def ceiling(nbytes):
        return ceil[min(sizes, key=lambda s: abs(np.log(s / max(nbytes, 1))))]

    rounds = 7
    per = collections.defaultdict(list)
    order = []
    for r in range(rounds):
        for i, (l, f) in enumerate(launches):
            if l.name.startswith("seg_gemm"):
                f()
                continue
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); f(); b.record()
            per[i].append((a, b))
            if r == 0:
                order.append(i)
        torch.cuda.synchronize()
    rows = []
    for i in order:
        l = launches[i][0]
        us = float(np.median([a.elapsed_time(b) * 1e3 for a, b in per[i]]))
        rows.append((l.name, l.tag, int(l.bytes), us))
    print(f"\n{'kernel':22s} {'tag':22s} {'MB':>8s} {'us':>8s} {'GB/s':>7s} {'of copy':>8s}")
    agg = collections.OrderedDict()
    for name, tag, nbytes, us in rows:
        if nbytes > 0:
            gbs = nbytes / us / 1e3
            print(f"{name:22s} {tag:22s} {nbytes / 1e6:8.1f} {us:8.1f} {gbs:7.0f} {gbs / ceiling(nbytes):8.2f}")
        a = agg.setdefault(name, [0, 0.0, 0, 0.0, 0.0])
        a[0] += 1; a[1] += us
        if nbytes > 0:
            a[2] += nbytes; a[3] += us; a[4] += nbytes / ceiling(nbytes) / 1e3  # us the same bytes take at copy speed
    print(f"\n{'kernel':22s} {'n':>3s} {'us':>8s} {'MB':>8s} {'GB/s':>7s} {'us at copy speed':>17s}  slack us")
    tot = tot_copy = tot_nob = 0.0
    for name, (n, us, nbytes, us_b, us_copy) in agg.items():
        if nbytes:
            print(f"{name:22s} {n:3d} {us:8.1f} {nbytes / 1e6:8.1f} {nbytes / us_b / 1e3:7.0f} {us_copy:17.1f}  {us_b - us_copy:7.1f}")
            tot += us_b; tot_copy += us_copy
        else:
            print(f"{name:22s} {n:3d} {us:8.1f}   (no byte count: {us / n:.1f} us per launch)")
            tot_nob += us
    print(f"\nstreaming passes {tot:.0f} us, the same bytes at copy speed {tot_copy:.0f} us: slack {tot - tot_copy:.0f} us;  "
          f"launch-floor population {tot_nob:.0f} us (upper bound: event pairs include gaps)")


if __name__ == "__main__":
    main()
