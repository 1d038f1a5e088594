#!/usr/bin/env python3
"""Verdict item 5 (fill the drains of the backward GEMM launches with independent work): how much is there to fill?
Library built with -DHYPEL_GEMM_CLK=2 (every block stamps its start / end on the 100 MHz wall counter; results are wrong, timing is not):
for every forward / data-gradient GEMM launch of the headline step, the idle capacity of the launch
    idle = integral over the launch of (1 - blocks alive(t) / peak blocks alive)      [full-machine microseconds]
split into ramp (before the peak is first reached), drain (after the last block has STARTED) and the middle (quantisation rounds).
The sum over the data-gradient launches is an upper bound on what appended filter-gradient tiles could recover.
  tools/exp/build_variant.sh clk2 -DHYPEL_GEMM_CLK=2
  tools/gpu.sh --timeout 600 -- 'HYPEL_LIB_PATH=$PWD/hypelcnn_amd/csrc/variants/clk2/libhypel_hip.so python tools/exp/drain_probe.py'"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from hypelcnn_amd.backend import HipBackend, Ref  # noqa: E402


def main():
    be = HipBackend()
    _, _, patch, chans, classes, nb, _ = bench.CLASSIFIER_WORKLOADS["hypelcnn"]
    ctx, train_step, lr, alg = bench.build_model(nb, be, "hypelcnn")
    ctx.capture_graphs = False
    ct = train_step.compiled(nb)
    ct.set_input("x", torch.rand((nb, patch, patch, chans)).cuda())
    ct.set_input("labels", torch.nn.functional.one_hot(torch.randint(0, classes, (nb,)), classes).float().cuda())
    launches = ct.serial_launches()
    for l, f in launches:  # realistic data in every buffer (the stamped launches below destroy their outputs' bias term only)
        f()
    torch.cuda.synchronize()
    print(f"{'launch':34s} {'blocks':>6s} {'peak':>5s} {'us':>7s} {'ramp':>6s} {'middle':>7s} {'drain':>6s} {'idle':>6s}  idle share")
    tot = {}
    for l, _ in launches:
        if l.name not in ("seg_gemm_f32", "seg_gemm_stats_f32", "seg_gemm_res_f32") or l.flops < 1e9:
            continue
        a = list(l.args)
        cap = int(a[12]) * ((int(a[8]) + 15) // 16) + 64
        dbg = torch.zeros(2 * cap, dtype=torch.int64, device="cuda")
        a[13] = Ref(dbg.view(torch.float32))
        f = be.bind(l.name, tuple(a))
        best = None
        for _ in range(3):  # keep the shortest of three stamped runs
            dbg.zero_()
            f()
            torch.cuda.synchronize()
            d = dbg.cpu().numpy().reshape(-1, 2)
            d = d[d[:, 1] > 0].astype(np.float64) / 100.0  # us
            if best is None or d[:, 1].max() - d[:, 0].min() < best[:, 1].max() - best[:, 0].min():
                best = d
        d = best
        t0, t1 = d[:, 0].min(), d[:, 1].max()
        ts = np.linspace(t0, t1, 2001)
        mid = 0.5 * (ts[1:] + ts[:-1])
        starts, ends = np.sort(d[:, 0]), np.sort(d[:, 1])
        alive = np.searchsorted(starts, mid, side="right") - np.searchsorted(ends, mid, side="right")
        peak = alive.max()
        dt = (t1 - t0) / 2000
        idle = (1.0 - alive / peak) * dt
        t_peak = mid[np.argmax(alive >= 0.98 * peak)]
        t_last = starts[-1]
        ramp, drain = idle[mid < t_peak].sum(), idle[mid >= t_last].sum()
        middle = idle.sum() - ramp - drain
        kind = l.tag.split(":")[0]
        tot.setdefault(kind, [0, 0.0, 0.0, 0.0, 0.0, 0.0])
        for i, v in enumerate((1, t1 - t0, ramp, middle, drain, idle.sum())):
            tot[kind][i] += v
        print(f"{l.tag:34s} {len(d):6d} {peak:5d} {t1 - t0:7.1f} {ramp:6.1f} {middle:7.1f} {drain:6.1f} {idle.sum():6.1f}  {idle.sum() / (t1 - t0):5.2f}")
    for kind, (n, us, ramp, middle, drain, idle) in tot.items():
        print(f"TOTAL {kind:6s} {n:2d} launches {us:8.1f} us  ramp {ramp:6.1f}  middle {middle:6.1f}  drain {drain:6.1f}  idle {idle:7.1f} us = {idle / us:.2f} of their time")


if __name__ == "__main__":
    main()
