#!/usr/bin/env python3
"""HBM traffic of every seg_gemm launch of the step vs its algorithmic bytes.
  python tools/pmc_traffic_per_launch.py <fetch_counter_collection.csv> <write_counter_collection.csv>
(two rocprofv3 --pmc passes of `bench.py --no-graph`; FETCH_SIZE doubled per MI355X_MICROARCH.md)."""
import collections
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load(path, counter):
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            disp[int(r["Dispatch_Id"])] = (r["Kernel_Name"], float(r["Counter_Value"]) * 1024)
    return [v for _, v in sorted(disp.items())]


def last_step(lst):
    hk = [d for d in lst if "anonymous namespace" in d[0]]
    starts = [i for i, d in enumerate(hk) if "nhwc_to_pnc" in d[0]]
    return hk[starts[-1]:]


def main():
    from tests.emu_backend import EmuBackend
    import bench
    workload = "hypelcnn"
    if "--workload" in sys.argv:
        i = sys.argv.index("--workload")
        workload = sys.argv[i + 1]
        del sys.argv[i:i + 2]
    fe, wr = last_step(load(sys.argv[1], "FETCH_SIZE")), last_step(load(sys.argv[2], "WRITE_SIZE"))
    nb = bench.CLASSIFIER_WORKLOADS[workload][5]
    ctx, ts, lr, alg = bench.build_model(nb, EmuBackend(), workload)
    ctx.capture_graphs = False
    plan = ts.compiled(nb).plan
    launches = [l for l in plan.fwd + plan.bwd if l.name not in ("_fork", "_join")]
    j = 0
    tot_alg = tot = 0.0
    print(f"{'launch':34s} {'algorithmic MB':>15s} {'read MB':>9s} {'write MB':>9s} {'ratio':>6s}")
    for l in launches:
        n_k = 2 if l.name in ("mse", "sum_f32") else 1
        f, w = fe[j][1] * 2, wr[j][1]
        j += n_k
        if not l.name.startswith("seg_gemm"):
            continue
        tot_alg += l.bytes
        tot += f + w
        if l.bytes > 20e6:
            print(f"{l.tag:34s} {l.bytes / 1e6:15.1f} {f / 1e6:9.1f} {w / 1e6:9.1f} {(f + w) / l.bytes:6.2f}")
    n = sum(1 for l in launches if l.name.startswith("seg_gemm"))
    print(f"all {n} seg_gemm launches: algorithmic {tot_alg / 1e6:.0f} MB, measured {tot / 1e6:.0f} MB "
          f"({tot / n / 1e6:.1f} MB per launch)")


if __name__ == "__main__":
    main()
