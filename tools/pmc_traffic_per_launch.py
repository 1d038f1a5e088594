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


def steps_of(lst):
    """The library's dispatches cut into steps (each begins with nhwc_to_pnc); only complete steps -- as many dispatches as
    the last one -- are kept."""
    hk = [d for d in lst if "anonymous namespace" in d[0]]
    starts = [i for i, d in enumerate(hk) if "nhwc_to_pnc" in d[0]] + [len(hk)]
    steps = [hk[a:b] for a, b in zip(starts[:-1], starts[1:])]
    n = len(steps[-1])
    return [st for st in steps if len(st) == n]


def write_calibration(steps_w, known_bytes):
    """Same rule as tools/pmc_traffic.py: WRITE_SIZE is calibrated on nhwc_to_pnc, which writes exactly one fp32 copy of the
    batch (the two tools then report the same bytes for the same passes)."""
    vals = [st[0][1] for st in steps_w if "nhwc_to_pnc" in st[0][0]]
    return known_bytes * len(vals) / sum(vals) if vals and sum(vals) > 0 else 1.0


def main():
    from tests.emu_backend import EmuBackend
    import bench
    workload = "hypelcnn"
    if "--workload" in sys.argv:
        i = sys.argv.index("--workload")
        workload = sys.argv[i + 1]
        del sys.argv[i:i + 2]
    fe_steps, wr_steps = steps_of(load(sys.argv[1], "FETCH_SIZE")), steps_of(load(sys.argv[2], "WRITE_SIZE"))
    _, _, patch, chans, _, nb, _ = bench.CLASSIFIER_WORKLOADS[workload]
    cal_w = write_calibration(wr_steps, nb * patch * patch * chans * 4)
    ctx, ts, lr, alg = bench.build_model(nb, EmuBackend(), workload)
    ctx.capture_graphs = False
    plan = ts.compiled(nb).plan
    launches = plan.fwd + plan.bwd

    def mean_at(steps, j):  # average over every complete step of the passes
        return sum(st[j][1] for st in steps) / len(steps)

    j = 0
    tot_alg = tot = 0.0
    print(f"{len(fe_steps)} / {len(wr_steps)} complete steps averaged; FETCH_SIZE x 2.0, WRITE_SIZE x {cal_w:.3f} "
          f"(calibrated on nhwc_to_pnc, as tools/pmc_traffic.py)")
    print(f"{'launch':34s} {'algorithmic MB':>15s} {'read MB':>9s} {'write MB':>9s} {'ratio':>6s}")
    for l in launches:
        n_k = 2 if l.name in ("mse", "sum_f32") else 1
        f, w = mean_at(fe_steps, j) * 2, mean_at(wr_steps, j) * cal_w
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
