#!/usr/bin/env python3
"""Matrix-core utilisation of every seg_gemm launch of the step from the SQ counters.
  python tools/pmc_sq_per_launch.py <counter_collection.csv> [--workload hypelcnn]
(one rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE pass of
`bench.py --no-graph`; no tracing in the same run).

  mfma busy % = SQ_VALU_MFMA_BUSY_CYCLES / (SIMD-cycles of the launch): the share of SIMD-cycles in which the matrix pipe was
                executing, against the clock the launch actually ran at (no assumed frequency).  rocprofv3 reports both
                counters summed over their instances: GRBM_GUI_ACTIVE over the 8 XCDs (= 8 x the launch's cycles), the SQ
                counter over all 1024 SIMDs -- so SIMD-cycles = GRBM_GUI_ACTIVE x 128 (SIMDs per XCD).  Cross-check: useful %
                x 157.3 TFLOP/s reproduces the launch durations of the kernel trace (profiles/r4_per_launch_hypelcnn.txt);
  useful %    = algorithmic FLOP / (SIMD-cycles x 64 FLOP per SIMD-cycle): what of that was exact-tap work (the rest:
                zero-filled k-tile tails, ragged row / column tiles);
  cyc / MFMA  = busy cycles per algorithmic v_mfma_f32_32x32x2_f32 (4096 FLOP; 64 = the instruction's 16 passes);
  wait_inst % = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES: wave-cycles spent waiting for an instruction's operands / results."""
import collections
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load(path):
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        d = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "vgpr": r.get("VGPR_Count", "?"),
                                                     "lds": r.get("LDS_Block_Size", "?"), "grid": r.get("Grid_Size", "?")})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return [v for _, v in sorted(disp.items())]


def main():
    from tests.emu_backend import EmuBackend
    import bench
    workload = "hypelcnn"
    if "--workload" in sys.argv:
        i = sys.argv.index("--workload")
        workload = sys.argv[i + 1]
        del sys.argv[i:i + 2]
    ds = [d for d in load(sys.argv[1]) if "anonymous namespace" in d["name"]]
    starts = [i for i, d in enumerate(ds) if "nhwc_to_pnc" in d["name"]] + [len(ds)]
    steps = [ds[a:b] for a, b in zip(starts[:-1], starts[1:])]
    steps = [st for st in steps if len(st) == len(steps[-1])]
    nb = bench.CLASSIFIER_WORKLOADS[workload][5]
    ctx, ts, lr, alg = bench.build_model(nb, EmuBackend(), workload)
    ctx.capture_graphs = False
    plan = ts.compiled(nb).plan
    launches = plan.fwd + plan.bwd

    def mean(j, key):
        return sum(st[j].get(key, 0.0) for st in steps) / len(steps)

    SIMDS = 128  # per XCD: GRBM_GUI_ACTIVE arrives summed over the 8 XCDs
    print(f"{len(steps)} complete steps averaged; SIMD-cycles = GRBM_GUI_ACTIVE (sum over 8 XCDs) x 128, 64 fp32 MFMA FLOP per SIMD-cycle")
    print(f"{'launch':34s} {'GFLOP':>7s} {'mfma busy %':>11s} {'useful %':>9s} {'cyc/MFMA':>9s} {'wait_inst %':>11s} {'vgpr':>5s} {'lds':>6s} {'grid':>8s}")
    j = 0
    tot_busy = tot_act = tot_flop = 0.0
    for l in launches:
        n_k = 2 if l.name in ("mse", "sum_f32") else 1
        k = j
        j += n_k
        if not l.name.startswith("seg_gemm"):
            continue
        busy, act = mean(k, "SQ_VALU_MFMA_BUSY_CYCLES"), mean(k, "GRBM_GUI_ACTIVE")
        wc, wi = mean(k, "SQ_WAVE_CYCLES"), mean(k, "SQ_WAIT_INST_ANY")
        if act <= 0:
            continue
        tot_busy += busy
        tot_act += act
        tot_flop += l.flops
        d = steps[-1][k]
        print(f"{l.tag:34s} {l.flops / 1e9:7.2f} {100 * busy / (SIMDS * act):11.1f} {100 * l.flops / (act * SIMDS * 64):9.1f} "
              f"{busy / (l.flops / 4096):9.1f} {100 * wi / wc if wc else 0:11.1f} {d['vgpr']:>5s} {d['lds']:>6s} {d['grid']:>8s}")
    print(f"all seg_gemm launches: mfma busy {100 * tot_busy / (SIMDS * tot_act):.1f} % of the SIMD-cycles they were resident, "
          f"useful {100 * tot_flop / (tot_act * SIMDS * 64):.1f} %")


if __name__ == "__main__":
    main()
