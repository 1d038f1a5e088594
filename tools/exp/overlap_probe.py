#!/usr/bin/env python3
"""How well do the filter-gradient GEMMs co-run with (a) the elementwise kernels and (b) the data-gradient GEMMs of the
backward pass when they are issued on two HIP streams at the same time?  (HYPELCNN, batch 1024, eager launches.)
Prints wall time of: each list alone, both lists back to back on one stream, both lists concurrently on two streams."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    from hypelcnn_amd.backend import HipBackend
    be = HipBackend()
    nb = 1024
    ctx, train_step, lr, alg = bench.build_model(nb, be, "hypelcnn")
    ctx.capture_graphs = False
    ct = train_step.compiled(nb)
    ct.set_input("x", torch.rand((nb, 7, 7, 145)).cuda())
    ct.set_input("labels", torch.nn.functional.one_hot(torch.randint(0, 15, (nb,)), 15).float().cuda())
    ct.forward_backward()
    torch.cuda.synchronize()
    bwd = [l for l in ct.plan.bwd if not l.name.startswith("_")]
    ew = [l for l in bwd if l.name.startswith(("bn_act_bwd", "bwd_reduce", "chanmap", "reduce_splits_f32"))]
    dg = [l for l in bwd if l.name.startswith("seg_gemm") and not l.name.startswith("seg_gemm_multi")]
    wg = [l for l in bwd if l.name.startswith("seg_gemm_multi")]
    print(f"elementwise {len(ew)} launches, dgrad {len(dg)}, wgrad-merged {len(wg)}")
    # (the package itself runs on one stream since round 5.)  SIDE_PRIO: priority of the side stream (0 = normal; the main
    # stream is the backend's, normal priority); MAIN_HI=1: run the main list on a high-priority stream of its own instead
    main_s = torch.cuda.Stream(be.device, priority=-1) if os.environ.get("MAIN_HI") == "1" else be.stream
    side_s = torch.cuda.Stream(be.device, priority=int(os.environ.get("SIDE_PRIO", "0")))
    handles = [main_s.cuda_stream, side_s.cuda_stream]

    def run(lists, reps=30):
        """lists: [(launch list, stream index)] started together; returns mean wall us."""
        bound = [[be.bind(l.name, l.args, handles[si]) for l in ls] for ls, si in lists]
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tot = 0.0
        for r in range(reps + 3):
            torch.cuda.synchronize()
            ev0.record(main_s)
            side_s.wait_event(ev0)
            # interleave the host-side issue so both queues fill at the same time
            n = max(len(b) for b in bound)
            for i in range(n):
                for b in bound:
                    if i < len(b):
                        b[i]()
            evs = torch.cuda.Event()
            evs.record(side_s)
            main_s.wait_event(evs)
            ev1.record(main_s)
            torch.cuda.synchronize()
            if r >= 3:
                tot += ev0.elapsed_time(ev1) * 1e3
        return tot / reps

    t_ew = run([(ew, 0)])
    t_dg = run([(dg, 0)])
    t_wg = run([(wg, 0)])
    print(f"alone: elementwise {t_ew:.0f} us, dgrad {t_dg:.0f} us, wgrad {t_wg:.0f} us")
    print(f"elementwise then wgrad, one stream: {run([(ew + wg, 0)]):.0f} us;  two streams: {run([(ew, 0), (wg, 1)]):.0f} us")
    print(f"dgrad then wgrad, one stream: {run([(dg + wg, 0)]):.0f} us;  two streams: {run([(dg, 0), (wg, 1)]):.0f} us")
    allm = [l for l in bwd if not l.name.startswith("seg_gemm_multi") and l.name != "reduce_splits_multi_f32"]
    print(f"whole backward chain (no wgrad) {run([(allm, 0)]):.0f} us; chain then wgrad {run([(allm + wg, 0)]):.0f} us; "
          f"chain || wgrad {run([(allm, 0), (wg, 1)]):.0f} us")


if __name__ == "__main__":
    main()
