#!/usr/bin/env python3
"""Do two independent branches of a captured HIP graph run CONCURRENTLY when the graph is replayed?  Two long launches with few
blocks each (8 blocks on 256 CUs: they cannot slow each other), on one stream vs forked onto a second stream, eager and captured.
Serial time = 2 x one launch, concurrent = 1 x."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from hypelcnn_amd.backend import HipBackend, Ref  # noqa: E402
from hypelcnn_amd.plan import GemmTables  # noqa: E402


def main():
    be = HipBackend()
    m, k, n = 1024, 32768, 128
    rng = np.random.default_rng(0)

    def launch_on(stream_handle):
        a = torch.from_numpy(rng.standard_normal((m, k)).astype(np.float32)).cuda()
        b = torch.from_numpy(rng.standard_normal((n, k)).astype(np.float32)).cuda()
        c = torch.zeros(m * n, device="cuda")
        tb = GemmTables()
        tb.add_group(0, [(0, 0, k)], m)
        g, sg, t, _ = tb.finalize(n)
        gt, st, tt = be.upload(g), be.upload(sg), be.upload(t)
        keep = (a, b, c, gt, st, tt)
        return be.bind("seg_gemm_f32", (Ref(a), k, 0, Ref(b), k, 1, Ref(c), n, n, Ref(gt), Ref(st), Ref(tt), len(t), None,
                                        0x8000 | (3 << 8)), stream_handle), keep

    main_s = be.stream
    side_s = torch.cuda.Stream(be.device)
    f_main1, k1 = launch_on(main_s.cuda_stream)
    f_main2, k2 = launch_on(main_s.cuda_stream)
    f_side2, k3 = launch_on(side_s.cuda_stream)
    ev_f, ev_j = torch.cuda.Event(), torch.cuda.Event()

    def serial():
        f_main1()
        f_main2()

    def forked():
        ev_f.record(main_s)
        side_s.wait_event(ev_f)
        f_main1()
        f_side2()
        ev_j.record(side_s)
        main_s.wait_event(ev_j)

    def timed(fn, reps=10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0.record(main_s)
            fn()
            e1.record(main_s)
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        return sorted(ts)[len(ts) // 2]

    with torch.cuda.stream(main_s):
        one = timed(f_main1)
        print(f"one launch (8 blocks, K = {k}): {one:9.1f} us")
        print(f"eager, one stream              : {timed(serial):9.1f} us")
        print(f"eager, forked onto 2 streams   : {timed(forked):9.1f} us")
        g_serial = be.capture([serial])
        g_forked = be.capture([forked])
        print(f"graph, one stream              : {timed(g_serial):9.1f} us")
        print(f"graph, forked branch           : {timed(g_forked):9.1f} us")
    for var in ("DEBUG_HIP_GRAPH_DOT_PRINT",):
        pass


if __name__ == "__main__":
    main()
