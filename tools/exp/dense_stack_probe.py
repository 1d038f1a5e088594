#!/usr/bin/env python3
"""tools/exp/dense_stack_probe.py: back-to-back launch time of hypel_dense_stack_fwd / _bwd for a few shapes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from hypelcnn_amd.backend import HipBackend, Ref  # noqa: E402

be = HipBackend()


def probe(widths, n, reps=300):
    L = len(widths) - 1
    wq = list(widths) + [0] * (5 - len(widths))
    wtot = sum(a * b for a, b in zip(widths, widths[1:]))
    btot = sum(widths[1:])
    x = torch.rand(n, widths[0], device="cuda")
    w = torch.randn(wtot, device="cuda") * 0.1
    b = torch.randn(btot, device="cuda") * 0.1
    out = torch.zeros(n, widths[-1], device="cuda")
    dout = torch.randn(n, widths[-1], device="cuda")
    dx = torch.zeros(n, widths[0], device="cuda")
    blocks = be.dense_stack_blocks(n)
    pw = torch.zeros(blocks * wtot, device="cuda")
    pb = torch.zeros(blocks * btot, device="cuda")
    f = be.bind("dense_stack_fwd", (Ref(x), widths[0], n, L, *wq, (1 << (L - 1)) - 1, 0.1, Ref(w), Ref(b), Ref(out), widths[-1]))
    g = be.bind("dense_stack_bwd", (Ref(x), widths[0], Ref(dout), widths[-1], n, L, *wq, (1 << (L - 1)) - 1, 0.1, Ref(w), Ref(b),
                                    Ref(dx), widths[0], 0, Ref(pw), Ref(pb)))
    res = []
    for fn in (f, g):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        res.append(a.elapsed_time(e) * 1e3 / reps)
    print(f"widths {widths} n {n}: fwd {res[0]:.1f} us  bwd {res[1]:.1f} us")


for widths, n in [((64, 64, 64, 32), 2048), ((64, 64), 2048), ((64, 64, 64, 32), 16), ((64, 64, 64, 32), 256),
                  ((64, 64, 64, 32), 4096), ((16, 16, 16, 8), 2048)]:
    probe(widths, n)
