"""Cost of one dependent launch inside a HIP-graph replay: N x step_inc / N x (fill of 1 MB) chains."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hypelcnn_amd.backend import HipBackend, Ref
be = HipBackend()
step = be.zeros(2, dtype=torch.int64)
buf = be.zeros(1 << 20)
for name, mk in (("step_inc", lambda: be.bind("step_inc", (Ref(step),))),
                 ("fill 4 MB", lambda: be.bind("fill_f32", (Ref(buf), 1 << 20, 1.0)))):
    for n in (50, 200):
        g = be.capture([mk() for _ in range(n)])
        for _ in range(3): g()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(be.stream)
        for _ in range(20): g()
        b.record(be.stream)
        torch.cuda.synchronize()
        print(f"{name:10s} chain of {n:3d}: {a.elapsed_time(b) / 20 / n * 1e3:.2f} us per launch")
