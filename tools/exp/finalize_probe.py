"""Per-launch cost of the BN finalisers in a dependent chain (HIP graph of 50 identical launches; a trivial kernel is 1.6 us)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hypelcnn_amd.backend import HipBackend, Ref
be = HipBackend()


def chain(mk, n=50, reps=20):
    g = be.capture([mk() for _ in range(n)])
    for _ in range(3):
        g()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(be.stream)
    for _ in range(reps):
        g()
    b.record(be.stream)
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps / n * 1e3


rows = 50176
for c, n_chunks, chunk in ((120, 392, 128), (480, 392, 128), (240, 256, 196), (15, 128, 392)):
    part = torch.randn(n_chunks * 2 * c, device=be.device).abs()
    mean, rstd = torch.zeros(c, device=be.device), torch.zeros(c, device=be.device)
    mm, mv = torch.zeros(c, device=be.device), torch.ones(c, device=be.device)
    sums, dpar = torch.zeros(2 * c, device=be.device), torch.zeros(c, device=be.device)
    f = chain(lambda: be.bind("bn_finalize", (Ref(part), n_chunks, chunk, rows, c, 1e-3, Ref(mean), Ref(rstd), Ref(mm), Ref(mv), 0.95)))
    b = chain(lambda: be.bind("bwd_reduce_finalize", (Ref(part), n_chunks, c, Ref(sums), Ref(dpar), 0)))
    print(f"c={c:4d} chunks={n_chunks:4d}  bn_finalize {f:6.2f} us   bwd_reduce_finalize {b:6.2f} us")
