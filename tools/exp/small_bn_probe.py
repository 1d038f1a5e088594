"""Per-launch cost of the small-rows BN kernels in a dependent chain (HIP graph of 50 identical launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hypelcnn_amd.backend import HipBackend, Ref
be = HipBackend()
rows = 1024


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


step = be.zeros(2, dtype=torch.int64)
print(f"step_inc                 {chain(lambda: be.bind('step_inc', (Ref(step),))):6.2f} us")
for c, use_mask in ((15, False), (108, True), (326, True), (980, True), (7105, False)):
    y = torch.randn(rows * c, device=be.device)
    z = torch.empty_like(y)
    dz = torch.randn_like(y)
    dy = torch.empty_like(y)
    mask = (torch.rand(rows * c, device=be.device) < 0.7).float() if use_mask else None
    beta = torch.zeros(c, device=be.device)
    mean, rstd = torch.zeros(c, device=be.device), torch.ones(c, device=be.device)
    mm, mv = torch.zeros(c, device=be.device), torch.ones(c, device=be.device)
    dbeta = torch.zeros(c, device=be.device)
    m = Ref(mask) if mask is not None else None
    f = chain(lambda: be.bind("bn_act_small_fwd", (Ref(y), c, rows, c, 1e-3, Ref(beta), 1, 0.18, m, c, Ref(mean), Ref(rstd),
                                                  Ref(mm), Ref(mv), 0.95, Ref(z), c)))
    b = chain(lambda: be.bind("bn_act_small_bwd", (Ref(dz), c, Ref(y), c, rows, c, Ref(mean), Ref(rstd), Ref(beta), 1, 0.18,
                                                  m, c, Ref(dy), c, Ref(dbeta), 0)))
    print(f"c={c:5d} mask={use_mask!s:5s}  fwd {f:6.2f} us   bwd {b:6.2f} us")
