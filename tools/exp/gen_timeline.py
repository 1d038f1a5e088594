#!/usr/bin/env python3
"""Occupancy timeline of the register-ring generator backward kernel (library built with -DGM_DIAG=6, HYPEL_LIB_PATH):
every block reports start / end on the 100 MHz wall clock and its HW_ID / XCC_ID."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from hypelcnn_amd.backend import HipBackend, Ref
be = HipBackend()
n, bands = int(os.environ.get("GP_N", 8192)), int(os.environ.get("GP_B", 360))
ks = [bands >> s for s in (0, 1, 2, 3, 2, 1, 0)]; wt = sum(ks)
rng = np.random.default_rng(0)
x = be.upload(rng.random((n, bands)).astype(np.float32)); d = be.upload(rng.standard_normal((n, bands)).astype(np.float32))
w = be.upload((rng.standard_normal(wt) * 0.05).astype(np.float32)); b = be.upload(np.zeros(8, np.float32))
blocks = be.gan_generator_blocks(n)
pw = be.zeros(blocks * wt); pb = be.zeros(blocks * 8); dx = be.zeros(n * bands); out = be.zeros(n * bands)
keep = be.zeros(be.gan_generator_keep_floats(n, bands, 0))
for _ in range(3):
    be.call("gan_generator_fwd_keep", Ref(x), bands, n, bands, Ref(w), Ref(b), 0, Ref(out), bands, Ref(keep))
    be.call("gan_generator_bwd_kept", Ref(x), bands, Ref(d), bands, n, bands, Ref(w), Ref(b), 0, Ref(dx), bands, 0, Ref(pw), Ref(pb), Ref(keep))
be.synchronize()
r = pb.cpu().numpy().view(np.int32).reshape(blocks, 8)
t0, t1, hw, xcc = r[:, 0].astype(np.int64) & 0xffffffff, r[:, 1].astype(np.int64) & 0xffffffff, r[:, 2], r[:, 3]
base = t0.min()
s, e = (t0 - base) / 100.0, (t1 - base) / 100.0  # us
cu = ((xcc & 0xf).astype(np.int64) << 12) | (hw & 0xff00)  # xcc | se / sh / cu bits of HW_ID
print(f"blocks {blocks}, launch span {e.max():.1f} us; block duration mean {np.mean(e - s):.1f} us, min {np.min(e - s):.1f}, max {np.max(e - s):.1f}")
print("start times (us) histogram:", np.histogram(s, bins=[0, 5, 20, 60, 100, 140, 180, 400])[0])
u, c = np.unique(cu, return_counts=True)
print(f"distinct (xcc, se, cu) ids {len(u)}; blocks per id: min {c.min()} max {c.max()}; ids with 1 / 2 / 3+ blocks: {(c == 1).sum()} / {(c == 2).sum()} / {(c >= 3).sum()}")
# concurrency on a CU: for ids with 2 blocks, overlap fraction
ov = []
for k in u[c == 2]:
    i = np.where(cu == k)[0]
    ov.append(max(0.0, min(e[i[0]], e[i[1]]) - max(s[i[0]], s[i[1]])) / max(e[i].max() - s[i].min(), 1e-9))
if ov: print(f"CUs with two blocks: mean overlap of the pair {np.mean(ov):.2f} of their span")
