#!/usr/bin/env python3
"""Shader-clock cycles per phase of the generator backward kernel (library built with -DGM_DIAG=5, HYPEL_LIB_PATH):
thread 0 of block 0 stamps the phase boundaries."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from hypelcnn_amd.backend import HipBackend, Ref
be = HipBackend()
n, bands = int(os.environ.get("GP_N", 4096)), int(os.environ.get("GP_B", 360))
ks = [bands >> s for s in (0, 1, 2, 3, 2, 1, 0)]
wt = sum(ks)
rng = np.random.default_rng(0)
x = be.upload(rng.random((n, bands)).astype(np.float32)); d = be.upload(rng.standard_normal((n, bands)).astype(np.float32))
w = be.upload((rng.standard_normal(wt) * 0.05).astype(np.float32)); b = be.upload(np.zeros(8, np.float32))
blocks = be.gan_generator_blocks(n)
pw = be.zeros(blocks * wt); pb = be.zeros(blocks * 8); dx = be.zeros(n * bands)
kept = os.environ.get("GP_KEPT", "0") == "1"  # the form the train ops run: backward from the forward pass's kept activations
out = be.zeros(n * bands)
for enc in (0, 1):
    keep = be.zeros(be.gan_generator_keep_floats(n, bands, enc)) if kept else None
    for _ in range(3):
        if kept:
            be.call("gan_generator_fwd_keep", Ref(x), bands, n, bands, Ref(w), Ref(b), enc, Ref(out), bands, Ref(keep))
            be.call("gan_generator_bwd_kept", Ref(x), bands, Ref(d), bands, n, bands, Ref(w), Ref(b), enc, Ref(dx), bands, 0,
                    Ref(pw), Ref(pb), Ref(keep))
        else:
            be.call("gan_generator_bwd", Ref(x), bands, Ref(d), bands, n, bands, Ref(w), Ref(b), enc, Ref(dx), bands, 0, Ref(pw), Ref(pb))
    be.synchronize()
    c = pb[:8].cpu().numpy()
    names = ["tile setup", "forward recompute", "dout load", "step A (dz, skips, X)", "step B products", "diagonal sums", "step C (dgrad)", "write-out"]
    print(f"only_encoder={enc}: total {c.sum():.0f} cycles = {c.sum() / 2400:.1f} us at 2.4 GHz")
    for nm, v in zip(names, c):
        print(f"   {nm:24s} {v:10.0f} cycles  {100 * v / c.sum():5.1f} %")
