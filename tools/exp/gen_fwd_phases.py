#!/usr/bin/env python3
"""Shader-clock cycles per phase of the generator forward kernel (library built with -DGM_DIAG=7, HYPEL_LIB_PATH): thread 0 of
block 0 stamps the phase boundaries, reported through out[0..7].  GP_N (4096), GP_B (360)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from hypelcnn_amd.backend import HipBackend, Ref
be = HipBackend()
n, bands = int(os.environ.get("GP_N", 4096)), int(os.environ.get("GP_B", 360))
ks = [bands >> s for s in (0, 1, 2, 3, 2, 1, 0)]; wt = sum(ks)
rng = np.random.default_rng(0)
x = be.upload(rng.random((n, bands)).astype(np.float32))
w = be.upload((rng.standard_normal(wt) * 0.05).astype(np.float32)); b = be.upload(np.zeros(8, np.float32))
out = be.zeros(n * bands)
names = ["block set-up", "tap table + barrier", "products", "epilogue", "layer hand-over", "tile end", "-", "-"]
for enc in (0, 1):
    keep = be.zeros(be.gan_generator_keep_floats(n, bands, enc))
    for kept in (0, 1):
        for _ in range(3):
            if kept: be.call("gan_generator_fwd_keep", Ref(x), bands, n, bands, Ref(w), Ref(b), enc, Ref(out), bands, Ref(keep))
            else: be.call("gan_generator_fwd", Ref(x), bands, n, bands, Ref(w), Ref(b), enc, Ref(out), bands)
        be.synchronize()
        c = out[:8].cpu().numpy()
        print(f"only_encoder={enc} keep={kept}: total {c.sum():.0f} cycles = {c.sum() / 2400:.1f} us at 2.4 GHz: " +
              ", ".join(f"{nm} {v:.0f}" for nm, v in zip(names[:6], c[:6])))
