#!/usr/bin/env python3
"""Isolated timings of the wide-spectrum generator kernels (torch events on the backend's stream; HYPEL_LIB_PATH = alt build).
  GP_N (4096), GP_B (360), GP_REPS (50).  Prints us per launch: forward plain / keeping, backward from kept activations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from hypelcnn_amd.backend import HipBackend, Ref
be = HipBackend()
bands, reps = int(os.environ.get("GP_B", 360)), int(os.environ.get("GP_REPS", 50))
ks = [bands >> s for s in (0, 1, 2, 3, 2, 1, 0)]
wt = sum(ks)
rng = np.random.default_rng(0)
tag = os.environ.get("GP_TAG", os.path.basename(os.environ.get("HYPEL_LIB_PATH", "tree")))
for n in [int(v) for v in os.environ.get("GP_N", "4096,8192").split(",")]:
    x = be.upload(rng.random((n, bands)).astype(np.float32)); d = be.upload(rng.standard_normal((n, bands)).astype(np.float32))
    w = be.upload((rng.standard_normal(wt) * 0.05).astype(np.float32)); b = be.upload(np.zeros(8, np.float32))
    blocks = be.gan_generator_blocks(n)
    pw = be.zeros(blocks * wt); pb = be.zeros(blocks * 8); dx = be.zeros(n * bands); out = be.zeros(n * bands)
    for enc in (0, 1):
        keep = be.zeros(be.gan_generator_keep_floats(n, bands, enc))
        def fwd(): be.call("gan_generator_fwd", Ref(x), bands, n, bands, Ref(w), Ref(b), enc, Ref(out), bands)
        def fwdk(): be.call("gan_generator_fwd_keep", Ref(x), bands, n, bands, Ref(w), Ref(b), enc, Ref(out), bands, Ref(keep))
        def bwdk(): be.call("gan_generator_bwd_kept", Ref(x), bands, Ref(d), bands, n, bands, Ref(w), Ref(b), enc, Ref(dx), bands, 0,
                            Ref(pw), Ref(pb), Ref(keep))
        res = []
        for f in (fwd, fwdk, bwdk):
            for _ in range(5): f()
            be.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): f()
            e1.record(); be.synchronize()
            res.append(e0.elapsed_time(e1) * 1000 / reps)
        print(f"{tag:28s} n={n:5d} B={bands} enc={enc}: fwd {res[0]:7.1f} us  fwd_keep {res[1]:7.1f} us  bwd_kept {res[2]:7.1f} us", flush=True)
