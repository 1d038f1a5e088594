"""Debug: determinism / consistency of full-size DUALCNN towers sharing one session."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests import parity_util as U
from hypelcnn_amd.backend import HipBackend
alg = json.load(open("hypelcnn_amd/nnmodel/modelconfigs/alg_param_dualcnn.json"))
hip = HipBackend()
rng = np.random.default_rng(2018)
built = U.build("DUALCNNModel", 11, 49, 20, alg, hip)
sess = built.ctx.session()
params = U.make_params("DUALCNNModel", 11, 49, 20, alg, rng)
U.inject(sess, params)
nb = 128
x = rng.random((nb, 11, 11, 49)).astype(np.float32)
onehot = np.eye(20, dtype=np.float32)[rng.integers(0, 20, nb)]
masks = U.make_masks(built, nb, rng)
def run(lo, hi):
    ct = U.run_train_step(built, x[lo:hi], onehot[lo:hi], {k: v[lo:hi] for k, v in masks.items()})
    torch.cuda.synchronize()
    return ct, sess.grads.clone(), ct.value(built.y_conv).clone()
order = sys.argv[1] if len(sys.argv) > 1 else "c0,c0,c1,big,big,c0"
res = {}
for name in order.split(","):
    lo, hi = {"c0": (0, 64), "c1": (64, 128), "big": (0, 128)}[name]
    ct, g, lg = run(lo, hi)
    if name in res:
        print(name, "repeat: grads equal", torch.equal(g, res[name][0]), "logits equal", torch.equal(lg, res[name][1]),
              "max dg", float((g - res[name][0]).abs().max()), "max dl", float((lg - res[name][1]).abs().max()))
    res[name] = (g, lg)
if all(k in res for k in ("c0", "c1", "big")):
    want = (res["c0"][0].double() + res["c1"][0].double()) / 2
    d = (res["big"][0].double() - want).abs()
    print("big vs chunks: grads max abs", float(d.max()), "of", float(want.abs().max()),
          "logits", float((res["big"][1] - torch.cat([res["c0"][1], res["c1"][1]])).abs().max()))
    # which variables differ most
    worst = []
    for v in sess.trainable:
        sl = slice(v.offset, v.offset + v.size)
        m = float(want[sl].abs().max())
        worst.append((float(d[sl].max()) / max(m, 1e-12), v.name))
    print(sorted(worst, reverse=True)[:8])
