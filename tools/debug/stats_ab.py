import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests import parity_util as U
from hypelcnn_amd import plan as P
from hypelcnn_amd.backend import HipBackend
alg = json.load(open("hypelcnn_amd/nnmodel/modelconfigs/alg_param_hypelcnn.json"))
hip = HipBackend()
nb = 64
def run(fused):
    P.STATS_EPILOGUE = fused
    rng = np.random.default_rng(1234)
    built = U.build("HYPELCNNModel", 7, 145, 15, alg, hip)
    sess = built.ctx.session()
    params = U.make_params("HYPELCNNModel", 7, 145, 15, alg, rng)
    U.inject(sess, params)
    x = rng.random((nb, 7, 7, 145)).astype(np.float32)
    onehot = np.eye(15, dtype=np.float32)[rng.integers(0, 15, nb)]
    masks = U.make_masks(built, nb, rng)
    ct = U.run_train_step(built, x, onehot, masks)
    torch.cuda.synchronize()
    return built, sess, ct
a = run(True); b = run(False)
for name in a[2].plan.buffers:
    if name in b[2].plan.buffers and name.split(":")[0] in ("y", "z", "mean", "rstd", "g", "dy"):
        ta, tb = a[2].plan.buffers[name], b[2].plan.buffers[name]
        n = min(ta.numel(), tb.numel())
        d = (ta[:n] - tb[:n]).abs()
        m = float(tb[:n].abs().max())
        if float(d.max()) > 1e-5 * max(m, 1e-30):
            i = int(d.argmax())
            print(f"{name:24s} max diff {float(d.max()):.3e} of {m:.3e} at {i} ({float(ta[i])} vs {float(tb[i])})  n_diff>1e-5: {int((d > 1e-5 * m).sum())}")
ga, gb = a[1].grads, b[1].grads
for v in a[1].trainable:
    sl = slice(v.offset, v.offset + v.size)
    d = float((ga[sl] - gb[sl]).abs().max()); m = float(gb[sl].abs().max())
    if d > 1e-4 * m:
        print("grad", v.name, d / m)
