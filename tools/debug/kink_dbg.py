import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests import parity_util as U
from oracle import train as OT
from hypelcnn_amd.backend import HipBackend
alg = json.load(open("hypelcnn_amd/nnmodel/modelconfigs/alg_param_hypelcnn.json"))
hip = HipBackend()
rng = np.random.default_rng(1234)
nb = 64
built = U.build("HYPELCNNModel", 7, 145, 15, alg, hip)
sess = built.ctx.session()
params = U.make_params("HYPELCNNModel", 7, 145, 15, alg, rng)
U.inject(sess, params)
x = rng.random((nb, 7, 7, 145)).astype(np.float32)
onehot = np.eye(15, dtype=np.float32)[rng.integers(0, 15, nb)]
masks = U.make_masks(built, nb, rng)
ct = U.run_train_step(built, x, onehot, masks)
def errs(r):
    out = {}
    for k, g in r["grads"].items():
        got = sess.get_gradient("nn_core/" + k)
        out[k] = np.abs(got - g).max() / max(np.abs(g).max(), 1e-6)
    return sorted(out.items(), key=lambda t: -t[1])[:4]
fa = {}
cur = OT.forward_backward("HYPELCNNModel", {k: v.copy() for k, v in params.items()}, x.astype(np.float64), onehot.astype(np.float64), 15, alg, True, masks)
for it in range(3):
    print("iter", it, "worst", errs(cur))
    force, n_amb, n_flip = U.product_kink_decisions(built, ct, cur, alg)
    print("  flips", {sc: d for sc, d in force.items()})
    for sc, d in force.items():
        for i in d:
            a = cur["trace"][sc].reshape(-1)[i]
            print("   ", sc, i, "oracle act", a)
        fa.setdefault(sc, {}).update(d)
    cur = OT.forward_backward("HYPELCNNModel", {k: v.copy() for k, v in params.items()}, x.astype(np.float64), onehot.astype(np.float64), 15, alg, True, masks, kink_force=fa)
print("final", errs(cur))
# near-kink candidates in the suspicious scope
from hypelcnn_amd import graph as G
plan = ct.plan
for idx, node in enumerate(built.train_tower.nodes):
    if not isinstance(node, G.LinearNode) or not node.has_bn or node.act is None:
        continue
    off = 0
    for b in node.branches:
        if b.scope == "connector_1_conv3x3":
            c = node.cout
            aux = plan.node_aux[idx]
            y = plan.buffers[aux["y"].buf][: node.out.npix * nb * c].reshape(node.out.npix, nb, c)
            mean = plan.buffers[f"mean:{idx}"][:c]; rstd = plan.buffers[f"rstd:{idx}"][:c]
            beta = sess.params[aux["beta"].offset:aux["beta"].offset + c]
            yh = ((y - mean) * rstd + beta).permute(1, 0, 2).cpu().numpy()[:, :, off:off + b.cout].reshape(-1)
            act64 = cur["trace"][b.scope].reshape(-1)
            pre64 = np.where(act64 > 0, act64, act64 / alg["lrelu_alpha"])
            order = np.argsort(np.abs(pre64))[:8]
            for i in order:
                print("cand", i, "pre64", pre64[i], "product yh", yh[i])
            print("max |yh - pre64|", np.abs(yh - pre64).max(), "ybuf numel", plan.buffers[aux["y"].buf].numel(), "expected", node.out.npix * nb * c)
        off += b.cout
