import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests import parity_util as U
from hypelcnn_amd.backend import HipBackend
alg = json.load(open("hypelcnn_amd/nnmodel/modelconfigs/alg_param_hypelcnn.json"))
hip = HipBackend()
nb = 64
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
plan = ct.plan
idx, c, ch = 8, 120, 35
aux = plan.node_aux[idx]
rows = 49 * nb
y = plan.buffers[aux["y"].buf][: rows * c].reshape(rows, c)[:, ch].clone()
dz = plan.buffers["g:z:8"][: rows * c].reshape(rows, c)[:, ch].clone()
dy = plan.buffers["dy:z:8"][: rows * c].reshape(rows, c)[:, ch].clone()
mean, rstd = plan.buffers[f"mean:{idx}"][ch], plan.buffers[f"rstd:{idx}"][ch]
beta = sess.params[aux["beta"].offset + ch]
xhat = (y - mean) * rstd
pre = xhat + beta
r = 45 * nb + 11
print("row", r, "y", float(y[r]), "xhat", float(xhat[r]), "pre", float(pre[r]), "beta", float(beta), "mean", float(mean), "rstd", float(rstd))
print("n exactly zero pre:", int((pre == 0).sum()), "n |pre|<1e-6:", int((pre.abs() < 1e-6).sum()))
a = alg["lrelu_alpha"]
for name, slope_at in (("alpha", a), ("one", 1.0)):
    slope = torch.where(pre > 0, torch.ones_like(pre), torch.full_like(pre, a))
    slope[r] = slope_at
    dyh = dz * slope
    s0, s1 = dyh.double().sum(), (dyh.double() * xhat.double()).sum()
    want = rstd * (dyh - float(s0) / rows - xhat * float(s1) / rows)
    print(name, "max |dy - want|", float((dy - want).abs().max()), "of", float(dy.abs().max()))
print("grad beta product", float(sess.grads[aux["beta"].offset + ch]))
