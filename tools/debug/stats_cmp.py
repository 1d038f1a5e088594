import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests import parity_util as U
from hypelcnn_amd.backend import HipBackend
from hypelcnn_amd import graph as G
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
torch.cuda.synchronize()
plan = ct.plan
out = {}
for idx, node in enumerate(built.train_tower.nodes):
    if isinstance(node, G.LinearNode) and node.has_bn and f"mean:{idx}" in plan.buffers:
        c = node.cout
        y = plan.buffers[plan.node_aux[idx]["y"].buf][: node.out.npix * nb * c].reshape(-1, c).double()
        m, r = plan.buffers[f"mean:{idx}"][:c].double(), plan.buffers[f"rstd:{idx}"][:c].double()
        tm, tr = y.mean(0), 1.0 / torch.sqrt(y.var(0, unbiased=False) + 1e-3)
        print(f"{idx:3d} {node.branches[0].scope:28s} fused={plan.node_aux[idx].get('stats_in_gemm')} "
              f"mean err {float((m - tm).abs().max()):.2e} (max {float(tm.abs().max()):.2e}) rstd rel err {float(((r - tr) / tr).abs().max()):.2e}")
np.save(sys.argv[1], sess.grads.cpu().numpy())
