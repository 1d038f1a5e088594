import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests import parity_util as U
from oracle import train as OT
from hypelcnn_amd.backend import HipBackend
alg = json.load(open("hypelcnn_amd/nnmodel/modelconfigs/alg_param_hypelcnn.json"))
hip = HipBackend()
rng = np.random.default_rng(360)
nb = 64
built = U.build("HYPELCNNModel", 7, 360, 2, alg, hip)
sess = built.ctx.session()
params = U.make_params("HYPELCNNModel", 7, 360, 2, alg, rng)
U.inject(sess, params)
x = rng.random((nb, 7, 7, 360)).astype(np.float32)
train_first = len(sys.argv) > 1 and sys.argv[1] == "train"
if train_first:
    onehot = np.eye(2, dtype=np.float32)[rng.integers(0, 2, nb)]
    masks = U.make_masks(built, nb, rng)
    U.run_train_step(built, x, onehot, masks)
li = U.run_eval(built, x)
ri = OT.forward_backward("HYPELCNNModel", {k: sess.get_variable("nn_core/" + k).astype(np.float64) for k in params}, x.astype(np.float64), None, 2, alg, False)
print("eval logits got", li[:2], "want", ri["logits"][:2])
ct = sess.compile(built.eval_tower, nb)
# walk the eval tower: first node whose output is all zero / differs
from hypelcnn_amd import graph as G
for idx, node in enumerate(built.eval_tower.nodes):
    v = ct.value(node.out)
    print(idx, type(node).__name__, getattr(node, "kind", ""), tuple(v.shape), "absmax", float(v.abs().max()), "finite", bool(torch.isfinite(v).all()))
