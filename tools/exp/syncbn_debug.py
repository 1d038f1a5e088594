import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from hypelcnn_amd.backend import HipBackend
from tests import parity_util as U
alg = {"drop_out_ratio": 0.7, "filter_count": 96, "learning_rate": 3e-4, "learning_rate_decay_factor": 0.96,
       "learning_rate_decay_step": 350, "lrelu_alpha": 0.18, "optimizer": "AdamOptimizer", "bn_decay": 0.95,
       "l2regularizer_scale": 1e-5, "spectral_hierarchy_level": 3, "spatial_hierarchy_level": 3,
       "degradation_coeff": 3, "use_residual": True}
rng = np.random.default_rng(5)
patch, ch, classes, nb = 7, 33, 6, 256
built = U.build("HYPELCNNModel", patch, ch, classes, alg, HipBackend(), with_eval=False)
sess = built.ctx.session()
params = U.make_params("HYPELCNNModel", patch, ch, classes, alg, rng)
x = rng.random((nb, patch, patch, ch)).astype(np.float32)
onehot = np.eye(classes, dtype=np.float32)[rng.integers(0, classes, nb)]
masks = U.make_masks(built, nb, rng)
res = {}
for name, kw in (("plain", dict(sync_bn=False)), ("sync", dict(sync_bn=True))):
    U.inject(sess, params)
    ct = sess.compile(built.train_tower, nb, loss=built.train_step.loss, external_masks=True, **kw)
    U.feed(ct, x, onehot, masks)
    ct.forward_backward()
    torch.cuda.synchronize()
    res[name] = (sess.grads.clone(), ct.value(built.y_conv).clone(), ct)
    try:
        U.compare_step(built, ct, params, x, onehot, masks, "HYPELCNNModel", classes, alg, tol_logit=1e-3, tol_grad=1e-3)
        print(name, "matches the oracle")
    except AssertionError as e:
        print(name, "ORACLE MISMATCH", str(e)[:300])
    print(name, "launches", len(ct.plan.fwd), len(ct.plan.bwd), "loss", ct.loss_value())
gp, gs = res["plain"][0], res["sync"][0]
print("logits diff", float((res["plain"][1] - res["sync"][1]).abs().max()))
for v in sess.trainable:
    a, b = gp[v.offset:v.offset + v.size], gs[v.offset:v.offset + v.size]
    d = float((a - b).abs().max()); s = float(a.abs().max())
    if d > 1e-5 * max(s, 1e-9):
        print(f"{v.name:50s} max {s:.3e} diff {d:.3e} rel {d / max(s, 1e-12):.2e}")
# forward intermediates: mean / rstd per node
pp, ps = res["plain"][2].plan, res["sync"][2].plan
for k in pp.buffers:
    if k.startswith(("mean:", "rstd:")) and k in ps.buffers:
        d = float((pp.buffers[k] - ps.buffers[k]).abs().max())
        if d > 1e-6:
            print(k, "diff", d)
dist.destroy_process_group()
