"""Shared harness for the classifier parity tests: build a model through the PRODUCT's plugin API on
a given backend (numpy emulation on CPU, HIP on the GPU box), inject oracle-generated parameters,
run one training step / one inference pass, and compare everything with the oracle (float64)."""
import numpy as np
import torch

from hypelcnn_amd.common import common_nn_ops as cno
from oracle import models as OM, train as OT


def to_pixel_major(a):
    """[N, H, W, C] -> flat [P, N, C]; [N, C] unchanged."""
    if a.ndim == 2:
        return a
    n, h, w, c = a.shape
    return a.reshape(n, h * w, c).transpose(1, 0, 2)


class Built:
    pass


def build(model_name, patch, channels, classes, alg, backend, with_eval=True):
    model = cno.get_model_from_name(model_name)
    template = cno.Template("nn_core", model.create_tensor_graph, class_count=classes)
    ctx = cno.GraphContext(template, backend)
    ctx.external_masks = True
    ctx.capture_graphs = False
    images = cno.Placeholder("x", (patch, patch), channels)
    labels = cno.Placeholder("labels", None, classes)
    b = Built()
    b.model, b.template, b.ctx = model, template, ctx
    b.y_conv, b.cross_entropy, b.lr, b.train_step = cno.optimize_nn(
        template, images, labels, "/gpu:0", "training", alg, model.get_loss_func, ctx=ctx)
    b.train_tower = template.towers[0]
    if with_eval:
        outs = template(cno.ModelInputParams(x=cno.Placeholder("x", (patch, patch), channels), y=None,
                                             device_id="/gpu:0", is_training=False), algorithm_params=alg)
        b.eval_out, b.eval_tower = outs, outs.tower
    return b


def inject(sess, params):
    for k, v in params.items():
        sess.set_variable("nn_core/" + k, v)


def make_params(model_name, patch, channels, classes, alg, rng):
    if model_name == "HYPELCNNModel":
        p = OM.hypelcnn_init_params(patch, channels, classes, alg, rng, np.float64)
        for k in p:
            if k.endswith("beta") or k.endswith("moving_mean"):
                p[k] = rng.standard_normal(p[k].shape) * 0.1
            if k.endswith("moving_variance"):
                p[k] = rng.random(p[k].shape) + 0.5
        return p
    table = (OM.dualcnn_layer_table if model_name == "DUALCNNModel" else OM.concnn_layer_table)(patch, channels,
                                                                                                classes, alg)
    p = OM.xavier_init_params(table, rng, np.float64)
    for k in p:
        if k.endswith("biases"):
            p[k] = rng.standard_normal(p[k].shape) * 0.05
    return p


def make_masks(built, nb, rng):
    """One {0, 1/keep} mask per dropout site, in call order; returns (oracle dict, list of arrays)."""
    masks = {}
    from hypelcnn_amd import graph as G
    for node in built.train_tower.nodes:
        keep = getattr(node, "dropout_keep", None)
        if keep is None:
            continue
        out = node.out
        shape = (nb, out.c) if out.hw is None else (nb, out.hw[0], out.hw[1], out.c)
        masks[f"dropout_{node.dropout_index}"] = (rng.random(shape) < keep) / keep
    return masks


def run_train_step(built, x, onehot, masks):
    sess = built.ctx.session()
    nb = x.shape[0]
    ct = built.train_step.compiled(nb)
    dev = ct.input("x").device
    ct.set_input("x", torch.as_tensor(x, dtype=torch.float32).to(dev))
    ct.set_input("labels", torch.as_tensor(onehot, dtype=torch.float32).to(dev))
    for key, m in masks.items():
        idx = int(key.split("_")[1])
        buf = ct.dropout_mask(idx)
        buf.copy_(torch.as_tensor(np.ascontiguousarray(to_pixel_major(m)).reshape(-1), dtype=torch.float32).to(dev))
    ct.forward_backward()
    return ct


def compare_step(built, ct, params, x, onehot, masks, model_name, classes, alg, tol_logit=1e-3, tol_grad=2e-3,
                 check=None):
    """Oracle (float64) vs product after one forward+backward.  Returns the oracle result."""
    sess = built.ctx.session()
    ref = OT.forward_backward(model_name, {k: v.copy() for k, v in params.items()}, x.astype(np.float64),
                              onehot.astype(np.float64), classes, alg, True, masks)
    logits = ct.value(built.y_conv).cpu().numpy()
    err = np.abs(logits - ref["logits"]).max()
    assert err < tol_logit, f"logits max abs err {err}"
    assert abs(ct.loss_value() - ref["loss"]) < tol_logit * max(1.0, abs(ref["loss"])), (ct.loss_value(), ref["loss"])
    worst = ("", 0.0)
    for k, g in ref["grads"].items():
        got = sess.get_gradient("nn_core/" + k)
        scale = max(np.abs(g).max(), 1e-6)
        e = np.abs(got - g).max() / scale
        if e > worst[1]:
            worst = (k, e)
    assert worst[1] < tol_grad, f"gradient {worst[0]} rel err {worst[1]}"
    for k, v in ref["new_moving"].items():
        got = sess.get_variable("nn_core/" + k)
        assert np.abs(got - v).max() < 1e-4 * max(1.0, np.abs(v).max()), k
    return ref, err, worst


def run_eval(built, x):
    sess = built.ctx.session()
    ct = sess.compile(built.eval_tower, x.shape[0])
    dev = ct.input("x").device
    ct.set_input("x", torch.as_tensor(x, dtype=torch.float32).to(dev))
    ct.forward()
    return ct.value(built.eval_out.y_conv).cpu().numpy()
