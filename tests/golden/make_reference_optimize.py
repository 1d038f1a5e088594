#!/usr/bin/env python3
"""Pin `optimize_nn` + `get_loss_func` by executing the reference's own text (build container only; needs /root/reference).

`common/common_nn_ops.py:208-240` (optimize_nn: mean of the per-sample loss, staircase exponential LR decay, the optimiser by
name, tf_slim's create_train_op) and the three `get_loss_func`s (`nnmodel/HYPELCNNModel.py:101-112`, `DUALCNNModel.py:87-89`,
`CONCNNModel.py:66-68`) run UNCHANGED under the float64 recording engine (`tf_standin.OracleEngine`) with a stand-in for the
handful of `tf.compat.v1.train` calls they make (restated: exponential_decay, AdamOptimizer / MomentumOptimizer records; the
global step is a symbol).  Written to tests/golden/reference_optimize.json / .npz per case: the optimiser's class, name and
hyper-parameters, the LR at ten steps, the loss value and the gradient of the train op's loss w.r.t. every trainable variable
-- through the REFERENCE's composition model -> get_loss_func -> reduce_mean.  `tests/test_reference_optimize.py` holds
`oracle/train.py`, `oracle/models.py::*_loss` and the product's `LearningRate` / optimiser settings to it.  Data only."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import tf_standin as S  # noqa: E402
from hypelcnn_amd import tf_facade as F  # noqa: E402
from hypelcnn_amd import tfgan_facade as TG  # noqa: E402

LR_STEPS = [0, 1, 349, 350, 351, 700, 1049, 1050, 3500, 35000]
CFG = os.path.join(S.REF, "nnmodel", "modelconfigs")
# (case, model, config, overrides, patch, channels, classes, batch)
CASES = [
    ("hypelcnn", "HYPELCNNModel", "alg_param_hypelcnn.json", {"filter_count": 48}, 5, 21, 5, 6),
    ("hypelcnn_nonres", "HYPELCNNModel", "alg_param_hypelcnn_nonres.json", {"filter_count": 96}, 3, 10, 4, 5),
    ("dualcnn", "DUALCNNModel", "alg_param_dualcnn.json", {"filter_count": 32}, 5, 9, 4, 4),
    ("concnn", "CONCNNModel", "alg_param_concnn.json", {"filter_count": 8}, 5, 12, 6, 4),
]


class Optimizer:
    def __init__(self, kind, learning_rate, name, **hyper):
        self.kind, self.learning_rate, self.name, self.hyper = kind, learning_rate, name, hyper


def exponential_decay(learning_rate, global_step, decay_steps, decay_rate, staircase=False, name=None):
    """tf.compat.v1.train.exponential_decay: lr * rate ^ (step / decay_steps), the exponent floored when staircase."""
    def lr(s):
        p = global_step(s) / decay_steps
        return learning_rate * decay_rate ** (np.floor(p) if staircase else p)
    return TG.GlobalStep(lr)


TRAIN_OPS = []


def create_train_op(total_loss, optimizer, global_step=None, **kw):
    op = {"loss": total_loss, "optimizer": optimizer, "global_step_is_the_shared_one": isinstance(global_step, TG.GlobalStep),
          "kwargs": sorted(kw)}
    TRAIN_OPS.append(op)
    return op


_GS = [None]


def _setup(module):
    n = module.__name__
    if n == "tensorflow.compat.v1.train":
        def gs():
            if _GS[0] is None:
                _GS[0] = TG.GlobalStep()
            return _GS[0]
        module.get_or_create_global_step = gs
        module.exponential_decay = exponential_decay
        module.AdamOptimizer = lambda learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8, use_locking=False, name="Adam": \
            Optimizer("AdamOptimizer", learning_rate, name, beta1=beta1, beta2=beta2, epsilon=epsilon)
        module.MomentumOptimizer = lambda learning_rate, momentum, use_locking=False, name="Momentum", use_nesterov=False: \
            Optimizer("MomentumOptimizer", learning_rate, name, momentum=momentum, use_nesterov=use_nesterov)
    elif n == "tensorflow.compat.v1":
        module.name_scope = TG.name_scope
        module.GraphKeys = TG.GraphKeys
        module.get_collection = lambda *a, **k: []   # (BN update ops only order the moving-average updates behind the step)
    elif n == "tf_slim.learning":
        module.create_train_op = create_train_op


def run_case(model_name, cfg, over, patch, ch, classes, nb, seed):
    import importlib
    from oracle import models as OM
    ref_ops = importlib.import_module("common.common_nn_ops")
    model = getattr(importlib.import_module("nnmodel." + model_name), model_name)()
    alg = dict(json.load(open(os.path.join(CFG, cfg))), **over)
    rng = np.random.default_rng(seed)
    if model_name == "HYPELCNNModel":
        params = OM.hypelcnn_init_params(patch, ch, classes, alg, rng, np.float64)
    else:
        table = (OM.dualcnn_layer_table if model_name == "DUALCNNModel" else OM.concnn_layer_table)(patch, ch, classes, alg)
        params = OM.xavier_init_params(table, rng, np.float64)
    for k in params:
        if k.endswith(("beta", "biases")):
            params[k] = rng.standard_normal(params[k].shape) * 0.1
    x = rng.random((nb, patch, patch, ch))
    onehot = np.eye(classes)[rng.integers(0, classes, nb)]
    # dropout masks: recorded shapes first (shape-only pass of the model file), then seeded masks
    import make_reference_graphs as MG
    probe, _ = MG.record_classifier(model_name, alg, patch, ch, classes, True)
    masks, di = {}, 0
    for r in probe.records:
        if r["op"] == "dropout" and r["is_training"]:
            keep = r["keep_prob"]
            masks[f"dropout_{di}"] = (rng.random((nb,) + tuple(r["shape"])) < keep) / keep
            di += 1
    eng = S.OracleEngine(params=params, is_training=True, dropout_masks=masks)
    del TRAIN_OPS[:]
    _GS[0] = None
    with S.use_engine(eng):
        images, labels = eng.placeholder(x, "x"), eng.placeholder(onehot, "labels")
        template = lambda model_input_params, algorithm_params: model.create_tensor_graph(  # noqa: E731  (tf.make_template's role)
            model_input_params, classes, algorithm_params)
        y_conv, cross_entropy, learning_rate, train_step = ref_ops.optimize_nn(
            template, images, labels, "/cpu:0", "training", alg, model.get_loss_func)
    assert len(TRAIN_OPS) == 1 and train_step is TRAIN_OPS[0] and train_step["loss"] is cross_entropy
    opt = train_step["optimizer"]
    assert opt.learning_rate is learning_rate
    # gradients of the train op's loss through the reference's composition
    vars_ = {}
    seen, stack = set(), [cross_entropy.var]
    while stack:
        n = stack.pop()
        if id(n) in seen:
            continue
        seen.add(id(n))
        n.g = None
        if n.name and not n.parents:
            vars_.setdefault(n.name, []).append(n)
        stack.extend(n.parents)
    eng.O.backward(cross_entropy.var)
    arrays = {"x": x, "labels": onehot, "loss": np.asarray(float(cross_entropy.var.v)), "logits": y_conv.var.v}
    for k, v in params.items():
        arrays["param/" + k] = v
    for k, m in masks.items():
        arrays["mask/" + k] = m
    grads = {}
    for name, nodes in vars_.items():
        if name in params and not name.endswith(("moving_mean", "moving_variance")):
            g = sum(n.g for n in nodes if n.g is not None)
            grads[name] = np.asarray(g, np.float64)
            arrays["grad/" + name] = grads[name]
    case = {"model": model_name, "config": cfg, "alg": alg, "patch": patch, "channels": ch, "classes": classes, "batch": nb,
            "optimizer": {"class": opt.kind, "name": opt.name, "hyper": opt.hyper},
            "global_step_is_the_shared_one": train_step["global_step_is_the_shared_one"], "create_train_op_kwargs": train_step["kwargs"],
            "lr_steps": LR_STEPS, "lr": [float(learning_rate(s)) for s in LR_STEPS], "loss": float(cross_entropy.var.v),
            "trained": sorted(grads), "loss_ops": [r["op"] for r in eng.records[-8:]]}
    return case, arrays


def main():
    F._Finder.EXTRA_SETUP.append(_setup)
    S.install()
    import importlib
    for name in ("tensorflow.compat.v1.train", "tf_slim.learning"):
        m = importlib.import_module(name)
        parent, _, attr = name.rpartition(".")
        setattr(importlib.import_module(parent), attr, m)
    out, arrays = {}, {}
    for i, (name, model, cfg, over, patch, ch, classes, nb) in enumerate(CASES):
        case, arr = run_case(model, cfg, over, patch, ch, classes, nb, seed=300 + i)
        out[name] = case
        for k, v in arr.items():
            arrays[f"{name}/{k}"] = v
        print(f"{name}: {case['optimizer']}, loss {case['loss']:.6f}, {len(case['trained'])} trained variables, "
              f"lr {case['lr'][:4]} ..., loss ops {case['loss_ops']}")
    with open(os.path.join(HERE, "reference_optimize.json"), "w") as f:
        json.dump(out, f, sort_keys=True, indent=0, separators=(",", ":"))
    np.savez_compressed(os.path.join(HERE, "reference_optimize.npz"), **arrays)
    print("wrote reference_optimize.json / .npz")


if __name__ == "__main__":
    main()
