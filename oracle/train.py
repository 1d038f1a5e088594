"""ORACLE (test infrastructure) -- optimisers and the classifier training step.

Follows common/common_nn_ops.py:208-240 (optimize_nn) and the TF1 optimiser update rules
(SURVEY Appendix A.10/A.11).  Parity unpinned at the TensorFlow boundary.
"""
import numpy as np

from . import models as M
from . import ops as O
from .host import exponential_decay_staircase


def trainable_names(params):
    return [k for k in params if not (k.endswith("moving_mean") or k.endswith("moving_variance"))]


def adam_tf1_step(p, g, m, v, lr, t, beta1=0.9, beta2=0.999, eps=1e-8):
    """tf.compat.v1.train.AdamOptimizer: epsilon OUTSIDE the bias-corrected sqrt
    (lr_t = lr*sqrt(1-b2^t)/(1-b1^t); p -= lr_t*m/(sqrt(v)+eps)).  t counts from 1."""
    m[...] = beta1 * m + (1 - beta1) * g
    v[...] = beta2 * v + (1 - beta2) * g * g
    lr_t = lr * np.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
    p[...] = p - lr_t * m / (np.sqrt(v) + eps)


def momentum_tf1_step(p, g, a, lr, mu):
    """tf.compat.v1.train.MomentumOptimizer: a <- mu*a + g; p <- p - lr*a."""
    a[...] = mu * a + g
    p[...] = p - lr * a


MODEL_FWD = {
    "HYPELCNNModel": (M.hypelcnn_forward, M.hypelcnn_loss),
    "DUALCNNModel": (M.dualcnn_forward, M.plain_xent_loss),
    "CONCNNModel": (M.concnn_forward, M.plain_xent_loss),
}


def forward_backward(model_name, params, x, labels_onehot, class_count, alg, is_training=True,
                     dropout_masks=None, kink_force=None):
    """One forward (+ backward when training) pass.  Returns dict(loss, per_sample, logits,
    grads{name}, new_moving{name}, trace{scope})."""
    fwd, loss_fn = MODEL_FWD[model_name]
    ctx = M.Ctx(params, is_training, dropout_masks)
    ctx.kink_force = kink_force or {}
    xin = O.Var(x)
    out = fwd(ctx, xin, class_count, alg)
    res = {"logits": out["y_conv"].v, "trace": ctx.trace, "outputs": out}
    if labels_onehot is None:
        return res
    loss, per_sample = loss_fn(out, labels_onehot)
    res["loss"] = float(loss.v)
    res["per_sample"] = per_sample.v
    if is_training:
        O.backward(loss)
        res["grads"] = {k: ctx.vars[k].g for k in trainable_names(params) if ctx.vars[k].g is not None}
        res["new_moving"] = ctx.new_moving
        res["dx"] = xin.g
    return res


class ClassifierTrainer:
    """optimize_nn + create_train_op (common/common_nn_ops.py:208-240): gradients of the mean
    loss only (regularisation losses never reach the optimiser, Appendix A.7), BN moving
    statistics updated every step, staircase exponential LR decay, global_step from 0."""

    def __init__(self, model_name, params, class_count, alg):
        self.model_name = model_name
        self.params = params
        self.class_count = class_count
        self.alg = alg
        self.step = 0
        names = trainable_names(params)
        self.slots = {k: (np.zeros_like(params[k]), np.zeros_like(params[k])) for k in names}
        opt = alg["optimizer"]
        self.momentum = opt[1] if isinstance(opt, (list, tuple)) else None

    def learning_rate(self):
        return exponential_decay_staircase(self.alg["learning_rate"], self.step,
                                           self.alg["learning_rate_decay_step"],
                                           self.alg["learning_rate_decay_factor"])

    def train_step(self, x, labels_onehot, dropout_masks=None):
        r = forward_backward(self.model_name, self.params, x, labels_onehot, self.class_count, self.alg,
                             True, dropout_masks)
        lr = self.learning_rate()
        for k, g in r["grads"].items():
            if self.momentum is None:
                m, v = self.slots[k]
                adam_tf1_step(self.params[k], g, m, v, lr, self.step + 1)
            else:
                momentum_tf1_step(self.params[k], g, self.slots[k][0], lr, self.momentum)
        for k, val in r["new_moving"].items():
            self.params[k] = val.astype(self.params[k].dtype)
        self.step += 1
        return r
