"""Recording `tensorflow_gan` + the `tf.*` calls of the reference's GAN wrappers (build container only).

TEST INFRASTRUCTURE.  `gan/wrappers/{gan_common,gan_wrapper,cycle_gan_wrapper,cut_wrapper,dcl_gan_wrapper,
dcl_cycle_gan_wrapper}.py` (1 700 lines: which tensor feeds which loss, with which weight, in which train op, over which
variable scope) import `tensorflow_gan`, which cannot be installed here.  This module extends the functional `tensorflow` /
`tf_slim` stand-in (`tf_standin.py`, `hypelcnn_amd/tf_facade.py`) with

  * TF1 variable scopes (`tf.compat.v1.variable_scope` incl. re-entering a captured scope, `reuse`, default-name
    uniquification that restarts when a scope is re-entered, `tf_slim.get_trainable_variables(scope)`, the
    regularisation-loss collection that `tf_slim` fills when a variable with a `weights_regularizer` is CREATED),
  * the few `tf.*` calls the wrappers make themselves (`matmul`, `transpose`, `eye`, `shape`, `nn.softmax_cross_entropy_
    with_logits`, `losses.compute_weighted_loss / absolute_difference / get_regularization_loss`, `cond`,
    `polynomial_decay`, the global step, `AdamOptimizer`, `tf_slim.learning.create_train_op`),
  * a RESTATEMENT of the tensorflow_gan functions they call -- `gan_model`, `cyclegan_model`, `gan_loss`, `cyclegan_loss`,
    `tuple_losses.*`, `args_to_gan_model`, `gan_train_ops`, `get_sequential_train_hooks`, `features.tensor_pool`,
    `namedtuples.*` -- after tensorflow_gan 2.1.0's published source.  tensorflow_gan's internals
    therefore stay restatements; what EXECUTES is the reference's own wrapper text on top of them.

Since round 6 that surface (the `tf.*` functions and the restated tensorflow_gan) is PRODUCT code -- `hypelcnn_amd/tfgan_facade.py`,
whose `GraphGanEngine` lets the reference's unchanged wrapper files build the product's train ops -- and this module keeps what
is test infrastructure: the recording float64 ENGINE behind the same surface.

Every tensor is evaluated in float64 with `oracle/ops.py` (a tape: gradients of every train op's loss w.r.t. its
`variables_to_train` come out of the same run) and carries a provenance label (`G[scope](x)`, `D[scope](...)`,
`pool(...)`) and, for scalars, its linear decomposition into primitive loss terms.  `make_reference_gan_wiring.py` turns a
run of `define_model -> define_loss -> define_train_ops -> get_train_hooks_fn()(train_ops)` into data.
"""
import math
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)
import tf_standin as S  # noqa: E402
from hypelcnn_amd import tf_facade as F  # noqa: E402
from hypelcnn_amd import tfgan_facade as TG  # noqa: E402  (the tf / tensorflow_gan surface itself is PRODUCT code since round 6)
from hypelcnn_amd.tfgan_facade import RunTrainOpsHook, Scope, Shape  # noqa: E402,F401


# ------------------------------------------------------------------------------------------------ engine
class WiringEngine(S.OracleEngine):
    """OracleEngine + variable scopes + collections + scalar arithmetic with provenance."""

    def __init__(self, rng=None):
        super().__init__(params=None, is_training=True, rng=rng or np.random.default_rng(0))
        self.scope_stack = []          # full names
        self.var_objs = {}             # full name -> ONE O.Var per variable (gradients of all applications accumulate)
        self.trainable = []            # full names, creation order
        self.reg_losses = []           # (variable full name, l2 scale): tf_slim adds regularizer(w) when w is created
        self.counts = {}               # (scope full name, base) -> next default-name index
        self._gstep = None
        self.pools = []                # labels of the tensors that went through tfgan.features.tensor_pool
        self.applications = []         # (network kind, variable scope, input label, output label)

    # -- tensors with arithmetic
    class T(S.OracleEngine.T):
        label = None     # provenance of a network-level tensor
        lin = None       # scalars: [(weight, term)] with term = (kind, operands..., params...)

        def get_shape(self):
            return Shape([F.Dim(None)] + [F.Dim(s) for s in self.var.v.shape[1:]]) if self.var.v.ndim else Shape([])

        @property
        def shape(self):
            return self.get_shape()

        def __add__(self, other):
            return self.eng.add_any(self, other)

        __radd__ = __add__

        def __sub__(self, other):
            return self.eng.sub_any(self, other)

        def __mul__(self, other):
            return self.eng.scale_any(self, other)

        __rmul__ = __mul__

        def __truediv__(self, other):
            return self.eng.scale_any(self, 1.0 / float(other), shown=("div", float(other)))

        def __neg__(self):
            return self.eng.scale_any(self, -1.0, shown=("neg",))

    def new(self, var, op, inputs=(), **attrs):
        t = WiringEngine.T(self, var, self.next_id)
        self.next_id += 1
        rec = {"op": op, "id": t.id, "inputs": [i.id for i in inputs], "shape": [int(s) for s in var.v.shape[1:]]}
        rec.update(attrs)
        self.records.append(rec)
        return t

    def placeholder(self, value, name):
        t = super().placeholder(value, name)
        t.label = name
        return t

    @staticmethod
    def label_of(t):
        return t.label if getattr(t, "label", None) else f"t{t.id}"

    # -- variable scopes
    def prefix(self):
        return self.scope_stack[-1] + "/" if self.scope_stack and self.scope_stack[-1] else ""

    def enter_scope(self, name_or_scope):
        full = name_or_scope.name if isinstance(name_or_scope, Scope) else self.prefix() + str(name_or_scope)
        self.scope_stack.append(full)
        # tf: entering a named scope restarts the default-name counters of everything below it
        for k in [k for k in self.counts if k[0] == full or k[0].startswith(full + "/")]:
            del self.counts[k]
        return Scope(full)

    def exit_scope(self):
        self.scope_stack.pop()

    def unique(self, base):
        key = (self.scope_stack[-1] if self.scope_stack else "", base)
        n = self.counts.get(key, 0)
        self.counts[key] = n + 1
        return base if n == 0 else f"{base}_{n}"

    def variable(self, name, shape, init):
        full = self.prefix() + name
        shape = tuple(int(s) for s in shape)
        if full not in self.var_objs:
            self.variables[full] = list(shape)
            if full not in self.params:
                # values that make every gradient non-trivial (the reference zero-initialises its generator): the
                # initialiser is recorded by the layer record, the VALUE is this stand-in's choice and travels in the fixture
                fan_in = max(1, int(np.prod(shape[:-1]))) if len(shape) > 1 else 1
                self.params[full] = (self.rng.standard_normal(shape) * (0.3 / math.sqrt(fan_in)) if len(shape) > 1
                                     else self.rng.standard_normal(shape) * 0.05)
            self.var_objs[full] = self.O.Var(np.asarray(self.params[full], np.float64), name=full)
            self.trainable.append(full)
        v = self.var_objs[full]
        assert v.v.shape == shape, (full, v.v.shape, shape)
        return v

    def _registered(self, t):
        """tf_slim: `weights_regularizer(weights)` joins REGULARIZATION_LOSSES when the variable is created."""
        rec = self.records[-1]
        name = self.prefix() + rec["scope"] + "/weights"
        if rec.get("regularizer") is not None and all(n != name for n, _ in self.reg_losses):
            self.reg_losses.append((name, float(rec["regularizer"])))
        rec["variable_scope"] = self.scope_stack[-1] if self.scope_stack else ""
        return t

    def fully_connected(self, *a, **k):
        return self._registered(super().fully_connected(*a, **k))

    def convolution1d(self, *a, **k):
        return self._registered(super().convolution1d(*a, **k))

    def conv2d(self, *a, **k):
        return self._registered(super().conv2d(*a, **k))

    # -- scalar / elementwise arithmetic with provenance
    def _lin(self, t):
        return t.lin

    def add_any(self, a, b):
        O = self.O
        if not isinstance(b, S.OracleEngine.T):
            if float(b) == 0.0:
                return a
            out = self.new(O.add(a.var, O.const(np.asarray(float(b)))), "add_const", [a], value=float(b))
            out.label = f"add({self.label_of(a)},{float(b)!r})"
            return out
        out = self.new(O.add(a.var, b.var), "add", [a, b])
        if a.lin is not None and b.lin is not None:
            out.lin = a.lin + b.lin
        out.label = f"add({self.label_of(a)},{self.label_of(b)})"
        return out

    def sub_any(self, a, b):
        O = self.O
        if not isinstance(b, S.OracleEngine.T):
            out = self.new(O.add(a.var, O.const(np.asarray(-float(b)))), "sub_const", [a], value=float(b))
            out.label = f"sub({self.label_of(a)},{float(b)!r})"
            return out
        out = self.new(O.sub(a.var, b.var), "sub", [a, b])
        if a.lin is not None and b.lin is not None:
            out.lin = a.lin + [(-w, t) for w, t in b.lin]
        out.label = f"sub({self.label_of(a)},{self.label_of(b)})"
        return out

    def scale_any(self, a, w, shown=None):
        if isinstance(w, S.OracleEngine.T):
            raise RuntimeError("tensor * tensor is not on the wrappers' path")
        w = float(w)
        out = self.new(self.O.scale(a.var, w), "scale", [a], weight=w)
        if a.lin is not None:
            out.lin = [(w * ww, t) for ww, t in a.lin]
        out.label = (f"{shown[0]}({self.label_of(a)}" + "".join(f",{v!r}" for v in shown[1:]) + ")") if shown else \
            f"scale({self.label_of(a)},{w!r})"
        return out

    def prim(self, var, kind, inputs, term):
        """A primitive scalar loss term: value `var`, linear decomposition [(1, term)]."""
        out = self.new(var, kind, inputs, term=list(term))
        out.lin = [(1.0, tuple(term))]
        out.label = f"{kind}(" + ",".join(str(x) for x in term[1:]) + ")"
        return out

    # -- the engine interface of hypelcnn_amd.tfgan_facade (the tf.* / tensorflow_gan calls of the wrappers land here)
    def trainable_variables(self, scope):
        return [n for n in self.trainable if not scope or n == scope or n.startswith(scope + "/")]

    def global_step(self):
        if self._gstep is None:
            self._gstep = TG.GlobalStep()
        return self._gstep

    def matmul_nt(self, a, b):
        out = self.new(self.O.matmul_nt_batched(a.var, b.var), "matmul_nt", [a, b])
        out.label = f"matmul_nt({self.label_of(a)},{self.label_of(b)})"
        return out

    def shape(self, t):
        return tuple(int(s) for s in t.var.v.shape)

    def flatten(self, x):
        out = self.reshape(x, (x.var.v.shape[0], -1), "flatten")
        out.label = self.label_of(x)
        return out

    def softmax_xent(self, lab, logits):
        assert lab.shape == logits.var.v.shape, (lab.shape, logits.var.v.shape)
        p = int(round(lab.shape[1] ** 0.5))
        is_eye = np.array_equal(lab.reshape(lab.shape[0], p, -1), TG.eye(p, batch_shape=[lab.shape[0]]))
        out = self.new(self.O.softmax_xent(logits.var, lab), "softmax_xent", [logits], labels="eye" if is_eye else "other")
        out.label = f"xent_eye({self.label_of(logits)})"
        return out

    def weighted_mean(self, losses, reduction):
        return self.prim(self.O.reduce_mean(losses.var), "mean", [losses], ("mean", self.label_of(losses), reduction))

    def abs_diff(self, labels, predictions):
        d = self.new(self.O.absolute(self.O.sub(predictions.var, labels.var)), "abs_diff", [labels, predictions])
        d.label = f"abs_diff({self.label_of(labels)},{self.label_of(predictions)})"
        return d

    def sqdiff_half(self, t, label):
        d = self.O.sub(t.var, self.O.const(np.asarray(float(label))))
        out = self.new(self.O.scale(self.O.square(d), 0.5), "sqdiff_half", [t], label=float(label))
        out.label = f"sqdiff_half({self.label_of(t)},{float(label)!r})"
        return out

    def regularization_loss(self, scope):
        total = None
        for vname, scale in self.reg_losses:
            if scope and not re.match(scope, vname):
                continue
            v = self.var_objs[vname]
            t = self.prim(self.O.scale(self.O.reduce_sum(self.O.square(v)), 0.5 * scale), "l2", [], ("l2", vname, scale))
            total = t if total is None else total + t
        return 0.0 if total is None else total

    def tensor_pool(self, values, pool_size, pooling_probability):
        out = []
        for t in values:
            p = self.new(self.O.Var(t.var.v.copy()), "tensor_pool", [t], pool_size=pool_size,
                         pooling_probability=pooling_probability)
            p.label = f"pool({self.label_of(t)})"
            self.pools.append(self.label_of(t))
            out.append(p)
        return tuple(out)




def eng():
    e = F.ENGINE[0]
    if not isinstance(e, WiringEngine):
        raise RuntimeError("the GAN wrappers run under a WiringEngine")
    return e


def install():
    """The facade's finder with the tf / tensorflow_gan surface of hypelcnn_amd.tfgan_facade on top + the reference on
    sys.path (build container only)."""
    TG.enable()
    S.install()
    TG.preload()


# ------------------------------------------------------------------------------------------------ labelled networks
def labelled(kind, fn):
    """Wrap a network function of the reference so that its output carries a provenance label and the application is
    recorded (kind G: generator, E when create_only_encoder; D: discriminator; F: feature discriminator)."""
    def wrapped(x, *a, **k):
        e = eng()
        out = fn(x, *a, **k)
        kk = kind
        if kind == "G" and (k.get("create_only_encoder") or (len(a) > 0 and a[0] is True)):
            kk = "E"
        scope = e.scope_stack[-1] if e.scope_stack else ""
        out.label = f"{kk}[{scope}]({e.label_of(x)})"
        e.applications.append((kk, scope, e.label_of(x), out.label))
        return out
    return wrapped
