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

Every tensor is evaluated in float64 with `oracle/ops.py` (a tape: gradients of every train op's loss w.r.t. its
`variables_to_train` come out of the same run) and carries a provenance label (`G[scope](x)`, `D[scope](...)`,
`pool(...)`) and, for scalars, its linear decomposition into primitive loss terms.  `make_reference_gan_wiring.py` turns a
run of `define_model -> define_loss -> define_train_ops -> get_train_hooks_fn()(train_ops)` into data.
"""
import collections
import inspect
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

AUTO_REUSE = "AUTO_REUSE"


# ------------------------------------------------------------------------------------------------ engine
class Scope:
    """tf.compat.v1.VariableScope: what `with variable_scope(...) as s` yields; `s.name` is the full name."""

    def __init__(self, name):
        self.name = name

    def __repr__(self):
        return f"Scope({self.name!r})"


class Shape(list):
    """TensorShape as far as the wrappers use it."""

    def is_compatible_with(self, other):
        return len(self) == len(other) and all(a.value is None or b.value is None or a.value == b.value
                                               for a, b in zip(self, other))


class VarRef:
    """What tf_slim.get_trainable_variables returns: the variable's name (the engine holds the value)."""

    def __init__(self, name):
        self.name = name
        self.op = self

    def __repr__(self):
        return f"VarRef({self.name!r})"


class WiringEngine(S.OracleEngine):
    """OracleEngine + variable scopes + collections + scalar arithmetic with provenance."""

    def __init__(self, rng=None):
        super().__init__(params=None, is_training=True, rng=rng or np.random.default_rng(0))
        self.scope_stack = []          # full names
        self.var_objs = {}             # full name -> ONE O.Var per variable (gradients of all applications accumulate)
        self.trainable = []            # full names, creation order
        self.reg_losses = []           # (variable full name, l2 scale): tf_slim adds regularizer(w) when w is created
        self.counts = {}               # (scope full name, base) -> next default-name index
        self.global_step = None
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


def eng():
    e = F.ENGINE[0]
    if not isinstance(e, WiringEngine):
        raise RuntimeError("the GAN wrappers run under a WiringEngine")
    return e


# ------------------------------------------------------------------------------------------------ tf surface
class variable_scope:
    def __init__(self, name_or_scope, default_name=None, values=None, reuse=None, **kw):
        self.arg = name_or_scope

    def __enter__(self):
        return eng().enter_scope(self.arg)

    def __exit__(self, *exc):
        eng().exit_scope()
        return False


class name_scope:
    def __init__(self, name=None, default_name=None, values=None):
        self.name = name or default_name or ""

    def __enter__(self):
        return self.name

    def __exit__(self, *exc):
        return False


class Reduction:
    NONE, SUM, MEAN = "none", "weighted_sum", "weighted_mean"
    SUM_OVER_BATCH_SIZE, SUM_BY_NONZERO_WEIGHTS = "weighted_sum_over_batch_size", "weighted_sum_by_nonzero_weights"
    SUM_OVER_NONZERO_WEIGHTS = SUM_BY_NONZERO_WEIGHTS


class GraphKeys:
    LOSSES, UPDATE_OPS, GLOBAL_VARIABLES, REGULARIZATION_LOSSES = "losses", "update_ops", "variables", "regularization_losses"


class Transposed:
    def __init__(self, t, perm):
        assert list(perm) == [0, 2, 1], perm
        self.t = t


def transpose(a=None, perm=None, **kw):
    return Transposed(a, perm)


def matmul(a, b, **kw):
    """[N, P, E] x transpose([N, Q, E]) -> [N, P, Q] (cut_wrapper.py:361)."""
    if not isinstance(b, Transposed):
        raise RuntimeError("tf.matmul on this path multiplies by a transposed operand")
    e = eng()
    out = e.new(e.O.matmul_nt_batched(a.var, b.t.var), "matmul_nt", [a, b.t])
    out.label = f"matmul_nt({e.label_of(a)},{e.label_of(b.t)})"
    return out


def shape(t, **kw):
    return tuple(int(s) for s in t.var.v.shape)


def eye(num_rows, num_columns=None, batch_shape=None, **kw):
    m = np.eye(int(num_rows), int(num_columns if num_columns is not None else num_rows))
    for b in reversed(list(batch_shape or [])):
        m = np.broadcast_to(m, (int(b),) + m.shape).copy()
    return m


def layers_flatten(x):
    """tensorflow.python.layers.core.flatten."""
    if isinstance(x, np.ndarray):
        return x.reshape(x.shape[0], -1)
    e = eng()
    out = e.reshape(x, (x.var.v.shape[0], -1), "flatten")
    out.label = e.label_of(x)
    return out


def softmax_cross_entropy_with_logits(labels=None, logits=None, **kw):
    e = eng()
    lab = np.asarray(labels, np.float64)
    assert lab.shape == logits.var.v.shape, (lab.shape, logits.var.v.shape)
    out = e.new(e.O.softmax_xent(logits.var, lab), "softmax_xent", [logits],
                labels="eye" if np.array_equal(lab.reshape(lab.shape[0], int(round(lab.shape[1] ** 0.5)), -1),
                                               eye(int(round(lab.shape[1] ** 0.5)), batch_shape=[lab.shape[0]])) else "other")
    out.label = f"xent_eye({e.label_of(logits)})"
    return out


def compute_weighted_loss(losses, weights=1.0, scope=None, loss_collection=GraphKeys.LOSSES,
                          reduction=Reduction.SUM_BY_NONZERO_WEIGHTS):
    """weights = 1.0 throughout the path: SUM_BY_NONZERO_WEIGHTS and SUM_OVER_BATCH_SIZE both are the mean over all elements."""
    if weights != 1.0 or reduction not in (Reduction.SUM_BY_NONZERO_WEIGHTS, Reduction.SUM_OVER_BATCH_SIZE):
        raise RuntimeError(f"compute_weighted_loss(weights={weights}, reduction={reduction}) is not on the path")
    e = eng()
    return e.prim(e.O.reduce_mean(losses.var), "mean", [losses], ("mean", e.label_of(losses), reduction))


def absolute_difference(labels, predictions, weights=1.0, scope=None, loss_collection=GraphKeys.LOSSES,
                        reduction=Reduction.SUM_BY_NONZERO_WEIGHTS):
    e = eng()
    d = e.new(e.O.absolute(e.O.sub(predictions.var, labels.var)), "abs_diff", [labels, predictions])
    d.label = f"abs_diff({e.label_of(labels)},{e.label_of(predictions)})"
    return compute_weighted_loss(d, weights, scope, loss_collection, reduction)


def get_regularization_loss(scope=None, name="total_regularization_loss"):
    e = eng()
    total = None
    for vname, scale in e.reg_losses:
        if scope and not re.match(scope, vname):
            continue
        v = e.var_objs[vname]
        t = e.prim(e.O.scale(e.O.reduce_sum(e.O.square(v)), 0.5 * scale), "l2", [], ("l2", vname, scale))
        total = t if total is None else total + t
    return 0.0 if total is None else total


def get_collection(key, scope=None):
    return []


class GlobalStep:
    """The global step as a symbol: `-`, `<` give functions of the step (the LR schedule is sampled afterwards)."""

    def __init__(self, fn=lambda s: s):
        self.fn = fn
        self.dtype = type("dt", (), {"base_dtype": "int64"})

    def __sub__(self, c):
        return GlobalStep(lambda s, f=self.fn: f(s) - c)

    def __lt__(self, c):
        return GlobalStep(lambda s, f=self.fn: f(s) < c)

    def assign_add(self, k):
        return ("global_step_inc", k)

    def __call__(self, s):
        return self.fn(s)


def get_or_create_global_step(*a, **k):
    e = eng()
    if e.global_step is None:
        e.global_step = GlobalStep()
    return e.global_step


def polynomial_decay(learning_rate, global_step, decay_steps, end_learning_rate=0.0001, power=1.0, cycle=False, name=None):
    assert not cycle

    def lr(s):
        g = min(global_step(s), decay_steps)
        return (learning_rate - end_learning_rate) * (1.0 - g / decay_steps) ** power + end_learning_rate
    return GlobalStep(lr)


def cond(pred=None, true_fn=None, false_fn=None, **kw):
    def pick(s):
        v = true_fn() if pred(s) else false_fn()
        return v(s) if isinstance(v, GlobalStep) else v
    return GlobalStep(pick)


class AdamOptimizer:
    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8, use_locking=False, name="Adam"):
        self.learning_rate, self.beta1, self.beta2, self.epsilon = learning_rate, beta1, beta2, epsilon


class SyncReplicasOptimizer:
    pass


class TrainOp:
    def __init__(self, total_loss, optimizer, variables_to_train, name_scope_):
        self.loss, self.optimizer = total_loss, optimizer
        self.variables = [v.name for v in variables_to_train]
        self.name_scope = name_scope_


_NAME_SCOPES = []


def create_train_op(total_loss, optimizer, global_step=None, update_ops=None, variables_to_train=None, check_numerics=True,
                    **kwargs):
    """tf_slim.learning.create_train_op: minimise total_loss over variables_to_train with `optimizer`."""
    return TrainOp(total_loss, optimizer, variables_to_train, None)


def get_trainable_variables(scope=None, suffix=None):
    e = eng()
    name = scope.name if isinstance(scope, Scope) else (scope or "")
    return [VarRef(n) for n in e.trainable if not name or n == name or n.startswith(name + "/")]


# ------------------------------------------------------------------------------------------------ tensorflow_gan, restated
class GANModel(collections.namedtuple("GANModel", (
        "generator_inputs", "generated_data", "generator_variables", "generator_scope", "generator_fn", "real_data",
        "discriminator_real_outputs", "discriminator_gen_outputs", "discriminator_variables", "discriminator_scope",
        "discriminator_fn"))):
    """tensorflow_gan.python.namedtuples.GANModel."""


class CycleGANModel(collections.namedtuple("CycleGANModel", ("model_x2y", "model_y2x", "reconstructed_x", "reconstructed_y"))):
    """tensorflow_gan.python.namedtuples.CycleGANModel (a class WITHOUT __slots__, as in tensorflow_gan: the reference's
    subclass assigns identity_x / identity_y as instance attributes)."""


class GANLoss(collections.namedtuple("GANLoss", ("generator_loss", "discriminator_loss"))):
    pass


class CycleGANLoss(collections.namedtuple("CycleGANLoss", ("loss_x2y", "loss_y2x"))):
    pass


class GANTrainOps(collections.namedtuple("GANTrainOps", ("generator_train_op", "discriminator_train_op", "global_step_inc_op",
                                                         "train_hooks"))):
    def __new__(cls, generator_train_op, discriminator_train_op, global_step_inc_op, train_hooks=()):
        return super().__new__(cls, generator_train_op, discriminator_train_op, global_step_inc_op, train_hooks)


class GANTrainSteps(collections.namedtuple("GANTrainSteps", ("generator_train_steps", "discriminator_train_steps"))):
    pass


class RunTrainOpsHook:
    """tensorflow_gan.python.train.RunTrainOpsHook: before every session.run of the loop, run `train_ops` `train_steps`
    times -- all of them in ONE session.run (same weights for every op of the hook)."""

    def __init__(self, train_ops, train_steps):
        self.train_ops = list(train_ops) if isinstance(train_ops, (list, tuple)) else [train_ops]
        self.train_steps = train_steps


def _convert_tensor_or_l_or_d(t):
    return t


def _validate_aux_loss_weight(w, name="weight"):
    if w is not None and float(w) < 0:
        raise ValueError(f"`{name}` must be non-negative")
    return w


def gan_model(generator_fn, discriminator_fn, real_data, generator_inputs, generator_scope="Generator",
              discriminator_scope="Discriminator", check_shapes=True):
    with variable_scope(generator_scope, reuse=AUTO_REUSE) as gen_scope:
        generated_data = generator_fn(generator_inputs)
    with variable_scope(discriminator_scope, reuse=AUTO_REUSE) as dis_scope:
        discriminator_gen_outputs = discriminator_fn(generated_data, generator_inputs)
    with variable_scope(dis_scope, reuse=True):
        discriminator_real_outputs = discriminator_fn(real_data, generator_inputs)
    if check_shapes and not generated_data.shape.is_compatible_with(real_data.shape):
        raise ValueError("generator output shape must be the same shape as real data")
    return GANModel(generator_inputs, generated_data, get_trainable_variables(gen_scope), gen_scope, generator_fn, real_data,
                    discriminator_real_outputs, discriminator_gen_outputs, get_trainable_variables(dis_scope), dis_scope,
                    discriminator_fn)


def cyclegan_model(generator_fn, discriminator_fn, data_x, data_y, generator_scope="Generator",
                   discriminator_scope="Discriminator", model_x2y_scope="ModelX2Y", model_y2x_scope="ModelY2X",
                   check_shapes=True):
    def partial_model(input_data, output_data):
        return gan_model(generator_fn=generator_fn, discriminator_fn=discriminator_fn, real_data=output_data,
                         generator_inputs=input_data, generator_scope=generator_scope, discriminator_scope=discriminator_scope,
                         check_shapes=check_shapes)
    with variable_scope(model_x2y_scope):
        model_x2y = partial_model(data_x, data_y)
    with variable_scope(model_y2x_scope):
        model_y2x = partial_model(data_y, data_x)
    with variable_scope(model_y2x.generator_scope, reuse=True):
        reconstructed_x = model_y2x.generator_fn(model_x2y.generated_data)
    with variable_scope(model_x2y.generator_scope, reuse=True):
        reconstructed_y = model_x2y.generator_fn(model_y2x.generated_data)
    return CycleGANModel(model_x2y, model_y2x, reconstructed_x, reconstructed_y)


def tensor_pool(input_values, pool_size=50, pooling_probability=0.5, name="tensor_pool"):
    """tfgan.features.tensor_pool: the returned tensors come out of a queue -- no gradient flows through them.  While the
    pool fills (and with probability 1 - pooling_probability afterwards) the values are the inputs themselves."""
    e = eng()
    out = []
    for t in input_values:
        p = e.new(e.O.Var(t.var.v.copy()), "tensor_pool", [t], pool_size=pool_size, pooling_probability=pooling_probability)
        p.label = f"pool({e.label_of(t)})"
        e.pools.append(e.label_of(t))
        out.append(p)
    return tuple(out)


def _tensor_pool_adjusted_model(model, tensor_pool_fn):
    if tensor_pool_fn is None:
        return model
    pooled_generator_inputs, pooled_generated_data = tensor_pool_fn((model.generator_inputs, model.generated_data))
    with variable_scope(model.discriminator_scope, reuse=True):
        dis_gen_outputs = model.discriminator_fn(pooled_generated_data, pooled_generator_inputs)
    return model._replace(generator_inputs=pooled_generator_inputs, generated_data=pooled_generated_data,
                          discriminator_gen_outputs=dis_gen_outputs)


def args_to_gan_model(loss_fn):
    """tensorflow_gan.python.losses.tuple_losses.args_to_gan_model: a loss function of named tensors becomes one of a model
    tuple -- required arguments are taken from the tuple's fields of the same name, the others from kwargs / defaults."""
    argspec = inspect.getfullargspec(loss_fn)
    defaults = argspec.defaults or []
    required_args = set(argspec.args[:-len(defaults)] if defaults else argspec.args)
    args_with_defaults = argspec.args[-len(defaults):] if defaults else []
    default_args_dict = dict(zip(args_with_defaults, defaults))

    def new_loss_fn(gan_model, **kwargs):  # pylint:disable=missing-docstring
        gan_model_dict = gan_model._asdict()
        gan_model_dict.update(getattr(gan_model, "__dict__", {}))
        args_from_tuple = set(argspec.args).intersection(set(gan_model_dict))
        required_args_not_from_tuple = required_args - args_from_tuple
        for arg in required_args_not_from_tuple:
            if arg not in kwargs:
                raise ValueError(f"`{arg}` must be supplied to {loss_fn.__name__} loss function.")
        ambiguous_args = set(gan_model_dict).intersection(set(kwargs.keys()))
        if ambiguous_args:
            raise ValueError(f"The following args are present in both the tuple and keyword args for {loss_fn.__name__}: "
                             f"{ambiguous_args}")
        for arg in required_args.intersection(args_from_tuple):
            assert arg not in kwargs
            kwargs[arg] = gan_model_dict[arg]
        for arg in default_args_dict:
            val_from_tuple = gan_model_dict[arg] if arg in gan_model_dict else None
            val_from_kwargs = kwargs[arg] if arg in kwargs else None
            assert not (val_from_tuple is not None and val_from_kwargs is not None)
            kwargs[arg] = (val_from_tuple if val_from_tuple is not None else
                           val_from_kwargs if val_from_kwargs is not None else default_args_dict[arg])
        return loss_fn(**kwargs)
    new_loss_fn.__name__ = loss_fn.__name__
    return new_loss_fn


# tensorflow_gan.python.losses.losses_impl, the four the path uses
def _wasserstein_generator_loss(discriminator_gen_outputs, weights=1.0, scope=None, loss_collection=GraphKeys.LOSSES,
                                reduction=Reduction.SUM_BY_NONZERO_WEIGHTS, add_summaries=False):
    return compute_weighted_loss(-discriminator_gen_outputs, weights, scope, loss_collection, reduction)


def _wasserstein_discriminator_loss(discriminator_real_outputs, discriminator_gen_outputs, real_weights=1.0,
                                    generated_weights=1.0, scope=None, loss_collection=GraphKeys.LOSSES,
                                    reduction=Reduction.SUM_BY_NONZERO_WEIGHTS, add_summaries=False):
    loss_on_generated = compute_weighted_loss(discriminator_gen_outputs, generated_weights, scope, None, reduction)
    loss_on_real = compute_weighted_loss(discriminator_real_outputs, real_weights, scope, None, reduction)
    return loss_on_generated - loss_on_real


def _squared_difference_half(t, label):
    e = eng()
    d = e.O.sub(t.var, e.O.const(np.asarray(float(label))))
    out = e.new(e.O.scale(e.O.square(d), 0.5), "sqdiff_half", [t], label=float(label))
    out.label = f"sqdiff_half({e.label_of(t)},{float(label)!r})"
    return out


def _least_squares_generator_loss(discriminator_gen_outputs, real_label=1, weights=1.0, scope=None,
                                  loss_collection=GraphKeys.LOSSES, reduction=Reduction.SUM_BY_NONZERO_WEIGHTS,
                                  add_summaries=False):
    return compute_weighted_loss(_squared_difference_half(discriminator_gen_outputs, real_label), weights, scope,
                                 loss_collection, reduction)


def _least_squares_discriminator_loss(discriminator_real_outputs, discriminator_gen_outputs, real_label=1, fake_label=0,
                                      real_weights=1.0, generated_weights=1.0, scope=None, loss_collection=GraphKeys.LOSSES,
                                      reduction=Reduction.SUM_BY_NONZERO_WEIGHTS, add_summaries=False):
    loss_on_real = compute_weighted_loss(_squared_difference_half(discriminator_real_outputs, real_label), real_weights, scope,
                                         None, reduction)
    loss_on_generated = compute_weighted_loss(_squared_difference_half(discriminator_gen_outputs, fake_label),
                                              generated_weights, scope, None, reduction)
    return loss_on_real + loss_on_generated


def _cycle_consistency_loss_impl(data_x, reconstructed_data_x, data_y, reconstructed_data_y, scope=None, add_summaries=False):
    loss_x = absolute_difference(data_x, reconstructed_data_x)
    loss_y = absolute_difference(data_y, reconstructed_data_y)
    return (loss_x + loss_y) / 2.0


wasserstein_generator_loss = args_to_gan_model(_wasserstein_generator_loss)
wasserstein_discriminator_loss = args_to_gan_model(_wasserstein_discriminator_loss)
least_squares_generator_loss = args_to_gan_model(_least_squares_generator_loss)
least_squares_discriminator_loss = args_to_gan_model(_least_squares_discriminator_loss)


def cycle_consistency_loss(cyclegan_model, scope=None, add_summaries=False):
    return _cycle_consistency_loss_impl(cyclegan_model.model_x2y.generator_inputs, cyclegan_model.reconstructed_x,
                                        cyclegan_model.model_y2x.generator_inputs, cyclegan_model.reconstructed_y, scope,
                                        add_summaries)


def _optional_kwargs(fn, possible_kwargs):
    spec = inspect.getfullargspec(fn)
    if spec.varkw is not None:
        return possible_kwargs
    return {k: v for k, v in possible_kwargs.items() if k in spec.args}


def gan_loss(model, generator_loss_fn=wasserstein_generator_loss, discriminator_loss_fn=wasserstein_discriminator_loss,
             gradient_penalty_weight=None, gradient_penalty_epsilon=1e-10, gradient_penalty_target=1.0,
             gradient_penalty_one_sided=False, mutual_information_penalty_weight=None, aux_cond_generator_weight=None,
             aux_cond_discriminator_weight=None, tensor_pool_fn=None, reduction=Reduction.SUM_BY_NONZERO_WEIGHTS,
             add_summaries=True):
    """tensorflow_gan.python.train.gan_loss without the auxiliary penalties the path never asks for."""
    for w in (gradient_penalty_weight, mutual_information_penalty_weight, aux_cond_generator_weight, aux_cond_discriminator_weight):
        if w is not None:
            raise RuntimeError("auxiliary GAN penalties are not on the path")
    kw = {"reduction": reduction, "add_summaries": add_summaries}
    gen_loss = generator_loss_fn(model, **_optional_kwargs(generator_loss_fn, kw))
    dis_loss = discriminator_loss_fn(_tensor_pool_adjusted_model(model, tensor_pool_fn),
                                     **_optional_kwargs(discriminator_loss_fn, kw))
    gen_reg_loss = get_regularization_loss(model.generator_scope.name) if model.generator_scope else 0
    dis_reg_loss = get_regularization_loss(model.discriminator_scope.name) if model.discriminator_scope else 0
    return GANLoss(gen_loss + gen_reg_loss, dis_loss + dis_reg_loss)


def cyclegan_loss(model, generator_loss_fn=least_squares_generator_loss, discriminator_loss_fn=least_squares_discriminator_loss,
                  cycle_consistency_loss_fn=cycle_consistency_loss, cycle_consistency_loss_weight=10.0, **kwargs):
    cycle_loss = cycle_consistency_loss_fn(model, add_summaries=kwargs.get("add_summaries", True))
    cycle_consistency_loss_weight = _validate_aux_loss_weight(cycle_consistency_loss_weight, "cycle_consistency_loss_weight")
    aux_loss = cycle_consistency_loss_weight * cycle_loss

    def _partial_loss(partial_model):
        partial_loss = gan_loss(partial_model, generator_loss_fn=generator_loss_fn, discriminator_loss_fn=discriminator_loss_fn,
                                **kwargs)
        return partial_loss._replace(generator_loss=partial_loss.generator_loss + aux_loss)
    with name_scope("cyclegan_loss_x2y"):
        loss_x2y = _partial_loss(model.model_x2y)
    with name_scope("cyclegan_loss_y2x"):
        loss_y2x = _partial_loss(model.model_y2x)
    return CycleGANLoss(loss_x2y, loss_y2x)


def gan_train_ops(model, loss, generator_optimizer, discriminator_optimizer, check_for_unused_update_ops=True, is_chief=True,
                  **kwargs):
    """tensorflow_gan.python.train.gan_train_ops: a CycleGAN gets the train ops of its two partial models, the generator
    (discriminator) pair run together."""
    if isinstance(model, CycleGANModel):
        x2y = gan_train_ops(model.model_x2y, loss.loss_x2y, generator_optimizer, discriminator_optimizer,
                            check_for_unused_update_ops, is_chief, **kwargs)
        y2x = gan_train_ops(model.model_y2x, loss.loss_y2x, generator_optimizer, discriminator_optimizer,
                            check_for_unused_update_ops, is_chief, **kwargs)
        return GANTrainOps((x2y.generator_train_op, y2x.generator_train_op),
                           (x2y.discriminator_train_op, y2x.discriminator_train_op),
                           get_or_create_global_step().assign_add(1), tuple(x2y.train_hooks) + tuple(y2x.train_hooks))
    global_step = get_or_create_global_step()
    gen_op = create_train_op(total_loss=loss.generator_loss, optimizer=generator_optimizer,
                             variables_to_train=model.generator_variables, global_step=None, check_numerics=False)
    dis_op = create_train_op(total_loss=loss.discriminator_loss, optimizer=discriminator_optimizer,
                             variables_to_train=model.discriminator_variables, global_step=None, check_numerics=False)
    return GANTrainOps(gen_op, dis_op, global_step.assign_add(1), ())


def get_sequential_train_hooks(train_steps=GANTrainSteps(1, 1)):
    def get_hooks(train_ops):
        return [RunTrainOpsHook(train_ops.generator_train_op, train_steps.generator_train_steps),
                RunTrainOpsHook(train_ops.discriminator_train_op, train_steps.discriminator_train_steps)] + \
            list(train_ops.train_hooks)
    return get_hooks


# ------------------------------------------------------------------------------------------------ module surface
def _setup(module):
    n = module.__name__
    if n == "tensorflow":
        module.matmul, module.transpose, module.shape, module.eye, module.cond = matmul, transpose, shape, eye, cond
        module.executing_eagerly = lambda: False
        module.float32 = "float32"
    elif n == "tensorflow.nn":
        module.softmax_cross_entropy_with_logits = softmax_cross_entropy_with_logits
    elif n == "tensorflow.compat.v1":
        module.variable_scope, module.name_scope, module.AUTO_REUSE = variable_scope, name_scope, AUTO_REUSE
        module.GraphKeys, module.get_collection = GraphKeys, get_collection
    elif n == "tensorflow.compat.v1.losses":
        module.Reduction, module.compute_weighted_loss = Reduction, compute_weighted_loss
        module.absolute_difference, module.get_regularization_loss = absolute_difference, get_regularization_loss
    elif n == "tensorflow.compat.v1.train":
        module.get_or_create_global_step = module.get_global_step = get_or_create_global_step
        module.SyncReplicasOptimizer = SyncReplicasOptimizer
    elif n == "tensorflow.compat.v1.summary":
        module.scalar = lambda *a, **k: None
    elif n == "tensorflow.python.layers.core":
        module.flatten = layers_flatten
    elif n == "tensorflow.python.training.adam":
        module.AdamOptimizer = AdamOptimizer
    elif n == "tensorflow.python.training.learning_rate_decay":
        module.polynomial_decay = polynomial_decay
    elif n == "tensorflow.python.training.training_util":
        module.get_or_create_global_step = module.get_global_step = get_or_create_global_step
    elif n == "tensorflow.python.training.session_run_hook":
        module.SessionRunHook = type("SessionRunHook", (), {})
    elif n == "tensorflow.python.summary.summary":
        module.scalar = lambda *a, **k: None
    elif n == "tf_slim":
        module.get_trainable_variables = get_trainable_variables
    elif n == "tf_slim.learning":
        module.create_train_op = create_train_op
    elif n == "tensorflow_gan":
        module.gan_model, module.cyclegan_model, module.gan_loss, module.cyclegan_loss = gan_model, cyclegan_model, gan_loss, cyclegan_loss
        module.gan_train_ops, module.get_sequential_train_hooks = gan_train_ops, get_sequential_train_hooks
        module.GANTrainSteps, module.CycleGANModel, module.GANModel = GANTrainSteps, CycleGANModel, GANModel
    elif n == "tensorflow_gan.features":
        module.tensor_pool = tensor_pool
    elif n == "tensorflow_gan.python.namedtuples":
        for c in (GANModel, CycleGANModel, GANLoss, CycleGANLoss, GANTrainOps, GANTrainSteps):
            setattr(module, c.__name__, c)
    elif n == "tensorflow_gan.python.losses.tuple_losses":
        module.args_to_gan_model = args_to_gan_model
        module.wasserstein_generator_loss, module.wasserstein_discriminator_loss = wasserstein_generator_loss, wasserstein_discriminator_loss
        module.least_squares_generator_loss = least_squares_generator_loss
        module.least_squares_discriminator_loss = least_squares_discriminator_loss
        module.cycle_consistency_loss = cycle_consistency_loss
    elif n == "tensorflow_gan.python.train":
        module._validate_aux_loss_weight, module._convert_tensor_or_l_or_d = _validate_aux_loss_weight, _convert_tensor_or_l_or_d
        module.RunTrainOpsHook, module.gan_loss = RunTrainOpsHook, gan_loss
        module.get_sequential_train_hooks = get_sequential_train_hooks


_PRELOAD = ("tensorflow.compat.v1.train", "tensorflow.compat.v1.summary", "tensorflow.python.layers", "tensorflow.python.layers.core",
            "tensorflow.python.training", "tensorflow.python.training.adam", "tensorflow.python.training.learning_rate_decay",
            "tensorflow.python.training.training_util", "tensorflow.python.training.session_run_hook",
            "tensorflow.python.summary", "tensorflow.python.summary.summary", "tf_slim.learning", "tensorflow_gan",
            "tensorflow_gan.features", "tensorflow_gan.python", "tensorflow_gan.python.namedtuples",
            "tensorflow_gan.python.losses", "tensorflow_gan.python.losses.tuple_losses", "tensorflow_gan.python.train")


def install():
    """The facade's finder with this module's surface on top + the reference on sys.path (build container only)."""
    import importlib
    if _setup not in F._Finder.EXTRA_SETUP:
        F._Finder.EXTRA_SETUP.append(_setup)
    S.install()
    for name in _PRELOAD:
        m = importlib.import_module(name)
        parent_name, _, attr = name.rpartition(".")
        if parent_name:
            setattr(importlib.import_module(parent_name), attr, m)


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
