"""`tensorflow_gan` facade: the reference's UNCHANGED GAN wrapper files build the product's train ops.

`gan/wrappers/{gan,cycle_gan,cut,dcl_gan,dcl_cycle_gan}_wrapper.py`, `gan/wrappers/gan_common.py` and `gan/wrapper_registry.py`
of the reference import `tensorflow`, `tf_slim` and `tensorflow_gan`.  On top of `tf_facade` (the import finder, the engine
switch) this module serves

  * the `tf.*` calls the wrappers make themselves -- `tf.compat.v1.variable_scope` (incl. re-entering a captured scope),
    `name_scope`, `matmul` / `transpose` / `eye` / `shape`, `nn.softmax_cross_entropy_with_logits`,
    `losses.compute_weighted_loss / absolute_difference / get_regularization_loss`, `cond`, `polynomial_decay`, the global
    step, `AdamOptimizer`, `tf_slim.get_trainable_variables`, `tf_slim.learning.create_train_op` --, each forwarding to the
    ENGINE in force, and
  * `tensorflow_gan`'s functions the wrappers call, RESTATED after tensorflow_gan 2.1.0's published source (SURVEY
    Appendix A.12): `gan_model`, `cyclegan_model`, `gan_loss`, `cyclegan_loss`, `tuple_losses.*`, `args_to_gan_model`,
    `gan_train_ops`, `get_sequential_train_hooks`, `features.tensor_pool`, `namedtuples.*`.

Two engines sit behind the surface: `GraphGanEngine` (here, product code) turns what the reference's wrapper text builds into
`hypelcnn_amd.gan.wrappers.gan_common.Phase` lists on a product Tower --

    from hypelcnn_amd import tfgan_facade
    wrapper = tfgan_facade.reference_wrapper("cut_x2y", "/path/to/hypelcnn", flags)   # the reference checkout, unchanged
    # ... use `wrapper` wherever hypelcnn_amd.gan.wrapper_registry.get_wrapper_dict(flags)["cut_x2y"] is used

-- and test infrastructure adds a recording float64 engine on the same surface (`tests/golden/tfgan_standin.py`), whose
output is the committed fixture the product's own wrappers are held to.  The network builders under the wrappers are the
product's fused ones (`hypelcnn_amd.gan.shadow_data_models`: the generator is ONE node); the reference's
`gan/shadow_data_models.py` runs under the recording engines only (tests/test_reference_wiring.py pins its layer tables).
"""
import collections
import inspect
import os
import re

import numpy as np

from . import tf_facade as F

AUTO_REUSE = "AUTO_REUSE"


def eng():
    e = F.ENGINE[0]
    if e is None or not hasattr(e, "enter_scope"):
        raise RuntimeError("the GAN wrappers run under a GAN engine (tfgan_facade.GraphGanEngine / the recording stand-in)")
    return e


# ------------------------------------------------------------------------------------------------ small value types
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


class AdamOptimizer:
    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8, use_locking=False, name="Adam"):
        self.learning_rate, self.beta1, self.beta2, self.epsilon = learning_rate, beta1, beta2, epsilon

    def lr_at(self, step):
        lr = self.learning_rate
        return float(lr(step) if callable(lr) else lr)


class SyncReplicasOptimizer:
    pass


class TrainOp:
    """tf_slim.learning.create_train_op: minimise total_loss over variables_to_train with `optimizer`."""

    def __init__(self, total_loss, optimizer, variables_to_train):
        self.loss, self.optimizer = total_loss, optimizer
        self.variables = [v.name for v in variables_to_train]


class RunTrainOpsHook:
    """tensorflow_gan.python.train.RunTrainOpsHook: before every session.run of the loop, run `train_ops` `train_steps`
    times -- all of them in ONE session.run (same weights for every op of the hook)."""

    def __init__(self, train_ops, train_steps):
        self.train_ops = list(train_ops) if isinstance(train_ops, (list, tuple)) else [train_ops]
        self.train_steps = train_steps


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


def transpose(a=None, perm=None, **kw):
    return Transposed(a, perm)


def matmul(a, b, **kw):
    """[N, P, E] x transpose([N, Q, E]) -> [N, P, Q] (cut_wrapper.py:361)."""
    if not isinstance(b, Transposed):
        raise RuntimeError("tf.matmul on this path multiplies by a transposed operand")
    return eng().matmul_nt(a, b.t)


def shape(t, **kw):
    return eng().shape(t)


def eye(num_rows, num_columns=None, batch_shape=None, **kw):
    m = np.eye(int(num_rows), int(num_columns if num_columns is not None else num_rows))
    for b in reversed(list(batch_shape or [])):
        m = np.broadcast_to(m, (int(b),) + m.shape).copy()
    return m


def layers_flatten(x):
    """tensorflow.python.layers.core.flatten."""
    if isinstance(x, np.ndarray):
        return x.reshape(x.shape[0], -1)
    return eng().flatten(x)


def softmax_cross_entropy_with_logits(labels=None, logits=None, **kw):
    return eng().softmax_xent(np.asarray(labels, np.float64), logits)


def compute_weighted_loss(losses, weights=1.0, scope=None, loss_collection=GraphKeys.LOSSES,
                          reduction=Reduction.SUM_BY_NONZERO_WEIGHTS):
    """weights = 1.0 throughout the path: SUM_BY_NONZERO_WEIGHTS and SUM_OVER_BATCH_SIZE both are the mean over all elements."""
    if weights != 1.0 or reduction not in (Reduction.SUM_BY_NONZERO_WEIGHTS, Reduction.SUM_OVER_BATCH_SIZE):
        raise RuntimeError(f"compute_weighted_loss(weights={weights}, reduction={reduction}) is not on the path")
    return eng().weighted_mean(losses, reduction)


def absolute_difference(labels, predictions, weights=1.0, scope=None, loss_collection=GraphKeys.LOSSES,
                        reduction=Reduction.SUM_BY_NONZERO_WEIGHTS):
    return compute_weighted_loss(eng().abs_diff(labels, predictions), weights, scope, loss_collection, reduction)


def get_regularization_loss(scope=None, name="total_regularization_loss"):
    return eng().regularization_loss(scope)


def get_collection(key, scope=None):
    return []


def get_or_create_global_step(*a, **k):
    return eng().global_step()


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


def create_train_op(total_loss, optimizer, global_step=None, update_ops=None, variables_to_train=None, check_numerics=True,
                    **kwargs):
    return TrainOp(total_loss, optimizer, variables_to_train)


def get_trainable_variables(scope=None, suffix=None):
    name = scope.name if isinstance(scope, Scope) else (scope or "")
    return [VarRef(n) for n in eng().trainable_variables(name)]


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
    return eng().tensor_pool(tuple(input_values), pool_size, pooling_probability)


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


# tensorflow_gan.python.losses.losses_impl, the four the path uses (+ the cycle-consistency loss)
def _wasserstein_generator_loss(discriminator_gen_outputs, weights=1.0, scope=None, loss_collection=GraphKeys.LOSSES,
                                reduction=Reduction.SUM_BY_NONZERO_WEIGHTS, add_summaries=False):
    return compute_weighted_loss(-discriminator_gen_outputs, weights, scope, loss_collection, reduction)


def _wasserstein_discriminator_loss(discriminator_real_outputs, discriminator_gen_outputs, real_weights=1.0,
                                    generated_weights=1.0, scope=None, loss_collection=GraphKeys.LOSSES,
                                    reduction=Reduction.SUM_BY_NONZERO_WEIGHTS, add_summaries=False):
    loss_on_generated = compute_weighted_loss(discriminator_gen_outputs, generated_weights, scope, None, reduction)
    loss_on_real = compute_weighted_loss(discriminator_real_outputs, real_weights, scope, None, reduction)
    return loss_on_generated - loss_on_real


def _least_squares_generator_loss(discriminator_gen_outputs, real_label=1, weights=1.0, scope=None,
                                  loss_collection=GraphKeys.LOSSES, reduction=Reduction.SUM_BY_NONZERO_WEIGHTS,
                                  add_summaries=False):
    return compute_weighted_loss(eng().sqdiff_half(discriminator_gen_outputs, real_label), weights, scope, loss_collection,
                                 reduction)


def _least_squares_discriminator_loss(discriminator_real_outputs, discriminator_gen_outputs, real_label=1, fake_label=0,
                                      real_weights=1.0, generated_weights=1.0, scope=None, loss_collection=GraphKeys.LOSSES,
                                      reduction=Reduction.SUM_BY_NONZERO_WEIGHTS, add_summaries=False):
    loss_on_real = compute_weighted_loss(eng().sqdiff_half(discriminator_real_outputs, real_label), real_weights, scope, None,
                                         reduction)
    loss_on_generated = compute_weighted_loss(eng().sqdiff_half(discriminator_gen_outputs, fake_label), generated_weights,
                                              scope, None, reduction)
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
        module.softmax_cross_entropy_with_logits = _xent_dispatch
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


def _xent_dispatch(labels=None, logits=None, **kw):
    """tf.nn.softmax_cross_entropy_with_logits: the classifiers' loss (tf_facade) or the patch-NCE of the GAN wrappers."""
    e = F.ENGINE[0]
    if hasattr(e, "enter_scope"):
        return softmax_cross_entropy_with_logits(labels=labels, logits=logits)
    return F.loss_softmax_xent(labels=labels, logits=logits)


PRELOAD = ("tensorflow.compat.v1.train", "tensorflow.compat.v1.summary", "tensorflow.python.layers", "tensorflow.python.layers.core",
           "tensorflow.python.training", "tensorflow.python.training.adam", "tensorflow.python.training.learning_rate_decay",
           "tensorflow.python.training.training_util", "tensorflow.python.training.session_run_hook",
           "tensorflow.python.summary", "tensorflow.python.summary.summary", "tf_slim.learning", "tensorflow_gan",
           "tensorflow_gan.features", "tensorflow_gan.python", "tensorflow_gan.python.namedtuples",
           "tensorflow_gan.python.losses", "tensorflow_gan.python.losses.tuple_losses", "tensorflow_gan.python.train")


def enable():
    """Add this module's surface to the stub modules `tf_facade`'s finder creates (idempotent)."""
    if _setup not in F._Finder.EXTRA_SETUP:
        F._Finder.EXTRA_SETUP.append(_setup)


def preload():
    """Inside an `installed(...)` context: import the stub submodules once so that attribute access on their parents works."""
    import importlib
    for name in PRELOAD:
        m = importlib.import_module(name)
        parent_name, _, attr = name.rpartition(".")
        if parent_name:
            setattr(importlib.import_module(parent_name), attr, m)


# ------------------------------------------------------------------------------------------------ the product's engine
class Loss:
    """A scalar the wrappers' loss code builds: a weighted sum of primitive terms (the product's LossTerm kinds) plus the
    regularisation losses of whole variable scopes (applied by the phase that trains the scope, plan_gan.py)."""

    def __init__(self, terms=(), reg_scopes=()):
        self.terms = list(terms)          # [(weight, (kind, a, b, target, tau))]  a / b: graph tensors
        self.reg_scopes = list(reg_scopes)

    def __add__(self, other):
        if isinstance(other, Loss):
            return Loss(self.terms + other.terms, self.reg_scopes + other.reg_scopes)
        if float(other) == 0.0:
            return self
        raise RuntimeError("adding a constant to a loss is not on the path")

    __radd__ = __add__

    def __sub__(self, other):
        if not isinstance(other, Loss) or other.reg_scopes:
            raise RuntimeError("loss - x: only a difference of primitive terms is on the path")
        return Loss(self.terms + [(-w, t) for w, t in other.terms], self.reg_scopes)

    def __mul__(self, w):
        if self.reg_scopes:
            raise RuntimeError("scaling a regularisation loss is not on the path")
        return Loss([(float(w) * ww, t) for ww, t in self.terms])

    __rmul__ = __mul__

    def __truediv__(self, w):
        return self * (1.0 / float(w))


class GraphGanEngine:
    """The `tf` / `tensorflow_gan` calls of the reference's wrappers, landing on `hypelcnn_amd.graph`."""

    def __init__(self, tower):
        from . import graph as G
        self.G, self.tower = G, tower
        self._saved = []
        self._gstep = None
        self.pools = []   # (placeholder name, source SymTensor) in creation order

    # -- tensors
    class T:
        """A product tensor ([N, B] SymTensor) with the little TensorFlow surface the wrappers touch."""

        def __init__(self, eng_, sym):
            self.eng, self.sym = eng_, sym

        @property
        def shape(self):
            return Shape([F.Dim(None), F.Dim(1), F.Dim(1), F.Dim(self.sym.c)])

        def get_shape(self):
            return self.shape

        def __neg__(self):
            return GraphGanEngine.Elem("neg", self)

    class Elem:
        """An elementwise expression that only ever reaches compute_weighted_loss."""

        def __init__(self, kind, a, b=None, value=None):
            self.kind, self.a, self.b, self.value = kind, a, b, value

        def __truediv__(self, c):  # matmul(...) / tau
            if self.kind != "logits":
                raise RuntimeError("only the patch-NCE logits are divided by a constant on this path")
            return GraphGanEngine.Elem("logits", self.a, self.b, (self.value or 1.0) * float(c))

    def wrap(self, sym):
        return GraphGanEngine.T(self, sym)

    # -- scopes (hypelcnn_amd.graph keeps the TF1 rules; a captured scope is re-entered by its full name)
    def enter_scope(self, name_or_scope):
        G = self.G
        full = name_or_scope.name if isinstance(name_or_scope, Scope) else \
            ((G._VSCOPE[-1] + "/" + str(name_or_scope)) if G._VSCOPE[-1] else str(name_or_scope))
        G._VSCOPE.append(full)
        for k in [k for k in G._DEFAULT_NAME_COUNTS if k[0] == full or k[0].startswith(full + "/")]:
            del G._DEFAULT_NAME_COUNTS[k]
        return Scope(full)

    def exit_scope(self):
        self.G._VSCOPE.pop()

    def trainable_variables(self, scope):
        return [v.name for v in self.tower.store.order if v.trainable and (not scope or v.name.startswith(scope + "/"))]

    def regularization_loss(self, scope):
        if any(v.l2_scale and (not scope or re.match(scope, v.name)) for v in self.tower.store.order):
            return Loss(reg_scopes=[scope])
        return 0.0

    def global_step(self):
        if self._gstep is None:
            self._gstep = GlobalStep()
        return self._gstep

    # -- the loss surface
    def matmul_nt(self, a, b):
        return GraphGanEngine.Elem("logits", a, b, 1.0)

    def shape(self, t):
        if isinstance(t, GraphGanEngine.Elem) and t.kind == "logits":
            parts = t.a.sym.parts
            return (-1, parts, parts)
        raise RuntimeError("tf.shape is only taken of the patch-NCE logits on this path")

    def flatten(self, x):
        return x

    def softmax_xent(self, labels, logits):
        if not (isinstance(logits, GraphGanEngine.Elem) and logits.kind == "logits"):
            raise RuntimeError("softmax cross entropy is the patch-NCE's on this path")
        p = logits.a.sym.parts
        lab = labels.reshape(labels.shape[0], -1)
        if not np.array_equal(lab, np.tile(np.eye(p).reshape(1, -1), (lab.shape[0], 1))):
            raise RuntimeError("patch-NCE labels other than the identity are not on the path")
        return GraphGanEngine.Elem("nce", logits.a, logits.b, logits.value)

    def sqdiff_half(self, t, label):
        return GraphGanEngine.Elem("sqdiff_half", t, None, float(label))

    def abs_diff(self, labels, predictions):
        return GraphGanEngine.Elem("abs_diff", labels, predictions)

    def weighted_mean(self, x, reduction):
        if isinstance(x, GraphGanEngine.T):
            return Loss([(1.0, ("mean", x.sym, None, 0.0, None))])
        if x.kind == "neg":
            return Loss([(-1.0, ("mean", x.a.sym, None, 0.0, None))])
        if x.kind == "sqdiff_half":
            return Loss([(0.5, ("mean_sq", x.a.sym, None, x.value, None))])
        if x.kind == "abs_diff":
            return Loss([(1.0, ("mean_abs", x.a.sym, x.b.sym, 0.0, None))])
        if x.kind == "nce":
            return Loss([(1.0, ("nce", x.a.sym, x.b.sym, 0.0, x.value))])
        raise RuntimeError(f"mean of {x.kind} is not on the path")

    def tensor_pool(self, values, pool_size, pooling_probability):
        inputs, generated = values
        scope = self.G.group_of(generated.sym.node.weights[0].name) if hasattr(generated.sym.node, "weights") else ""
        name = {"ModelX2Y": "pool_fake_y", "ModelY2X": "pool_fake_x"}.get(scope.split("/")[-2] if "/" in scope else "", "pool_fake")
        if name in self.tower.inputs:
            raise RuntimeError(f"a second tensor pool for {scope}")
        ph = self.tower.placeholder(name, None, generated.sym.c)
        self.pools.append((name, generated.sym))
        return inputs, self.wrap(ph)


class ReferenceGANWrapper:
    """The `Wrapper` contract (`gan/wrappers/wrapper.py`) served by the REFERENCE's wrapper class: `define_model`,
    `define_loss` and `define_train_ops` execute the reference's text through the facade onto the Tower of the input
    tensors; what comes back from `define_train_ops` is the product's `GANTrainOps` (ordered phases + LR schedules +
    `run_step`), built from the RunTrainOpsHooks the reference's `get_train_hooks_fn()` returns."""

    def __init__(self, gan_type, reference_root, flags):
        import importlib
        from .gan import shadow_data_models as nets
        enable()
        self._ctx = F.installed(reference_root)
        self.gan_type = gan_type
        self.backend = None
        self._eng = None
        with self._ctx:
            preload()
            registry = importlib.import_module("gan.wrapper_registry")
            if not os.path.abspath(getattr(registry, "__file__", "") or "").startswith(os.path.abspath(reference_root) + os.sep):
                raise RuntimeError(f"gan.wrapper_registry resolved to {getattr(registry, '__file__', None)}, not to the reference checkout")
            # the registry's text binds three names it imported from gan/shadow_data_models.py: give it the product's
            # fused network builders instead (same signatures), wrapped for the facade's tensors
            saved = (registry.shadowdata_generator_model, registry.shadowdata_discriminator_model,
                     registry.shadowdata_feature_discriminator_model)
            registry.shadowdata_generator_model = self._net(nets.shadowdata_generator_model)
            registry.shadowdata_discriminator_model = self._net(nets.shadowdata_discriminator_model, second_tensor=True)
            registry.shadowdata_feature_discriminator_model = self._net(nets.shadowdata_feature_discriminator_model)
            try:
                self.reference = registry.get_wrapper_dict(flags)[gan_type]
            finally:
                (registry.shadowdata_generator_model, registry.shadowdata_discriminator_model,
                 registry.shadowdata_feature_discriminator_model) = saved

    def _net(self, fn, second_tensor=False):
        def call(x, *a, **k):
            if second_tensor and a:
                a = (a[0].sym if hasattr(a[0], "sym") else a[0],) + tuple(a[1:])
            return self._eng.wrap(fn(x.sym, *a, **k))
        return call

    def define_model(self, images_x, images_y):
        self._eng = GraphGanEngine(images_x.tower)
        with self._ctx, F.use_engine(self._eng):
            return self.reference.define_model(self._eng.wrap(images_x), self._eng.wrap(images_y))

    def define_loss(self, model):
        with self._ctx, F.use_engine(self._eng):
            return self.reference.define_loss(model)

    def define_train_ops(self, model, loss, max_number_of_steps, **kwargs):
        from .gan.wrappers import gan_common as C
        from . import graph as G
        with self._ctx, F.use_engine(self._eng):
            ref_ops = self.reference.define_train_ops(model, loss, max_number_of_steps, **kwargs)
            hooks = self.reference.get_train_hooks_fn()(ref_ops)
        hooks = [h for h in hooks if isinstance(h, RunTrainOpsHook)]
        by_name = {v.name: v for v in self._eng.tower.store.order}
        kinds = {"Generator": "gen", "Discriminator": "dis", "FeatDiscriminator": "feat"}
        infos = []
        for h in hooks:
            if h.train_steps != 1:
                raise RuntimeError("train steps other than 1 per hook are not on the path")
            groups, terms, lr_key, opt = [], {}, None, None
            for op in h.train_ops:
                if (op.optimizer.beta1, op.optimizer.beta2, op.optimizer.epsilon) != (0.5, 0.999, 1e-8):
                    raise RuntimeError("the GAN train ops run Adam(beta1 = 0.5) (gan_common.py:264-265)")
                gs = []
                for n in op.variables:
                    g = G.group_of(n)
                    if g not in gs:
                        gs.append(g)
                trained = set(op.variables)
                # tfgan / cut_loss add the regularisation losses of the scope the op trains; the product applies exactly
                # those (plan_gan.py::_emit_regularisers) -- anything else cannot be expressed
                for sc in op.loss.reg_scopes:
                    regd = {v.name for v in by_name.values() if v.l2_scale and re.match(sc, v.name)}
                    if not regd <= trained:
                        raise RuntimeError(f"regularisation loss of {sc} in a train op that does not train it")
                want = {v.name for v in by_name.values() if v.l2_scale and v.name in trained}
                have = {v.name for sc in op.loss.reg_scopes for v in by_name.values() if v.l2_scale and re.match(sc, v.name)}
                if want != have:
                    raise RuntimeError(f"train op over {gs} leaves out the regularisation loss of {sorted(want - have)[:2]}")
                for w, t in op.loss.terms:
                    if w == 0.0:
                        continue  # (a term the reference multiplies by 0.0: same loss, same gradients without it)
                    key = (t[0], id(t[1]), None if t[2] is None else id(t[2]), t[3], t[4])
                    if key in terms and abs(terms[key][0] - w) > 1e-12:
                        raise RuntimeError("two train ops of one hook weigh the same term differently")
                    terms[key] = (w, t)
                groups += [g for g in gs if g not in groups]
                k = kinds[groups[-1].rsplit("/", 1)[-1]]
                if lr_key is not None and (k != lr_key or op.optimizer is not opt):
                    raise RuntimeError("one hook, two optimisers")
                lr_key, opt = k, op.optimizer
            infos.append((groups, terms, lr_key, opt))
        # phase names as the product's own wrappers give them: the kind, prefixed by the direction when every kind has
        # one hook per direction (DCL)
        per_direction = len(infos) > 3
        phases, lrs = [], {}
        for groups, terms, lr_key, opt in infos:
            name = lr_key
            if per_direction:
                name = ("x2y:" if "/ModelX2Y/" in groups[0] + "/" else "y2x:") + lr_key
            lts = []
            used = set()
            for w, (kind, a, b, target, tau) in terms.values():
                kw = {}
                if kind == "nce":
                    kw = dict(tau=tau, parts=a.parts, embed=a.c // a.parts)
                lts.append(G.LossTerm(kind, a, b, target=target, weight=w, **kw))
                used.update(id(x) for x in (a, b) if x is not None)
            pool = [(n, src) for n, src in self._eng.pools if self._reaches(lts, self._eng.tower.inputs[n])]
            phases.append(C.Phase(name, lts, groups, lr_key, pool or None))
            lrs.setdefault(lr_key, opt.lr_at)
        outs = [src for _, src in self._eng.pools] or [model.generated_data.sym] if hasattr(model, "generated_data") else \
            [src for _, src in self._eng.pools]
        gan_loss_ = C.GANLoss(phases, self._eng.tower, outs)
        return C.GANTrainOps(gan_loss_, lrs, C.GanContext(self._eng.tower, self.backend), use_pool=bool(self._eng.pools))

    @staticmethod
    def _reaches(terms, placeholder):
        """Does any term's operand depend on `placeholder`?"""
        from . import graph as G
        seen = set()

        def walk(t):
            if t is None or id(t) in seen:
                return False
            seen.add(id(t))
            if t is placeholder:
                return True
            if t.root is not None and walk(t.root):
                return True
            n = t.node
            if n is None:
                return False
            srcs = list(getattr(n, "sources", ())) + list(getattr(n, "srcs", ())) + ([n.src] if hasattr(n, "src") else [])
            return any(walk(s) for s in srcs)
        return any(walk(x) for t in terms for x in (t.a, t.b))

    def get_train_hooks_fn(self):
        return lambda train_ops: [train_ops.run_step]


def reference_wrapper(gan_type, reference_root, flags):
    return ReferenceGANWrapper(gan_type, reference_root, flags)
