"""Shared GAN machinery (reference gan/wrappers/gan_common.py): scope names, the tfgan-style model records,
the LR schedule, the tensor pool, and the phase executor that replaces tfgan's RunTrainOpsHook sequence."""
import collections

import numpy
import torch

from hypelcnn_amd import graph as G

model_generator_name = "Generator"
model_base_name = "Model"
input_x_tensor_name = "x"
input_y_tensor_name = "y"


def adj_shadow_ratio(shadow_ratio, is_shadow):
    return 1. / shadow_ratio if is_shadow else shadow_ratio


class GANModel(collections.namedtuple("GANModel", (
        "tower", "generator_inputs", "generated_data", "generator_scope", "real_data", "discriminator_real_outputs",
        "discriminator_gen_outputs", "discriminator_pool_outputs", "pool_input", "discriminator_scope"))):
    """What tfgan.gan_model returns, plus the discriminator applied to the tensor-pool placeholder."""


CycleGANModel = collections.namedtuple("CycleGANModel", ("model_x2y", "model_y2x", "reconstructed_x",
                                                          "reconstructed_y", "identity_x", "identity_y"))

Phase = collections.namedtuple("Phase", ("name", "terms", "train_groups", "lr_key", "pool"))
GANLoss = collections.namedtuple("GANLoss", ("phases", "tower", "generate_outputs"))


def new_gan_tower(bands):
    """One recorded graph per GAN (TF builds one graph too); variables carry the full TF names (no template prefix)."""
    tower = G.Tower(G.VariableStore(prefix=""), True, name="gan")
    x = tower.placeholder(input_x_tensor_name, None, bands)
    y = tower.placeholder(input_y_tensor_name, None, bands)
    return tower, x, y


def build_gan_model(tower, generator_fn, discriminator_fn, generator_inputs, real_data, pool_name,
                    generator_scope="Generator", discriminator_scope="Discriminator"):
    """tfgan.gan_model: G(inputs), D(G(inputs)), D(real) with shared discriminator variables -- and a third
    application of D on the tensor-pool placeholder (tfgan.gan_loss(tensor_pool_fn=...) re-applies D on pooled data)."""
    with G.variable_scope(generator_scope) as gs:
        generated = generator_fn(generator_inputs)
    with G.variable_scope(discriminator_scope) as ds:
        d_gen = discriminator_fn(generated, generator_inputs)
    with G.variable_scope(discriminator_scope):
        d_real = discriminator_fn(real_data, generator_inputs)
    pool_in = d_pool = None
    if pool_name:
        pool_in = tower.placeholder(pool_name, None, generated.c)
        with G.variable_scope(discriminator_scope):
            d_pool = discriminator_fn(pool_in, generator_inputs)
    return GANModel(tower, generator_inputs, generated, gs, real_data, d_real, d_gen, d_pool, pool_in, ds)


def _get_lr(base_lr, max_number_of_steps):
    """reference :222-244 -- constant for the first half, then polynomial_decay(power=1) to 0."""
    half = max_number_of_steps // 2

    def lr(global_step):
        if global_step < half:
            return base_lr
        decay_steps = max_number_of_steps - half
        s = min(global_step - half, decay_steps)
        return base_lr * (1 - s / decay_steps)

    return lr


class TensorPool:
    """tfgan.features.tensor_pool(pool_size=50, pooling_probability=0.5): until the pool is full the input is
    stored and returned; afterwards with probability 0.5 the input is returned unchanged, otherwise a random pooled
    element is returned and replaced by the input.  One element = one whole batch tensor, as in tfgan."""

    def __init__(self, pool_size=50, pooling_probability=0.5, seed=1234):
        self.pool_size, self.prob = pool_size, pooling_probability
        self.items = []
        self.rng = numpy.random.default_rng(seed)

    def query(self, value):
        if self.pool_size <= 0:
            return value
        if len(self.items) < self.pool_size:
            self.items.append(value.clone())
            return value
        if self.rng.random() >= self.prob:
            return value
        i = int(self.rng.integers(0, len(self.items)))
        out = self.items[i]
        self.items[i] = value.clone()
        return out


class GANTrainOps:
    """What define_train_ops returns: the ordered phases (tfgan sequential hooks), LR schedules, and `run_step`,
    the counterpart of one `session.run(global_step_inc_op)` with its RunTrainOpsHook sequence
    (gan_train_for_shadow.py:141-142)."""

    def __init__(self, loss, lrs, ctx, use_pool=True):
        self.loss = loss
        self.lrs = lrs
        self.ctx = ctx
        if not ctx.group_affinity:  # the variable groups of one train op next to each other in the flat buffers
            ctx.group_affinity = [list(p.train_groups) for p in loss.phases]
        self.pools = {}
        self.use_pool = use_pool
        self.last_losses = {}
        self.capture_graphs = True
        self.pool_override = None  # tests: callable(name, fresh) -> tensor fed to the discriminator

    def _compiled(self, sess, phase, nb):
        ct = sess.compile_phase(self.loss.tower, nb, terms=phase.terms, train_groups=phase.train_groups,
                                key=phase.name)
        if self.capture_graphs and getattr(sess.backend, "name", "") == "hip" and ct._graph_all is None:
            ct.capture()
        return ct

    def _feed(self, ct, x, y, fed=None):
        """Copy the batch into the phase's input buffers.  The buffers are shared between the phases of a session
        (PhasePlan); `fed` (a set owned by one run_step call) makes only the first phase that reads an input pay for
        the copy."""
        b = ct.plan.buffers
        todo = []
        for name, t in (("x", x), ("y", y)):
            if "in:" + name in b:
                key = (name, b["in:" + name].data_ptr())
                if fed is None or key not in fed:
                    todo.append((name, t))
                    if fed is not None:
                        fed.add(key)
        ct.set_inputs(todo)  # (both batches in one launch)

    def run_step(self, x, y):
        sess = self.ctx.session()
        fed = set()
        nb = x.shape[0]
        step = sess.global_step
        for phase in self.loss.phases:
            ct = self._compiled(sess, phase, nb)
            self._feed(ct, x, y, fed)
            if phase.pool:
                gen = sess.compile_phase(self.loss.tower, nb, outputs=[t for _, t in phase.pool], key="generate")
                self._feed(gen, x, y, fed)
                gen.forward()
                pooled = []
                for name, t in phase.pool:
                    fresh = gen.value(t, copy=False)  # consumed (pool query / set_input copy) before the next forward
                    if self.pool_override is not None:
                        val = self.pool_override(name, fresh)
                    elif self.use_pool:
                        val = self.pools.setdefault(name, TensorPool(seed=sess.seed)).query(fresh)
                    else:
                        val = fresh
                    pooled.append((name, val))
                ct.set_inputs(pooled)
            ct.forward_backward()
            sess.allreduce_group_gradients(phase.train_groups)
            sess.adam_step_groups(phase.train_groups, self.lrs[phase.lr_key](step), step + 1, beta1=0.5)
            self.last_losses[phase.name] = ct
        sess.global_step += 1

    def losses(self):
        return {k: ct.loss_value() for k, ct in self.last_losses.items()}


class GanContext:
    """Holds the recorded tower's variable store and (lazily) the device session."""

    def __init__(self, tower, backend=None, seed=1234):
        self.tower = tower
        self.backend = backend
        self.seed = seed
        self.group_affinity = []
        self._session = None

    def session(self):
        if self._session is None:
            from hypelcnn_amd.runtime import Session
            if self.backend is None:
                from hypelcnn_amd.backend import HipBackend
                self.backend = HipBackend()
            self._session = Session(self.tower.store, self.backend, seed=self.seed)
            self._session.finalize_variables(group_affinity=self.group_affinity)
            self._session.init_data_parallel()
        return self._session


def define_standard_train_ops(gan_model, gan_loss, max_number_of_steps, generator_lr, discriminator_lr, backend=None):
    """reference :247-279: Adam(beta1=0.5) for generator and discriminator, sequential G-then-D phases."""
    lrs = {"gen": _get_lr(generator_lr, max_number_of_steps), "dis": _get_lr(discriminator_lr, max_number_of_steps)}
    return GANTrainOps(gan_loss, lrs, GanContext(gan_loss.tower, backend))


def ls_terms_generator(d_gen):
    return [G.LossTerm("mean_sq", d_gen, target=1.0, weight=0.5)]


def ls_terms_discriminator(d_real, d_gen):
    return [G.LossTerm("mean_sq", d_real, target=1.0, weight=0.5), G.LossTerm("mean_sq", d_gen, target=0.0, weight=0.5)]
