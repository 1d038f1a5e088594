"""ORACLE (test infrastructure) -- the shadow-GAN stacks restated over oracle/ops.py.

Parity unpinned at the TensorFlow / tensorflow_gan boundary (neither is installable here); semantics follow
tensorflow_gan's published losses and the reference's wrappers:

  gan/wrappers/gan_wrapper.py:14-66          vanilla GAN (tfgan.gan_loss defaults => Wasserstein losses)
  gan/wrappers/cycle_gan_wrapper.py:48-333   CycleGAN (+ "identity" loss |x - G(x)|, :303-333)
  gan/wrappers/cut_wrapper.py:90-665         CUT: LS-GAN + patch-NCE on feature-discriminator embeddings
  gan/wrappers/dcl_gan_wrapper.py, dcl_cycle_gan_wrapper.py   two independent CUT models (the coupling
                                             `_replace(...)` results are discarded: :189-190 / :149-150)
  gan/wrappers/gan_common.py:222-279         LR schedule, Adam(beta1=0.5), sequential G-then-D steps

Variable names are the TF ones: Model[/ModelX2Y|/ModelY2X]/{Generator,Discriminator,FeatDiscriminator}/...
A training step is a sequence of PHASES; each phase recomputes the forward pass with the current weights,
differentiates its loss w.r.t. its own variable scope only, and applies one TF1-Adam update (tfgan
RunTrainOpsHook semantics: one session.run per phase).
"""
import numpy as np

from . import models as M
from . import ops as O
from .host import gan_lr
from .train import adam_tf1_step


# ----------------------------------------------------------------------------- tfgan losses (Appendix A.12)
def mean(v):
    return O.reduce_mean(v)


def ls_generator_loss(d_gen):
    """least_squares_generator_loss: mean((D(G) - 1)^2) / 2."""
    return O.scale(mean(O.square(O.sub(d_gen, O.const(np.asarray(1.0, d_gen.v.dtype))))), 0.5)


def ls_discriminator_loss(d_real, d_gen):
    """least_squares_discriminator_loss: mean((D(x)-1)^2)/2 + mean(D(G)^2)/2."""
    a = O.scale(mean(O.square(O.sub(d_real, O.const(np.asarray(1.0, d_real.v.dtype))))), 0.5)
    b = O.scale(mean(O.square(d_gen)), 0.5)
    return O.add(a, b)


def wasserstein_generator_loss(d_gen):
    return O.scale(mean(d_gen), -1.0)


def wasserstein_discriminator_loss(d_real, d_gen):
    return O.sub(mean(d_gen), mean(d_real))


def abs_diff(a, b):
    """tf.compat.v1.losses.absolute_difference, SUM_BY_NONZERO_WEIGHTS = mean |a - b|."""
    return mean(O.absolute(O.sub(a, b)))


def cycle_consistency(x, rec_x, y, rec_y):
    return O.scale(O.add(abs_diff(x, rec_x), abs_diff(y, rec_y)), 0.5)


def nce_loss(feat_gen, feat_real, tau):
    """cut_wrapper.py:360-420: logits = gen @ real^T / tau per sample ([P,P]), labels = eye(P); both flattened to
    [N, P*P]; softmax-CE per sample (labels sum to P); SUM_OVER_BATCH_SIZE => mean over the batch."""
    logits = O.scale(O.matmul_nt_batched(feat_gen, feat_real), 1.0 / tau)
    n, p, _ = logits.v.shape
    labels = np.tile(np.eye(p, dtype=logits.v.dtype).reshape(1, -1), (n, 1))
    ce = O.softmax_xent(O.reshape(logits, (n, p * p)), labels)
    return mean(ce)


def l2_reg(ctx, names, scale):
    """tf_slim.l2_regularizer(scale)(w) = scale * sum(w^2) / 2 summed over the scope's regularised weights."""
    total = None
    for nme in names:
        t = O.scale(O.reduce_sum(O.square(ctx.p(nme))), 0.5 * scale)
        total = t if total is None else O.add(total, t)
    return total


# ----------------------------------------------------------------------------- parameter sets
def gan_param_names(kind, bands, patches=6, embed=2):
    """{scope prefix: {"gen": [...], "dis": [...], "feat": [...]}} for a wrapper kind."""
    def one(prefix, with_feat):
        gen = [f"{prefix}Generator/net{i}/{s}" for i in range(1, 8) for s in ("weights", "biases")]
        dis = [f"{prefix}Discriminator/{sc}/{s}" for sc, _, _ in M.discriminator_layer_table(bands)
               for s in ("weights", "biases")]
        out = {"gen": gen, "dis": dis}
        if with_feat:
            out["feat"] = [f"{prefix}FeatDiscriminator/{sc}/{s}"
                           for sc, _, _ in M.feature_discriminator_layer_table(bands, patches, embed)
                           for s in ("weights", "biases")]
        return out

    if kind in ("gan_x2y", "gan_y2x"):
        return {"Model/": one("Model/", False)}
    if kind in ("cut_x2y", "cut_y2x"):
        return {"Model/": one("Model/", True)}
    feat = kind in ("dcl_gan", "dcl_cycle_gan")
    return {"Model/ModelX2Y/": one("Model/ModelX2Y/", feat), "Model/ModelY2X/": one("Model/ModelY2X/", feat)}


def init_gan_params(kind, bands, rng, patches=6, embed=2, dtype=np.float32, zero_generator=True):
    """Generators zero-initialised (shadow_data_models.py:47) unless asked otherwise (tests want non-trivial
    gradients); discriminators He-truncated-normal, zero biases (:95,128)."""
    params = {}
    for prefix, groups in gan_param_names(kind, bands, patches, embed).items():
        g = M.generator_init_params(bands, prefix + "Generator/", dtype)
        if not zero_generator:
            for k in g:
                g[k] = (rng.standard_normal(g[k].shape) * (0.3 / max(g[k].shape[0], 1) ** 0.5)).astype(dtype)
        params.update(g)
        params.update(M.he_fc_init(M.discriminator_layer_table(bands), rng, prefix + "Discriminator/", dtype))
        if "feat" in groups:
            params.update(M.he_fc_init(M.feature_discriminator_layer_table(bands, patches, embed), rng,
                                       prefix + "FeatDiscriminator/", dtype))
    return params


def regularised_dis_weights(prefix):
    """fully_connected and fully_connected_1 carry l2_regularizer(scale); the last layer passes
    weights_regularizer=None (shadow_data_models.py:115-121)."""
    return [prefix + "Discriminator/fully_connected/weights", prefix + "Discriminator/fully_connected_1/weights"]


def regularised_feat_weights(prefix, bands, patches, embed):
    return [f"{prefix}FeatDiscriminator/{sc}/weights"
            for sc, _, _ in M.feature_discriminator_layer_table(bands, patches, embed)]


# ----------------------------------------------------------------------------- phase losses
class GanConfig:
    def __init__(self, kind, bands, cycle_weight=10.0, identity_weight=0.5, use_identity=True, nce_weight=10.0,
                 tau=0.07, patches=6, embed=2, dis_reg=1e-5, feat_reg=1e-4, generator_lr=2e-4, discriminator_lr=1e-4,
                 gen_discriminator_lr=1e-4, max_steps=1000):
        self.__dict__.update(locals())
        del self.__dict__["self"]


def _G(ctx, x, prefix, only_encoder=False):
    return M.generator_forward(ctx, x, only_encoder=only_encoder, prefix=prefix + "Generator/")


def _D(ctx, x, prefix):
    return M.discriminator_forward(ctx, x, prefix=prefix + "Discriminator/")


def _F(ctx, x, prefix, cfg):
    return M.feature_discriminator_forward(ctx, x, cfg.patches, cfg.embed, prefix=prefix + "FeatDiscriminator/")


def phase_list(kind):
    """Order of the sequential train ops of one global step."""
    if kind in ("gan_x2y", "gan_y2x", "cycle_gan"):
        return ["gen", "dis"]
    if kind in ("cut_x2y", "cut_y2x"):
        return ["gen", "dis", "feat"]
    return ["x2y:gen", "x2y:dis", "x2y:feat", "y2x:gen", "y2x:dis", "y2x:feat"]  # dcl_gan_wrapper.py:224-227


def _cut_terms(ctx, cfg, prefix, gen_in, real):
    """cut_model (cut_wrapper.py:256-356) for one direction."""
    fake = _G(ctx, gen_in, prefix)
    d_gen, d_real = _D(ctx, fake, prefix), _D(ctx, real, prefix)
    f_gen = _F(ctx, _G(ctx, fake, prefix, True), prefix, cfg)
    f_x = _F(ctx, _G(ctx, gen_in, prefix, True), prefix, cfg)
    f_y = _F(ctx, _G(ctx, real, prefix, True), prefix, cfg)
    idt = _G(ctx, real, prefix)
    f_idt = _F(ctx, _G(ctx, idt, prefix, True), prefix, cfg)
    return dict(fake=fake, d_gen=d_gen, d_real=d_real, nce_x=nce_loss(f_gen, f_x, cfg.tau),
                nce_id=nce_loss(f_idt, f_y, cfg.tau))


def phase_loss(cfg, params, x, y, phase, pooled=None):
    """Returns (loss Var, ctx, trained variable names) for one phase.  `pooled`: optional dict of tensor-pool
    outputs {"fake_y": ..., "fake_x": ...} fed to the discriminators (tfgan.features.tensor_pool); when absent the
    freshly generated data is used (pool still filling / pass-through draw)."""
    kind = cfg.kind
    ctx = M.Ctx(params, True)
    xv, yv = O.Var(x), O.Var(y)
    names = gan_param_names(kind, cfg.bands, cfg.patches, cfg.embed)
    id_w = cfg.identity_weight if cfg.use_identity else 0.0

    if kind in ("gan_x2y", "gan_y2x"):
        gen_in, real = (yv, xv) if kind == "gan_y2x" else (xv, yv)
        pre = "Model/"
        fake = _G(ctx, gen_in, pre)
        if phase == "gen":
            return wasserstein_generator_loss(_D(ctx, fake, pre)), ctx, names[pre]["gen"]
        fk = O.Var(pooled["fake"]) if pooled else O.Var(fake.v)
        loss = O.add(wasserstein_discriminator_loss(_D(ctx, real, pre), _D(ctx, fk, pre)),
                     l2_reg(ctx, regularised_dis_weights(pre), cfg.dis_reg))
        return loss, ctx, names[pre]["dis"]

    if kind == "cycle_gan":
        px, py = "Model/ModelX2Y/", "Model/ModelY2X/"
        fake_y, fake_x = _G(ctx, xv, px), _G(ctx, yv, py)
        if phase == "gen":
            rec_x, rec_y = _G(ctx, fake_y, py), _G(ctx, fake_x, px)
            aux = O.scale(cycle_consistency(xv, rec_x, yv, rec_y), cfg.cycle_weight)
            if cfg.use_identity:  # identity_x = G_x2y(x), identity_y = G_y2x(y)  (cycle_gan_wrapper.py:303-333)
                aux = O.add(aux, O.scale(O.add(abs_diff(xv, fake_y), abs_diff(yv, fake_x)), cfg.identity_weight))
            # each generator is trained on ITS adversarial term + the full aux loss; the two adversarial terms have
            # disjoint variable support, so one combined loss yields both gradient sets
            loss = O.add(O.add(ls_generator_loss(_D(ctx, fake_y, px)), ls_generator_loss(_D(ctx, fake_x, py))), aux)
            return loss, ctx, names[px]["gen"] + names[py]["gen"]
        fy = O.Var(pooled["fake_y"]) if pooled else O.Var(fake_y.v)
        fx = O.Var(pooled["fake_x"]) if pooled else O.Var(fake_x.v)
        # model_x2y: real_data = y, discriminator sees (y, G(x)); model_y2x: real_data = x
        loss = O.add(ls_discriminator_loss(_D(ctx, yv, px), _D(ctx, fy, px)),
                     ls_discriminator_loss(_D(ctx, xv, py), _D(ctx, fx, py)))
        loss = O.add(loss, O.add(l2_reg(ctx, regularised_dis_weights(px), cfg.dis_reg),
                                 l2_reg(ctx, regularised_dis_weights(py), cfg.dis_reg)))
        return loss, ctx, names[px]["dis"] + names[py]["dis"]

    # ---- CUT family ----
    if kind in ("cut_x2y", "cut_y2x"):
        pre = "Model/"
        gen_in, real = (yv, xv) if kind == "cut_y2x" else (xv, yv)
        sub = phase
    else:
        direction, sub = phase.split(":")
        pre = "Model/ModelX2Y/" if direction == "x2y" else "Model/ModelY2X/"
        gen_in, real = (xv, yv) if direction == "x2y" else (yv, xv)
    t = _cut_terms(ctx, cfg, pre, gen_in, real)
    if sub == "gen":
        loss = O.add(ls_generator_loss(t["d_gen"]),
                     O.add(O.scale(t["nce_x"], cfg.nce_weight), O.scale(t["nce_id"], id_w)))
        return loss, ctx, names[pre]["gen"]
    if sub == "dis":
        loss = O.add(ls_discriminator_loss(t["d_real"], t["d_gen"]),
                     l2_reg(ctx, regularised_dis_weights(pre), cfg.dis_reg))
        return loss, ctx, names[pre]["dis"]
    loss = O.add(t["nce_x"], l2_reg(ctx, regularised_feat_weights(pre, cfg.bands, cfg.patches, cfg.embed),
                                    cfg.feat_reg))
    return loss, ctx, names[pre]["feat"]


def phase_gradients(cfg, params, x, y, phase, pooled=None):
    loss, ctx, trained = phase_loss(cfg, params, x, y, phase, pooled)
    O.backward(loss)
    grads = {}
    for k in trained:
        g = ctx.vars[k].g
        grads[k] = np.zeros_like(params[k]) if g is None else g
    return float(loss.v), grads


class GanTrainer:
    """Sequential phases with TF1 Adam(beta1=0.5) and the constant-then-linear-decay LR (gan_common.py:222-279)."""

    def __init__(self, cfg, params):
        self.cfg = cfg
        self.params = params
        self.slots = {k: (np.zeros_like(v), np.zeros_like(v)) for k, v in params.items()}
        self.t = {}  # Adam step count per optimiser (phase kind)
        self.global_step = 0

    def _lr(self, phase):
        sub = phase.split(":")[-1]
        base = {"gen": self.cfg.generator_lr, "dis": self.cfg.discriminator_lr,
                "feat": self.cfg.gen_discriminator_lr}[sub]
        return gan_lr(base, self.global_step, self.cfg.max_steps)

    def step(self, x, y, pooled_fn=None):
        losses = {}
        for phase in phase_list(self.cfg.kind):
            pooled = pooled_fn(phase, self.params) if pooled_fn else None
            loss, grads = phase_gradients(self.cfg, self.params, x, y, phase, pooled)
            losses[phase] = loss
            # every variable has its own beta-power accumulators in TF; a variable is updated once per global
            # step, so its Adam t equals global_step + 1
            for k, g in grads.items():
                m, v = self.slots[k]
                adam_tf1_step(self.params[k], g, m, v, self._lr(phase), self.global_step + 1, beta1=0.5)
        self.global_step += 1
        return losses
