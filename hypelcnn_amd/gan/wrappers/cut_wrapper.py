"""CUT wrapper (reference gan/wrappers/cut_wrapper.py:90-665): LS-GAN + patch-NCE between feature-discriminator
embeddings of encoder features; three sequential train ops (generator, discriminator, feature discriminator)."""
import collections

from hypelcnn_amd import graph as G
from hypelcnn_amd.gan.wrappers import gan_common as C
from hypelcnn_amd.gan.wrappers.wrapper import Wrapper

CUTTrainSteps = collections.namedtuple("CUTTrainSteps", ("generator_train_steps", "discriminator_train_steps",
                                                         "gen_discriminator_train_steps"))
CUTModel = collections.namedtuple("CUTModel", (
    "tower", "generator_inputs", "generated_data", "generator_scope", "real_data", "discriminator_real_outputs",
    "discriminator_gen_outputs", "discriminator_scope", "feat_discriminator_gen_data",
    "feat_discriminator_gen_data_scope", "feat_discriminator_real_data_x", "feat_discriminator_real_data_y",
    "feat_discriminator_generated_data_y", "patches", "embed"))


def cut_model(generator_fn, discriminator_fn, feat_discriminator_fn, real_data, generator_inputs,
              generator_scope="Generator", discriminator_scope="Discriminator",
              feat_discriminator_scope="FeatDiscriminator"):
    """reference :256-356.  The generator is applied 7 times (4 of them encoder-only), the feature discriminator 4
    times, all sharing variables."""
    tower = generator_inputs.tower

    def gen(t, only_encoder):
        with G.variable_scope(generator_scope) as s:
            return generator_fn(t, create_only_encoder=only_encoder), s

    def feat(t):
        with G.variable_scope(feat_discriminator_scope) as s:
            return feat_discriminator_fn(t), s

    generated, gs = gen(generator_inputs, False)
    with G.variable_scope(discriminator_scope) as ds:
        d_gen = discriminator_fn(generated, generator_inputs)
    with G.variable_scope(discriminator_scope):
        d_real = discriminator_fn(real_data, generator_inputs)
    f_gen, fs = feat(gen(generated, True)[0])
    f_x, _ = feat(gen(generator_inputs, True)[0])
    f_y, _ = feat(gen(real_data, True)[0])
    identity = gen(real_data, False)[0]
    f_idt, _ = feat(gen(identity, True)[0])
    parts = f_gen.parts
    return CUTModel(tower, generator_inputs, generated, gs, real_data, d_real, d_gen, ds, f_gen, fs, f_x, f_y, f_idt,
                    parts, f_gen.c // parts)


def cut_phases(model, nce_loss_weight, nce_identity_loss_weight, tau, prefix=""):
    """cut_loss (reference :90-208) split into its three train ops (cut_train_ops :467-584)."""
    def nce(a, b, w):
        return G.LossTerm("nce", a, b, weight=w, tau=tau, parts=model.patches, embed=model.embed)

    gen_terms = C.ls_terms_generator(model.discriminator_gen_outputs) + \
        [nce(model.feat_discriminator_gen_data, model.feat_discriminator_real_data_x, nce_loss_weight)]
    if nce_identity_loss_weight:
        gen_terms.append(nce(model.feat_discriminator_generated_data_y, model.feat_discriminator_real_data_y,
                             nce_identity_loss_weight))
    return [
        C.Phase(prefix + "gen", gen_terms, [model.generator_scope], "gen", None),
        C.Phase(prefix + "dis", C.ls_terms_discriminator(model.discriminator_real_outputs,
                                                         model.discriminator_gen_outputs),
                [model.discriminator_scope], "dis", None),
        C.Phase(prefix + "feat", [nce(model.feat_discriminator_gen_data, model.feat_discriminator_real_data_x, 1.0)],
                [model.feat_discriminator_gen_data_scope], "feat", None)]


def cut_train_ops(loss, max_number_of_steps, kwargs, backend=None):
    lrs = {"gen": C._get_lr(kwargs["generator_lr"], max_number_of_steps),
           "dis": C._get_lr(kwargs["discriminator_lr"], max_number_of_steps),
           "feat": C._get_lr(kwargs["gen_discriminator_lr"], max_number_of_steps)}
    return C.GANTrainOps(loss, lrs, C.GanContext(loss.tower, backend), use_pool=False)


class CUTWrapper(Wrapper):
    def __init__(self, nce_loss_weight, identity_loss_weight, use_identity_loss, tau, batch_size, swap_inputs,
                 generator_fn, discriminator_fn, feat_discriminator_fn):
        self._nce_loss_weight = nce_loss_weight
        self._identity_loss_weight = 0.0 if not use_identity_loss else identity_loss_weight
        self._swap_inputs = swap_inputs
        self._tau = tau
        self._batch_size = batch_size
        self._generator_fn, self._discriminator_fn = generator_fn, discriminator_fn
        self._feat_discriminator_fn = feat_discriminator_fn
        self.backend = None

    def define_model(self, images_x, images_y):
        gen_in, real = (images_y, images_x) if self._swap_inputs else (images_x, images_y)
        with G.variable_scope(C.model_base_name):
            return cut_model(self._generator_fn, self._discriminator_fn, self._feat_discriminator_fn,
                             generator_inputs=gen_in, real_data=real)

    def define_loss(self, model):
        return C.GANLoss(cut_phases(model, self._nce_loss_weight, self._identity_loss_weight, self._tau), model.tower,
                         [model.generated_data])

    def define_train_ops(self, model, loss, max_number_of_steps, **kwargs):
        return cut_train_ops(loss, max_number_of_steps, kwargs, backend=self.backend)

    def get_train_hooks_fn(self):
        return lambda train_ops: [train_ops.run_step]
