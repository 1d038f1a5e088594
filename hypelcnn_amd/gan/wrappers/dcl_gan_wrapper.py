"""DCLGAN wrapper (reference gan/wrappers/dcl_gan_wrapper.py:28-319): two CUT models in ModelX2Y / ModelY2X.
The reference's attempt to couple the two generator losses is a discarded `_replace(...)` (:189-190), so the six
train ops (x2y gen/dis/feat, then y2x gen/dis/feat, :224-227) are those of two independent CUT models."""
import collections

from hypelcnn_amd import graph as G
from hypelcnn_amd.gan.wrappers import gan_common as C
from hypelcnn_amd.gan.wrappers.cut_wrapper import cut_model, cut_phases, cut_train_ops
from hypelcnn_amd.gan.wrappers.wrapper import Wrapper

DCLGANModel = collections.namedtuple("DCLGANModel", ("model_x2y", "model_y2x"))


def dcl_gan_model(generator_fn, discriminator_fn, feat_discriminator_fn, image_x, image_y,
                  model_x2y_scope="ModelX2Y", model_y2x_scope="ModelY2X"):
    with G.variable_scope(model_x2y_scope):
        m_x2y = cut_model(generator_fn, discriminator_fn, feat_discriminator_fn, generator_inputs=image_x,
                          real_data=image_y)
    with G.variable_scope(model_y2x_scope):
        m_y2x = cut_model(generator_fn, discriminator_fn, feat_discriminator_fn, generator_inputs=image_y,
                          real_data=image_x)
    return DCLGANModel(m_x2y, m_y2x)


class DCLGANWrapper(Wrapper):
    def __init__(self, nce_loss_weight, identity_loss_weight, use_identity_loss, tau, batch_size, generator_fn,
                 discriminator_fn, feat_discriminator_fn):
        self._nce_loss_weight = nce_loss_weight
        self._identity_loss_weight = 0.0 if not use_identity_loss else identity_loss_weight
        self._tau = tau
        self._batch_size = batch_size
        self._generator_fn, self._discriminator_fn = generator_fn, discriminator_fn
        self._feat_discriminator_fn = feat_discriminator_fn
        self.backend = None

    def define_model(self, images_x, images_y):
        with G.variable_scope(C.model_base_name):
            return dcl_gan_model(self._generator_fn, self._discriminator_fn, self._feat_discriminator_fn, images_x,
                                 images_y)

    def define_loss(self, model):
        phases = cut_phases(model.model_x2y, self._nce_loss_weight, self._identity_loss_weight, self._tau, "x2y:") + \
            cut_phases(model.model_y2x, self._nce_loss_weight, self._identity_loss_weight, self._tau, "y2x:")
        return C.GANLoss(phases, model.model_x2y.tower, [model.model_x2y.generated_data, model.model_y2x.generated_data])

    def define_train_ops(self, model, loss, max_number_of_steps, **kwargs):
        return cut_train_ops(loss, max_number_of_steps, kwargs, backend=self.backend)

    def get_train_hooks_fn(self):
        return lambda train_ops: [train_ops.run_step]
