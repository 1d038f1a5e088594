"""DCL-CycleGAN wrapper (reference gan/wrappers/dcl_cycle_gan_wrapper.py:16-206).  The cycle-consistency loss is
computed but the `_replace(...)` that would add it to the generator losses is discarded (:149-150), so training is
identical to DCLGAN; the reconstructions exist only as model outputs."""
import collections

from hypelcnn_amd import graph as G
from hypelcnn_amd.gan.wrappers import gan_common as C
from hypelcnn_amd.gan.wrappers.dcl_gan_wrapper import DCLGANWrapper, dcl_gan_model

DCLCycleGANModel = collections.namedtuple("DCLCycleGANModel", ("model_x2y", "model_y2x", "reconstructed_x",
                                                                "reconstructed_y"))


class DCLCycleGANWrapper(DCLGANWrapper):
    def __init__(self, nce_loss_weight, identity_loss_weight, cycle_consistency_loss_weight, use_identity_loss, tau,
                 batch_size, generator_fn, discriminator_fn, feat_discriminator_fn):
        super().__init__(nce_loss_weight, identity_loss_weight, use_identity_loss, tau, batch_size, generator_fn,
                         discriminator_fn, feat_discriminator_fn)
        self._cycle_consistency_loss_weight = cycle_consistency_loss_weight

    def define_model(self, images_x, images_y):
        with G.variable_scope(C.model_base_name):
            m = dcl_gan_model(self._generator_fn, self._discriminator_fn, self._feat_discriminator_fn, images_x,
                              images_y)
            with G.variable_scope("ModelY2X"), G.variable_scope("Generator"):
                rec_x = self._generator_fn(m.model_x2y.generated_data, create_only_encoder=False)
            with G.variable_scope("ModelX2Y"), G.variable_scope("Generator"):
                rec_y = self._generator_fn(m.model_y2x.generated_data, create_only_encoder=False)
        return DCLCycleGANModel(m.model_x2y, m.model_y2x, rec_x, rec_y)
