"""Vanilla GAN wrapper (reference gan/wrappers/gan_wrapper.py:14-66).  tfgan.gan_loss defaults => Wasserstein
generator / discriminator losses, tensor pool on the discriminator's generated inputs."""
from hypelcnn_amd import graph as G
from hypelcnn_amd.gan.wrappers import gan_common as C
from hypelcnn_amd.gan.wrappers.wrapper import Wrapper


class GANWrapper(Wrapper):
    def __init__(self, identity_loss_weight, use_identity_loss, swap_inputs, generator_fn, discriminator_fn):
        self._identity_loss_weight = identity_loss_weight
        self._use_identity_loss = use_identity_loss
        self._swap_inputs = swap_inputs
        self._generator_fn, self._discriminator_fn = generator_fn, discriminator_fn
        self.backend = None

    def define_model(self, images_x, images_y):
        tower = images_x.tower
        gen_in, real = (images_y, images_x) if self._swap_inputs else (images_x, images_y)
        with G.variable_scope(C.model_base_name):
            return C.build_gan_model(tower, self._generator_fn, self._discriminator_fn, gen_in, real, "pool_fake")

    def define_loss(self, model):
        gen = C.Phase("gen", [G.LossTerm("mean", model.discriminator_gen_outputs, weight=-1.0)],
                      [model.generator_scope], "gen", None)
        dis = C.Phase("dis", [G.LossTerm("mean", model.discriminator_pool_outputs, weight=1.0),
                              G.LossTerm("mean", model.discriminator_real_outputs, weight=-1.0)],
                      [model.discriminator_scope], "dis", [("pool_fake", model.generated_data)])
        return C.GANLoss([gen, dis], model.tower, [model.generated_data])

    def define_train_ops(self, model, loss, max_number_of_steps, **kwargs):
        return C.define_standard_train_ops(model, loss, max_number_of_steps, kwargs["generator_lr"],
                                           kwargs["discriminator_lr"], backend=self.backend)

    def get_train_hooks_fn(self):
        return lambda train_ops: [train_ops.run_step]
