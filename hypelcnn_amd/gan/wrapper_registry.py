"""Name -> wrapper / sampler dictionaries (reference gan/wrapper_registry.py:13-94), same keys."""
from functools import partial

from hypelcnn_amd.gan.gan_sampling_methods import DummySampler, NeighborhoodBasedSampler, RandomBasedSampler, \
    TargetBasedSampler
from hypelcnn_amd.gan.shadow_data_models import shadowdata_discriminator_model, \
    shadowdata_feature_discriminator_model, shadowdata_generator_model
from hypelcnn_amd.gan.wrappers.cut_wrapper import CUTWrapper
from hypelcnn_amd.gan.wrappers.cycle_gan_wrapper import CycleGANInferenceWrapper, CycleGANWrapper
from hypelcnn_amd.gan.wrappers.dcl_cycle_gan_wrapper import DCLCycleGANWrapper
from hypelcnn_amd.gan.wrappers.dcl_gan_wrapper import DCLGANWrapper
from hypelcnn_amd.gan.wrappers.gan_wrapper import GANWrapper


def get_sampling_map():
    return {"target": TargetBasedSampler(margin=5), "random": RandomBasedSampler(multiply_shadowed_data=True),
            "neighbour": NeighborhoodBasedSampler(neighborhood_size=20, margin=2),
            "dummy": DummySampler(element_count=2000, fill_value=0.5, coefficient=2)}


def get_infer_wrapper_dict():
    generator_fn = partial(shadowdata_generator_model, create_only_encoder=False, is_training=False)
    cyc = CycleGANInferenceWrapper(shadow_generator_fn=generator_fn)
    return {"cycle_gan": cyc, "dcl_gan": cyc, "dcl_cycle_gan": cyc}


def get_wrapper_dict(flags):
    generator_fn = partial(shadowdata_generator_model, is_training=True)
    generator_fn_gan = partial(shadowdata_generator_model, create_only_encoder=False, is_training=True)
    discriminator_fn = partial(shadowdata_discriminator_model, is_training=True, scale=flags.discriminator_reg_scale)
    feat_discriminator_fn = partial(shadowdata_feature_discriminator_model,
                                    embedded_feature_size=flags.embedded_feat_size, patch_count=flags.patches,
                                    is_training=True, scale=flags.gen_disc_reg_scale)
    common = dict(identity_loss_weight=flags.identity_loss_weight, use_identity_loss=flags.use_identity_loss)
    cut = dict(nce_loss_weight=flags.nce_loss_weight, tau=flags.tau, batch_size=flags.batch_size,
               generator_fn=generator_fn, discriminator_fn=discriminator_fn,
               feat_discriminator_fn=feat_discriminator_fn, **common)
    return {
        "cycle_gan": CycleGANWrapper(cycle_consistency_loss_weight=flags.cycle_consistency_loss_weight,
                                     generator_fn=generator_fn_gan, discriminator_fn=discriminator_fn, **common),
        "gan_x2y": GANWrapper(swap_inputs=False, generator_fn=generator_fn_gan, discriminator_fn=discriminator_fn,
                              **common),
        "gan_y2x": GANWrapper(swap_inputs=True, generator_fn=generator_fn_gan, discriminator_fn=discriminator_fn,
                              **common),
        "cut_x2y": CUTWrapper(swap_inputs=False, **cut),
        "cut_y2x": CUTWrapper(swap_inputs=True, **cut),
        "dcl_gan": DCLGANWrapper(**cut),
        "dcl_cycle_gan": DCLCycleGANWrapper(cycle_consistency_loss_weight=flags.cycle_consistency_loss_weight, **cut)}
