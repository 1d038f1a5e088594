"""Harness for the GAN parity tests: build a wrapper through the product API on a backend, inject the oracle's
parameters, run phases, compare with oracle/gan.py."""
from types import SimpleNamespace

import numpy as np
import torch

from hypelcnn_amd.gan.wrapper_registry import get_wrapper_dict
from hypelcnn_amd.gan.wrappers import gan_common as C
from oracle import gan as OG


def flags_for(cfg, batch):
    return SimpleNamespace(discriminator_reg_scale=cfg.dis_reg, gen_disc_reg_scale=cfg.feat_reg,
                           embedded_feat_size=cfg.embed, patches=cfg.patches,
                           cycle_consistency_loss_weight=cfg.cycle_weight, identity_loss_weight=cfg.identity_weight,
                           use_identity_loss=cfg.use_identity, nce_loss_weight=cfg.nce_weight, tau=cfg.tau,
                           batch_size=batch)


def build(cfg, batch, backend):
    wrapper = get_wrapper_dict(flags_for(cfg, batch))[cfg.kind]
    wrapper.backend = backend
    tower, x, y = C.new_gan_tower(cfg.bands)
    model = wrapper.define_model(x, y)
    loss = wrapper.define_loss(model)
    ops = wrapper.define_train_ops(model, loss, max_number_of_steps=cfg.max_steps, generator_lr=cfg.generator_lr,
                                   discriminator_lr=cfg.discriminator_lr, gen_discriminator_lr=cfg.gen_discriminator_lr)
    ops.capture_graphs = False
    return wrapper, model, loss, ops


def inject(sess, params):
    names = set(sess.variable_names())
    assert names == set(params), (sorted(names - set(params))[:5], sorted(set(params) - names)[:5])
    for k, v in params.items():
        sess.set_variable(k, v)


def fp32(params):
    return {k: v.astype(np.float32).astype(np.float64) for k, v in params.items()}


def check_phase_gradients(cfg, ops, params, x, y, tol=2e-4):
    """Every phase at the SAME parameters (no optimiser): loss value and the trained groups' gradients."""
    sess = ops.ctx.session()
    dev = sess.params.device
    xt = torch.as_tensor(x.reshape(x.shape[0], -1), dtype=torch.float32).to(dev)
    yt = torch.as_tensor(y.reshape(y.shape[0], -1), dtype=torch.float32).to(dev)
    worst = 0.0
    for phase in ops.loss.phases:
        ct = ops._compiled(sess, phase, x.shape[0])
        ops._feed(ct, xt, yt)
        if phase.pool:  # pool pass-through: the discriminator sees the freshly generated data
            gen = sess.compile_phase(ops.loss.tower, x.shape[0], outputs=[t for _, t in phase.pool], key="generate")
            ops._feed(gen, xt, yt)
            gen.forward()
            for name, t in phase.pool:
                ct.set_input(name, gen.value(t))
        ct.forward_backward()
        ref_loss, ref_grads = OG.phase_gradients(cfg, params, x, y, phase.name)
        got_loss = ct.loss_value()
        assert abs(got_loss - ref_loss) < tol * max(1.0, abs(ref_loss)), (phase.name, got_loss, ref_loss)
        g_all = max(np.abs(g).max() for g in ref_grads.values())
        for k, g in ref_grads.items():
            got = sess.get_gradient(k)
            if np.abs(g).max() <= 1e-9 * g_all:
                # RULE for analytically-zero gradients (last bias of a Wasserstein critic: +1/N per real and -1/N per fake
                # sample; the float64 oracle gives 1e-17): the product sums contributions of total magnitude <= the loss
                # weight (1.0: N terms of 1/N on each side) in fp32, and whether the two halves cancel EXACTLY depends on
                # the summation tree -- two separate applications whose slab sums mirror each other do, one application
                # on the row-concatenated batch [real; fake] (round 4) leaves the rounding of a sum of unit magnitude.
                # The check is therefore absolute: |got| <= 8 eps_fp32 * max(1, largest gradient of the phase).
                lim = 8 * np.finfo(np.float32).eps * max(1.0, g_all)
                assert np.abs(got - g).max() <= lim, (phase.name, k, np.abs(got - g).max(), lim)
                continue
            scale = max(np.abs(g).max(), 1e-5 * g_all, 1e-7)
            err = np.abs(got - g).max() / scale
            worst = max(worst, err)
            assert err < tol, (phase.name, k, err)
    return worst
