"""CPU (numpy kernel emulation): the GAN wrappers of the product vs oracle/gan.py -- per-phase losses and
gradients for every wrapper kind, then multi-step training (sequential phases, TF1 Adam beta1=0.5, LR schedule)."""
import numpy as np
import pytest
import torch

from oracle import gan as OG
from tests import gan_util as U
from tests.emu_backend import EmuBackend


def _data(n, b, seed):
    rng = np.random.default_rng(seed)
    return rng.random((n, 1, 1, b)).astype(np.float32).astype(np.float64), \
        (rng.random((n, 1, 1, b)) * 0.5).astype(np.float32).astype(np.float64)


@pytest.mark.parametrize("kind,bands,patches", [("cycle_gan", 16, 4), ("gan_x2y", 16, 4), ("gan_y2x", 24, 4),
                                                ("cut_x2y", 24, 6), ("cut_y2x", 64, 6), ("dcl_gan", 16, 4),
                                                ("dcl_cycle_gan", 16, 4)])
def test_phase_gradients_match_oracle(kind, bands, patches):
    n = 6
    cfg = OG.GanConfig(kind, bands, patches=patches, max_steps=20)
    params = U.fp32(OG.init_gan_params(kind, bands, np.random.default_rng(2), patches=patches, dtype=np.float64,
                                       zero_generator=False))
    x, y = _data(n, bands, 4)
    wrapper, model, loss, ops = U.build(cfg, n, EmuBackend())
    assert [p.name for p in loss.phases] == OG.phase_list(kind)
    sess = ops.ctx.session()
    U.inject(sess, params)
    U.check_phase_gradients(cfg, ops, params, x, y, tol=5e-5)


@pytest.mark.parametrize("kind", ["cycle_gan", "cut_x2y"])
def test_training_steps_track_oracle_trainer(kind):
    bands, n, steps = 16, 6, 6
    cfg = OG.GanConfig(kind, bands, patches=4, max_steps=8)   # LR decays from step 4 on
    params = U.fp32(OG.init_gan_params(kind, bands, np.random.default_rng(3), patches=4, dtype=np.float64,
                                       zero_generator=False))
    wrapper, model, loss, ops = U.build(cfg, n, EmuBackend())
    ops.pool_override = lambda name, fresh: fresh            # pass-through pool (the oracle trainer does the same)
    sess = ops.ctx.session()
    U.inject(sess, params)
    trainer = OG.GanTrainer(cfg, {k: v.copy() for k, v in params.items()})
    for s in range(steps):
        x, y = _data(n, bands, 100 + s)
        ops.run_step(torch.as_tensor(x.reshape(n, -1), dtype=torch.float32),
                     torch.as_tensor(y.reshape(n, -1), dtype=torch.float32))
        ref_losses = trainer.step(x, y)
        got = ops.losses()
        for ph, v in ref_losses.items():
            assert abs(got[ph] - v) < 2e-3 * max(1.0, abs(v)), (s, ph, got[ph], v)
    assert sess.global_step == steps
    for k, v in trainer.params.items():
        got = sess.get_variable(k)
        assert np.abs(got - v).max() < 2e-3 * max(np.abs(v).max(), 1e-3), k


def test_zero_init_generator_known_answers():
    """K1: zero weights -> encoder output 5x, generator output tanh(0) = 0 (reference zero-initialises, :47)."""
    from hypelcnn_amd.gan.shadow_data_models import shadowdata_generator_model
    from hypelcnn_amd.gan.wrappers import gan_common as C
    from hypelcnn_amd import graph as G
    tower, x, _ = C.new_gan_tower(16)
    with G.variable_scope("Model"), G.variable_scope("Generator"):
        enc = shadowdata_generator_model(x, True)
    with G.variable_scope("Model"), G.variable_scope("Generator"):
        full = shadowdata_generator_model(x, False)
    ctx = C.GanContext(tower, EmuBackend())
    sess = ctx.session()
    assert len(sess.variable_names()) == 14 and float(sess.params.abs().sum()) == 0.0
    ct = sess.compile_phase(tower, 3, outputs=[enc, full], key="k1")
    xv = torch.rand(3, 16)
    ct.set_input("x", xv)
    ct.forward()
    torch.testing.assert_close(ct.value(enc), 5 * xv)
    assert float(ct.value(full).abs().max()) == 0.0


def test_tensor_pool_semantics():
    from hypelcnn_amd.gan.wrappers.gan_common import TensorPool
    pool = TensorPool(pool_size=3, pooling_probability=0.5, seed=0)
    vals = [torch.full((2,), float(i)) for i in range(40)]
    outs = [pool.query(v) for v in vals]
    assert all(torch.equal(o, v) for o, v in zip(outs[:3], vals[:3]))           # filling: pass-through
    later = [float(o[0]) for o in outs[3:]]
    same = sum(1 for o, v in zip(outs[3:], vals[3:]) if torch.equal(o, v))
    assert 8 < same < 30 and any(l < i + 3 for i, l in enumerate(later))        # some historical samples returned
    assert len(pool.items) == 3


def _phase_launches(ops, sess, n):
    out = {}
    for phase in ops.loss.phases:
        plan = ops._compiled(sess, phase, n).plan
        out[phase.name] = [l.name for l in plan.fwd + plan.bwd]
    return out


@pytest.mark.parametrize("kind,bands", [("cut_x2y", 24), ("cycle_gan", 16), ("dcl_gan", 16)])
def test_same_weight_applications_run_as_one_row_concatenated_application(monkeypatch, kind, bands):
    """Round 4 (plan_gan.PhasePlan._schedule_units): the same-weight applications of a train op -- G([x; y]), enc on the four
    inputs of CUT, D([real; fake]), the feature-discriminator layers, the feature stack with per-application norms -- run
    as ONE application on the row-concatenated batch, and every per-block gradient slab of the op is summed by one launch.
    Fewer launches, same losses and gradients as the oracle; the unbatched form stays available and agrees too.
    CycleGAN's G_xy(G_yx(y)) next to G_xy(x) must NOT be grouped (the unit graph would be cyclic)."""
    from hypelcnn_amd import plan_gan
    n = 6
    cfg = OG.GanConfig(kind, bands, patches=4 if bands == 16 else 6, max_steps=20)
    params = U.fp32(OG.init_gan_params(kind, bands, np.random.default_rng(2), patches=cfg.patches, dtype=np.float64,
                                       zero_generator=False))
    x, y = _data(n, bands, 4)
    counts = {}
    for batched in (True, False):
        monkeypatch.setattr(plan_gan, "BATCH_APPS", batched)
        monkeypatch.setattr(plan_gan, "SLAB_REDUCE_MULTI", batched)
        wrapper, model, loss, ops = U.build(cfg, n, EmuBackend())
        sess = ops.ctx.session()
        U.inject(sess, params)
        U.check_phase_gradients(cfg, ops, params, x, y, tol=5e-5)
        counts[batched] = _phase_launches(ops, sess, n)
    tot = {b: sum(len(v) for v in counts[b].values()) for b in counts}
    assert tot[True] < tot[False], tot
    gen_b = [l for l in counts[True]["gen" if "gen" in counts[True] else list(counts[True])[0]] if "generator_fwd" in l]
    gen_u = [l for l in counts[False]["gen" if "gen" in counts[False] else list(counts[False])[0]] if "generator_fwd" in l]
    if kind == "cut_x2y":
        # unbatched: six applications.  Batched: G([x; y]) whose n_4 doubles as enc(x), enc(y) (the encoder tap: one
        # launch writes both, one backward launch takes both gradients), and enc([G(x); G(y)])
        assert len(gen_u) == 6 and sorted(gen_b) == ["gan_generator_fwd_keep", "gan_generator_fwd_tap"], (gen_u, gen_b)
        assert "gan_generator_bwd_tap" in counts[True]["gen"] and "gan_generator_fwd_tap" in counts[True]["feat"]
        assert tot[True] * 2 <= tot[False] + 10, tot
    if kind == "cycle_gan":
        # the four generator applications depend on each other pairwise: the same-weight ones cannot share a launch, but
        # G_x2y(x) | G_y2x(y), then G_y2x(fake_y) | G_x2y(fake_x) -- same shape, different variables -- do (hypel.h: *_apps),
        # and so do the two critics of either phase
        assert len(gen_u) == 4 and gen_b == ["gan_generator_fwd_apps"] * 2, (gen_u, gen_b)
        for ph in ("gen", "dis"):
            assert counts[True][ph].count("dense_stack_bwd_apps") == 1 and "dense_stack_bwd" not in counts[True][ph]
        assert counts[True]["gen"].count("gan_generator_bwd_apps") == 2
        # the batched inputs are neighbouring row blocks of the tower's input slab (tower.input_layout): nothing to gather
        assert not any(l == "copy_blocks_f32" for v in counts[True].values() for l in v), counts[True]
        assert sum(l == "reduce_splits_wave_multi_f32" for v in counts[True].values() for l in v) == 2


@pytest.mark.parametrize("kind,bands", [("cut_x2y", 24), ("cut_x2y", 144), ("cycle_gan", 144)])
def test_unbatched_applications_with_the_one_launch_slab_reduction(monkeypatch, kind, bands):
    """BATCH_APPS off, SLAB_REDUCE_MULTI on (round-4 advisor finding): a BN-less layer applied twice as separate units leaves
    two chunk-sum entries for ONE bias gradient; they must not share a reduction launch (the emulation asserts that no two
    entries of a launch write the same output) -- the second entry goes to a second launch, and the gradients stay the
    oracle's."""
    from hypelcnn_amd import plan_gan
    monkeypatch.setattr(plan_gan, "BATCH_APPS", False)
    monkeypatch.setattr(plan_gan, "SLAB_REDUCE_MULTI", True)
    n = 6
    cfg = OG.GanConfig(kind, bands, patches=6, max_steps=20)
    params = U.fp32(OG.init_gan_params(kind, bands, np.random.default_rng(2), patches=cfg.patches, dtype=np.float64,
                                       zero_generator=False))
    x, y = _data(n, bands, 4)
    wrapper, model, loss, ops = U.build(cfg, n, EmuBackend())
    U.inject(ops.ctx.session(), params)
    U.check_phase_gradients(cfg, ops, params, x, y, tol=5e-5)


def test_two_variable_sets_in_one_launch_can_be_switched_off(monkeypatch):
    """HYPEL_GAN_BATCH_HETERO=0: CycleGAN's generators and critics run one launch per variable set again; both forms give
    the oracle's phase gradients (the default form is covered by test_phase_gradients_match_oracle)."""
    from hypelcnn_amd import plan_gan
    n, bands = 6, 16
    cfg = OG.GanConfig("cycle_gan", bands, patches=4, max_steps=20)
    params = U.fp32(OG.init_gan_params("cycle_gan", bands, np.random.default_rng(2), patches=4, dtype=np.float64,
                                       zero_generator=False))
    x, y = _data(n, bands, 4)
    monkeypatch.setattr(plan_gan, "BATCH_HETERO", False)
    wrapper, model, loss, ops = U.build(cfg, n, EmuBackend())
    sess = ops.ctx.session()
    U.inject(sess, params)
    U.check_phase_gradients(cfg, ops, params, x, y, tol=5e-5)
    names = [l for v in _phase_launches(ops, sess, n).values() for l in v]
    assert not any(l.endswith("_apps") for l in names)
    assert names.count("gan_generator_bwd_kept") == 4 and names.count("dense_stack_bwd") == 4, names


def test_bias_and_leaky_relu_ride_in_the_product(monkeypatch):
    """A tf_slim.fully_connected without a normaliser (the feature-discriminator layers, the wide critic:
    shadow_data_models.py:95-149) is ONE launch: the product's epilogue applies bias + leaky-ReLU (HYPEL_GEMM_ACT_*), the
    backward pass reads act' from the sign of the layer output.  HYPEL_ACT_IN_GEMM=0 restores product -> post-op; both
    forms give the oracle's phase gradients."""
    from hypelcnn_amd import plan
    n, bands = 6, 144  # (wide critic: the 144-band stack runs layer by layer)
    cfg = OG.GanConfig("cut_x2y", bands, patches=6, max_steps=20)
    params = U.fp32(OG.init_gan_params("cut_x2y", bands, np.random.default_rng(2), patches=6, dtype=np.float64,
                                       zero_generator=False))
    x, y = _data(n, bands, 4)
    posts = {}
    for fused in (True, False):
        monkeypatch.setattr(plan, "ACT_IN_GEMM", fused)
        wrapper, model, loss, ops = U.build(cfg, n, EmuBackend())
        sess = ops.ctx.session()
        U.inject(sess, params)
        U.check_phase_gradients(cfg, ops, params, x, y, tol=5e-5)
        plans = [ops._compiled(sess, ph, n).plan for ph in ops.loss.phases]
        posts[fused] = sum(l.name == "bn_act_fwd" for p in plans for l in p.fwd)
        flagged = sum(l.name == "seg_gemm_f32" and (int(l.args[14]) >> 16) & 7 == 1 for p in plans for l in p.fwd)
        assert (flagged > 0) == fused
    assert posts[True] < posts[False], posts
