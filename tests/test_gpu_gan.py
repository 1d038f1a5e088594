"""-m gpu: GAN wrappers on the HIP backend vs oracle/gan.py (fp64): per-phase loss + gradients for every wrapper
kind at the BASELINE band counts (Gulfport 64, GRSS2013 144, AVON 360), HIP-graph replay of a phase, and an
end-to-end CycleGAN run on the DummySampler known-answer pairs (x = 2*y: ideal generator halves its input)."""
import numpy as np
import pytest
import torch

from oracle import gan as OG
from tests import gan_util as U

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from hypelcnn_amd.backend import HipBackend
    return HipBackend()


def _data(n, b, seed):
    rng = np.random.default_rng(seed)
    return rng.random((n, 1, 1, b)).astype(np.float32).astype(np.float64), \
        (rng.random((n, 1, 1, b)) * 0.5).astype(np.float32).astype(np.float64)


@pytest.mark.parametrize("kind,bands,patches,n", [("cycle_gan", 64, 6, 256), ("gan_x2y", 64, 6, 100),
                                                  ("gan_y2x", 144, 6, 64), ("cut_x2y", 64, 6, 128),
                                                  ("cut_y2x", 360, 6, 48), ("dcl_gan", 64, 6, 96),
                                                  ("dcl_cycle_gan", 144, 6, 40),
                                                  # the sizes BASELINE quotes: the per-slice l2_normalize of the feature
                                                  # discriminator (shadow_data_models.py:147) and the patch-NCE / batch
                                                  # means couple ALL N samples, so small N does not cover them
                                                  ("cut_x2y", 360, 6, 4096), ("cycle_gan", 64, 6, 2048)])
def test_phase_gradients_match_oracle(hip, kind, bands, patches, n):
    cfg = OG.GanConfig(kind, bands, patches=patches, max_steps=20)
    params = U.fp32(OG.init_gan_params(kind, bands, np.random.default_rng(2), patches=patches, dtype=np.float64,
                                       zero_generator=False))
    x, y = _data(n, bands, 4)
    wrapper, model, loss, ops = U.build(cfg, n, hip)
    sess = ops.ctx.session()
    U.inject(sess, params)
    worst = U.check_phase_gradients(cfg, ops, params, x, y, tol=5e-5)
    print(f"\n{kind} B={bands}: worst phase-gradient error {worst:.2e} (relative to the tensor maximum)")
    # narrow discriminators (every width <= 128) run as ONE launch per direction, wide ones layer by layer
    names = set()
    for phase in ops.loss.phases:
        plan = ops._compiled(sess, phase, n).plan
        names |= {l.name for l in plan.fwd + plan.bwd}
    assert any(n_.startswith("dense_stack_bwd") for n_ in names) == (bands <= 128), sorted(names)  # (plain or *_apps form)


def test_cyclegan_graph_replay_and_training_on_dummy_pairs(hip):
    """cfg4 shape (B=64, batch 2048), DummySampler pairs: x == 1.0, y == 0.5.  Phases are replayed as HIP graphs.
    After training, G_x2y(x) must have moved from 0 toward y and the cycle must reconstruct better than at init."""
    bands, n = 64, 2048
    cfg = OG.GanConfig("cycle_gan", bands, max_steps=400, generator_lr=2e-3, discriminator_lr=1e-3)
    wrapper, model, loss, ops = U.build(cfg, n, hip)
    ops.capture_graphs = True
    sess = ops.ctx.session()
    x = torch.full((n, bands), 1.0).cuda()
    y = torch.full((n, bands), 0.5).cuda()
    ops.run_step(x, y)
    l0 = ops.losses()
    assert abs(l0["gen"] - (1.0 + 10.0 * 0.75 + 0.5 * 1.5)) < 0.2   # K12-style closed form at zero-init generators
    for _ in range(300):
        ops.run_step(x, y)
    l1 = ops.losses()
    assert np.isfinite(l1["gen"]) and l1["gen"] < 0.5 * l0["gen"], (l0, l1)
    gen = sess.compile_phase(loss.tower, n, outputs=loss.generate_outputs, key="generate")
    gen.set_input("x", x); gen.set_input("y", y)
    gen.forward()
    fake_y = gen.value(loss.generate_outputs[0])
    assert float((fake_y - 0.5).abs().mean()) < 0.2, float(fake_y.mean())
    assert sess.global_step == 301


def test_joint_gan_augmentation_and_classifier_loop_on_gpu(tmp_path):
    """SURVEY cfg5 on the device: CycleGAN session -> checkpoint -> classifier whose input pipeline applies the
    trained generator per pixel (one fused generator launch + the fused augmentation kernel)."""
    from hypelcnn_amd.backend import HipBackend
    from tests.test_training_loop_emu import run_joint_loop
    res, seen = run_joint_loop(tmp_path, HipBackend, gan_steps=30, cls_steps=60)
    assert res.test_accuracy > 0.5


def _snapshot(sess):
    return {k: getattr(sess, k).clone() for k in ("params", "slot_m", "slot_v", "state")}, sess.global_step


def _restore(sess, snap):
    for k, v in snap[0].items():
        getattr(sess, k).copy_(v)
    sess.global_step = snap[1]


def test_cut_full_step_batch4096_eager_equals_graph(hip):
    """GAN half of BASELINE configs[4] at its size (CUT x2y on [4096,1,1,360] pairs, patches 6, E 2, tau 0.07): the
    float64 oracle checks the same phases at N = 48 (test_phase_gradients_match_oracle[cut_y2x-360]); at N = 4096 a
    full step (generator, discriminator and feature-discriminator phases + 3 Adam updates) is checked by
    size-independent properties: eager launches == HIP-graph replay == a second run, bit for bit, all finite, and
    every optimiser group moved."""
    bands, n = 360, 4096
    cfg = OG.GanConfig("cut_x2y", bands, patches=6, max_steps=1000)
    params = U.fp32(OG.init_gan_params("cut_x2y", bands, np.random.default_rng(7), patches=6, dtype=np.float64,
                                       zero_generator=False))
    x, y = _data(n, bands, 9)
    xt = torch.as_tensor(x.reshape(n, -1), dtype=torch.float32).cuda()
    yt = torch.as_tensor(y.reshape(n, -1), dtype=torch.float32).cuda()
    wrapper, model, loss, ops = U.build(cfg, n, hip)
    ops.use_pool = False  # the tensor pool draws from a host RNG: pass the fresh fakes through
    sess = ops.ctx.session()
    U.inject(sess, params)
    snap = _snapshot(sess)
    ops.capture_graphs = False
    ops.run_step(xt, yt)
    torch.cuda.synchronize()
    p_eager, l_eager = sess.params.clone(), ops.losses()
    assert torch.isfinite(p_eager).all() and all(np.isfinite(v) for v in l_eager.values()), l_eager
    for gname, (lo, hi) in sess.group_ranges.items():
        assert not torch.equal(p_eager[lo:hi], snap[0]["params"][lo:hi]), f"group {gname} did not move"
    _restore(sess, snap)
    ops.run_step(xt, yt)
    torch.cuda.synchronize()
    assert torch.equal(sess.params, p_eager), "a CUT step must be deterministic"
    _restore(sess, snap)
    ops.capture_graphs = True
    ops.run_step(xt, yt)  # captures every phase, then replays
    torch.cuda.synchronize()
    _restore(sess, snap)
    ops.run_step(xt, yt)
    torch.cuda.synchronize()
    assert torch.equal(sess.params, p_eager), "HIP-graph replay must equal the eager step"
    assert ops.losses() == l_eager


def test_cfg5_joint_loop_at_the_per_gpu_batch_512(tmp_path):
    """BASELINE configs[4] at its per-GPU size (batch 4096 over 8 GPUs = 512 each): the joint loop with
    --batch_size 512 on 7x7x360 patches, the CUT-based (DCL-GAN) generator applied to all 49 pixels of a patch w.p. 0.5
    (gan/gan_utilities.py:30-43, gan_common.py:282-304 -- 512 x 49 = 25 088 generator rows per augmented batch, through
    the matrix-core generator kernel)."""
    import json
    import os
    from hypelcnn_amd.backend import HipBackend
    from tests.test_training_loop_emu import run_joint_loop
    cfg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "hypelcnn_amd", "nnmodel", "modelconfigs")
    alg = json.load(open(os.path.join(cfg_dir, "alg_param_hypelcnn.json")))
    alg["learning_rate"] = 1e-3
    res, seen = run_joint_loop(tmp_path, HipBackend, gan_steps=10, cls_steps=12, gan_type="dcl_gan",
                               scene="avon:h=60:w=80:bands=360:samples=0.6", neighborhood=3, alg=alg, batch=512,
                               gan_batch=512)
    assert np.isfinite(res.loss), res.loss


def test_cfg5_joint_loop_at_avon_shape_on_gpu(tmp_path):
    """BASELINE configs[4] end to end at the AVON shape: B = 360 bands, no LiDAR, 2 classes; the CUT-based DCL-GAN
    trains the shadow generators, the classifier (full alg_param_hypelcnn.json, 7x7x360 patches) then trains with the
    generator applied to all 49 pixels of a patch w.p. 0.5 (gan/gan_utilities.py:30-43, gan_common.py:282-304)."""
    import json
    import os
    from hypelcnn_amd.backend import HipBackend
    from tests.test_training_loop_emu import run_joint_loop
    cfg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "hypelcnn_amd", "nnmodel", "modelconfigs")
    alg = json.load(open(os.path.join(cfg_dir, "alg_param_hypelcnn.json")))
    alg["learning_rate"] = 1e-3
    res, seen = run_joint_loop(tmp_path, HipBackend, gan_steps=20, cls_steps=80, gan_type="dcl_gan",
                               scene="avon:h=40:w=50:bands=360:samples=0.6", neighborhood=3, alg=alg, batch=64,
                               gan_batch=64)
    assert np.isfinite(res.loss) and res.test_accuracy > 0.6, (res.loss, res.test_accuracy)


@pytest.mark.parametrize("kind,bands,n", [("cycle_gan", 64, 96), ("cut_x2y", 64, 64), ("dcl_gan", 64, 48)])
def test_training_steps_track_oracle_trainer_on_gpu(hip, kind, bands, n):
    """Six whole GAN steps on the MI355X (every train op replayed as a HIP graph: two-variable-set launches, slab inputs,
    one Adam launch per train op, the LR decay from step 4 on) against oracle/gan.py::GanTrainer on the same batches:
    losses of every train op and every variable after the last step."""
    steps = 6
    cfg = OG.GanConfig(kind, bands, patches=6, max_steps=8)
    params = U.fp32(OG.init_gan_params(kind, bands, np.random.default_rng(3), patches=6, dtype=np.float64,
                                       zero_generator=False))
    wrapper, model, loss, ops = U.build(cfg, n, hip)
    ops.capture_graphs = True
    ops.pool_override = lambda name, fresh: fresh  # pass-through pool (the oracle trainer does the same)
    sess = ops.ctx.session()
    U.inject(sess, params)
    trainer = OG.GanTrainer(cfg, {k: v.copy() for k, v in params.items()})
    for s in range(steps):
        x, y = _data(n, bands, 100 + s)
        ops.run_step(torch.as_tensor(x.reshape(n, -1), dtype=torch.float32).cuda(),
                     torch.as_tensor(y.reshape(n, -1), dtype=torch.float32).cuda())
        ref_losses = trainer.step(x, y)
        got = ops.losses()
        for ph, v in ref_losses.items():
            assert abs(got[ph] - v) < 2e-3 * max(1.0, abs(v)), (s, ph, got[ph], v)
    assert sess.global_step == steps
    for k, v in trainer.params.items():
        got = sess.get_variable(k)
        assert np.abs(got - v).max() < 2e-3 * max(np.abs(v).max(), 1e-3), k
