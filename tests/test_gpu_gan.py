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
                                                  ("dcl_cycle_gan", 144, 6, 40)])
def test_phase_gradients_match_oracle(hip, kind, bands, patches, n):
    cfg = OG.GanConfig(kind, bands, patches=patches, max_steps=20)
    params = U.fp32(OG.init_gan_params(kind, bands, np.random.default_rng(2), patches=patches, dtype=np.float64,
                                       zero_generator=False))
    x, y = _data(n, bands, 4)
    wrapper, model, loss, ops = U.build(cfg, n, hip)
    sess = ops.ctx.session()
    U.inject(sess, params)
    U.check_phase_gradients(cfg, ops, params, x, y, tol=2e-3)


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
