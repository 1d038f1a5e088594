"""-m gpu: the parity tests proper.  The product (plugin API -> planner -> HIP kernels through the C-ABI)
against the oracle (float64) on identical seeded inputs, weights and dropout masks.

Tolerances (BASELINE.json north_star): logits within 1e-3 (fp32), class labels bit-exact.
Gradients are compared relative to each tensor's max magnitude (2e-3)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import train as OT
from tests import parity_util as U

pytestmark = pytest.mark.gpu
CFG = os.path.join(os.path.dirname(os.path.abspath(U.cno.__file__)), "..", "nnmodel", "modelconfigs")


@pytest.fixture(scope="module")
def hip():
    from hypelcnn_amd.backend import HipBackend
    return HipBackend()


def _alg(name):
    return json.load(open(os.path.join(CFG, name)))


def _case(hip, model_name, patch, ch, classes, alg, nb, seed):
    rng = np.random.default_rng(seed)
    built = U.build(model_name, patch, ch, classes, alg, hip)
    sess = built.ctx.session()
    params = U.make_params(model_name, patch, ch, classes, alg, rng)
    U.inject(sess, params)
    x = rng.random((nb, patch, patch, ch)).astype(np.float32)
    onehot = np.eye(classes, dtype=np.float32)[rng.integers(0, classes, nb)]
    masks = U.make_masks(built, nb, rng)
    return built, sess, params, x, onehot, masks


SMALL_H = {"drop_out_ratio": 0.7, "filter_count": 48, "learning_rate": 3e-4, "learning_rate_decay_factor": 0.96,
           "learning_rate_decay_step": 350, "lrelu_alpha": 0.18, "optimizer": "AdamOptimizer", "bn_decay": 0.95,
           "l2regularizer_scale": 1e-5, "spectral_hierarchy_level": 3, "spatial_hierarchy_level": 3,
           "degradation_coeff": 3, "use_residual": True}


@pytest.mark.parametrize("model_name,patch,ch,classes,alg,nb", [
    ("HYPELCNNModel", 5, 11, 4, SMALL_H, 6),                      # tiny, ragged everything
    ("HYPELCNNModel", 7, 9, 4, dict(SMALL_H, filter_count=96), 37),  # fc_0 K=588 -> split-K forward
    ("HYPELCNNModel", 7, 21, 5, SMALL_H, 150),                    # rows per pixel > one 128-row tile
    ("HYPELCNNModel", 3, 7, 3, dict(SMALL_H, use_residual=False, spectral_hierarchy_level=2), 1),  # batch of one
    ("DUALCNNModel", 5, 7, 3, {"drop_out_ratio": 0.7, "lrelu_alpha": 0.18, "filter_count": 32, "hs_lidar_diff": 1,
                               "optimizer": "AdamOptimizer", "learning_rate": 3e-4,
                               "learning_rate_decay_factor": 0.96, "learning_rate_decay_step": 350}, 9),
    # GRSS2018 geometry of BASELINE configs[2] (11x11 patch, 48 HSI + 1 LiDAR bands, 20 classes, all nine kernel
    # sizes 1..9 with their valid-tap tables) at a filter count the float64 oracle finishes in seconds
    ("DUALCNNModel", 11, 49, 20, {"drop_out_ratio": 0.7, "lrelu_alpha": 0.18, "filter_count": 32, "hs_lidar_diff": 1,
                                  "optimizer": "AdamOptimizer", "learning_rate": 3e-4,
                                  "learning_rate_decay_factor": 0.96, "learning_rate_decay_step": 350}, 12),
    ("CONCNNModel", 5, 9, 3, {"drop_out_ratio": 0.5, "filter_count": 6, "optimizer": ["MomentumOptimizer", 0.9],
                              "learning_rate": 1e-3, "learning_rate_decay_factor": 0.01,
                              "learning_rate_decay_step": 33333}, 8),
])
def test_small_models_train_step(hip, model_name, patch, ch, classes, alg, nb):
    built, sess, params, x, onehot, masks = _case(hip, model_name, patch, ch, classes, alg, nb, 11)
    if nb == 1 and model_name == "HYPELCNNModel":
        # batch-norm over a single FC row has zero variance; only the inference tower is meaningful
        li = U.run_eval(built, x)
        ri = OT.forward_backward(model_name, params, x.astype(np.float64), None, classes, alg, False)
        assert np.abs(li - ri["logits"]).max() < 1e-3 * max(1.0, np.abs(ri["logits"]).max())
        return
    ct = U.run_train_step(built, x, onehot, masks)
    U.compare_step(built, ct, params, x, onehot, masks, model_name, classes, alg)


def test_grss2013_hypelcnn_batch64_cfg1(hip):
    """BASELINE configs[0]: GRSS2013 HYPELCNN, 7x7x(144+1), batch 64 -- full model vs oracle."""
    alg = _alg("alg_param_hypelcnn.json")
    built, sess, params, x, onehot, masks = _case(hip, "HYPELCNNModel", 7, 145, 15, alg, 64, 1234)
    ct = U.run_train_step(built, x, onehot, masks)
    ref, err, worst = U.compare_step(built, ct, params, x, onehot, masks, "HYPELCNNModel", 15, alg)
    # class labels bit-exact (training-mode logits)
    got = ct.value(built.y_conv).cpu().numpy()
    assert (got.argmax(1) == ref["logits"].argmax(1)).all()
    # one TF1-Adam step: the first step moves every weight by ~lr*sign(g) (K4), so a looser absolute bound
    # (2*lr) holds even where a kink flip changed a tiny gradient's sign
    trainer_p = {k: v.copy() for k, v in params.items()}
    tr = OT.ClassifierTrainer("HYPELCNNModel", trainer_p, 15, alg)
    tr.train_step(x.astype(np.float64), onehot.astype(np.float64), masks)
    sess.adam_step(built.lr.eval(0))
    n_bad = n_all = 0
    for k, v in tr.params.items():
        gotp = sess.get_variable("nn_core/" + k)
        assert np.abs(gotp - v).max() <= 2.01 * alg["learning_rate"] + 1e-4 * np.abs(v).max(), k
        n_bad += int((np.abs(gotp - v) > 1e-5 * max(1.0, np.abs(v).max())).sum())
        n_all += v.size
    assert n_bad / n_all < 0.02, (n_bad, n_all)
    # inference tower: labels bit-exact, logits within 1e-3 relative to their scale
    li = U.run_eval(built, x)
    p2 = {k: sess.get_variable("nn_core/" + k).astype(np.float64) for k in params}
    ri = OT.forward_backward("HYPELCNNModel", p2, x.astype(np.float64), None, 15, alg, False)
    assert np.abs(li - ri["logits"]).max() < 1e-3 * max(1.0, np.abs(ri["logits"]).max())
    assert (li.argmax(1) == ri["logits"].argmax(1)).all()


def test_grss2013_hypelcnn_batch1024_properties(hip):
    """BASELINE configs[1] size (batch 1024): size-independent properties instead of the (slow) oracle:
    (1) run-to-run bit-exact determinism of a full training step;
    (2) inference on 1024 patches == 16 independent inference runs of 64 (moving statistics decouple samples):
        labels bit-exact, logits equal to fp32 rounding;
    (3) the gradient buffer is exactly linear in the upstream loss scale is NOT assumed; instead the
        flat gradient of batch [A;A] (two copies of a 512 batch) equals that of the same data in swapped order."""
    alg = _alg("alg_param_hypelcnn.json")
    built, sess, params, x, onehot, masks = _case(hip, "HYPELCNNModel", 7, 145, 15, alg, 1024, 99)
    ct = U.run_train_step(built, x, onehot, masks)
    g1 = sess.grads.clone()
    l1 = ct.value(built.y_conv).clone()
    U.inject(sess, params)  # restore moving statistics
    ct = U.run_train_step(built, x, onehot, masks)
    assert torch.equal(g1, sess.grads) and torch.equal(l1, ct.value(built.y_conv)), "training step must be deterministic"
    assert torch.isfinite(sess.grads).all()
    # (2)
    U.inject(sess, params)
    big = U.run_eval(built, x)
    parts = np.concatenate([U.run_eval(built, x[i:i + 64]) for i in range(0, 1024, 64)])
    # split-K factors depend on the batch size, so the fp32 summation order may differ: labels must still be
    # bit-exact, logits equal to fp32 rounding
    assert np.array_equal(big.argmax(1), parts.argmax(1))
    assert np.abs(big - parts).max() <= 2e-5 * max(1.0, np.abs(big).max())
    # (3) permutation equivariance of the batch: swapping the two halves leaves the mean-loss gradient unchanged
    perm = np.concatenate([np.arange(512, 1024), np.arange(0, 512)])
    U.inject(sess, params)
    ct = U.run_train_step(built, x[perm], onehot[perm], {k: v[perm] for k, v in masks.items()})
    rel = float((sess.grads - g1).abs().max() / g1.abs().max())
    # a different summation order may flip 1-2 leaky-ReLU kink decisions among 4e7 activations (see parity_util)
    assert rel < 2e-2, rel
    assert np.array_equal(ct.value(built.y_conv).cpu().numpy().argmax(1), l1.cpu().numpy().argmax(1)[perm])


def test_grss2018_dualcnn_full_size_properties(hip):
    """BASELINE configs[2] model (alg_param_dualcnn.json: filter_count 480, 258 M parameters) at a reduced batch:
    the float64 oracle needs ~1e11 MAC per patch, so the full-size check is by properties -- bit-exact determinism
    of the training step, and inference logits independent of how the batch is cut."""
    alg = _alg("alg_param_dualcnn.json")
    built, sess, params, x, onehot, masks = _case(hip, "DUALCNNModel", 11, 49, 20, alg, 48, 2018)
    ct = U.run_train_step(built, x, onehot, masks)
    g1 = sess.grads.clone()
    l1 = ct.value(built.y_conv).clone()
    assert torch.isfinite(g1).all() and float(g1.abs().max()) > 0
    ct = U.run_train_step(built, x, onehot, masks)
    assert torch.equal(g1, sess.grads) and torch.equal(l1, ct.value(built.y_conv)), "training step must be deterministic"
    big = U.run_eval(built, x)
    parts = np.concatenate([U.run_eval(built, x[i:i + 16]) for i in range(0, 48, 16)])
    assert np.array_equal(big.argmax(1), parts.argmax(1))
    assert np.abs(big - parts).max() <= 2e-5 * max(1.0, np.abs(big).max())


def test_hip_graph_replay_equals_eager(hip):
    built, sess, params, x, onehot, masks = _case(hip, "HYPELCNNModel", 7, 21, 5, SMALL_H, 96, 5)
    ct = U.run_train_step(built, x, onehot, masks)
    g_eager = sess.grads.clone()
    U.inject(sess, params)
    ct.capture()
    U.inject(sess, params)
    ct.forward_backward()
    torch.cuda.synchronize()
    assert torch.equal(g_eager, sess.grads)
    mm = sess.state.clone()
    ct.forward_backward()  # a second replay keeps updating the moving statistics
    torch.cuda.synchronize()
    assert not torch.equal(mm, sess.state)


def test_end_to_end_training_on_gpu(tmp_path):
    """flags -> SyntheticDataLoader -> InMemoryImporter -> create_graph -> run_monitored_session on the real
    backend (HIP-graph replay of the step, device-side metrics): the classifier learns the synthetic scene."""
    from hypelcnn_amd.classify import train_for_classification as T
    from tests.test_training_loop_emu import ALG, _flags
    flags = _flags(tmp_path, 151)
    model = T.get_model_from_name(flags.model_name)
    log_dir = os.path.join(flags.base_log_path, T.get_log_suffix(flags))
    res = T.perform_an_episode(flags, dict(ALG), model, log_dir)
    assert np.isfinite(res.loss) and res.test_accuracy > 0.85 and res.validation_accuracy > 0.85
    assert "model.ckpt-150.npz" in os.listdir(log_dir)
