"""Host-side logic of the product (graph recording, fusion, planning, GEMM tables, hand-derived backward,
optimiser) executed on the numpy emulation of the kernel ABI and compared with the oracle.  CPU only."""
import numpy as np
import pytest

from oracle import train as OT
from tests import parity_util as U
from tests.emu_backend import EmuBackend

ALG_H = {"drop_out_ratio": 0.7, "filter_count": 48, "learning_rate": 3e-4, "learning_rate_decay_factor": 0.96,
         "learning_rate_decay_step": 350, "lrelu_alpha": 0.18, "optimizer": "AdamOptimizer", "bn_decay": 0.95,
         "l2regularizer_scale": 1e-5, "spectral_hierarchy_level": 3, "spatial_hierarchy_level": 3,
         "degradation_coeff": 3, "use_residual": True, "batch_size": 6}
ALG_D = {"drop_out_ratio": 0.7, "lrelu_alpha": 0.18, "filter_count": 32, "hs_lidar_diff": 1,
         "optimizer": "AdamOptimizer", "learning_rate": 3e-4, "learning_rate_decay_factor": 0.96,
         "learning_rate_decay_step": 350}
ALG_C = {"drop_out_ratio": 0.5, "filter_count": 6, "optimizer": ["MomentumOptimizer", 0.9], "learning_rate": 1e-3,
         "learning_rate_decay_factor": 0.01, "learning_rate_decay_step": 33333}


def _case(model_name, patch, ch, classes, alg, nb, seed):
    rng = np.random.default_rng(seed)
    built = U.build(model_name, patch, ch, classes, alg, EmuBackend())
    sess = built.ctx.session()
    params = U.make_params(model_name, patch, ch, classes, alg, rng)
    U.inject(sess, params)
    x = rng.random((nb, patch, patch, ch)).astype(np.float32)
    onehot = np.eye(classes, dtype=np.float32)[rng.integers(0, classes, nb)]
    masks = U.make_masks(built, nb, rng)
    return built, sess, params, x, onehot, masks


@pytest.mark.parametrize("model_name,patch,ch,classes,alg,nb", [
    ("HYPELCNNModel", 5, 11, 4, ALG_H, 6),
    ("HYPELCNNModel", 7, 9, 4, dict(ALG_H, filter_count=96), 5),  # fc_0 has K=588: exercises the split-K path
    ("HYPELCNNModel", 3, 7, 3, dict(ALG_H, use_residual=False, spectral_hierarchy_level=2), 5),
    ("DUALCNNModel", 5, 7, 3, ALG_D, 4),
    ("CONCNNModel", 5, 9, 3, ALG_C, 4),
])
def test_train_step_matches_oracle(model_name, patch, ch, classes, alg, nb):
    built, sess, params, x, onehot, masks = _case(model_name, patch, ch, classes, alg, nb, 11)
    ct = U.run_train_step(built, x, onehot, masks)
    ref, err, worst = U.compare_step(built, ct, params, x, onehot, masks, model_name, classes, alg,
                                     tol_logit=2e-5, tol_grad=2e-4)
    # inference tower shares the (now updated) moving statistics
    p2 = {k: sess.get_variable("nn_core/" + k).astype(np.float64) for k in params}
    li = U.run_eval(built, x)
    ri = OT.forward_backward(model_name, p2, x.astype(np.float64), None, classes, alg, False)
    assert np.abs(li - ri["logits"]).max() < 2e-6 * max(1.0, np.abs(ri["logits"]).max())


def test_three_adam_steps_track_oracle_trainer():
    model_name, patch, ch, classes, nb = "HYPELCNNModel", 5, 11, 4, 6
    built, sess, params, x, onehot, masks = _case(model_name, patch, ch, classes, ALG_H, nb, 5)
    trainer = OT.ClassifierTrainer(model_name, {k: v.copy() for k, v in params.items()}, classes, ALG_H)
    for step in range(3):
        ct = U.run_train_step(built, x, onehot, masks)
        sess.adam_step(built.lr.eval(sess.global_step))
        trainer.train_step(x.astype(np.float64), onehot.astype(np.float64), masks)
    for k, v in trainer.params.items():
        got = sess.get_variable("nn_core/" + k)
        assert np.abs(got - v).max() < 5e-5 * max(1.0, np.abs(v).max()), k
    assert sess.global_step == 3


def test_exact_tap_flop_count_matches_survey():
    """Planned GEMM work for the full GRSS2013 HYPELCNN = SURVEY Appendix B.1's exact-tap MAC count."""
    import json, os
    alg = json.load(open(os.path.join(os.path.dirname(U.cno.__file__), "..", "nnmodel", "modelconfigs",
                                      "alg_param_hypelcnn.json")))
    built = U.build("HYPELCNNModel", 7, 145, 15, alg, EmuBackend(), with_eval=False)
    ct = built.train_step.compiled(2)
    fwd, bwd = ct.flops()
    assert fwd == 2 * 2 * 78579058
    n_params = built.ctx.session().params.numel()
    assert n_params == 8160297


def test_tap_split_forward_matches_oracle(monkeypatch):
    """Heavy branches (k >= 5) of a level have their tap list split into chunks that write partial outputs, summed
    by the strided reduce; force the path at a tiny batch."""
    from hypelcnn_amd import plan
    monkeypatch.setattr(plan, "TAP_SPLIT_MIN_BATCH", 1)
    monkeypatch.setattr(plan, "MAX_TAPS_PER_TILE", 4)
    built, sess, params, x, onehot, masks = _case("HYPELCNNModel", 7, 9, 4, ALG_H, 5, 21)
    ct = U.run_train_step(built, x, onehot, masks)
    assert any(l.tag == "tap-split-reduce" for l in ct.plan.fwd)
    U.compare_step(built, ct, params, x, onehot, masks, "HYPELCNNModel", 4, ALG_H, tol_logit=2e-5, tol_grad=2e-4)
