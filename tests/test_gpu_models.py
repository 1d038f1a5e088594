"""-m gpu: the parity tests proper.  The product (plugin API -> planner -> HIP kernels through the C-ABI)
against the oracle (float64) on identical seeded inputs, weights and dropout masks.

Tolerances (BASELINE.json north_star): logits within 1e-3 (fp32), class labels bit-exact.
Gradients are compared relative to each tensor's max magnitude: 1e-4 on the small cases, 5e-4 at the full model
sizes (fp32 sums over up to 2e5 products)."""
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
    ref, err, worst = U.compare_step(built, ct, params, x, onehot, masks, model_name, classes, alg, tol_logit=2e-4,
                                     tol_grad=1e-4)
    print(f"\n{model_name} nb={nb}: logits {err:.2e}, worst grad {worst}")


def test_grss2013_hypelcnn_batch64_cfg1(hip):
    """BASELINE configs[0]: GRSS2013 HYPELCNN, 7x7x(144+1), batch 64 -- full model vs oracle."""
    alg = _alg("alg_param_hypelcnn.json")
    built, sess, params, x, onehot, masks = _case(hip, "HYPELCNNModel", 7, 145, 15, alg, 64, 1234)
    ct = U.run_train_step(built, x, onehot, masks)
    ref, err, worst = U.compare_step(built, ct, params, x, onehot, masks, "HYPELCNNModel", 15, alg)
    # class labels bit-exact (training-mode logits)
    got = ct.value(built.y_conv).cpu().numpy()
    assert (got.argmax(1) == ref["logits"].argmax(1)).all()
    # one TF1-Adam step at model level.  The first step moves every weight by ~lr*sign(g) (K4), so the update is
    # discontinuous where a gradient is within fp32 rounding of zero, and a leaky-ReLU kink flip (see parity_util)
    # moves a few gradients by a discrete amount.  The device update is therefore checked (a) exactly -- oracle Adam
    # applied to the DEVICE gradients (pinned to the oracle's by compare_step above) must give the device parameters
    # to fp32 rounding -- and (b) against the oracle trainer's parameters: > 99 % of all elements within 1 % of lr
    g_dev = {k: sess.get_gradient("nn_core/" + k).astype(np.float64) for k in ref["grads"]}
    trainer_p = {k: v.copy() for k, v in params.items()}
    tr = OT.ClassifierTrainer("HYPELCNNModel", trainer_p, 15, alg)
    tr.train_step(x.astype(np.float64), onehot.astype(np.float64), masks)
    lr0 = built.lr.eval(0)
    sess.adam_step(lr0)
    n_close = n_all = 0
    for k, g in ref["grads"].items():
        gotp = sess.get_variable("nn_core/" + k)
        exp = params[k].copy()
        OT.adam_tf1_step(exp, g_dev[k], np.zeros_like(exp), np.zeros_like(exp), lr0, 1)
        assert np.abs(gotp - exp).max() <= 2e-7 * max(1.0, np.abs(exp).max()) + 1e-3 * lr0, k
        assert np.abs(gotp - tr.params[k]).max() <= 2.01 * lr0 + 2e-7 * max(1.0, np.abs(exp).max()), k
        n_close += int((np.abs(gotp - tr.params[k]) <= 1e-2 * lr0 + 2e-7 * max(1.0, np.abs(exp).max())).sum())
        n_all += g.size
    print(f"\nmodel-level Adam vs oracle trainer: {n_all - n_close} of {n_all} elements differ by more than 1 % of lr")
    assert n_close / n_all > 0.99, (n_close, n_all)
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


def test_grss2013_hypelcnn_batch1024_split6_vs_fp32_kernels(hip, monkeypatch):
    """The benchmarked configuration with every eligible product on the split-operand kernels (HYPEL_GEMM_SPLIT=6) against
    the same plan on the fp32 MFMA kernels, same weights / inputs: logits and gradients equal to fp32 rounding (two
    fp32-grade evaluations of the same sums), labels identical, run-to-run bit-exact."""
    from hypelcnn_amd import plan
    alg = _alg("alg_param_hypelcnn.json")
    monkeypatch.setattr(plan, "GEMM_SPLIT", 0)
    built0, sess0, params, x, onehot, masks = _case(hip, "HYPELCNNModel", 7, 145, 15, alg, 1024, 99)
    ct0 = U.run_train_step(built0, x, onehot, masks)
    g0, l0 = sess0.grads.clone(), ct0.value(built0.y_conv).clone()
    monkeypatch.setattr(plan, "GEMM_SPLIT", 6)
    built1, sess1, _, _, _, _ = _case(hip, "HYPELCNNModel", 7, 145, 15, alg, 1024, 99)
    ct1 = U.run_train_step(built1, x, onehot, masks)
    launches = ct1.plan.fwd + ct1.plan.bwd
    split = [l.tag for l in launches if l.name.startswith("seg_gemm") and
             (l.args[3] & 0x100 if l.name == "seg_gemm_multi_f32" else l.args[14] & 0x8000)]
    assert any(t.startswith("fwd:") for t in split) and any(t.startswith("dgrad:") for t in split) and \
        any(t.startswith("wgrad-merged/s") for t in split), split
    g1, l1 = sess1.grads.clone(), ct1.value(built1.y_conv).clone()
    assert float((l1 - l0).abs().max()) <= 2e-5 * max(1.0, float(l0.abs().max()))
    assert torch.equal(l1.argmax(1), l0.argmax(1))
    # a different rounding may flip 1-2 leaky-ReLU kink decisions among 4e7 activations (see parity_util)
    assert float((g1 - g0).abs().max() / g0.abs().max()) < 2e-2
    assert float((g1 - g0).abs().median()) < 1e-6 * float(g0.abs().max())
    U.inject(sess1, params)
    ct1 = U.run_train_step(built1, x, onehot, masks)
    assert torch.equal(sess1.grads, g1) and torch.equal(ct1.value(built1.y_conv), l1)


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


# ---------------------------------------------------------------------------------------------------------------
# Large-batch planner paths against the oracle: the split policies of plan.py switch on at
# nb >= 64 (forward tap / channel-part split incl. biased convolutions, data-gradient segment split) and at
# nb >= 128 on device-filling launches (64-row filter-gradient ranges).
def _tags(ct):
    return [l.tag for l in ct.plan.fwd + ct.plan.bwd]


@pytest.fixture(scope="module")
def dual_full(hip):
    """BASELINE configs[2] model: alg_param_dualcnn.json (filter_count 480, 258 M parameters), 11x11x(48+1), 20
    classes; 512 seeded patches with dropout masks."""
    alg = _alg("alg_param_dualcnn.json")
    built, sess, params, x, onehot, masks = _case(hip, "DUALCNNModel", 11, 49, 20, alg, 512, 2018)
    return alg, built, sess, params, x, onehot, masks


def _run_chunk(built, x, onehot, masks, lo, hi):
    ct = U.run_train_step(built, x[lo:hi], onehot[lo:hi], {k: v[lo:hi] for k, v in masks.items()})
    sess = built.ctx.session()
    return ct, sess.grads.clone(), ct.value(built.y_conv).clone(), ct.loss_value()


def test_grss2018_dualcnn_full_size_batch64_vs_oracle(hip, dual_full):
    """Full-size DUALCNN at nb = 64 against the float64 oracle (torch-CPU composition of oracle/torch_ref.py): logits
    1e-3, labels exact, every gradient tensor.  The plan must contain the split launches whose numbers DESIGN quotes."""
    alg, built, sess, params, x, onehot, masks = dual_full
    ct, g, logits, loss = _run_chunk(built, x, onehot, masks, 0, 64)
    tags = _tags(ct)
    assert "tap-split-reduce" in tags and any(t.endswith("/split") and t.startswith("fwd:") for t in tags)
    assert "dgrad-split-reduce" in tags
    assert any(l.kparts > 1 for l in ct.plan.fwd), "no level forward was cut into channel parts"
    import time
    t0 = time.time()
    ref = U.torch_reference_step("DUALCNNModel", params, x[:64], onehot[:64], {k: v[:64] for k, v in masks.items()},
                                 20, alg, threads=min(64, os.cpu_count() or 8))
    t_ref = time.time() - t0
    # ~6e8 leaky-ReLU pre-activations without batch norm: a handful lie within fp32 rounding of the kink and the
    # fp32 product may take the other branch than float64, which moves a gradient by a discrete amount.  Same
    # protocol as parity_util.compare_step: decisions may differ ONLY where |float64 pre-activation| < 1e-4; those
    # are pinned to the product's own choice and the oracle re-run
    force, n_amb, n_flip = U.product_kink_decisions_biased(built, ct, ref["pre"])
    if n_flip:
        ref = U.torch_reference_step("DUALCNNModel", params, x[:64], onehot[:64],
                                     {k: v[:64] for k, v in masks.items()}, 20, alg,
                                     threads=min(64, os.cpu_count() or 8), kink_force=force)
    print(f"\nleaky-ReLU kinks: {n_amb} ambiguous pre-activations, {n_flip} decided differently in fp32 (pinned)")
    err, worst, errs = U.compare_with_reference(built, ct, ref, tol_logit=1e-3, tol_grad=5e-4)
    assert (logits.cpu().numpy().argmax(1) == ref["logits"].argmax(1)).all()
    print(f"\nDUALCNN full size nb=64 vs fp64 oracle ({t_ref:.0f} s): logits {err:.2e}, worst grad {worst}")


@pytest.mark.parametrize("nb", [128, 512])
def test_grss2018_dualcnn_large_batches_equal_oracle_checked_chunks(hip, dual_full, nb, monkeypatch):
    """nb = 128 (adds the 64-row filter-gradient ranges) and nb = 512 (the per-GPU batch bench.py --workload dualcnn
    and configs[2] run at): DUALCNN has no batch statistics, so the step on nb patches must equal the 64-patch steps
    (the first of which the previous test pins to the oracle) -- logits row for row, gradient = mean of the chunk
    gradients -- up to fp32 summation order.  The comparison needs the SAME arithmetic per sample at every batch size (a
    forward value that differs by one rounding can take the other leaky-ReLU branch): since round 6 the planner chooses
    between the fp32 MFMA and the split-operand kernels per LAYER (a launch is priced at a nominal batch,
    plan.SPLIT_NOMINAL_BATCH), so the default plan is used as it is at 64, 128 and 512 (round 5 had to take the
    batch-dependent FLOP threshold out here)."""
    alg = dual_full[0]
    built, sess, params, x, onehot, masks = _case(hip, "DUALCNNModel", 11, 49, 20, alg, 512, 2018)
    g_sum, logit_parts, loss_sum = None, [], 0.0
    for lo in range(0, nb, 64):
        _, g, lg, ls = _run_chunk(built, x, onehot, masks, lo, lo + 64)
        g_sum = g.double() if g_sum is None else g_sum + g.double()
        logit_parts.append(lg)
        loss_sum += ls
    ct, g, logits, loss = _run_chunk(built, x, onehot, masks, 0, nb)
    tags = _tags(ct)
    assert "wgrad-reduce" in tags and "dgrad-split-reduce" in tags and "tap-split-reduce" in tags
    rows = [S for l in ct.plan.bwd if l.tag == "wgrad-reduce" for S in (l.meta.get("splits") or [l.args[2]])]
    assert max(rows) >= nb // 64, f"no filter gradient was cut into 64-row ranges: {rows}"
    want = (g_sum / (nb // 64)).float()
    scale = float(want.abs().max())
    rel = float((g - want).abs().max()) / scale
    # per variable, relative to that variable's largest gradient
    worst = ("", 0.0)
    for v in sess.trainable:
        sl = slice(v.offset, v.offset + v.size)
        m = float(want[sl].abs().max())
        e = float((g[sl] - want[sl]).abs().max()) / max(m, 1e-12)
        if e > worst[1]:
            worst = (v.name, e)
    lg = torch.cat(logit_parts)
    lerr = float((logits - lg).abs().max())
    print(f"\nDUALCNN nb={nb} vs 64-patch chunks: grads {rel:.2e} of max, worst variable {worst}, logits {lerr:.2e}")
    assert worst[1] < 2e-4, worst
    assert lerr < 1e-4 * max(1.0, float(lg.abs().max()))
    assert torch.equal(logits.argmax(1), lg.argmax(1))
    assert abs(loss - loss_sum / (nb // 64)) < 1e-5 * max(1.0, abs(loss))


def test_grss2013_hypelcnn_batch1024_vs_oracle(hip):
    """BASELINE configs[1] exactly as benchmarked (batch 1024, alg_param_hypelcnn.json): ONE full forward+backward
    against the float64 numpy oracle -- 16-slab filter-gradient splits, channel parts, folded shortcut epilogues at
    392-tile grids.  Batch norm couples all 1024 samples, so there is no cheaper decomposition."""
    alg = _alg("alg_param_hypelcnn.json")
    built, sess, params, x, onehot, masks = _case(hip, "HYPELCNNModel", 7, 145, 15, alg, 1024, 77)
    ct = U.run_train_step(built, x, onehot, masks)
    tags = _tags(ct)
    assert "tap-split-reduce" in tags and "wgrad-reduce" in tags and "splitk-reduce" in tags
    # the default plan (round 6): merged forward of the 30- and 15-filter levels, merged data gradients and merged filter
    # gradients (packed image + scatter) of all three levels
    assert "level-pack" in tags and sum(1 for t in tags if t.startswith("fwd:") and t.endswith("/merged")) == 2 and \
        sum(1 for t in tags if t.startswith("dgrad:") and t.endswith("/merged")) == 3, sorted(set(tags))
    launches = ct.plan.fwd + ct.plan.bwd
    merged_products = [p for l in launches if l.name == "seg_gemm_multi_f32" for p in l.meta["products"] if p.endswith("/merged")]
    assert len(set(merged_products)) == 3 and "level-unpack" in tags, merged_products
    # ... and it is the split-operand plan that is held to the oracle here: the heavy launches of every pass carry the flag
    split = [l.tag for l in launches if l.name.startswith("seg_gemm") and
             (l.args[3] & 0x100 if l.name == "seg_gemm_multi_f32" else l.args[14] & 0x8000)]
    for must in ("fwd:conv_dec_0", "fwd:connector_0_conv1x1", "fwd:connector_1_conv1x1/merged", "dgrad:conv_dec_0",
                 "dgrad:connector_0_conv1x1/merged", "dgrad:connector_1_conv1x1/merged", "wgrad-merged/s128", "wgrad-merged/s64"):
        assert must in split, (must, split)
    ref, err, worst = U.compare_step(built, ct, params, x, onehot, masks, "HYPELCNNModel", 15, alg,
                                     tol_logit=1e-3, tol_grad=5e-4)
    got = ct.value(built.y_conv).cpu().numpy()
    assert (got.argmax(1) == ref["logits"].argmax(1)).all()
    print(f"\nHYPELCNN nb=1024 vs fp64 oracle: logits {err:.2e}, worst grad {worst}")


@pytest.mark.parametrize("nb", [64, 1024])
def test_grss2013_hypelcnn_every_level_pass_merged_vs_oracle(hip, monkeypatch, nb):
    """The merged form of the multi-kernel levels forced for ALL three levels and all three passes (forward on blocks with
    per-tile column counts -- 16x16x4 MFMA for the 15-filter level, the split-operand kernels for the wide ones --, merged
    data-gradient segments, per-offset filter gradients into a packed image + scatter) at the benchmark's shapes, against the
    float64 oracle."""
    from hypelcnn_amd import plan
    monkeypatch.setattr(plan, "MERGE_LEVELS", {"fwd", "dgrad", "wgrad"})
    monkeypatch.setattr(plan, "MERGE_LEVELS_MAX_COUT", 1 << 20)
    monkeypatch.setattr(plan, "MERGE_PASS_MAX_COUT", {"fwd": 1 << 20, "dgrad": 1 << 20, "wgrad": 1 << 20})
    monkeypatch.setattr(plan, "MERGE_WGRAD_MIN_COUT", 1)
    monkeypatch.setattr(plan, "MERGE_WGRAD_MAX_COUT", 1 << 20)
    alg = _alg("alg_param_hypelcnn.json")
    built, sess, params, x, onehot, masks = _case(hip, "HYPELCNNModel", 7, 145, 15, alg, nb, 78)
    ct = U.run_train_step(built, x, onehot, masks)
    tags = _tags(ct)
    assert sum(1 for t in tags if t.startswith("fwd:") and t.endswith("/merged")) == 3, sorted(set(tags))
    assert sum(1 for t in tags if t.startswith("dgrad:") and t.endswith("/merged")) == 3
    assert "level-pack" in tags and "level-unpack" in tags
    merged_products = {p for l in ct.plan.bwd if l.name == "seg_gemm_multi_f32" for p in l.meta["products"] if p.endswith("/merged")}
    assert len(merged_products) == 3, merged_products  # the packed filter gradients of all three levels
    ref, err, worst = U.compare_step(built, ct, params, x, onehot, masks, "HYPELCNNModel", 15, alg,
                                     tol_logit=1e-3, tol_grad=5e-4)
    got = ct.value(built.y_conv).cpu().numpy()
    assert (got.argmax(1) == ref["logits"].argmax(1)).all()
    print(f"\nHYPELCNN nb={nb}, every level merged, vs fp64 oracle: logits {err:.2e}, worst grad {worst}")


def test_avon_hypelcnn_hsi_only_two_classes_vs_oracle(hip):
    """BASELINE configs[4] classifier shape: AVON has no LiDAR (loader/AVONDataLoader.py:32) -> HYPELCNN on 7x7x360
    HSI-only patches, 2 classes (:95-110), full alg_param_hypelcnn.json, batch 64."""
    alg = _alg("alg_param_hypelcnn.json")
    built, sess, params, x, onehot, masks = _case(hip, "HYPELCNNModel", 7, 360, 2, alg, 64, 360)
    ct = U.run_train_step(built, x, onehot, masks)
    ref, err, worst = U.compare_step(built, ct, params, x, onehot, masks, "HYPELCNNModel", 2, alg,
                                     tol_logit=1e-3, tol_grad=5e-4)
    got = ct.value(built.y_conv).cpu().numpy()
    assert (got.argmax(1) == ref["logits"].argmax(1)).all()
    li = U.run_eval(built, x)
    p2 = {k: sess.get_variable("nn_core/" + k).astype(np.float64) for k in params}
    ri = OT.forward_backward("HYPELCNNModel", p2, x.astype(np.float64), None, 2, alg, False)
    assert np.abs(li - ri["logits"]).max() < 1e-3 * max(1.0, np.abs(ri["logits"]).max())
    assert (li.argmax(1) == ri["logits"].argmax(1)).all()
    print(f"\nAVON-shape HYPELCNN nb=64 vs fp64 oracle: logits {err:.2e}, worst grad {worst}")


def test_avon_hypelcnn_per_gpu_batch512_vs_oracle(hip):
    """BASELINE configs[4] classifier at its PER-GPU batch (4096 over 8 GPUs = 512): HYPELCNN on 7x7x360 HSI-only
    patches, 2 classes, full alg_param_hypelcnn.json -- K = 360 first-layer segments and other tile-width hints / split
    counts than the 145-channel model at batch 1024.  One forward + backward against the float64 numpy oracle."""
    alg = _alg("alg_param_hypelcnn.json")
    built, sess, params, x, onehot, masks = _case(hip, "HYPELCNNModel", 7, 360, 2, alg, 512, 512)
    ct = U.run_train_step(built, x, onehot, masks)
    tags = _tags(ct)
    assert "tap-split-reduce" in tags and "wgrad-reduce" in tags and "splitk-reduce" in tags, sorted(set(tags))
    ref, err, worst = U.compare_step(built, ct, params, x, onehot, masks, "HYPELCNNModel", 2, alg,
                                     tol_logit=1e-3, tol_grad=5e-4)
    got = ct.value(built.y_conv).cpu().numpy()
    assert (got.argmax(1) == ref["logits"].argmax(1)).all()
    print(f"\nAVON-shape HYPELCNN nb=512 vs fp64 oracle: logits {err:.2e}, worst grad {worst}")


def test_three_adam_steps_track_oracle_trainer_on_gpu(hip):
    """Three full train steps (forward, backward, TF1 Adam, staircase LR, moving statistics) against
    oracle/train.py::ClassifierTrainer: every variable within 5e-5 after the third update."""
    model_name, patch, ch, classes, nb = "HYPELCNNModel", 5, 11, 4, 6
    alg = dict(SMALL_H)
    built, sess, params, x, onehot, masks = _case(hip, model_name, patch, ch, classes, alg, nb, 5)
    trainer = OT.ClassifierTrainer(model_name, {k: v.copy() for k, v in params.items()}, classes, alg)
    for step in range(3):
        U.run_train_step(built, x, onehot, masks)
        sess.adam_step(built.lr.eval(sess.global_step))
        trainer.train_step(x.astype(np.float64), onehot.astype(np.float64), masks)
    worst = 0.0
    for k, v in trainer.params.items():
        got = sess.get_variable("nn_core/" + k)
        e = np.abs(got - v).max() / max(1.0, np.abs(v).max())
        worst = max(worst, e)
        assert e < 5e-5, (k, e)
    assert sess.global_step == 3
    print(f"\n3 Adam steps on GPU vs oracle trainer: worst variable error {worst:.2e}")


def test_backend_objects_share_the_device_stream_pair(hip):
    """torch's current stream is process-global: a second HipBackend must not move the first one's copies onto a
    stream its kernels are not ordered with (found by the AVON test reading zeros after an end-to-end test had
    created its own backend)."""
    from hypelcnn_amd.backend import HipBackend
    other = HipBackend()
    assert other.stream is hip.stream
    assert torch.cuda.current_stream().cuda_stream == hip.stream.cuda_stream
    built, sess, params, x, onehot, masks = _case(hip, "HYPELCNNModel", 5, 11, 4, SMALL_H, 6, 11)
    ct = U.run_train_step(built, x, onehot, masks)
    U.compare_step(built, ct, params, x, onehot, masks, "HYPELCNNModel", 4, SMALL_H, tol_logit=2e-4, tol_grad=1e-4)


def test_concnn_shipped_config_vs_oracle(hip):
    """CONCNNModel at its SHIPPED configuration (modelconfigs/alg_param_concnn.json: filter_count 128, i.e. three 128-filter
    branches 1x1 / 3x3 / 5x5 -> a 384-channel concat under a radius-5 LRN, eight 384-wide 1x1 convolutions, ReLU, biases,
    two dropouts, dense head; /root/reference/nnmodel/CONCNNModel.py:23-64) on GRSS2013-shaped 5x5x145 patches, 15 classes,
    batch 64, against the float64 oracle: logits, loss, every gradient, then one MomentumOptimizer(0.9) step."""
    alg = _alg("alg_param_concnn.json")
    assert alg["filter_count"] == 128 and alg["optimizer"][0] == "MomentumOptimizer"
    nb = 64
    built, sess, params, x, onehot, masks = _case(hip, "CONCNNModel", 5, 145, 15, alg, nb, 2024)
    ct = U.run_train_step(built, x, onehot, masks)
    names = {l.name for l in ct.plan.fwd + ct.plan.bwd}
    assert {"lrn_fwd", "lrn_bwd"} <= names, sorted(names)
    ref, err, worst = U.compare_step(built, ct, params, x, onehot, masks, "CONCNNModel", 15, alg, tol_logit=1e-3,
                                     tol_grad=1e-4)
    got = ct.value(built.y_conv).cpu().numpy()
    assert (got.argmax(1) == ref["logits"].argmax(1)).all()
    # one TF1 Momentum step (accum = mu * accum + g; var -= lr * accum): exact on the device gradients, and within
    # lr * |g_dev - g_oracle| of the oracle trainer's parameters
    g_dev = {k: sess.get_gradient("nn_core/" + k).astype(np.float64) for k in ref["grads"]}
    lr0 = built.lr.eval(0)
    mu = alg["optimizer"][1]
    sess.momentum_step(lr0, mu)
    assert sess.optimizer_kind == "momentum"
    for k, g in ref["grads"].items():
        gotp = sess.get_variable("nn_core/" + k)
        exp = params[k].copy()
        OT.momentum_tf1_step(exp, g_dev[k], np.zeros_like(exp), lr0, mu)
        assert np.abs(gotp - exp).max() <= 2e-7 * max(1.0, np.abs(exp).max()), k
        exp_o = params[k].copy()
        OT.momentum_tf1_step(exp_o, g, np.zeros_like(exp_o), lr0, mu)
        assert np.abs(gotp - exp_o).max() <= lr0 * 1e-4 * max(np.abs(g).max(), 1e-6) + 2e-7 * max(1.0, np.abs(exp).max()), k
    print(f"\nCONCNN (384-channel LRN, batch {nb}): logits {err:.2e}, worst grad {worst}")


@pytest.mark.gpu
def test_nonfinite_loss_flag_reaches_the_host_through_the_pinned_slot(hip):
    """The GPU twin of tests/test_host_plan_emu.py::test_nonfinite_loss_is_flagged_on_the_device_and_the_update_refused
    (NanTensorHook, monitored_session_runner.py:151): the flag behind the gradient buffer travels to the host through the
    library's own copy kernel writing into a pinned, device-mapped slot (no runtime blit in the step); a NaN input must
    refuse the update on the device and be reported with its step number."""
    import torch
    built, sess, params, x, onehot, masks = _case(hip, "HYPELCNNModel", 5, 11, 4, SMALL_H, 6, 3)
    for _ in range(6):  # more steps than the ring has slots: slots are recycled
        U.run_train_step(built, x, onehot, masks)
        sess.adam_step(1e-3)
    assert sess.nonfinite_step(sync=True) is None
    p1, m1 = sess.params.clone(), sess.slot_m.clone()
    bad = x.copy()
    bad[0, 0, 0, 0] = np.nan
    U.run_train_step(built, bad, onehot, masks)
    sess.adam_step(1e-3)
    torch.cuda.synchronize()
    assert torch.equal(sess.params, p1) and torch.equal(sess.slot_m, m1), "a non-finite step must not update anything"
    assert sess.nonfinite_step(sync=True) == 7
