"""Pin the oracle's TF-semantics restatement three ways: literal C loops,
an independent torch autograd composition, and closed-form known-answer tests (Appendix C)."""
import math

import numpy as np
import pytest
import torch

from oracle import cref, host, models as M, ops as O, train as T
from oracle import torch_ref as R

ALG_SMALL = {"drop_out_ratio": 0.7, "filter_count": 48, "learning_rate": 3e-4, "learning_rate_decay_factor": 0.96,
             "learning_rate_decay_step": 350, "lrelu_alpha": 0.18, "optimizer": "AdamOptimizer", "bn_decay": 0.95,
             "l2regularizer_scale": 1e-5, "spectral_hierarchy_level": 3, "spatial_hierarchy_level": 3,
             "degradation_coeff": 3, "use_residual": True}


@pytest.mark.parametrize("k,ci,co,hw", [(1, 5, 7, 3), (3, 4, 6, 5), (5, 3, 2, 7), (7, 6, 5, 7), (9, 2, 3, 9), (11, 1, 2, 11)])
def test_conv_numpy_vs_literal_c(k, ci, co, hw):
    rng = np.random.RandomState(k)
    x = rng.randn(3, hw, hw, ci).astype(np.float32)
    w = rng.randn(k, k, ci, co).astype(np.float32)
    b = rng.randn(co).astype(np.float32)
    xv, wv, bv = O.Var(x.astype(np.float64)), O.Var(w.astype(np.float64)), O.Var(b.astype(np.float64))
    y = O.conv2d_same(xv, wv, bv)
    np.testing.assert_allclose(cref.conv2d_same_fwd(x, w, b), y.v, rtol=2e-5, atol=2e-5)
    g = rng.randn(*y.v.shape).astype(np.float32)
    O.backward(y, g.astype(np.float64))
    np.testing.assert_allclose(cref.conv2d_same_bwd_input(g, w, ci), xv.g, rtol=2e-5, atol=2e-5)
    dw, db = cref.conv2d_same_bwd_filter(x, g, k, k, True)
    np.testing.assert_allclose(dw, wv.g, rtol=2e-5, atol=5e-5)
    np.testing.assert_allclose(db, bv.g, rtol=2e-5, atol=2e-5)


def test_same_padding_tap_counts():
    # K9 / Appendix B.1 footnote: valid (pixel, tap) pairs per axis
    def count(n, k):
        pb = (k - 1) // 2
        return sum(1 for o in range(n) for j in range(k) if 0 <= o + j - pb < n)
    assert [count(7, k) for k in (1, 3, 5, 7)] == [7, 19, 29, 37]
    assert [count(9, k) for k in (1, 3, 5, 7, 9)] == [9, 25, 39, 51, 61]
    assert [count(11, k) for k in (1, 3, 5, 7, 9, 11)] == [11, 31, 49, 65, 79, 91]


@pytest.mark.parametrize("k", [1, 2, 3, 4, 8, 9])
def test_conv1d_even_kernel_padding(k):
    rng = np.random.RandomState(k)
    x = rng.randn(2, 9, 1).astype(np.float32)
    w = rng.randn(k, 1, 1).astype(np.float32)
    y = O.conv1d_same(O.Var(x.astype(np.float64)), O.Var(w.astype(np.float64)))
    np.testing.assert_allclose(cref.conv1d_same_fwd(x, w), y.v, rtol=1e-5, atol=1e-5)
    # explicit: pad_left = (k-1)//2
    pl = (k - 1) // 2
    xp = np.pad(x[:, :, 0].astype(np.float64), ((0, 0), (pl, k - 1 - pl)))
    ref = np.stack([sum(xp[:, o + j] * w[j, 0, 0] for j in range(k)) for o in range(9)], 1)
    np.testing.assert_allclose(ref, y.v[:, :, 0], rtol=1e-12, atol=1e-12)


def test_lrn_vs_c_and_torch():
    rng = np.random.RandomState(0)
    x = rng.randn(4, 3, 3, 17).astype(np.float32)
    xv = O.Var(x.astype(np.float64))
    y = O.lrn(xv)
    np.testing.assert_allclose(cref.lrn_fwd(x), y.v, rtol=1e-5, atol=1e-6)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    yt = R.lrn(xt)
    g = rng.randn(*x.shape)
    yt.backward(torch.tensor(g))
    O.backward(y, g)
    np.testing.assert_allclose(yt.detach().numpy(), y.v, rtol=1e-12)
    np.testing.assert_allclose(xt.grad.numpy(), xv.g, rtol=1e-9, atol=1e-12)


def _to_t(P):
    return {k: torch.tensor(v, dtype=torch.float64, requires_grad=not k.endswith(("moving_mean", "moving_variance")))
            for k, v in P.items()}


def test_hypelcnn_oracle_vs_torch():
    rng = np.random.RandomState(7)
    patch, ch, classes, n = 5, 11, 4, 6
    P = M.hypelcnn_init_params(patch, ch, classes, ALG_SMALL, rng, np.float64)
    for k in P:  # non-trivial betas / moving stats
        if k.endswith("beta"):
            P[k] = rng.randn(*P[k].shape) * 0.1
        if k.endswith("moving_mean"):
            P[k] = rng.randn(*P[k].shape) * 0.1
        if k.endswith("moving_variance"):
            P[k] = rng.rand(*P[k].shape) + 0.5
    x = rng.rand(n, patch, patch, ch)
    lab = np.eye(classes)[rng.randint(0, classes, n)]
    flat = patch * patch * (ALG_SMALL["filter_count"] // 4 // 2 // 4 * 3)
    table = [l for l in M.hypelcnn_layer_table(patch, ch, classes, ALG_SMALL) if l[0].startswith("fc_") and l[0] != "fc_final"]
    masks = {f"dropout_{i}": (rng.rand(n, l[4]) < 0.3) / 0.3 for i, l in enumerate(table)}
    r = T.forward_backward("HYPELCNNModel", P, x, lab, classes, ALG_SMALL, True, masks)
    Pt = _to_t(P)
    xt = torch.tensor(x)
    mt = {k: torch.tensor(v) for k, v in masks.items()}
    logits, img, nm = R.hypelcnn(Pt, xt, classes, ALG_SMALL, True, mt)
    loss = R.hypelcnn_loss(logits, img, xt, torch.tensor(lab))
    loss.backward()
    np.testing.assert_allclose(logits.detach().numpy(), r["logits"], rtol=1e-9, atol=1e-10)
    assert abs(float(loss) - r["loss"]) < 1e-10
    for k, g in r["grads"].items():
        np.testing.assert_allclose(Pt[k].grad.numpy(), g, rtol=1e-7, atol=1e-10, err_msg=k)
    for k, v in r["new_moving"].items():
        np.testing.assert_allclose(nm[k].numpy(), v, rtol=1e-10, atol=1e-12, err_msg=k)
    # inference tower
    ri = T.forward_backward("HYPELCNNModel", P, x, None, classes, ALG_SMALL, False)
    li, _, _ = R.hypelcnn(Pt, xt, classes, ALG_SMALL, False)
    np.testing.assert_allclose(li.detach().numpy(), ri["logits"], rtol=1e-9, atol=1e-10)


def test_dualcnn_oracle_vs_torch():
    rng = np.random.RandomState(3)
    alg = {"drop_out_ratio": 0.7, "lrelu_alpha": 0.18, "filter_count": 32, "hs_lidar_diff": 1,
           "optimizer": "AdamOptimizer", "learning_rate": 3e-4, "learning_rate_decay_factor": 0.96,
           "learning_rate_decay_step": 350}
    patch, ch, classes, n = 5, 7, 3, 4
    P = M.xavier_init_params(M.dualcnn_layer_table(patch, ch, classes, alg), rng, np.float64)
    for k in P:
        if k.endswith("biases"):
            P[k] = rng.randn(*P[k].shape) * 0.1
    x = rng.rand(n, patch, patch, ch)
    lab = np.eye(classes)[rng.randint(0, classes, n)]
    masks = {f"dropout_{i}": (rng.rand(n, classes * m) < 0.7) / 0.7 for i, m in enumerate((9, 6, 3))}
    r = T.forward_backward("DUALCNNModel", P, x, lab, classes, alg, True, masks)
    Pt = _to_t(P)
    logits = R.dualcnn(Pt, torch.tensor(x), classes, alg, True, {k: torch.tensor(v) for k, v in masks.items()})
    loss = (-(torch.tensor(lab) * torch.log_softmax(logits, -1)).sum(-1)).mean()
    loss.backward()
    np.testing.assert_allclose(logits.detach().numpy(), r["logits"], rtol=1e-9, atol=1e-10)
    for k, g in r["grads"].items():
        np.testing.assert_allclose(Pt[k].grad.numpy(), g, rtol=1e-7, atol=1e-11, err_msg=k)


def test_concnn_oracle_vs_torch():
    rng = np.random.RandomState(5)
    alg = {"drop_out_ratio": 0.5, "filter_count": 6, "optimizer": ["MomentumOptimizer", 0.9], "learning_rate": 1e-3,
           "learning_rate_decay_factor": 0.01, "learning_rate_decay_step": 33333}
    patch, ch, classes, n = 5, 9, 3, 4
    P = M.xavier_init_params(M.concnn_layer_table(patch, ch, classes, alg), rng, np.float64)
    x = rng.rand(n, patch, patch, ch)
    lab = np.eye(classes)[rng.randint(0, classes, n)]
    masks = {f"dropout_{i}": (rng.rand(n, patch, patch, 18) < 0.5) / 0.5 for i in range(2)}
    r = T.forward_backward("CONCNNModel", P, x, lab, classes, alg, True, masks)
    Pt = _to_t(P)
    logits = R.concnn(Pt, torch.tensor(x), classes, alg, True, {k: torch.tensor(v) for k, v in masks.items()})
    loss = (-(torch.tensor(lab) * torch.log_softmax(logits, -1)).sum(-1)).mean()
    loss.backward()
    np.testing.assert_allclose(logits.detach().numpy(), r["logits"], rtol=1e-9, atol=1e-10)
    for k, g in r["grads"].items():
        np.testing.assert_allclose(Pt[k].grad.numpy(), g, rtol=1e-7, atol=1e-11, err_msg=k)


# ------------------------------------------------------------------ closed-form KATs (Appendix C)
def test_K1_generator_zero_weights_fibonacci():
    b = 16
    P = M.generator_init_params(b, dtype=np.float64)
    x = np.random.RandomState(0).rand(3, 1, 1, b)
    ctx = M.Ctx(P, True)
    enc = M.generator_forward(ctx, O.Var(x), only_encoder=True)
    np.testing.assert_allclose(enc.v, 5 * x)  # n1=x, n2=2x, n3=3x, n4=5x
    full = M.generator_forward(M.Ctx(P, True), O.Var(x))
    assert np.all(full.v == 0)  # tanh(0)


def test_K3_bn_constant_tensor():
    x = O.Var(np.full((8, 3, 3, 4), 2.5))
    beta = O.Var(np.array([0.1, -0.2, 0.3, 0.0]))
    y, mean, var, m = O.batch_norm_train(x, beta)
    np.testing.assert_allclose(y.v, np.broadcast_to(beta.v, y.v.shape), atol=1e-12)
    assert np.allclose(O.moving_average_update(np.zeros(4), mean, 0.95), 0.05 * 2.5)
    assert np.allclose(O.moving_average_update(np.ones(4), var * m / (m - 1), 0.95), 0.95)


def test_K4_adam_first_step():
    g = np.array([0.3, -2.0, 1e-3])
    p, m, v = np.zeros(3), np.zeros(3), np.zeros(3)
    T.adam_tf1_step(p, g, m, v, 3e-4, 1)
    np.testing.assert_allclose(p, -3e-4 * g / (np.abs(g) + 1e-8 / math.sqrt(1 - 0.999)), rtol=1e-9)


def test_K6_softmax_xent_and_nce():
    z = O.Var(np.zeros((5, 15)))
    np.testing.assert_allclose(O.softmax_xent(z, np.eye(15)[:5]).v, math.log(15))
    p = 6
    logits = O.Var(np.full((2, p * p), 0.37))
    lab = np.tile(np.eye(p).reshape(1, -1), (2, 1))
    np.testing.assert_allclose(O.softmax_xent(logits, lab).v, p * math.log(p * p))


def test_K8_identity_conv_with_residual():
    rng = np.random.RandomState(1)
    x = rng.randn(4, 3, 3, 6)
    w = np.eye(6).reshape(1, 1, 6, 6)
    xv = O.Var(x)
    y, _, _, _ = O.batch_norm_train(O.conv2d_same(xv, O.Var(w)), O.Var(np.zeros(6)))
    out = O.add(O.leaky_relu(y, 0.18), O.gather_channels(xv, host.scale_in_to_out_index(6, 6)))
    f = x.reshape(-1, 6)
    xh = (f - f.mean(0)) / np.sqrt(f.var(0) + 1e-3)
    np.testing.assert_allclose(out.v.reshape(-1, 6), np.where(xh > 0, xh, 0.18 * xh) + f, rtol=1e-12)


def test_K10_layer_arithmetic():
    alg = dict(ALG_SMALL, filter_count=480)
    t = M.hypelcnn_layer_table(7, 145, 15, alg)
    d = {s: (ci, co) for s, _, _, ci, co in t}
    assert d["conv_enc_0"] == (145, 120) and d["conv_enc_2"] == (240, 480) and d["conv_dec_2"] == (240, 120)
    assert d["connector_0_conv7x7"] == (120, 60) and d["connector_1_conv1x1"] == (240, 30)
    assert d["connector_conv_2"] == (60, 60)
    assert d["fc_0"] == (2940, 980) and d["fc_1"] == (980, 326) and d["fc_2"] == (326, 108)
    assert d["fc_final"] == (108, 15) and d["image_gen_net_4"] == (405, 7105)
    n_params = sum(ci * co * (max(k, 1) ** 2) + co for _, _, k, ci, co in t)
    assert n_params == 8160297  # SURVEY Appendix B.1 (weights + beta)
    td = M.dualcnn_layer_table(11, 49, 20, {"filter_count": 480, "hs_lidar_diff": 1})
    assert sum(ci * co * (max(k, 1) ** 2) + co for _, _, k, ci, co in td) == 258079549  # Appendix B.3


def test_feature_discriminator_ragged_slices():
    sl, ps = M.feature_discriminator_slices(64, 6)
    assert ps == 10 and len(sl) == 7 and sl[-1] == (60, 64)
    t = M.feature_discriminator_layer_table(64, 6, 2)
    assert sum(ci * co + co for _, ci, co in t) == 1053


# ---------------------------------------------------------------------------------------------------------------
# Third-party pins.  TensorFlow cannot run here, so the three TF conventions that SURVEY Appendix A states in prose are
# additionally checked against PyTorch's OWN operators (not the torch restatement in oracle/torch_ref.py): an
# independent implementation of the same published semantics.

@pytest.mark.parametrize("k", [2, 4, 6, 3])
def test_even_kernel_same_padding_matches_torch_conv2d_same(k):
    """TF SAME, stride 1: pad_before = (k-1)//2, the extra pixel of an even kernel goes to the bottom / right.  PyTorch's
    conv2d(padding='same') documents the same rule (total = k-1, left = total//2, the remainder right)."""
    rng = np.random.RandomState(10 + k)
    x = rng.randn(2, 6, 7, 3)
    w = rng.randn(k, k, 3, 4)
    b = rng.randn(4)
    xv, wv, bv = O.Var(x), O.Var(w), O.Var(b)
    y = O.conv2d_same(xv, wv, bv)
    xt = torch.tensor(x).permute(0, 3, 1, 2).requires_grad_()
    wt = torch.tensor(w).permute(3, 2, 0, 1).requires_grad_()
    bt = torch.tensor(b, requires_grad=True)
    yt = torch.nn.functional.conv2d(xt, wt, bt, padding="same")
    np.testing.assert_allclose(y.v, yt.permute(0, 2, 3, 1).detach().numpy(), rtol=1e-12, atol=1e-12)
    seed = rng.randn(*y.v.shape)
    O.backward(y, seed)
    yt.backward(torch.tensor(seed).permute(0, 3, 1, 2))
    np.testing.assert_allclose(xv.g, xt.grad.permute(0, 2, 3, 1).numpy(), rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(wv.g, wt.grad.permute(2, 3, 1, 0).numpy(), rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(bv.g, bt.grad.numpy(), rtol=1e-11, atol=1e-11)


def test_batch_norm_moments_and_bessel_correction_match_torch_batch_norm():
    """Fused batch norm: normalise with the BIASED batch variance, feed the moving variance the UNBIASED one
    (N/(N-1)), moving <- moving*decay + batch*(1-decay).  torch.nn.functional.batch_norm(training=True) implements the
    same three conventions with momentum = 1 - decay."""
    rng = np.random.RandomState(3)
    x = rng.randn(5, 3, 3, 6) * 2 + 1
    beta = rng.randn(6)
    decay, eps = 0.95, 1e-3
    mm0, mv0 = rng.randn(6), rng.rand(6) + 0.5
    xv, bv = O.Var(x), O.Var(beta)
    y, mean, var, m = O.batch_norm_train(xv, bv, eps=eps)
    mm1 = O.moving_average_update(mm0, mean, decay)
    mv1 = O.moving_average_update(mv0, var * m / (m - 1), decay)
    xt = torch.tensor(x).permute(0, 3, 1, 2).requires_grad_()
    bt = torch.tensor(beta, requires_grad=True)
    rm, rv = torch.tensor(mm0.copy()), torch.tensor(mv0.copy())
    yt = torch.nn.functional.batch_norm(xt, rm, rv, weight=None, bias=bt, training=True, momentum=1 - decay, eps=eps)
    np.testing.assert_allclose(y.v, yt.permute(0, 2, 3, 1).detach().numpy(), rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(mm1, rm.numpy(), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(mv1, rv.numpy(), rtol=1e-12, atol=1e-12)
    seed = rng.randn(*x.shape)
    O.backward(y, seed)
    yt.backward(torch.tensor(seed).permute(0, 3, 1, 2))
    np.testing.assert_allclose(xv.g, xt.grad.permute(0, 2, 3, 1).numpy(), rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(bv.g, bt.grad.numpy(), rtol=1e-11, atol=1e-11)
    # inference mode uses the moving statistics as they are
    yi = O.batch_norm_infer(O.Var(x), O.Var(beta), mm1, mv1, eps=eps)
    yti = torch.nn.functional.batch_norm(torch.tensor(x).permute(0, 3, 1, 2), rm, rv, None, torch.tensor(beta), False,
                                         0.0, eps)
    np.testing.assert_allclose(yi.v, yti.permute(0, 2, 3, 1).numpy(), rtol=1e-11, atol=1e-11)


def test_tf1_adam_epsilon_placement_against_torch_adam():
    """TF1 Adam (Kingma & Ba section 2's "epsilon hat" form, the docstring of tf.compat.v1.train.AdamOptimizer):
    theta -= lr * sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps).  torch.optim.Adam puts eps next to the bias-CORRECTED
    sqrt: theta -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps') -- the same update iff eps' = eps / sqrt(1-b2^t).
    Driving torch's own optimiser with that per-step eps must reproduce the oracle's trajectory; with the constant
    eps (the naive port) it must not, for gradients as small as eps."""
    rng = np.random.RandomState(8)
    g_seq = [rng.randn(7) * 10.0 ** rng.uniform(-9, 0, 7) for _ in range(6)]
    lr, b1, b2, eps = 3e-4, 0.9, 0.999, 1e-8
    p, m, v = rng.randn(7), np.zeros(7), np.zeros(7)
    pt = torch.tensor(p.copy(), requires_grad=True)
    pn = torch.tensor(p.copy(), requires_grad=True)
    opt = torch.optim.Adam([pt], lr=lr, betas=(b1, b2), eps=eps)
    naive = torch.optim.Adam([pn], lr=lr, betas=(b1, b2), eps=eps)
    for t, g in enumerate(g_seq, start=1):
        T.adam_tf1_step(p, g, m, v, lr, t, b1, b2, eps)
        opt.param_groups[0]["eps"] = eps / math.sqrt(1 - b2 ** t)
        pt.grad = torch.tensor(g.copy())
        opt.step()
        pn.grad = torch.tensor(g.copy())
        naive.step()
    np.testing.assert_allclose(p, pt.detach().numpy(), rtol=1e-12, atol=1e-15)
    assert np.abs(p - pn.detach().numpy()).max() > 1e-6  # the placement is observable


def test_softmax_xent_and_lrelu_match_torch_operators():
    rng = np.random.RandomState(4)
    z = rng.randn(9, 15) * 3
    lab = np.eye(15)[rng.randint(0, 15, 9)]
    zv = O.Var(z)
    loss = O.softmax_xent(zv, lab)
    zt = torch.tensor(z, requires_grad=True)
    lt = torch.nn.functional.cross_entropy(zt, torch.tensor(lab), reduction="none")
    np.testing.assert_allclose(loss.v, lt.detach().numpy(), rtol=1e-12, atol=1e-12)
    O.backward(O.reduce_mean(loss))
    lt.mean().backward()
    np.testing.assert_allclose(zv.g, zt.grad.numpy(), rtol=1e-11, atol=1e-13)
    x = rng.randn(50)
    xv = O.Var(x)
    y = O.leaky_relu(xv, 0.18)
    np.testing.assert_allclose(y.v, torch.nn.functional.leaky_relu(torch.tensor(x), 0.18).numpy(), rtol=0, atol=0)
