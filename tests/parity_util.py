"""Shared harness for the classifier parity tests: build a model through the PRODUCT's plugin API on
a given backend (numpy emulation on CPU, HIP on the GPU box), inject oracle-generated parameters,
run one training step / one inference pass, and compare everything with the oracle (float64)."""
import numpy as np
import torch

from hypelcnn_amd.common import common_nn_ops as cno
from oracle import models as OM, train as OT


def to_pixel_major(a):
    """[N, H, W, C] -> flat [P, N, C]; [N, C] unchanged."""
    if a.ndim == 2:
        return a
    n, h, w, c = a.shape
    return a.reshape(n, h * w, c).transpose(1, 0, 2)


class Built:
    pass


def build(model_name, patch, channels, classes, alg, backend, with_eval=True, model=None):
    model = model or cno.get_model_from_name(model_name)
    template = cno.Template("nn_core", model.create_tensor_graph, class_count=classes)
    ctx = cno.GraphContext(template, backend)
    ctx.external_masks = True
    ctx.capture_graphs = False
    images = cno.Placeholder("x", (patch, patch), channels)
    labels = cno.Placeholder("labels", None, classes)
    b = Built()
    b.model, b.template, b.ctx = model, template, ctx
    b.y_conv, b.cross_entropy, b.lr, b.train_step = cno.optimize_nn(
        template, images, labels, "/gpu:0", "training", alg, model.get_loss_func, ctx=ctx)
    b.train_tower = template.towers[0]
    if with_eval:
        outs = template(cno.ModelInputParams(x=cno.Placeholder("x", (patch, patch), channels), y=None,
                                             device_id="/gpu:0", is_training=False), algorithm_params=alg)
        b.eval_out, b.eval_tower = outs, outs.tower
    return b


def inject(sess, params):
    for k, v in params.items():
        sess.set_variable("nn_core/" + k, v)


def make_params(model_name, patch, channels, classes, alg, rng):
    if model_name == "HYPELCNNModel":
        p = OM.hypelcnn_init_params(patch, channels, classes, alg, rng, np.float64)
        for k in p:
            if k.endswith("beta") or k.endswith("moving_mean"):
                p[k] = rng.standard_normal(p[k].shape) * 0.1
            if k.endswith("moving_variance"):
                p[k] = rng.random(p[k].shape) + 0.5
        return {k: v.astype(np.float32).astype(np.float64) for k, v in p.items()}  # fp32-representable
    table = (OM.dualcnn_layer_table if model_name == "DUALCNNModel" else OM.concnn_layer_table)(patch, channels,
                                                                                                classes, alg)
    p = OM.xavier_init_params(table, rng, np.float64)
    for k in p:
        if k.endswith("biases"):
            p[k] = rng.standard_normal(p[k].shape) * 0.05
    return {k: v.astype(np.float32).astype(np.float64) for k, v in p.items()}  # fp32-representable


def make_masks(built, nb, rng):
    """One {0, 1/keep} mask per dropout site, in call order; returns (oracle dict, list of arrays)."""
    masks = {}
    from hypelcnn_amd import graph as G
    for node in built.train_tower.nodes:
        keep = getattr(node, "dropout_keep", None)
        if keep is None:
            continue
        out = node.out
        shape = (nb, out.c) if out.hw is None else (nb, out.hw[0], out.hw[1], out.c)
        m = ((rng.random(shape) < keep) / keep).astype(np.float32)  # fp32-representable, like the device mask
        masks[f"dropout_{node.dropout_index}"] = m.astype(np.float64)
    return masks


def run_train_step(built, x, onehot, masks):
    built.ctx.session()
    ct = built.train_step.compiled(x.shape[0])
    feed(ct, x, onehot, masks)
    ct.forward_backward()
    return ct


def feed(ct, x, onehot, masks):
    """Inputs and external dropout masks of one step into a compiled training tower."""
    dev = ct.input("x").device
    ct.set_input("x", torch.as_tensor(x, dtype=torch.float32).to(dev))
    ct.set_input("labels", torch.as_tensor(onehot, dtype=torch.float32).to(dev))
    for key, m in masks.items():
        idx = int(key.split("_")[1])
        buf = ct.dropout_mask(idx)
        buf.copy_(torch.as_tensor(np.ascontiguousarray(to_pixel_major(m)).reshape(-1), dtype=torch.float32).to(dev))


def compare_step(built, ct, params, x, onehot, masks, model_name, classes, alg, tol_logit=1e-3, tol_grad=2e-3,
                 check=None):
    """Oracle (float64) vs product after one forward+backward.  Returns the oracle result."""
    sess = built.ctx.session()
    ref = OT.forward_backward(model_name, {k: v.copy() for k, v in params.items()}, x.astype(np.float64),
                              onehot.astype(np.float64), classes, alg, True, masks)
    logits = ct.value(built.y_conv).cpu().numpy()
    err = np.abs(logits - ref["logits"]).max()
    assert err < tol_logit, f"logits max abs err {err}"
    assert abs(ct.loss_value() - ref["loss"]) < tol_logit * max(1.0, abs(ref["loss"])), (ct.loss_value(), ref["loss"])
    def grad_errors(r):
        out = {}
        for k, g in r["grads"].items():
            got = sess.get_gradient("nn_core/" + k)
            out[k] = np.abs(got - g).max() / max(np.abs(g).max(), 1e-6)
        return out

    errs = grad_errors(ref)
    worst = max(((k, e) for k, e in errs.items()), key=lambda t: t[1])
    if worst[1] >= tol_grad:
        # Piecewise-linear activations: an fp32 pre-activation within rounding of a leaky-ReLU kink may take the
        # other branch than the fp64 oracle (observed: 1-2 of ~2.5e6 elements at the full GRSS2013 size), which
        # changes the gradient by a discrete amount.  For elements the ORACLE itself flags as ambiguous
        # (|pre-activation| < 1e-4) the product's own branch decision is read back from its device buffers and
        # pinned in the oracle; everywhere else the decisions must already agree.  Then gradients must match.
        # Pinning a decision shifts everything downstream by a rounding-sized amount, which can move another
        # near-zero pre-activation across the kink: repeat against the re-run oracle until the decisions agree
        force_all, cur, history = {}, ref, []
        for _ in range(4):
            force, n_amb, n_flip = product_kink_decisions(built, ct, cur, alg)
            # (an element that is already pinned is reported again: its float64 pre-activation is reconstructed from
            # the oracle's activation output, whose sign the pin has changed -- only NEW elements count)
            new = sum(1 for sc, d in force.items() for i in d if i not in force_all.get(sc, {}))
            history.append(new)
            if new == 0:
                break
            for sc, d in force.items():
                force_all.setdefault(sc, {}).update(d)
            cur = OT.forward_backward(model_name, {k: v.copy() for k, v in params.items()}, x.astype(np.float64),
                                      onehot.astype(np.float64), classes, alg, True, masks, kink_force=force_all)
        assert history[0] > 0, f"gradient {worst[0]} rel err {worst[1]} vs fp64 oracle and no kink flip explains it"
        errs2 = grad_errors(cur)
        best = max(((k, e) for k, e in errs2.items()), key=lambda t: t[1])
        assert best[1] < tol_grad, (f"gradient {worst[0]} rel err {worst[1]} vs fp64 oracle; after pinning kink flips "
                                    f"{history} (of {n_amb} ambiguous): {best}")
        worst = best
    for k, v in ref["new_moving"].items():
        got = sess.get_variable("nn_core/" + k)
        assert np.abs(got - v).max() < 1e-4 * max(1.0, np.abs(v).max()), k
    return ref, err, worst


def product_kink_decisions(built, ct, ref, alg):
    """For every leaky-ReLU layer: recompute the product's pre-activation yh = (Y - mean) * rstd + beta from ITS
    buffers (fp32, same expression as the kernel), compare branch decisions with the oracle's forward, and return
    {scope: {flat NHWC index: bool}} for the oracle-ambiguous elements that differ."""
    from hypelcnn_amd import graph as G
    plan = ct.plan
    sess = built.ctx.session()
    nb = plan.nb
    alpha = alg.get("lrelu_alpha", 0.0)
    force, n_amb, n_flip = {}, 0, 0
    for idx, node in enumerate(built.train_tower.nodes):
        if not isinstance(node, G.LinearNode) or node.act is None or node.act.kind != "lrelu" or not node.has_bn:
            continue
        aux = plan.node_aux[idx]
        c = node.cout
        out = node.out
        y = plan.buffers[aux["y"].buf][: out.npix * nb * c].reshape(out.npix, nb, c)
        mean = plan.buffers[f"mean:{idx}"][:c]
        rstd = plan.buffers[f"rstd:{idx}"][:c]
        beta = sess.params[aux["beta"].offset:aux["beta"].offset + c]
        yh = ((y - mean) * rstd + beta).permute(1, 0, 2).cpu().numpy()  # [nb, P, C]
        off = 0
        for b in node.branches:
            got = yh[:, :, off:off + b.cout].reshape(-1)
            act64 = ref["trace"][b.scope].reshape(-1)
            pre64 = np.where(act64 > 0, act64, act64 / alpha)
            amb = np.abs(pre64) < 1e-4
            differ = (got > 0) != (pre64 > 0)
            assert not (differ & ~amb).any(), f"{b.scope}: branch decision differs outside the ambiguous zone"
            n_amb += int(amb.sum())
            idxs = np.flatnonzero(differ)
            if len(idxs):
                n_flip += len(idxs)
                force[b.scope] = {int(i): bool(got[i] > 0) for i in idxs}
            off += b.cout
    return force, n_amb, n_flip


def run_eval(built, x):
    sess = built.ctx.session()
    ct = sess.compile(built.eval_tower, x.shape[0])
    dev = ct.input("x").device
    ct.set_input("x", torch.as_tensor(x, dtype=torch.float32).to(dev))
    ct.forward()
    return ct.value(built.eval_out.y_conv).cpu().numpy()


def torch_reference_step(model_name, params, x, onehot, masks, classes, alg, threads=None, kink_force=None):
    """The float64 oracle through oracle/torch_ref.py (torch-CPU autograd composition; agrees with oracle/models.py
    to 1e-12, tests/test_oracle_selfcheck.py) -- used where the numpy tape is too slow (full-size DUALCNN: 81 GFLOP
    per patch).  Same result dict as oracle.train.forward_backward (no per-layer trace)."""
    from oracle import torch_ref as TR
    old = torch.get_num_threads()
    if threads:
        torch.set_num_threads(threads)
    try:
        P = {k: torch.tensor(v, dtype=torch.float64,
                             requires_grad=not k.endswith(("moving_mean", "moving_variance"))) for k, v in params.items()}
        xt = torch.tensor(x, dtype=torch.float64)
        oh = torch.tensor(onehot, dtype=torch.float64)
        mk = {k: torch.tensor(v, dtype=torch.float64) for k, v in (masks or {}).items()}
        nm = {}
        if model_name == "HYPELCNNModel":
            logits, img, nm = TR.hypelcnn(P, xt, classes, alg, True, mk)
            loss = TR.hypelcnn_loss(logits, img, xt, oh)
        else:
            trace = {}
            if model_name == "DUALCNNModel":
                logits = TR.dualcnn(P, xt, classes, alg, True, mk, trace=trace, kink_force=kink_force)
            else:
                logits = TR.concnn(P, xt, classes, alg, True, mk)
            loss = (-(oh * torch.log_softmax(logits, -1)).sum(-1)).mean()
        names = [k for k, v in P.items() if v.requires_grad]
        grads = torch.autograd.grad(loss, [P[k] for k in names])
        out = {"logits": logits.detach().numpy(), "loss": float(loss.detach()),
               "grads": {k: g.numpy() for k, g in zip(names, grads)},
               "new_moving": {k: v.detach().numpy() for k, v in nm.items()}}
        if model_name != "HYPELCNNModel":
            out["pre"] = trace  # scope -> float64 pre-activation (torch tensors, NHWC / [N, C])
        return out
    finally:
        torch.set_num_threads(old)


def compare_with_reference(built, ct, ref, tol_logit=1e-3, tol_grad=2e-3):
    """Product after one forward+backward vs a precomputed oracle result dict.  Returns (logit err, worst (name, rel
    err), {name: rel err})."""
    sess = built.ctx.session()
    logits = ct.value(built.y_conv).cpu().numpy()
    err = float(np.abs(logits - ref["logits"]).max())
    assert err < tol_logit * max(1.0, np.abs(ref["logits"]).max()), f"logits max abs err {err}"
    assert abs(ct.loss_value() - ref["loss"]) < tol_logit * max(1.0, abs(ref["loss"])), (ct.loss_value(), ref["loss"])
    errs = {}
    for k, g in ref["grads"].items():
        got = sess.get_gradient("nn_core/" + k)
        errs[k] = float(np.abs(got - g).max() / max(np.abs(g).max(), 1e-6))
    worst = max(errs.items(), key=lambda t: t[1])
    assert worst[1] < tol_grad, f"gradient {worst[0]} rel err {worst[1]}"
    for k, v in ref["new_moving"].items():
        got = sess.get_variable("nn_core/" + k)
        assert np.abs(got - v).max() < 1e-4 * max(1.0, np.abs(v).max()), k
    return err, worst, errs


def product_kink_decisions_biased(built, ct, ref_pre, alpha_zone=1e-4):
    """Leaky-ReLU layers WITHOUT batch norm (DUALCNN: conv/fc + bias -> lrelu): the product's pre-activation is its
    Y buffer (bias already added by the GEMM epilogue or the split reduce).  Returns ({scope: bool tensor of the
    product's branch decisions where they may legitimately differ, oracle's elsewhere}, n_ambiguous, n_flipped);
    asserts that decisions differ only where the float64 pre-activation is within `alpha_zone` of the kink."""
    from hypelcnn_amd import graph as G
    plan = ct.plan
    nb = plan.nb
    force, n_amb, n_flip = {}, 0, 0
    for idx, node in enumerate(built.train_tower.nodes):
        if not isinstance(node, G.LinearNode) or node.act is None or node.act.kind != "lrelu" or node.has_bn:
            continue
        aux = plan.node_aux[idx]
        c = node.cout
        out = node.out
        y = plan.buffers[aux["y"].buf][: out.npix * nb * c].reshape(out.npix, nb, c).permute(1, 0, 2).cpu()
        off = 0
        for b in node.branches:
            got = y[:, :, off:off + b.cout]
            pre64 = ref_pre[b.scope].reshape(nb, out.npix, b.cout)
            amb = pre64.abs() < alpha_zone
            differ = (got > 0) != (pre64 > 0)
            assert not bool((differ & ~amb).any()), f"{b.scope}: branch decision differs outside the ambiguous zone"
            n_amb += int(amb.sum())
            if bool(differ.any()):
                n_flip += int(differ.sum())
                force[b.scope] = torch.where(differ, got > 0, pre64 > 0).reshape(ref_pre[b.scope].shape)
            off += b.cout
    return force, n_amb, n_flip
