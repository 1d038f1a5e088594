"""ORACLE (test infrastructure, never shipped, never measured as the product).

CPU restatement, in numpy, of the TensorFlow / tf_slim operator semantics the reference's
hot path is written against.  PARITY UNPINNED at the TensorFlow
boundary: the reference has no tests/golden vectors and tensorflow is not installable
here, so these semantics are taken from tf_slim 1.1.0 / TF 2.9 behaviour and
cross-checked three ways (tests/): (1) the literal-definition C loops in
oracle/conv_ref.c, (2) an independent torch-CPU autograd composition, (3) closed-form
known-answer cases.  The numpy-side host functions of the reference
(channel maps, patch extraction, metrics) ARE pinned by golden vectors captured from the
reference itself (tests/golden/).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

A tiny reverse-mode tape: every op returns a `Var`; `backward(loss)` fills `.g`.
dtype follows the inputs (float64 = ground truth, float32 = CPU baseline timing).
"""
import numpy as np


class Var:
    __slots__ = ("v", "g", "parents", "back", "name")

    def __init__(self, v, parents=(), back=None, name=None):
        self.v = v
        self.g = None
        self.parents = parents
        self.back = back
        self.name = name

    @property
    def shape(self):
        return self.v.shape

    def __add__(self, other):
        return add(self, other)


def const(v):
    return Var(np.asarray(v))


def _acc(var, g):
    if var.g is None:
        var.g = g.copy() if isinstance(g, np.ndarray) else np.asarray(g)
    else:
        var.g = var.g + g


def backward(root, seed=None):
    order, seen = [], set()

    def visit(n):
        stack = [(n, False)]
        while stack:
            node, done = stack.pop()
            if done:
                order.append(node)
                continue
            if id(node) in seen:
                continue
            seen.add(id(node))
            stack.append((node, True))
            for p in node.parents:
                if id(p) not in seen:
                    stack.append((p, False))

    visit(root)
    root.g = np.ones_like(root.v) if seed is None else np.asarray(seed, dtype=root.v.dtype)
    for node in reversed(order):
        if node.back is not None and node.g is not None:
            node.back(node.g)


# --------------------------------------------------------------------------- linear ops
def same_pad(k):
    """TF SAME, stride 1: pad_before = floor((k-1)/2), pad_after = k-1-pad_before (Appendix A.1)."""
    pb = (k - 1) // 2
    return pb, k - 1 - pb


def conv2d_same(x, w, b=None):
    """tf_slim.conv2d core: NHWC x [N,H,W,Ci], HWIO w [kh,kw,Ci,Co], stride 1, SAME zero padding.

    Exact-tap formulation: only (output pixel, tap) pairs that read a real input pixel
    contribute.  Reference call sites: nnmodel/HYPELCNNModel.py:136,157,177;
    nnmodel/DUALCNNModel.py:38-83,99; nnmodel/CONCNNModel.py:33-60.
    """
    xv, wv = x.v, w.v
    n, h, wd, ci = xv.shape
    kh, kw, _, co = wv.shape
    pt, _ = same_pad(kh)
    pl, _ = same_pad(kw)
    out = np.zeros((n, h, wd, co), dtype=xv.dtype)
    taps = []
    for i in range(kh):
        dy = i - pt
        y0, y1 = max(0, -dy), min(h, h - dy)
        if y1 <= y0:
            continue
        for j in range(kw):
            dx = j - pl
            x0, x1 = max(0, -dx), min(wd, wd - dx)
            if x1 <= x0:
                continue
            taps.append((i, j, dy, dx, y0, y1, x0, x1))
            out[:, y0:y1, x0:x1, :] += xv[:, y0 + dy:y1 + dy, x0 + dx:x1 + dx, :] @ wv[i, j]
    if b is not None:
        out += b.v

    def back(g):
        dxv = np.zeros_like(xv)
        dwv = np.zeros_like(wv)
        for (i, j, dy, dx, y0, y1, x0, x1) in taps:
            gs = g[:, y0:y1, x0:x1, :]
            xs = xv[:, y0 + dy:y1 + dy, x0 + dx:x1 + dx, :]
            dxv[:, y0 + dy:y1 + dy, x0 + dx:x1 + dx, :] += gs @ wv[i, j].T
            dwv[i, j] += xs.reshape(-1, ci).T @ gs.reshape(-1, co)
        _acc(x, dxv)
        _acc(w, dwv)
        if b is not None:
            _acc(b, g.reshape(-1, co).sum(0))

    parents = (x, w) if b is None else (x, w, b)
    return Var(out, parents, back)


def dense(x, w, b=None):
    """tf_slim.fully_connected core: [N,in] x [in,out] (+bias)."""
    out = x.v @ w.v
    if b is not None:
        out = out + b.v

    def back(g):
        _acc(x, g @ w.v.T)
        _acc(w, x.v.T @ g)
        if b is not None:
            _acc(b, g.sum(0))

    parents = (x, w) if b is None else (x, w, b)
    return Var(out, parents, back)


def conv1d_same(x, w, b=None):
    """tf_slim.convolution1d, SAME, stride 1: x [N,L,Ci], w [k,Ci,Co] (gan/shadow_data_models.py:62-86).
    Even kernels pad floor((k-1)/2) on the left and the rest on the right (Appendix A.1)."""
    xv, wv = x.v, w.v
    n, L, ci = xv.shape
    k, _, co = wv.shape
    pl, _ = same_pad(k)
    if ci == 1 and co == 1:
        # the generator's single-channel layers: the same sum over taps written as ONE product with the banded Toeplitz
        # matrix T[l + d][l] = w[j], d = j - pl (a [N x L] x [L x L] BLAS call instead of k batched 1x1 products --
        # what lets the float64 oracle run the GAN phases at the batch sizes BASELINE quotes, N = 2048 / 4096)
        T = np.zeros((L, L), dtype=xv.dtype)
        ds = []
        for j in range(k):
            d = j - pl
            l0, l1 = max(0, -d), min(L, L - d)
            if l1 <= l0:
                continue
            ds.append((j, d))
            idx = np.arange(l0, l1)
            T[idx + d, idx] = wv[j, 0, 0]
        x2 = xv[:, :, 0]
        out = (x2 @ T)[:, :, None]
        if b is not None:
            out = out + b.v

        def back1(g):
            g2 = g[:, :, 0]
            _acc(x, (g2 @ T.T)[:, :, None])
            M = x2.T @ g2  # M[i][l] = sum_n x[n][i] g[n][l]; dw[j] = sum_l M[l + d][l]
            dwv = np.zeros_like(wv)
            for (j, d) in ds:
                dwv[j, 0, 0] = np.trace(M, offset=-d)
            _acc(w, dwv)
            if b is not None:
                _acc(b, g.reshape(-1, co).sum(0))

        return Var(out, (x, w) if b is None else (x, w, b), back1)
    out = np.zeros((n, L, co), dtype=xv.dtype)
    taps = []
    for j in range(k):
        d = j - pl
        l0, l1 = max(0, -d), min(L, L - d)
        if l1 <= l0:
            continue
        taps.append((j, d, l0, l1))
        out[:, l0:l1, :] += xv[:, l0 + d:l1 + d, :] @ wv[j]
    if b is not None:
        out += b.v

    def back(g):
        dxv = np.zeros_like(xv)
        dwv = np.zeros_like(wv)
        for (j, d, l0, l1) in taps:
            gs = g[:, l0:l1, :]
            dxv[:, l0 + d:l1 + d, :] += gs @ wv[j].T
            dwv[j] += xv[:, l0 + d:l1 + d, :].reshape(-1, ci).T @ gs.reshape(-1, co)
        _acc(x, dxv)
        _acc(w, dwv)
        if b is not None:
            _acc(b, g.reshape(-1, co).sum(0))

    parents = (x, w) if b is None else (x, w, b)
    return Var(out, parents, back)


# --------------------------------------------------------------------------- batch norm
BN_EPS = 0.001  # tf_slim.batch_norm default epsilon (Appendix A.3)


def batch_norm_train(x, beta, eps=BN_EPS):
    """tf_slim.batch_norm(is_training=True, center=True, scale=False), fused semantics:
    y = (x - mean_B) / sqrt(var_B + eps) + beta, var_B biased over all axes but the last.
    Returns (y, batch_mean, batch_var_biased, count)."""
    xv = x.v
    c = xv.shape[-1]
    flat = xv.reshape(-1, c)
    m = flat.shape[0]
    mean = flat.mean(0)
    var = ((flat - mean) ** 2).mean(0)
    rstd = 1.0 / np.sqrt(var + eps)
    xhat = (flat - mean) * rstd
    out = (xhat + beta.v).reshape(xv.shape).astype(xv.dtype)

    def back(g):
        gf = g.reshape(-1, c)
        _acc(beta, gf.sum(0))
        dx = rstd * (gf - gf.mean(0) - xhat * (gf * xhat).mean(0))
        _acc(x, dx.reshape(xv.shape))

    return Var(out, (x, beta), back), mean, var, m


def batch_norm_infer(x, beta, moving_mean, moving_var, eps=BN_EPS):
    xv = x.v
    rstd = 1.0 / np.sqrt(moving_var + eps)
    out = ((xv - moving_mean) * rstd + beta.v).astype(xv.dtype)

    def back(g):
        _acc(beta, g.reshape(-1, xv.shape[-1]).sum(0))
        _acc(x, g * rstd)

    return Var(out, (x, beta), back)


def moving_average_update(moving, batch_value, decay):
    """assign_moving_average(zero_debias=False): m <- m*decay + value*(1-decay)."""
    return moving * decay + batch_value * (1.0 - decay)


# --------------------------------------------------------------------------- elementwise
def leaky_relu(x, alpha, force=None):
    """gen_nn_ops.leaky_relu: x>0 ? x : alpha*x; gradient uses the same x>0 test.
    `force` ({flat index: bool}) pins the branch of individual elements: parity tests use it to enumerate the
    admissible branch assignments of elements whose pre-activation is within fp32 rounding of the kink."""
    pos = x.v > 0
    if force:
        pos = pos.copy()
        for i, b in force.items():
            pos.reshape(-1)[i] = b
    out = np.where(pos, x.v, x.v * alpha).astype(x.v.dtype)
    return Var(out, (x,), lambda g: _acc(x, np.where(pos, g, g * alpha)))


def relu(x):
    pos = x.v > 0
    return Var(np.where(pos, x.v, 0).astype(x.v.dtype), (x,), lambda g: _acc(x, np.where(pos, g, 0)))


def sigmoid(x):
    s = 1.0 / (1.0 + np.exp(-x.v))
    s = s.astype(x.v.dtype)
    return Var(s, (x,), lambda g: _acc(x, g * s * (1 - s)))


def tanh(x):
    t = np.tanh(x.v)
    return Var(t, (x,), lambda g: _acc(x, g * (1 - t * t)))


def add(a, b):
    out = a.v + b.v

    def back(g):
        _acc(a, _unbroadcast(g, a.v.shape))
        _acc(b, _unbroadcast(g, b.v.shape))

    return Var(out, (a, b), back)


def sub(a, b):
    out = a.v - b.v

    def back(g):
        _acc(a, _unbroadcast(g, a.v.shape))
        _acc(b, -_unbroadcast(g, b.v.shape))

    return Var(out, (a, b), back)


def scale(a, s):
    return Var(a.v * s, (a,), lambda g: _acc(a, g * s))


def _unbroadcast(g, shape):
    g = np.asarray(g)
    while g.ndim > len(shape):
        g = g.sum(0)
    for ax, s in enumerate(shape):
        if s == 1 and g.shape[ax] != 1:
            g = g.sum(ax, keepdims=True)
    return g.reshape(shape)


def gather_channels(x, idx):
    """tf.gather(x, idx, axis=-1) / tf.repeat / identity as one index map
    (common/common_nn_ops.py:546-564)."""
    idx = np.asarray(idx, dtype=np.int64)
    out = x.v[..., idx]

    def back(g):
        dx = np.zeros_like(x.v)
        np.add.at(dx, (Ellipsis, idx), g)
        _acc(x, dx)

    return Var(out, (x,), back)


def concat(vs, axis):
    out = np.concatenate([v.v for v in vs], axis=axis)
    sizes = [v.v.shape[axis] for v in vs]

    def back(g):
        off = 0
        for v, s in zip(vs, sizes):
            sl = [slice(None)] * g.ndim
            sl[axis] = slice(off, off + s)
            _acc(v, g[tuple(sl)])
            off += s

    return Var(out, tuple(vs), back)


def slice_(x, sl):
    out = x.v[sl]

    def back(g):
        dx = np.zeros_like(x.v)
        dx[sl] = g
        _acc(x, dx)

    return Var(out, (x,), back)


def reshape(x, shape):
    out = x.v.reshape(shape)
    return Var(out, (x,), lambda g: _acc(x, g.reshape(x.v.shape)))


def flatten(x):
    """tf_slim.flatten: keep batch, row-major (h, w, c) order."""
    return reshape(x, (x.v.shape[0], -1))


def dropout(x, scaled_mask):
    """tf_slim.dropout(keep_prob, is_training=True) with an explicit mask in {0, 1/keep_prob}
    (TF's random stream is not reproducible; tests inject the mask)."""
    out = x.v * scaled_mask
    return Var(out.astype(x.v.dtype), (x,), lambda g: _acc(x, g * scaled_mask))


def lrn(x, depth_radius=5, bias=1.0, alpha=1.0, beta=0.5):
    """tf.nn.local_response_normalization defaults (nnmodel/CONCNNModel.py:37,41; Appendix A.13)."""
    xv = x.v
    c = xv.shape[-1]
    sq = xv * xv
    cs = np.concatenate([np.zeros(xv.shape[:-1] + (1,), xv.dtype), np.cumsum(sq, -1)], -1)
    lo = np.maximum(np.arange(c) - depth_radius, 0)
    hi = np.minimum(np.arange(c) + depth_radius + 1, c)
    s = bias + alpha * (cs[..., hi] - cs[..., lo])
    p = s ** (-beta)
    out = xv * p

    def back(g):
        # d out_i / d x_j = delta_ij p_i - 2 alpha beta x_i x_j s_i^(-beta-1) [|i-j|<=r]
        t = g * xv * (s ** (-beta - 1))
        ct = np.concatenate([np.zeros(xv.shape[:-1] + (1,), xv.dtype), np.cumsum(t, -1)], -1)
        win = ct[..., hi] - ct[..., lo]
        _acc(x, g * p - 2.0 * alpha * beta * xv * win)

    return Var(out.astype(xv.dtype), (x,), back)


def l2_normalize_global(x, eps=1e-12):
    """tf.math.l2_normalize(x) with axis=None: divide by the norm of the WHOLE tensor
    (gan/shadow_data_models.py:147; Appendix A.16).  x * rsqrt(max(sum(x^2), eps))."""
    ss = float((x.v.astype(np.float64) ** 2).sum())
    d = max(ss, eps)
    inv = d ** -0.5
    out = (x.v * inv).astype(x.v.dtype)

    def back(g):
        if ss > eps:
            dot = float((g * x.v).sum())
            _acc(x, g * inv - x.v * (dot * inv ** 3))
        else:
            _acc(x, g * inv)

    return Var(out, (x,), back)


# --------------------------------------------------------------------------- losses
def softmax_xent(logits, labels):
    """tf.nn.softmax_cross_entropy_with_logits: per row -sum(labels * log_softmax(logits));
    labels need not sum to one (Appendix A.8).  labels: ndarray."""
    z = logits.v
    zmax = z.max(-1, keepdims=True)
    e = np.exp(z - zmax)
    se = e.sum(-1, keepdims=True)
    logsm = z - zmax - np.log(se)
    lab = labels.astype(z.dtype)
    out = -(lab * logsm).sum(-1)
    sm = e / se

    def back(g):
        _acc(logits, g[..., None] * (sm * lab.sum(-1, keepdims=True) - lab))

    return Var(out, (logits,), back)


def reduce_mean(x):
    n = x.v.size
    return Var(np.asarray(x.v.mean(), dtype=x.v.dtype), (x,),
               lambda g: _acc(x, np.full(x.v.shape, g / n, dtype=x.v.dtype)))


def reduce_sum(x):
    return Var(np.asarray(x.v.sum(), dtype=x.v.dtype), (x,),
               lambda g: _acc(x, np.full(x.v.shape, g, dtype=x.v.dtype)))


def square(x):
    return Var(x.v * x.v, (x,), lambda g: _acc(x, 2.0 * x.v * g))


def absolute(x):
    return Var(np.abs(x.v), (x,), lambda g: _acc(x, np.sign(x.v) * g))


def matmul_nt_batched(a, b):
    """[N,P,E] x [N,Q,E]^T -> [N,P,Q] (gan/wrappers/cut_wrapper.py:360-363 tf.matmul(transpose_b=True))."""
    out = np.einsum("npe,nqe->npq", a.v, b.v)

    def back(g):
        _acc(a, np.einsum("npq,nqe->npe", g, b.v))
        _acc(b, np.einsum("npq,npe->nqe", g, a.v))

    return Var(out, (a, b), back)
