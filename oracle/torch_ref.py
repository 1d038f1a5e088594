"""ORACLE (test infrastructure). Independent torch-CPU autograd composition of the same graphs, used to
cross-check oracle/ (a second implementation written against torch.nn.functional, NCHW
convolutions with explicit asymmetric padding).  Not an oracle for the product by itself."""
import math

import torch
import torch.nn.functional as F


def _idx(cin, cout):
    inv = 1 / (cin / cout)
    if float(inv).is_integer():
        return torch.arange(cout) // int(inv)
    return torch.tensor([min(round(o * (cin / cout)), cin - 1) for o in range(cout)])


def conv_same(x, w, b=None):
    # x NHWC, w HWIO
    kh, kw = w.shape[:2]
    pt, pb = (kh - 1) // 2, kh - 1 - (kh - 1) // 2
    pl, pr = (kw - 1) // 2, kw - 1 - (kw - 1) // 2
    xn = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    y = F.conv2d(xn.contiguous(), w.permute(3, 2, 0, 1).contiguous(), b)
    return y.permute(0, 2, 3, 1)


def bn(x, P, scope, training, decay, new_moving, eps=1e-3):
    beta = P[scope + "/BatchNorm/beta"]
    c = x.shape[-1]
    flat = x.reshape(-1, c)
    if training:
        mean = flat.mean(0)
        var = flat.var(0, unbiased=False)
        m = flat.shape[0]
        new_moving[scope + "/BatchNorm/moving_mean"] = (
            P[scope + "/BatchNorm/moving_mean"] * decay + mean.detach() * (1 - decay))
        new_moving[scope + "/BatchNorm/moving_variance"] = (
            P[scope + "/BatchNorm/moving_variance"] * decay + var.detach() * (m / max(m - 1, 1)) * (1 - decay))
    else:
        mean, var = P[scope + "/BatchNorm/moving_mean"], P[scope + "/BatchNorm/moving_variance"]
    return (x - mean) / torch.sqrt(var + eps) + beta


def hypelcnn(P, x, class_count, alg, training, masks=None):
    a = alg["lrelu_alpha"]
    act = lambda t: F.leaky_relu(t, a)
    decay = alg["bn_decay"]
    nm = {}
    res = lambda s, d: s[..., _idx(s.shape[-1], d.shape[-1])]
    cba = lambda t, sc, f=act: (lambda y: f(y) if f else y)(bn(conv_same(t, P[sc + "/weights"]), P, sc, training, decay, nm))
    fba = lambda t, sc, f=act: (lambda y: f(y) if f else y)(bn(t @ P[sc + "/weights"], P, sc, training, decay, nm))
    n_spec = alg["spectral_hierarchy_level"]
    net0 = x
    net = net0
    for i in range(n_spec):
        nx = cba(net, f"conv_enc_{i}")
        net = nx + res(net, nx)
    net1 = net + res(net0, net)
    net = net1
    for i in range(n_spec):
        nx = cba(net, f"conv_dec_{i}")
        net = nx + res(net, nx)
    net2 = net + res(net1, net)
    net = net2
    patch = x.shape[1]
    for i in range(alg["spatial_hierarchy_level"]):
        el = [cba(net, f"connector_{i}_conv{k}x{k}") for k in range(1, patch + 1, 2)]
        nx = torch.cat(el, 3)
        nx = nx + res(net, nx)
        net = cba(nx, f"connector_conv_{i}") + nx
    net3 = net + res(net2, net)
    h = net3.reshape(net3.shape[0], -1)
    stages = math.floor(math.log(h.shape[1] / class_count, alg["degradation_coeff"]))
    for i in range(stages - 1):
        h = fba(h, f"fc_{i}")
        if training and masks is not None and f"dropout_{i}" in masks:
            h = h * masks[f"dropout_{i}"]
    logits = fba(h, "fc_final", None)
    img = None
    if training:
        g = fba(logits, "image_gen_net_1")
        g = fba(g, "image_gen_net_2")
        g = fba(g, "image_gen_net_3")
        img = fba(g, "image_gen_net_4", torch.sigmoid)
    return logits, img, nm


def hypelcnn_loss(logits, img, x, onehot):
    ce = -(onehot * F.log_softmax(logits, -1)).sum(-1)
    if img is None:
        return ce.mean()
    rec = ((img - x.reshape(x.shape[0], -1)) ** 2).mean()
    return (ce + rec).mean()


def dualcnn(P, x, class_count, alg, training, masks=None, trace=None, kink_force=None):
    """trace (dict): receives every layer's PRE-activation by scope.  kink_force {scope: bool tensor}: the branch
    (pre > 0) each leaky-ReLU element takes, pinned by the caller (parity tests pin elements whose fp32
    pre-activation lies within rounding of the kink to the product's own decision, see tests/parity_util.py)."""
    a = alg["lrelu_alpha"]

    def act(t, sc):
        if trace is not None:
            trace[sc] = t.detach()
        if kink_force is not None and sc in kink_force:
            pos = kink_force[sc]
            return torch.where(pos, t, a * t)
        return F.leaky_relu(t, a)

    def cb(t, sc, f=act):
        y = conv_same(t, P[sc + "/weights"], P[sc + "/biases"])
        return f(y, sc) if f else y

    def fb(t, sc, f=act):
        y = t @ P[sc + "/weights"] + P[sc + "/biases"]
        return f(y, sc) if f else y
    c = x.shape[3]
    hs, lidar = x[..., :c - 1], x[..., c - 1:]
    d = alg["hs_lidar_diff"]
    if x.shape[1] > 1 or x.shape[2] > 1:
        hs = hs[:, d:x.shape[1] - d, d:x.shape[2] - d, :]
    f = alg["filter_count"]
    net = hs
    for li in range(1, 9):
        net = torch.cat([cb(net, f"level{li}_conv{k}x{k}") for k in range(1, net.shape[1] + 1, 2)], 3)
        net = cb(net, f"connector_conv{li}")
    hsn = net
    net = lidar
    for li in range(1, 4):
        net = torch.cat([cb(net, f"lidar_level{li}_conv{k}x{k}") for k in range(1, net.shape[1] + 1, 2)], 3)
        net = cb(net, f"lidar_connector_conv{li}")
    net = torch.cat([hsn.reshape(hsn.shape[0], -1), net.reshape(net.shape[0], -1)], 1)
    for i, sc in enumerate(("fc1", "fc2", "fc3")):
        net = fb(net, sc)
        if training and masks is not None and f"dropout_{i}" in masks:
            net = net * masks[f"dropout_{i}"]
    return fb(net, "fc4", None)


def lrn(x, r=5, bias=1.0, alpha=1.0, beta=0.5):
    sq = x * x
    c = x.shape[-1]
    pad = F.pad(sq, (r, r))
    s = sum(pad[..., i:i + c] for i in range(2 * r + 1))
    return x / (bias + alpha * s) ** beta


def concnn(P, x, class_count, alg, training, masks=None):
    cb = lambda t, sc: F.relu(conv_same(t, P[sc + "/weights"], P[sc + "/biases"]))
    n0 = lrn(torch.cat([cb(x, "conv0_1x1"), cb(x, "conv0_3x3"), cb(x, "conv0_5x5")], 3))
    n11 = lrn(cb(n0, "conv11"))
    n13 = cb(cb(n11, "conv12"), "conv13") + n11
    n22 = cb(cb(n13, "conv21"), "conv22") + n13
    n31 = cb(n22, "conv31")
    if training and masks is not None and "dropout_0" in masks:
        n31 = n31 * masks["dropout_0"]
    n32 = cb(n31, "conv32")
    if training and masks is not None and "dropout_1" in masks:
        n32 = n32 * masks["dropout_1"]
    n33 = cb(n32, "conv33")
    return n33.reshape(n33.shape[0], -1) @ P["fc/weights"] + P["fc/biases"]


# ----------------------------------------------------------------------------- GAN stacks (cross-check of oracle/gan.py)
def gen_t(P, x, pre, only_encoder=False):
    n, b = x.shape[0], x.shape[-1]
    a = [x.reshape(n, 1, b)]

    def c1d(inp, i):
        w = P[f"{pre}net{i}/weights"]
        k = w.shape[0]
        pl = (k - 1) // 2
        return F.conv1d(F.pad(inp, (pl, k - 1 - pl)), w.reshape(1, 1, k)) + P[f"{pre}net{i}/biases"]

    last = 4 if only_encoder else 6
    for i in range(1, last + 1):
        h = F.leaky_relu(c1d(a[-1], i), 0.1)
        a.append(h + a[-1] + (a[-2] if i >= 2 else 0))
    out = a[-1] if only_encoder else torch.tanh(c1d(a[-1], 7))
    return out.reshape(n, 1, 1, b)


def dis_t(P, x, pre):
    h = x.reshape(x.shape[0], -1)
    h = F.leaky_relu(h @ P[pre + "fully_connected/weights"] + P[pre + "fully_connected/biases"], 0.1)
    h = F.leaky_relu(h @ P[pre + "fully_connected_1/weights"] + P[pre + "fully_connected_1/biases"], 0.1)
    return h @ P[pre + "fully_connected_2/weights"] + P[pre + "fully_connected_2/biases"]


def feat_t(P, x, pre, patches, embed):
    n, b = x.shape[0], x.shape[-1]
    flat = x.reshape(n, b)
    ps = b // patches
    outs, idx = [], 0
    for s in range(0, b, ps):
        cur = flat[:, s:s + ps]
        for _ in range(4):
            sc = pre + "fully_connected" + ("" if idx == 0 else f"_{idx}")
            cur = F.leaky_relu(cur @ P[sc + "/weights"] + P[sc + "/biases"], 0.1)
            idx += 1
        outs.append((cur / torch.sqrt(torch.clamp((cur * cur).sum(), min=1e-12))).unsqueeze(1))
    return torch.cat(outs, 1)


def nce_t(fg, fr, tau):
    logits = torch.matmul(fg, fr.transpose(1, 2)) / tau
    n, p, _ = logits.shape
    lab = torch.eye(p, dtype=logits.dtype).reshape(1, -1).repeat(n, 1)
    return (-(lab * F.log_softmax(logits.reshape(n, -1), -1)).sum(-1)).mean()
