"""ORACLE (test infrastructure) -- the reference's model graphs restated over oracle/ops.py.

Parity unpinned at the TensorFlow boundary (see oracle/ops.py header).  Every function
cites the reference lines it follows.  Parameters are a dict keyed by the TF variable
names the reference would create under the `nn_core` template
(`<scope>/weights`, `/biases`, `/BatchNorm/{beta,moving_mean,moving_variance}`).
"""
import math

import numpy as np

from . import ops as O
from .host import scale_in_to_out_index


# ----------------------------------------------------------------------------- helpers
class Ctx:
    """Forward context: parameters as Vars, BN bookkeeping, dropout masks."""

    def __init__(self, params, is_training, dropout_masks=None, bn_decay=0.999):
        self.vars = {k: O.Var(v, name=k) for k, v in params.items()}
        self.params = params
        self.is_training = is_training
        self.dropout_masks = dropout_masks or {}
        self.bn_decay = bn_decay
        self.new_moving = {}  # name -> updated moving stat (training only)
        self.trace = {}  # scope -> forward value (for layer-wise parity)
        self.drop_idx = 0
        self.kink_force = {}  # scope -> {flat index: bool} (see ops.leaky_relu)

    def p(self, name):
        return self.vars[name]


def _residual(src, dst):
    """scale_in_to_out(src, dst, axis_no=3) (common/common_nn_ops.py:546-564)."""
    return O.gather_channels(src, scale_in_to_out_index(src.v.shape[-1], dst.v.shape[-1]))


def _bn(ctx, x, scope):
    beta = ctx.p(scope + "/BatchNorm/beta")
    mm_name, mv_name = scope + "/BatchNorm/moving_mean", scope + "/BatchNorm/moving_variance"
    if ctx.is_training:
        y, mean, var, m = O.batch_norm_train(x, beta)
        # fused batch norm feeds the Bessel-corrected variance to the moving average (Appendix A.3)
        unbiased = var * (m / max(m - 1, 1))
        ctx.new_moving[mm_name] = O.moving_average_update(ctx.params[mm_name], mean, ctx.bn_decay)
        ctx.new_moving[mv_name] = O.moving_average_update(ctx.params[mv_name], unbiased, ctx.bn_decay)
        return y
    return O.batch_norm_infer(x, beta, ctx.params[mm_name], ctx.params[mv_name])


def _conv_bn_act(ctx, x, scope, act):
    """tf_slim.conv2d under the HYPELCNN arg_scope: normalizer_fn=batch_norm => no bias,
    BN before the activation (Appendix A.1; nnmodel/HYPELCNNModel.py:40-45)."""
    y = O.conv2d_same(x, ctx.p(scope + "/weights"))
    y = _bn(ctx, y, scope)
    y = _apply_act(act, y, scope)
    ctx.trace[scope] = y.v
    return y


def _apply_act(act, y, scope):
    if act is None:
        return y
    try:
        return act(y, scope=scope)
    except TypeError:
        return act(y)


def _fc_bn_act(ctx, x, scope, act):
    y = O.dense(x, ctx.p(scope + "/weights"))
    y = _bn(ctx, y, scope)
    y = _apply_act(act, y, scope)
    ctx.trace[scope] = y.v
    return y


def _dropout(ctx, x, keep_prob):
    if not ctx.is_training:
        return x
    key = f"dropout_{ctx.drop_idx}"
    ctx.drop_idx += 1
    mask = ctx.dropout_masks.get(key)
    if mask is None:  # keep everything, scaled: deterministic fallback for tests that do not care
        mask = np.ones_like(x.v)
    return O.dropout(x, mask.astype(x.v.dtype))


# ----------------------------------------------------------------------------- HYPELCNN
def hypelcnn_layer_table(patch, channels, class_count, alg):
    """Layer list (scope, kind, k, cin, cout) implied by nnmodel/HYPELCNNModel.py:34-183."""
    f = alg["filter_count"]
    n_spec, n_spat = alg["spectral_hierarchy_level"], alg["spatial_hierarchy_level"]
    layers = []
    c = channels
    for i in range(n_spec):  # :146-164 encoder
        co = f // (2 ** ((n_spec - 1) - i))
        layers.append((f"conv_enc_{i}", "conv", 1, c, co))
        c = co
    for i in range(n_spec):  # decoder
        co = f // (2 ** i)
        layers.append((f"conv_dec_{i}", "conv", 1, c, co))
        c = co
    level_final = c // 2  # :67-68
    for i in range(n_spat):  # :128-143
        lf = level_final // (2 ** i)
        ks = [k for k in range(1, patch + 1) if k % 2 == 1]  # :170-175 odd, square kernels only
        for k in ks:
            layers.append((f"connector_{i}_conv{k}x{k}", "conv", k, c, lf))
        c = lf * len(ks)
        layers.append((f"connector_conv_{i}", "conv", 1, c, c))
    flat = patch * patch * c
    stages = math.floor(math.log(flat / class_count, alg["degradation_coeff"]))  # :117
    e = flat
    for i in range(stages - 1):
        e2 = e // alg["degradation_coeff"]
        layers.append((f"fc_{i}", "fc", 0, e, e2))
        e = e2
    layers.append(("fc_final", "fc", 0, e, class_count))
    img = patch * patch * channels
    layers.append(("image_gen_net_1", "fc", 0, class_count, class_count * 3))
    layers.append(("image_gen_net_2", "fc", 0, class_count * 3, class_count * 9))
    layers.append(("image_gen_net_3", "fc", 0, class_count * 9, class_count * 27))
    layers.append(("image_gen_net_4", "fc", 0, class_count * 27, img))
    return layers


def hypelcnn_init_params(patch, channels, class_count, alg, rng, dtype=np.float32):
    """variance_scaling(scale=2.0): fan_in, truncated normal, stddev=sqrt(2/fan_in)/0.87962566
    (Appendix A.6).  RNG streams are not comparable with TF; parity tests inject these."""
    params = {}
    for scope, kind, k, cin, cout in hypelcnn_layer_table(patch, channels, class_count, alg):
        fan_in = cin * (k * k if kind == "conv" else 1)
        std = math.sqrt(2.0 / fan_in) / 0.87962566103423978
        shape = (k, k, cin, cout) if kind == "conv" else (cin, cout)
        w = rng.standard_normal(shape)
        w = np.clip(w, -2.0, 2.0) * std
        params[scope + "/weights"] = w.astype(dtype)
        params[scope + "/BatchNorm/beta"] = np.zeros(cout, dtype)
        params[scope + "/BatchNorm/moving_mean"] = np.zeros(cout, dtype)
        params[scope + "/BatchNorm/moving_variance"] = np.ones(cout, dtype)
    return params


def hypelcnn_forward(ctx, x, class_count, alg):
    """nnmodel/HYPELCNNModel.py:34-99.  x: Var [N,P,P,C] NHWC.  Returns dict with y_conv,
    image_output (training only), image_original and the four histogram tensors."""
    ctx.bn_decay = alg["bn_decay"]
    alpha = alg["lrelu_alpha"]
    use_res = alg["use_residual"]

    class _LRelu:  # scope-aware so that tests can pin individual kink decisions
        def __call__(self, t, scope=None):
            return O.leaky_relu(t, alpha, ctx.kink_force.get(scope))

    lrelu = _LRelu()
    f = alg["filter_count"]
    n_spec = alg["spectral_hierarchy_level"]

    def spectral(net_in, encoding):  # :146-164
        net = net_in
        for i in range(n_spec):
            nxt = _conv_bn_act(ctx, net, ("conv_enc_" if encoding else "conv_dec_") + str(i), lrelu)
            if use_res:
                nxt = O.add(nxt, _residual(net, nxt))
            net = nxt
        return net

    net0 = x
    net1 = spectral(net0, True)  # :54-56
    if use_res:
        net1 = O.add(net1, _residual(net0, net1))  # :57-58
    net2 = spectral(net1, False)  # :60-62
    if use_res:
        net2 = O.add(net2, _residual(net1, net2))  # :63-64

    patch = net2.v.shape[1]
    level_final = net2.v.shape[3] // 2
    net = net2
    for i in range(alg["spatial_hierarchy_level"]):  # :128-143
        lf = level_final // (2 ** i)
        elems = [_conv_bn_act(ctx, net, f"connector_{i}_conv{k}x{k}", lrelu)
                 for k in range(1, patch + 1) if k % 2 == 1]  # :167-183
        nxt = O.concat(elems, axis=3)
        if use_res:
            nxt = O.add(nxt, _residual(net, nxt))
        conn = _conv_bn_act(ctx, nxt, f"connector_conv_{i}", lrelu)
        if use_res:
            conn = O.add(conn, nxt)  # :139-140 plain add
        net = conn
    net3 = net
    if use_res:
        net3 = O.add(net3, _residual(net2, net3))  # :71-72

    net4 = O.flatten(net3)  # :74
    flat = net4.v.shape[1]
    stages = math.floor(math.log(flat / class_count, alg["degradation_coeff"]))  # :117
    net5 = net4
    for i in range(stages - 1):  # :119-124
        net5 = _fc_bn_act(ctx, net5, f"fc_{i}", lrelu)
        net5 = _dropout(ctx, net5, 1 - alg["drop_out_ratio"])
    net6 = _fc_bn_act(ctx, net5, "fc_final", None)  # :80-81 logits are batch-normed too

    image_out = None
    if ctx.is_training:  # :84-94
        g = _fc_bn_act(ctx, net6, "image_gen_net_1", lrelu)
        g = _fc_bn_act(ctx, g, "image_gen_net_2", lrelu)
        g = _fc_bn_act(ctx, g, "image_gen_net_3", lrelu)
        image_out = _fc_bn_act(ctx, g, "image_gen_net_4", O.sigmoid)
    return {"y_conv": net6, "image_output": image_out, "image_original": net0,
            "net1": net1, "net2": net2, "net3": net3, "net5": net5}


def hypelcnn_loss(out, labels_onehot):
    """nnmodel/HYPELCNNModel.py:101-112 + common/common_nn_ops.py:214:
    mean over the batch of (softmax-CE_i + scalar reconstruction MSE)."""
    ce = O.softmax_xent(out["y_conv"], labels_onehot)
    if out["image_output"] is None:
        return O.reduce_mean(ce), ce
    orig = O.reshape(out["image_original"], (out["image_original"].v.shape[0], -1))
    rec = O.reduce_mean(O.square(O.sub(out["image_output"], orig)))
    per_sample = O.add(ce, rec)
    return O.reduce_mean(per_sample), per_sample


# ----------------------------------------------------------------------------- DUALCNN
def _odd_kernels(size):
    return [k for k in range(1, size + 1) if k % 2 == 1]


def dualcnn_layer_table(patch, channels, class_count, alg):
    """nnmodel/DUALCNNModel.py:11-104: (scope, kind, k, cin, cout), biases everywhere."""
    f = alg["filter_count"]
    d = alg["hs_lidar_diff"]
    hs_patch = patch - 2 * d if patch > 1 else patch
    layers = []
    c = channels - 1
    for li, fc in enumerate([f // 4, f // 2, f, f // 2, f // 4, f // 8, f // 16, f // 32], start=1):
        ks = _odd_kernels(hs_patch)
        for k in ks:
            layers.append((f"level{li}_conv{k}x{k}", "conv", k, c, fc))
        c = fc * len(ks)
        layers.append((f"connector_conv{li}", "conv", 1, c, c))
    hs_flat = hs_patch * hs_patch * c
    c = 1
    for li, fc in enumerate([2, 4, 8], start=1):
        ks = _odd_kernels(patch)
        for k in ks:
            layers.append((f"lidar_level{li}_conv{k}x{k}", "conv", k, c, fc))
        c = fc * len(ks)
        layers.append((f"lidar_connector_conv{li}", "conv", 1, c, c))
    flat = hs_flat + patch * patch * c
    for name, co in (("fc1", class_count * 9), ("fc2", class_count * 6), ("fc3", class_count * 3),
                     ("fc4", class_count)):
        layers.append((name, "fc", 0, flat, co))
        flat = co
    return layers


def xavier_init_params(layer_table, rng, dtype=np.float32):
    """tf_slim default xavier_initializer(): uniform +-sqrt(6/(fan_in+fan_out)); zero biases."""
    params = {}
    for scope, kind, k, cin, cout in layer_table:
        rf = k * k if kind == "conv" else 1
        lim = math.sqrt(6.0 / (cin * rf + cout * rf))
        shape = (k, k, cin, cout) if kind == "conv" else (cin, cout)
        params[scope + "/weights"] = rng.uniform(-lim, lim, size=shape).astype(dtype)
        params[scope + "/biases"] = np.zeros(cout, dtype)
    return params


def _conv_bias_act(ctx, x, scope, act):
    y = O.conv2d_same(x, ctx.p(scope + "/weights"), ctx.p(scope + "/biases"))
    y = act(y) if act is not None else y
    ctx.trace[scope] = y.v
    return y


def _fc_bias_act(ctx, x, scope, act):
    y = O.dense(x, ctx.p(scope + "/weights"), ctx.p(scope + "/biases"))
    y = act(y) if act is not None else y
    ctx.trace[scope] = y.v
    return y


def dualcnn_forward(ctx, x, class_count, alg):
    """nnmodel/DUALCNNModel.py:11-33."""
    alpha = alg["lrelu_alpha"]
    lrelu = lambda t: O.leaky_relu(t, alpha)
    n, ph, pw, c = x.v.shape
    hs = O.slice_(x, (slice(None), slice(None), slice(None), slice(0, c - 1)))  # :20
    lidar = O.slice_(x, (slice(None), slice(None), slice(None), slice(c - 1, c)))
    d = alg["hs_lidar_diff"]
    if ph > 1 or pw > 1:  # :23-26
        hs = O.slice_(hs, (slice(None), slice(d, ph - d), slice(d, pw - d), slice(None)))

    def level(net, fc, name):  # :92-104
        return O.concat([_conv_bias_act(ctx, net, f"{name}_conv{k}x{k}", lrelu)
                         for k in _odd_kernels(net.v.shape[1])], axis=3)

    f = alg["filter_count"]
    net = hs
    for li, fc in enumerate([f // 4, f // 2, f, f // 2, f // 4, f // 8, f // 16, f // 32], start=1):  # :58-85
        net = level(net, fc, f"level{li}")
        net = _conv_bias_act(ctx, net, f"connector_conv{li}", lrelu)
    hs_net = net
    net = lidar
    for li, fc in enumerate([2, 4, 8], start=1):  # :36-43
        net = level(net, fc, f"lidar_level{li}")
        net = _conv_bias_act(ctx, net, f"lidar_connector_conv{li}", lrelu)
    net = O.concat([O.flatten(hs_net), O.flatten(net)], axis=1)  # :31
    keep = alg["drop_out_ratio"]  # :49 keep_prob = drop_out_ratio itself (Appendix A.5)
    net = _dropout(ctx, _fc_bias_act(ctx, net, "fc1", lrelu), keep)
    net = _dropout(ctx, _fc_bias_act(ctx, net, "fc2", lrelu), keep)
    net = _dropout(ctx, _fc_bias_act(ctx, net, "fc3", lrelu), keep)
    net = _fc_bias_act(ctx, net, "fc4", None)
    return {"y_conv": net, "image_output": None, "image_original": None}


# ----------------------------------------------------------------------------- CONCNN
def concnn_layer_table(patch, channels, class_count, alg):
    """nnmodel/CONCNNModel.py:23-64."""
    f0 = alg["filter_count"]
    f1 = f0 * 3
    layers = [("conv0_1x1", "conv", 1, channels, f0), ("conv0_3x3", "conv", 3, channels, f0),
              ("conv0_5x5", "conv", 5, channels, f0)]
    for name in ("conv11", "conv12", "conv13", "conv21", "conv22", "conv31", "conv32", "conv33"):
        layers.append((name, "conv", 1, f1, f1))
    layers.append(("fc", "fc", 0, patch * patch * f1, class_count))
    return layers


def concnn_forward(ctx, x, class_count, alg):
    """nnmodel/CONCNNModel.py:23-64: default tf_slim activation (ReLU), biases, LRN defaults."""
    keep = alg["drop_out_ratio"]
    a = _conv_bias_act(ctx, x, "conv0_1x1", O.relu)
    b = _conv_bias_act(ctx, x, "conv0_3x3", O.relu)
    c = _conv_bias_act(ctx, x, "conv0_5x5", O.relu)
    net0 = O.lrn(O.concat([a, b, c], axis=3))  # :36-37
    net11 = O.lrn(_conv_bias_act(ctx, net0, "conv11", O.relu))  # :40-41
    net12 = _conv_bias_act(ctx, net11, "conv12", O.relu)
    net13 = O.add(_conv_bias_act(ctx, net12, "conv13", O.relu), net11)  # :44
    net21 = _conv_bias_act(ctx, net13, "conv21", O.relu)
    net22 = O.add(_conv_bias_act(ctx, net21, "conv22", O.relu), net13)  # :49
    net31 = _dropout(ctx, _conv_bias_act(ctx, net22, "conv31", O.relu), keep)
    net32 = _dropout(ctx, _conv_bias_act(ctx, net31, "conv32", O.relu), keep)
    net33 = _conv_bias_act(ctx, net32, "conv33", O.relu)
    y = _fc_bias_act(ctx, O.flatten(net33), "fc", None)
    return {"y_conv": y, "image_output": None, "image_original": None}


def plain_xent_loss(out, labels_onehot):
    """DUALCNN / CONCNN get_loss_func (DUALCNNModel.py:87-89, CONCNNModel.py:66-68)."""
    ce = O.softmax_xent(out["y_conv"], labels_onehot)
    return O.reduce_mean(ce), ce


# ----------------------------------------------------------------------------- GAN stacks
def generator_kernel_sizes(bands, only_encoder=False):
    """gan/shadow_data_models.py:57-86: B, B/2, B/4, B/8 (, B/4, B/2, B)."""
    ks = [bands, bands // 2, bands // 4, bands // 8]
    if not only_encoder:
        ks += [bands // 4, bands // 2, bands]
    return ks


def generator_init_params(bands, prefix="", dtype=np.float32):
    """weights_initializer=zeros, biases zeros (shadow_data_models.py:47)."""
    p = {}
    for i, k in enumerate(generator_kernel_sizes(bands), start=1):
        p[f"{prefix}net{i}/weights"] = np.zeros((k, 1, 1), dtype)
        p[f"{prefix}net{i}/biases"] = np.zeros((1,), dtype)
    return p


def generator_forward(ctx, x, only_encoder=False, prefix=""):
    """shadowdata_generator_model (gan/shadow_data_models.py:43-90).  x: Var [N,1,1,B] -> [N,1,1,B]."""
    n = x.v.shape[0]
    b = x.v.shape[3]
    lrelu = lambda t: O.leaky_relu(t, 0.1)

    def c1d(inp, i, act):
        y = O.conv1d_same(inp, ctx.p(f"{prefix}net{i}/weights"), ctx.p(f"{prefix}net{i}/biases"))
        return act(y)

    net0 = O.reshape(x, (n, b, 1))
    net1 = O.add(c1d(net0, 1, lrelu), net0)
    net2 = O.add(O.add(c1d(net1, 2, lrelu), net1), net0)
    net3 = O.add(O.add(c1d(net2, 3, lrelu), net2), net1)
    net4 = O.add(O.add(c1d(net3, 4, lrelu), net3), net2)
    res = net4
    if not only_encoder:
        net5 = O.add(O.add(c1d(net4, 5, lrelu), net4), net3)
        net6 = O.add(O.add(c1d(net5, 6, lrelu), net5), net4)
        res = c1d(net6, 7, O.tanh)
    return O.reshape(res, (n, 1, 1, b))


def discriminator_layer_table(bands):
    return [("fully_connected", bands, bands), ("fully_connected_1", bands, bands),
            ("fully_connected_2", bands, bands // 2)]


def he_fc_init(table, rng, prefix="", dtype=np.float32):
    """tf.compat.v1.initializers.variance_scaling(scale=2.0): fan_in, truncated normal (A.6)."""
    p = {}
    for scope, cin, cout in table:
        std = math.sqrt(2.0 / cin) / 0.87962566103423978
        p[f"{prefix}{scope}/weights"] = (np.clip(rng.standard_normal((cin, cout)), -2, 2) * std).astype(dtype)
        p[f"{prefix}{scope}/biases"] = np.zeros(cout, dtype)
    return p


def discriminator_forward(ctx, x, prefix=""):
    """shadowdata_discriminator_model (gan/shadow_data_models.py:93-123): [N,1,1,B] -> [N,1,1,B/2]."""
    n = x.v.shape[0]
    lrelu = lambda t: O.leaky_relu(t, 0.1)
    net = O.reshape(x, (n, -1))
    net = lrelu(O.dense(net, ctx.p(prefix + "fully_connected/weights"), ctx.p(prefix + "fully_connected/biases")))
    net = lrelu(O.dense(net, ctx.p(prefix + "fully_connected_1/weights"), ctx.p(prefix + "fully_connected_1/biases")))
    net = O.dense(net, ctx.p(prefix + "fully_connected_2/weights"), ctx.p(prefix + "fully_connected_2/biases"))
    return O.reshape(net, (n, 1, 1, -1))


def feature_discriminator_slices(bands, patch_count):
    """shadow_data_models.py:136-141: patch_size = B // patch_count, slices start at 0, ps, 2ps, ...
    (the last one may be ragged)."""
    ps = bands // patch_count
    return [(s, min(s + ps, bands)) for s in range(0, bands, ps)], ps


def feature_discriminator_layer_table(bands, patch_count, embed):
    slices, ps = feature_discriminator_slices(bands, patch_count)
    table = []
    idx = 0
    for (s, e) in slices:
        for cin, cout in ((e - s, ps), (ps, ps // 4), (ps // 4, ps // 2), (ps // 2, embed)):
            table.append(("fully_connected" + ("" if idx == 0 else f"_{idx}"), cin, cout))
            idx += 1
    return table


def feature_discriminator_forward(ctx, x, patch_count, embed, prefix=""):
    """shadowdata_feature_discriminator_model (gan/shadow_data_models.py:126-149): per band-slice
    4-layer MLP (lrelu 0.1 on every layer), l2_normalize over the WHOLE [N,E] tensor, stack."""
    n, b = x.v.shape[0], x.v.shape[3]
    lrelu = lambda t: O.leaky_relu(t, 0.1)
    flat = O.reshape(x, (n, b))
    slices, _ = feature_discriminator_slices(b, patch_count)
    outs, idx = [], 0
    for (s, e) in slices:
        cur = O.slice_(flat, (slice(None), slice(s, e)))
        for _ in range(4):
            scope = prefix + "fully_connected" + ("" if idx == 0 else f"_{idx}")
            cur = lrelu(O.dense(cur, ctx.p(scope + "/weights"), ctx.p(scope + "/biases")))
            idx += 1
        outs.append(O.reshape(O.l2_normalize_global(cur), (n, 1, -1)))
    return O.concat(outs, axis=1)
