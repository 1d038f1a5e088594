"""Deferred symbolic graph -- the build's counterpart of the TF1 graph the reference assembles.

The reference's models are written against tf_slim in graph mode: `create_tensor_graph` only
DESCRIBES the network (nnmodel/NNModel.py:10-12); a session executes it later.  This module
keeps that contract: layer calls record coarse nodes on a `Tower`; `hypelcnn_amd.plan` lowers a
tower to a flat list of HIP kernel launches (forward, and a hand-derived backward).

Fusion happens while the graph is being recorded (peephole, no pattern-matching pass):
  * conv2d / fully_connected with a normaliser and an activation is ONE LinearNode
    (GEMM -> batch-norm -> activation), as tf_slim orders them;
  * concat(axis=3) of sibling convolutions over the same input merges them into one
    multi-branch LinearNode that writes straight into the concatenated layout
    (HYPELCNNModel.py:167-183, DUALCNNModel.py:92-104, CONCNNModel.py:33-36);
  * `t + scale_in_to_out(src, t)` / `t + other` appends a residual (with its channel index
    map) to the producing node's epilogue; dropout folds the same way.
"""
import math

import os

import numpy as np

ACT_CODES = {None: 0, "lrelu": 1, "relu": 2, "sigmoid": 3, "tanh": 4}


class Activation:
    def __init__(self, kind, alpha=0.0):
        assert kind in ACT_CODES
        self.kind = kind
        self.alpha = float(alpha)

    @property
    def code(self):
        return ACT_CODES[self.kind]

    def __eq__(self, other):
        return isinstance(other, Activation) and (self.kind, self.alpha) == (other.kind, other.alpha)

    def __hash__(self):
        return hash((self.kind, self.alpha))


def leaky_relu(alpha):
    return Activation("lrelu", alpha)


relu = Activation("relu")
sigmoid = Activation("sigmoid")
tanh = Activation("tanh")


# ------------------------------------------------------------------------------- variables
def variance_scaling_init(scale=2.0):
    """tensorflow.initializers.variance_scaling(scale=2.0): fan_in, truncated normal with
    stddev = sqrt(scale/fan_in)/0.87962566 (HYPELCNNModel.py:41)."""

    def init(rng, shape):
        fan_in = int(np.prod(shape[:-1]))
        std = math.sqrt(scale / max(fan_in, 1)) / 0.87962566103423978
        w = rng.standard_normal(shape)
        bad = np.abs(w) > 2.0
        while bad.any():  # truncated normal: resample outside two sigma
            w[bad] = rng.standard_normal(int(bad.sum()))
            bad = np.abs(w) > 2.0
        return (w * std).astype(np.float32)

    return init


def xavier_init():
    """tf_slim default weights_initializer: uniform +-sqrt(6/(fan_in+fan_out))."""

    def init(rng, shape):
        rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
        lim = math.sqrt(6.0 / (shape[-2] * rf + shape[-1] * rf))
        return rng.uniform(-lim, lim, size=shape).astype(np.float32)

    return init


def zeros_init():
    return lambda rng, shape: np.zeros(shape, np.float32)


def ones_init():
    return lambda rng, shape: np.ones(shape, np.float32)


class Variable:
    def __init__(self, name, shape, init, trainable):
        self.name = name
        self.shape = tuple(int(s) for s in shape)
        self.init = init
        self.trainable = trainable
        self.size = int(np.prod(self.shape))
        self.offset = None  # element offset in the flat parameter / state buffer (set by the session)
        self.group = "default"  # optimiser group: variables of one group are contiguous in the flat buffers
        self.l2_scale = 0.0  # tf_slim.l2_regularizer scale attached to this variable (0 = none)
        self.order_key = None  # layout position override (tuple); None = creation order (see Session.finalize_variables)


# tf.compat.v1.variable_scope: name prefixes + default-name uniquification.  Re-entering a scope restarts the
# default-name counters, which is how AUTO_REUSE shares un-named tf_slim layers (fully_connected, _1, ...).
_VSCOPE = [""]
_DEFAULT_NAME_COUNTS = {}


class variable_scope:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        full = (_VSCOPE[-1] + "/" + self.name) if _VSCOPE[-1] else self.name
        _VSCOPE.append(full)
        for k in [k for k in _DEFAULT_NAME_COUNTS if k[0].startswith(full)]:
            del _DEFAULT_NAME_COUNTS[k]
        return full

    def __exit__(self, *exc):
        _VSCOPE.pop()
        return False


def current_scope():
    return _VSCOPE[-1]


def unique_default_name(base):
    key = (_VSCOPE[-1], base)
    n = _DEFAULT_NAME_COUNTS.get(key, 0)
    _DEFAULT_NAME_COUNTS[key] = n + 1
    return base if n == 0 else f"{base}_{n}"


def _scoped(name):
    return (_VSCOPE[-1] + "/" + name) if _VSCOPE[-1] else name


GROUP_MARKERS = ("Generator", "Discriminator", "FeatDiscriminator")


def group_of(full_name):
    parts = full_name.split("/")
    for i, p in enumerate(parts):
        if p in GROUP_MARKERS:
            return "/".join(parts[:i + 1])
    return "default"


class VariableStore:
    """Variables shared by every tower built from one template (tf.make_template("nn_core", ...),
    common_nn_ops.py:333)."""

    def __init__(self, prefix="nn_core"):
        self.prefix = prefix
        self.vars = {}
        self.order = []

    def get(self, name, shape, init, trainable=True):
        name = _scoped(name)
        full = f"{self.prefix}/{name}" if self.prefix else name
        v = self.vars.get(full)
        if v is None:
            v = Variable(full, shape, init, trainable)
            v.group = group_of(full)
            self.vars[full] = v
            self.order.append(v)
        elif v.shape != tuple(shape):
            raise ValueError(f"variable {full}: shape {v.shape} vs {tuple(shape)}")
        return v


# ------------------------------------------------------------------------------- tensors
class SymTensor:
    """[N, H, W, C] (hw = (H, W)) or [N, C] (hw = None).  A tensor either owns a pixel-major
    buffer (root is None) or is a view (channel slice / spatial crop) of its root."""

    def __init__(self, tower, hw, c, node=None, root=None, pixmap=None, ch_off=0, name=None):
        self.tower = tower
        self.hw = hw
        self.c = int(c)
        self.node = node
        self.root = root
        self.pixmap = pixmap
        self.ch_off = ch_off
        self.name = name
        self.consumers = 0
        self.needs_grad = True
        self.absorbed = False

    @property
    def npix(self):
        return 1 if self.hw is None else self.hw[0] * self.hw[1]

    @property
    def owner(self):
        return self if self.root is None else self.root

    def get_shape(self):
        return [None, self.c] if self.hw is None else [None, self.hw[0], self.hw[1], self.c]

    def use(self):
        if self.absorbed:
            raise RuntimeError("this tensor handle was fused into a later op; use the value returned by that op")
        self.consumers += 1
        if self.root is not None:
            self.root.consumers += 1
        return self

    def __add__(self, other):
        return add(self, other)

    def slice_channels(self, start, stop):
        """x[..., start:stop] -- zero-copy view (tf.split, DUALCNNModel.py:20)."""
        assert 0 <= start < stop <= self.c
        return SymTensor(self.tower, self.hw, stop - start, node=None, root=self.owner, pixmap=self.pixmap,
                         ch_off=self.ch_off + start)

    def crop(self, d):
        """x[:, d:-d, d:-d, :] -- zero-copy view (DUALCNNModel.py:23-26)."""
        h, w = self.hw
        base = self.pixmap if self.pixmap is not None else list(range(h * w))
        pm = [base[y * w + x] for y in range(d, h - d) for x in range(d, w - d)]
        return SymTensor(self.tower, (h - 2 * d, w - 2 * d), self.c, node=None, root=self.owner, pixmap=pm,
                         ch_off=self.ch_off)


class FlatTensor:
    """tf_slim.flatten of one or more tensors, concatenated on axis 1 (HYPELCNNModel.py:74;
    DUALCNNModel.py:31).  Feature order per source is (h, w, c)."""

    def __init__(self, sources):
        self.sources = list(sources)
        self.tower = sources[0].tower

    @property
    def features(self):
        return sum(s.npix * s.c for s in self.sources)

    def get_shape(self):
        return [None, self.features]


class ChanMap:
    """Result of scale_in_to_out(src, dst): `src` re-indexed on channels (no weights)."""

    def __init__(self, src, idx):
        self.src = src
        self.idx = idx  # None = identity


# ------------------------------------------------------------------------------- nodes
class Branch:
    def __init__(self, k, cout, w, bias=None, bn=None, scope=None):
        self.k = k
        self.cout = cout
        self.w = w
        self.bias = bias
        self.bn = bn  # (beta, moving_mean, moving_variance) Variables
        self.scope = scope


class LinearNode:
    def __init__(self, kind, sources, branches, act, bn_decay, bn_eps, training):
        self.kind = kind  # "conv" | "dense"
        self.sources = sources
        self.branches = branches
        self.act = act
        self.bn_decay = bn_decay
        self.bn_eps = bn_eps
        self.training = training
        self.dropout_keep = None
        self.residuals = []
        self.out = None
        self.in_slices = None  # kind "blockdense": (channel offset, width) of the source slice each branch reads

    @property
    def has_bn(self):
        return self.branches[0].bn is not None

    @property
    def has_bias(self):
        return self.branches[0].bias is not None

    @property
    def cout(self):
        return sum(b.cout for b in self.branches)

    @property
    def has_post(self):
        return self.has_bn or (self.act is not None) or self.dropout_keep is not None or bool(self.residuals)


class PostNode:
    """Stand-alone elementwise epilogue: out = act(src) * mask + residuals."""

    def __init__(self, src, act=None):
        self.src = src
        self.act = act
        self.dropout_keep = None
        self.residuals = []
        self.out = None


class LRNNode:
    def __init__(self, src, radius, bias, alpha, beta):
        self.src = src
        self.radius, self.bias, self.alpha, self.beta = radius, bias, alpha, beta
        self.out = None


class GeneratorNode:
    """Whole shadowdata_generator_model as ONE fused op (gan/shadow_data_models.py:43-90): [N,B] -> [N,B]."""

    def __init__(self, src, weights, biases, only_encoder):
        self.src = src
        self.weights = weights  # 7 Variables [k,1,1]
        self.biases = biases  # 7 Variables [1]
        self.only_encoder = only_encoder
        self.out = None


class DenseStackNode:
    """A chain of narrow tf_slim.fully_connected layers (biases, optional leaky-ReLU) as ONE fused op per direction
    (hypel_dense_stack_fwd / _bwd): shadowdata_discriminator_model at narrow band counts (shadow_data_models.py:95-121)."""

    def __init__(self, src, layers, alpha):
        self.src = src
        self.layers = layers  # [(weights Variable [cin, cout], biases Variable [cout], leaky: bool)]
        self.alpha = alpha
        self.out = None

    @property
    def weights(self):
        return [w for w, _, _ in self.layers]

    @property
    def biases(self):
        return [b for _, b, _ in self.layers]

    @property
    def widths(self):
        return [self.layers[0][0].shape[0]] + [w.shape[1] for w, _, _ in self.layers]


class FeatStackNode:
    """tf.math.l2_normalize (whole-tensor norm) of each slice embedding [N,E], stacked to [N, P*E]
    (gan/shadow_data_models.py:147-149)."""

    def __init__(self, srcs):
        self.srcs = srcs
        self.out = None


class Tower:
    """One instantiation of the model template (train / test / validation towers share variables)."""

    def __init__(self, store, is_training, name="tower"):
        self.store = store
        self.is_training = is_training
        self.name = name
        self.nodes = []
        self.inputs = {}
        self.n_dropout = 0

    def placeholder(self, name, hw, c):
        t = SymTensor(self, hw, c, node=None, name=name)
        t.needs_grad = False
        self.inputs[name] = t
        return t


# ------------------------------------------------------------------------------- arg_scope
_SCOPE_STACK = [{}]


class arg_scope:
    """tf_slim.arg_scope for conv2d / fully_connected keyword defaults."""

    def __init__(self, funcs, **kwargs):
        self.kwargs = kwargs

    def __enter__(self):
        merged = dict(_SCOPE_STACK[-1])
        merged.update(self.kwargs)
        _SCOPE_STACK.append(merged)
        return self

    def __exit__(self, *exc):
        _SCOPE_STACK.pop()
        return False


def _defaults(kwargs):
    out = dict(_SCOPE_STACK[-1])
    out.update({k: v for k, v in kwargs.items() if v is not _UNSET})
    return out


MERGE_PATCH_MLPS = True
_UNSET = object()
batch_norm = "batch_norm"  # normalizer_fn marker (tf_slim.batch_norm: center=True, scale=False, eps=1e-3)


def _make_branch(tower, scope, k, cin, cout, is_conv, opts):
    st = tower.store
    w_init = opts.get("weights_initializer") or xavier_init()
    shape = (k, k, cin, cout) if is_conv else (cin, cout)
    w = st.get(f"{scope}/weights", shape, w_init, True)
    reg = opts.get("weights_regularizer")
    if reg:
        w.l2_scale = float(reg)
    bias = bn = None
    if opts.get("normalizer_fn") == batch_norm:
        # a normaliser suppresses the bias (tf_slim layers.convolution / fully_connected)
        bn = (st.get(f"{scope}/BatchNorm/beta", (cout,), zeros_init(), True),
              st.get(f"{scope}/BatchNorm/moving_mean", (cout,), zeros_init(), False),
              st.get(f"{scope}/BatchNorm/moving_variance", (cout,), ones_init(), False))
    else:
        bias = st.get(f"{scope}/biases", (cout,), opts.get("biases_initializer") or zeros_init(), True)
    return Branch(k, cout, w, bias, bn, scope)


def _bn_opts(opts, tower):
    p = opts.get("normalizer_params") or {}
    return p.get("decay", 0.999), p.get("epsilon", 0.001), bool(p.get("is_training", tower.is_training))


def conv2d(inputs, num_outputs, kernel_size, scope, activation_fn=_UNSET, normalizer_fn=_UNSET,
           normalizer_params=_UNSET, weights_initializer=_UNSET, biases_initializer=_UNSET, weights_regularizer=_UNSET,
           data_format=None):
    """tf_slim.conv2d: NHWC, stride 1, SAME, square kernel (HYPELCNNModel.py:136,157,177)."""
    opts = _defaults(dict(activation_fn=activation_fn, normalizer_fn=normalizer_fn, normalizer_params=normalizer_params,
                          weights_initializer=weights_initializer, biases_initializer=biases_initializer,
                          weights_regularizer=weights_regularizer))
    k = kernel_size[0] if isinstance(kernel_size, (list, tuple)) else int(kernel_size)
    if isinstance(kernel_size, (list, tuple)) and kernel_size[0] != kernel_size[1]:
        raise NotImplementedError("only square kernels are on the hot path (HYPELCNNModel.py:174)")
    tower = inputs.tower
    act = opts.get("activation_fn", relu)
    br = _make_branch(tower, scope, k, inputs.c, int(num_outputs), True, opts)
    decay, eps, training = _bn_opts(opts, tower)
    node = LinearNode("conv", [inputs.use()], [br], act, decay, eps, training)
    node.out = SymTensor(tower, inputs.hw, br.cout, node=node)
    tower.nodes.append(node)
    return node.out


def l2_regularizer(scale):
    """tf_slim.l2_regularizer(scale): scale * sum(w^2) / 2; recorded on the variable, applied by GAN phases
    (tfgan.gan_loss adds the scope's regularisation losses; the classifier path never does, Appendix A.7)."""
    return float(scale)


def fully_connected(inputs, num_outputs, scope=None, activation_fn=_UNSET, normalizer_fn=_UNSET,
                    normalizer_params=_UNSET, weights_initializer=_UNSET, biases_initializer=_UNSET,
                    weights_regularizer=_UNSET):
    """tf_slim.fully_connected on [N, F]; F may be a flatten / axis-1 concat of patch tensors."""
    opts = _defaults(dict(activation_fn=activation_fn, normalizer_fn=normalizer_fn, normalizer_params=normalizer_params,
                          weights_initializer=weights_initializer, biases_initializer=biases_initializer,
                          weights_regularizer=weights_regularizer))
    if scope is None:
        scope = unique_default_name("fully_connected")
    if isinstance(inputs, FlatTensor):
        sources, feats, tower = inputs.sources, inputs.features, inputs.tower
    else:
        if inputs.hw is not None and inputs.npix != 1:
            raise ValueError("fully_connected expects a rank-2 input; call flatten() first")
        sources, feats, tower = [inputs], inputs.c, inputs.tower
    act = opts.get("activation_fn", relu)
    br = _make_branch(tower, scope, 0, feats, int(num_outputs), False, opts)
    decay, eps, training = _bn_opts(opts, tower)
    node = LinearNode("dense", [s.use() for s in sources], [br], act, decay, eps, training)
    node.out = SymTensor(tower, None, br.cout, node=node)
    tower.nodes.append(node)
    return node.out


def flatten(inputs):
    if isinstance(inputs, FlatTensor):
        return inputs
    return FlatTensor([inputs])


def concat(values, axis):
    values = list(values)
    if axis == 1 and all(isinstance(v, FlatTensor) for v in values):
        return FlatTensor([s for v in values for s in v.sources])
    if axis != 3:
        raise NotImplementedError("concat on the hot path is channel (axis=3) or flattened (axis=1)")
    nodes = [v.node for v in values]
    first = nodes[0]
    ok = all(isinstance(n, LinearNode) and n.kind == "conv" and len(n.branches) == 1 for n in nodes)
    ok = ok and all(n.sources[0] is first.sources[0] for n in nodes)
    ok = ok and all(n.act == first.act and n.has_bn == first.has_bn and n.bn_decay == first.bn_decay
                    and n.dropout_keep is None and not n.residuals for n in nodes)
    ok = ok and all(v.consumers == 0 for v in values)
    if not ok:
        raise NotImplementedError("concat(axis=3) is supported for sibling convolutions over one input")
    tower = first.out.tower
    merged = LinearNode("conv", first.sources, [n.branches[0] for n in nodes], first.act, first.bn_decay, first.bn_eps,
                        first.training)
    src = first.sources[0]
    for n in nodes[1:]:  # the siblings each counted one use of the shared input
        src.consumers -= 1
        if src.root is not None:
            src.root.consumers -= 1
    pos = tower.nodes.index(first)
    for n in nodes:
        tower.nodes.remove(n)
        n.out.absorbed = True
    merged.out = SymTensor(tower, src.hw, merged.cout, node=merged)
    tower.nodes.insert(pos, merged)
    return merged.out


def _foldable(t):
    return (t.root is None and t.consumers == 0 and isinstance(t.node, (LinearNode, PostNode))
            and len(t.node.residuals) < 2)


def _refresh(t):
    """The node's output changes meaning after folding: hand back a fresh handle."""
    node = t.node
    t.absorbed = True
    out = SymTensor(t.tower, t.hw, t.c, node=node)
    node.out = out
    return out


def add(a, b):
    if isinstance(a, ChanMap) and not isinstance(b, ChanMap):
        a, b = b, a
    if isinstance(b, ChanMap):
        src, idx = b.src, b.idx
    else:
        src, idx = b, None
    if src.c != a.c and idx is None:
        raise ValueError("channel mismatch in add")
    if (a.hw or (1, 1)) != (src.hw or (1, 1)):
        raise ValueError("spatial mismatch in add")
    if not _foldable(a):
        node = PostNode(a.use())
        node.out = SymTensor(a.tower, a.hw, a.c, node=node)
        a.tower.nodes.append(node)
        a = node.out
    a.node.residuals.append((src.use(), idx))
    return _refresh(a)


def dropout(inputs, keep_prob, is_training):
    """tf_slim.dropout: identity at inference; mask in {0, 1/keep_prob} when training."""
    if not is_training:
        return inputs
    t = inputs
    if not (_foldable(t) and not t.node.residuals and t.node.dropout_keep is None):
        node = PostNode(t.use())
        node.out = SymTensor(t.tower, t.hw, t.c, node=node)
        t.tower.nodes.append(node)
        t = node.out
    t.node.dropout_keep = float(keep_prob)
    t.node.dropout_index = t.tower.n_dropout
    t.tower.n_dropout += 1
    return _refresh(t)


def local_response_normalization(inputs, depth_radius=5, bias=1.0, alpha=1.0, beta=0.5):
    node = LRNNode(inputs.use(), depth_radius, bias, alpha, beta)
    node.out = SymTensor(inputs.tower, inputs.hw, inputs.c, node=node)
    inputs.tower.nodes.append(node)
    return node.out


def he_truncated_init():
    """tf.compat.v1.initializers.variance_scaling(scale=2.0): fan_in, truncated normal (shadow_data_models.py:95)."""
    return variance_scaling_init(2.0)


def shadow_generator(netinput, create_only_encoder, is_training=True):
    """Records the fused generator; variables net1..net7/{weights [k,1,1], biases [1]} are zero-initialised
    (shadow_data_models.py:47).  The full generator and its encoder-only prefix share net1..net4."""
    tower = netinput.tower
    st = tower.store
    b = netinput.c
    ks = [b, b // 2, b // 4, b // 8, b // 4, b // 2, b]
    ws, bs = [], []
    for i, k in enumerate(ks, start=1):  # all seven exist in TF as soon as the full generator was built once
        ws.append(st.get(f"net{i}/weights", (k, 1, 1), zeros_init(), True))
    for i in range(1, 8):
        bs.append(st.get(f"net{i}/biases", (1,), zeros_init(), True))
    node = GeneratorNode(netinput.use(), ws, bs, bool(create_only_encoder))
    node.out = SymTensor(tower, None, b, node=node)
    tower.nodes.append(node)
    return node.out


def _dense_chain(t):
    """The plain fully-connected layers between a channel slice of some tensor and `t` (each consumed only by the
    next one), first layer first; None when `t` is not the end of such a chain."""
    chain, cur = [], t
    while True:
        n = cur.node
        if not (isinstance(n, LinearNode) and n.kind == "dense" and len(n.branches) == 1 and len(n.sources) == 1
                and not n.has_bn and n.has_bias and not n.residuals and n.dropout_keep is None and cur.root is None
                and cur.consumers == (0 if not chain else 1)):
            return None
        chain.append(n)
        src = n.sources[0]
        if src.root is not None and src.node is None:  # the channel slice the chain starts from
            return chain[::-1]
        cur = src


def _merge_patch_mlps(embeddings):
    """shadowdata_feature_discriminator_model runs one small MLP per band slice (shadow_data_models.py:133-146): P
    chains of L tiny fully-connected layers = P*L GEMM launches plus as many epilogues, each far below a microsecond
    of work.  Layer l of all P chains has the same output width, so the P products are ONE grouped GEMM (a block
    diagonal matrix; groups differ in the input slice and the weights) and their bias/activation ONE epilogue on the
    [N, P*cout] result.  The nodes are replaced by L "blockdense" nodes; variables keep their TensorFlow names but
    are laid out layer-major so that the biases (and weights) of a merged layer are contiguous."""
    chains = [_dense_chain(e) for e in embeddings]
    if any(c is None for c in chains) or len(chains) < 2 or len({len(c) for c in chains}) != 1:
        return None
    depth = len(chains[0])
    tower = embeddings[0].tower
    first_srcs = [c[0].sources[0] for c in chains]
    root = first_srcs[0].root
    if any(s.root is not root or s.pixmap is not None or s.hw is not None for s in first_srcs):
        return None
    slices = [(s.ch_off - root.ch_off if root.root is not None else s.ch_off, s.c) for s in first_srcs]
    if (sorted(slices) != slices or any(a[0] + a[1] != b[0] for a, b in zip(slices, slices[1:])) or slices[0][0] != 0
            or slices[-1][0] + slices[-1][1] != root.c):
        return None  # the slices must tile the whole source in patch order
    for l in range(depth):
        nodes = [c[l] for c in chains]
        if len({n.branches[0].cout for n in nodes}) != 1 or len({(n.act.code, n.act.alpha) if n.act else None
                                                                  for n in nodes}) != 1:
            return None
    # ---- rewrite ----
    all_vars = [v for c in chains for n in c for v in (n.branches[0].w, n.branches[0].bias)]
    anchor = min(tower.store.order.index(v) for v in all_vars)
    pos = min(tower.nodes.index(n) for c in chains for n in c)
    for c in chains:
        for n in c:
            tower.nodes.remove(n)
            n.out.absorbed = True
    for s in first_srcs:  # the slices each counted one use of the root
        root.consumers -= 1
    src, prev_cout, out = root, None, None
    for l in range(depth):
        nodes = [c[l] for c in chains]
        branches = [n.branches[0] for n in nodes]
        for pi, b in enumerate(branches):
            if b.w.order_key is None:
                b.w.order_key = (anchor, 1, l, pi)
                b.bias.order_key = (anchor, 0, l, pi)
        merged = LinearNode("blockdense", [src.use()], branches, nodes[0].act, nodes[0].bn_decay, nodes[0].bn_eps,
                            nodes[0].training)
        cout = branches[0].cout
        merged.in_slices = slices if l == 0 else [(pi * prev_cout, prev_cout) for pi in range(len(branches))]
        merged.out = SymTensor(tower, None, cout * len(branches), node=merged)
        tower.nodes.insert(pos + l, merged)
        src, prev_cout, out = merged.out, cout, merged.out
    return out, len(chains), prev_cout


DENSE_STACK = True
DENSE_STACK_MAX_WIDTH = 128  # hypel.h: hypel_dense_stack_supported


def dense_stack_fits(widths):
    """hypel_dense_stack_supported's rule (csrc/dense_stack.hip): activations, gradients and EVERY layer's weights of
    a 16-sample tile live in the 160 KB of LDS."""
    if not (2 <= len(widths) <= 5 and all(1 <= v <= DENSE_STACK_MAX_WIDTH for v in widths)):
        return False
    w16 = (max(widths) + 15) // 16 * 16
    p = w16 // 32 * 32 + 18
    p = p if p >= w16 else p + 32
    return (8 * 16 * p + sum((c + 3) // 4 * 4 * p for c in widths[:-1]) + sum(widths[1:])) * 4 <= 160 * 1024


def fuse_dense_stack(t):
    """`t` = the output of a chain of plain fully-connected layers (biases, no normaliser, activation None or leaky-ReLU
    with one alpha, every intermediate consumed by the next layer only) on a [N, C] tensor: when every width is at most
    128 the chain is re-recorded as ONE DenseStackNode -- one launch per direction instead of ~5 forward / ~12 backward
    launches of a few microseconds each (the CycleGAN step at 64 bands is nothing but such launches).  Returns the tensor
    to use in place of `t` (`t` itself when the chain does not qualify).  Variables keep their TensorFlow names."""
    if not DENSE_STACK:
        return t
    chain, cur = [], t
    while len(chain) < 4:
        n = cur.node
        if not (isinstance(n, LinearNode) and n.kind == "dense" and len(n.branches) == 1 and len(n.sources) == 1
                and not n.has_bn and n.has_bias and not n.residuals and n.dropout_keep is None and cur.root is None
                and cur.consumers == (0 if not chain else 1)):
            break
        if n.act is not None and n.act.kind != "lrelu":
            break
        src = n.sources[0]
        if src.hw is not None or src.pixmap is not None:
            break
        chain.append(n)
        cur = src
    chain = chain[::-1]
    if len(chain) < 2:
        return t
    alphas = {n.act.alpha for n in chain if n.act is not None}
    widths = [chain[0].sources[0].c] + [n.branches[0].cout for n in chain]
    if len(alphas) > 1 or not dense_stack_fits(widths):
        return t
    tower = t.tower
    pos = tower.nodes.index(chain[0])
    for n in chain:
        tower.nodes.remove(n)
        n.out.absorbed = True
    node = DenseStackNode(chain[0].sources[0], [(n.branches[0].w, n.branches[0].bias, n.act is not None) for n in chain],
                          alphas.pop() if alphas else 0.0)
    node.out = SymTensor(tower, None, widths[-1], node=node)
    tower.nodes.insert(pos, node)
    return node.out


def feature_stack(embeddings):
    tower = embeddings[0].tower
    merged = _merge_patch_mlps(embeddings) if MERGE_PATCH_MLPS else None
    if merged is not None:
        out, parts, width = merged
        embeddings = [out.slice_channels(i * width, (i + 1) * width) for i in range(parts)]
    node = FeatStackNode([e.use() for e in embeddings])
    node.out = SymTensor(tower, None, sum(e.c for e in embeddings), node=node)
    node.out.parts = len(embeddings)
    tower.nodes.append(node)
    return node.out


# GAN loss terms: (kind, tensors..., weight).  A phase minimises the sum of its terms w.r.t. its variable groups.
class LossTerm:
    def __init__(self, kind, a, b=None, target=0.0, weight=1.0, tau=None, parts=None, embed=None):
        self.kind = kind  # "mean_sq" | "mean_abs" | "mean" | "nce"
        self.a, self.b = a, b
        self.target, self.weight, self.tau, self.parts, self.embed = target, weight, tau, parts, embed


# ------------------------------------------------------------------------------- loss expressions
class PerSampleXent:
    """tf.nn.softmax_cross_entropy_with_logits(labels, logits) -> [N]."""

    def __init__(self, logits, labels):
        self.logits = logits.use()
        self.labels = labels
        self.extra_mse = None

    def __add__(self, other):
        if isinstance(other, ScalarMSE):  # scalar broadcast-added to every sample (HYPELCNNModel.py:109)
            self.extra_mse = other
            return self
        return NotImplemented


class ScalarMSE:
    """tf.reduce_mean(tf.square(a - flatten(b)))."""

    def __init__(self, a, b_flat):
        self.a = a.use()
        self.b = b_flat


class MeanLoss:
    """tf.reduce_mean over the per-sample loss (common_nn_ops.py:214)."""

    def __init__(self, per_sample):
        self.per_sample = per_sample


def softmax_cross_entropy_with_logits(labels, logits):
    return PerSampleXent(logits, labels)


def mean_squared_reconstruction(image_output, image_original):
    return ScalarMSE(image_output, flatten(image_original))


def reduce_mean(expr):
    return MeanLoss(expr)
