"""Functional `tensorflow` / `tf_slim` stand-in under which the REFERENCE's own model files run (build container only).

TEST INFRASTRUCTURE.  The reference's plugins (`/root/reference/nnmodel/*.py`, `gan/shadow_data_models.py`) import
`tensorflow` and `tf_slim`, which cannot be installed here.  This module installs a meta-path finder that serves
those names with just enough behaviour for `create_tensor_graph` and the three `shadowdata_*_model` builders to EXECUTE
UNCHANGED, through one of two engines:

  * `OracleEngine`  -- every layer call is (a) RECORDED (scope, op, kernel, num_outputs, normaliser, activation,
    keep_prob, regulariser, input ids, output shape) and (b) EVALUATED in float64 with `oracle/ops.py`.  What it pins:
    the WIRING of `oracle/models.py` (layer order, scopes, widths, kernel lists, the batch norm on the logits, both
    dropout keep-prob conventions, `fc_stage_count`, the ragged last feature-discriminator slice) -- by execution of
    the reference's text instead of by reading it.  What it does NOT pin: the operator semantics themselves (the
    stand-in ops ARE oracle/ops.py; SURVEY Appendix A stays unpinned at the TensorFlow boundary).
  * `GraphEngine`   -- the same calls routed into `hypelcnn_amd.graph`: the `tf_slim` facade through which the
    reference's unchanged plugin files record a product `Tower`; tests compare it node for node with the Tower the
    product's own plugins record.

Only data leaves the container: `make_reference_graphs.py` writes the recorded layer tables, the canonical Tower dumps
and small-shape float64 outputs to `tests/golden/reference_graphs.json` / `.npz`.
"""
import importlib.abc
import importlib.machinery
import math
import os
import sys
import types

import numpy as np

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
STUB_ROOTS = ("tensorflow", "tf_slim", "tifffile", "tqdm", "tensorflow_gan", "numba", "sklearn")


# ------------------------------------------------------------------------------------------------ permissive fallback
class _Anything(types.ModuleType):
    """Module whose every unknown attribute is another permissive stand-in (import-time names the path never calls)."""

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        child = _Anything(self.__name__ + "." + item)
        setattr(self, item, child)
        return child

    def __call__(self, *a, **k):
        return _Anything(self.__name__ + "()")

    def __mro_entries__(self, bases):
        return (object,)


class Dim:
    """tf.compat.v1.Dimension: `.value`, products of dimensions, usable where an int is expected."""

    def __init__(self, v):
        self.value = None if v is None else int(v)

    def __mul__(self, o):
        return Dim(self.value * (o.value if isinstance(o, Dim) else int(o)))

    __rmul__ = __mul__

    def __floordiv__(self, o):
        return Dim(self.value // (o.value if isinstance(o, Dim) else int(o)))

    def __int__(self):
        return self.value

    __index__ = __int__

    def __eq__(self, o):
        return self.value == (o.value if isinstance(o, Dim) else o)

    def __hash__(self):
        return hash(self.value)


def _int(v):
    return v.value if isinstance(v, Dim) else int(v)


class ActProbe:
    """Fed to an `activation_fn` lambda to learn WHICH activation it applies (leaky_relu(alpha) / sigmoid / tanh / relu)."""


class ActDesc:
    def __init__(self, kind, alpha=0.0):
        self.kind, self.alpha = kind, float(alpha)

    def key(self):
        return [self.kind, self.alpha] if self.kind == "leaky_relu" else [self.kind]


def describe_activation(fn):
    if fn is None:
        return None
    d = fn(ActProbe())
    if not isinstance(d, ActDesc):
        raise TypeError("activation_fn did not resolve to a known activation")
    return d


# ------------------------------------------------------------------------------------------------ engines
class OracleEngine:
    """Records every call and evaluates it with oracle/ops.py in float64."""

    def __init__(self, params=None, is_training=True, rng=None, dropout_masks=None):
        sys.path.insert(0, ROOT) if ROOT not in sys.path else None
        from oracle import ops as O
        self.O = O
        self.records = []
        self.variables = {}          # name -> shape (creation order)
        self.params = dict(params) if params else {}
        self.rng = rng or np.random.default_rng(0)
        self.is_training = is_training
        self.dropout_masks = dropout_masks or {}
        self.n_dropout = 0
        self.names = {}
        self.next_id = 0

    # -- tensors
    class T:
        def __init__(self, eng, var, tid):
            self.eng, self.var, self.id = eng, var, tid

        def get_shape(self):
            return [Dim(None)] + [Dim(s) for s in self.var.v.shape[1:]]

        @property
        def shape(self):
            return self.get_shape()

        def __add__(self, other):
            return self.eng.add(self, other)

        def __sub__(self, other):
            return self.eng.sub(self, other)

        def __getitem__(self, sl):
            return self.eng.slice(self, sl)

    def new(self, var, op, inputs=(), **attrs):
        t = OracleEngine.T(self, var, self.next_id)
        self.next_id += 1
        rec = {"op": op, "id": t.id, "inputs": [i.id for i in inputs], "shape": [int(s) for s in var.v.shape[1:]]}
        rec.update(attrs)
        self.records.append(rec)
        return t

    def placeholder(self, value, name):
        return self.new(self.O.Var(np.asarray(value, np.float64), name=name), "placeholder", name=name)

    def unique(self, base):
        n = self.names.get(base, 0)
        self.names[base] = n + 1
        return base if n == 0 else f"{base}_{n}"

    def variable(self, name, shape, init):
        shape = tuple(int(s) for s in shape)
        if name not in self.variables:
            self.variables[name] = list(shape)
            if name not in self.params:
                self.params[name] = init(self.rng, shape)
        v = np.asarray(self.params[name], np.float64)
        assert v.shape == shape, (name, v.shape, shape)
        return self.O.Var(v, name=name)

    # -- layers (opts: the merged arg_scope keywords)
    def _post(self, y, scope, opts, cout):
        O = self.O
        norm = opts.get("normalizer_fn")
        if norm is not None:
            p = opts.get("normalizer_params") or {}
            beta = self.variable(scope + "/BatchNorm/beta", (cout,), lambda r, s: np.zeros(s))
            mm = self.variable(scope + "/BatchNorm/moving_mean", (cout,), lambda r, s: np.zeros(s))
            mv = self.variable(scope + "/BatchNorm/moving_variance", (cout,), lambda r, s: np.ones(s))
            if p.get("is_training", True):
                y = O.batch_norm_train(y, beta)[0]
            else:
                y = O.batch_norm_infer(y, beta, mm.v, mv.v)
        act = describe_activation(opts.get("activation_fn", _relu))
        if act is not None:
            y = {"leaky_relu": lambda t: O.leaky_relu(t, act.alpha), "relu": O.relu, "sigmoid": O.sigmoid,
                 "tanh": O.tanh}[act.kind](y)
        return y, act

    def _layer_attrs(self, scope, opts, act, **more):
        p = opts.get("normalizer_params") or {}
        a = {"scope": scope, "normalizer": "batch_norm" if opts.get("normalizer_fn") is not None else None,
             "bn_decay": p.get("decay") if opts.get("normalizer_fn") is not None else None,
             "bn_is_training": p.get("is_training") if opts.get("normalizer_fn") is not None else None,
             "activation": act.key() if act else None,
             "regularizer": opts.get("weights_regularizer"),
             "initializer": getattr(opts.get("weights_initializer"), "desc", "xavier"),
             "trainable": bool(opts.get("trainable", True))}
        a.update(more)
        return a

    def conv2d(self, x, num_outputs, kernel_size, scope, opts):
        O = self.O
        kh, kw = (kernel_size if isinstance(kernel_size, (list, tuple)) else (kernel_size, kernel_size))
        kh, kw = _int(kh), _int(kw)
        cin, cout = x.var.v.shape[3], _int(num_outputs)
        scope = scope or self.unique("Conv")
        w = self.variable(scope + "/weights", (kh, kw, cin, cout), opts.get("weights_initializer") or _xavier)
        b = None
        if opts.get("normalizer_fn") is None:
            b = self.variable(scope + "/biases", (cout,), lambda r, s: np.zeros(s))
        assert kh == kw, "square kernels only on the path"
        y = O.conv2d_same(x.var, w, b)
        y, act = self._post(y, scope, opts, cout)
        return self.new(y, "conv2d", [x], **self._layer_attrs(scope, opts, act, kernel=[kh, kw], num_outputs=cout,
                                                              biases=b is not None))

    def fully_connected(self, x, num_outputs, scope, opts):
        O = self.O
        cin, cout = x.var.v.shape[1], _int(num_outputs)
        assert x.var.v.ndim == 2
        scope = scope or self.unique("fully_connected")
        w = self.variable(scope + "/weights", (cin, cout), opts.get("weights_initializer") or _xavier)
        b = None
        if opts.get("normalizer_fn") is None:
            b = self.variable(scope + "/biases", (cout,), lambda r, s: np.zeros(s))
        y = O.dense(x.var, w, b)
        y, act = self._post(y, scope, opts, cout)
        return self.new(y, "fully_connected", [x], **self._layer_attrs(scope, opts, act, num_outputs=cout,
                                                                       biases=b is not None))

    def convolution1d(self, x, num_outputs, kernel_size, scope, padding, opts):
        O = self.O
        k, cin, cout = _int(kernel_size), x.var.v.shape[2], _int(num_outputs)
        scope = scope or self.unique("Conv")
        assert padding == "SAME" and cin == 1 and cout == 1, "the path's generator: SAME, one channel in and out"
        w = self.variable(scope + "/weights", (k, cin, cout), opts.get("weights_initializer") or _xavier)
        b = None
        if opts.get("normalizer_fn") is None:
            b = self.variable(scope + "/biases", (cout,), lambda r, s: np.zeros(s))
        y = O.conv1d_same(x.var, w, b)
        y, act = self._post(y, scope, opts, cout)
        return self.new(y, "convolution1d", [x], **self._layer_attrs(scope, opts, act, kernel=[k], num_outputs=cout,
                                                                     padding=padding, biases=b is not None))

    def dropout(self, x, keep_prob, is_training):
        if not is_training:
            return self.new(x.var, "dropout", [x], keep_prob=float(keep_prob), is_training=False)
        key = f"dropout_{self.n_dropout}"
        self.n_dropout += 1
        mask = self.dropout_masks.get(key)
        if mask is None:
            mask = np.ones_like(x.var.v)
        return self.new(self.O.dropout(x.var, mask.astype(np.float64)), "dropout", [x], keep_prob=float(keep_prob),
                        is_training=True)

    def flatten(self, x):
        return self.new(self.O.flatten(x.var), "flatten", [x])

    def concat(self, values, axis):
        return self.new(self.O.concat([v.var for v in values], axis=axis), "concat", values, axis=int(axis))

    def split(self, value, sizes, axis):
        out, pos = [], 0
        for s in sizes:
            sl = [slice(None)] * value.var.v.ndim
            sl[axis] = slice(pos, pos + _int(s))
            out.append(self.new(self.O.slice_(value.var, tuple(sl)), "split_part", [value], axis=int(axis), start=pos,
                                size=_int(s)))
            pos += _int(s)
        return out

    def slice(self, x, sl):
        if not isinstance(sl, tuple):
            sl = (sl,)
        sl = tuple(sl) + (slice(None),) * (x.var.v.ndim - len(sl))
        norm = []
        for s, dim in zip(sl, x.var.v.shape):
            a, b, _ = s.indices(dim)
            norm.append([a, b])
        return self.new(self.O.slice_(x.var, sl), "slice", [x], ranges=norm[1:])

    def add(self, a, b):
        return self.new(self.O.add(a.var, b.var), "add", [a, b])

    def sub(self, a, b):
        return self.new(self.O.sub(a.var, b.var), "sub", [a, b])

    def gather(self, x, idx, axis):
        assert axis == x.var.v.ndim - 1
        return self.new(self.O.gather_channels(x.var, np.asarray(idx, np.int64)), "gather", [x], axis=int(axis),
                        indices=[int(i) for i in idx])

    def repeat(self, x, axis, repeats):
        assert axis == x.var.v.ndim - 1
        idx = np.arange(x.var.v.shape[-1] * repeats) // repeats
        return self.new(self.O.gather_channels(x.var, idx), "repeat", [x], axis=int(axis), repeats=int(repeats))

    def lrn(self, x, **kw):
        return self.new(self.O.lrn(x.var, **kw), "local_response_normalization", [x], **kw)

    def reshape(self, x, shape, op="reshape"):
        return self.new(self.O.reshape(x.var, shape), op, [x])

    def l2_normalize(self, x):
        return self.new(self.O.l2_normalize_global(x.var), "l2_normalize", [x])


class GraphEngine:
    """Routes the same calls into hypelcnn_amd.graph (the tf_slim facade of the product)."""

    def __init__(self, tower):
        from hypelcnn_amd import graph as G
        from hypelcnn_amd.common import common_nn_ops as P
        self.G, self.P, self.tower = G, P, tower

    class T:
        """Wrapper that gives a graph.SymTensor / FlatTensor the TensorFlow tensor surface the reference files use."""

        def __init__(self, eng, sym):
            self.eng, self.sym = eng, sym

        def get_shape(self):
            return [Dim(s) for s in self.sym.get_shape()]

        @property
        def shape(self):
            return self.get_shape()

        def __add__(self, other):
            G = self.eng.G
            return GraphEngine.T(self.eng, G.add(self.sym, other.sym))

        def __getitem__(self, sl):
            return self.eng.slice(self, sl)

    def wrap(self, sym):
        return GraphEngine.T(self, sym)

    def _act(self, opts):
        G = self.G
        d = describe_activation(opts.get("activation_fn", _relu))
        if d is None:
            return None
        return {"leaky_relu": lambda: G.leaky_relu(d.alpha), "relu": lambda: G.relu, "sigmoid": lambda: G.sigmoid,
                "tanh": lambda: G.tanh}[d.kind]()

    def _kw(self, opts):
        G = self.G
        init = getattr(opts.get("weights_initializer"), "desc", None)
        kw = {"activation_fn": self._act(opts),
              "normalizer_fn": G.batch_norm if opts.get("normalizer_fn") is not None else None,
              "normalizer_params": opts.get("normalizer_params"),
              "weights_regularizer": opts.get("weights_regularizer")}
        if init is not None and init[0] == "variance_scaling":
            kw["weights_initializer"] = G.variance_scaling_init(scale=init[1])
        elif init is not None and init[0] == "zeros":
            kw["weights_initializer"] = G.zeros_init()
        return kw

    def conv2d(self, x, num_outputs, kernel_size, scope, opts):
        ks = [_int(k) for k in kernel_size] if isinstance(kernel_size, (list, tuple)) else _int(kernel_size)
        return self.wrap(self.G.conv2d(x.sym, _int(num_outputs), ks, scope=scope, **self._kw(opts)))

    def fully_connected(self, x, num_outputs, scope, opts):
        return self.wrap(self.G.fully_connected(x.sym, _int(num_outputs), scope=scope, **self._kw(opts)))

    def dropout(self, x, keep_prob, is_training):
        return self.wrap(self.G.dropout(x.sym, keep_prob=keep_prob, is_training=is_training))

    def flatten(self, x):
        return self.wrap(self.G.flatten(x.sym))

    def concat(self, values, axis):
        return self.wrap(self.G.concat([v.sym for v in values], axis=axis))

    def split(self, value, sizes, axis):
        assert axis == 3
        out, pos = [], 0
        for s in sizes:
            out.append(self.wrap(value.sym.slice_channels(pos, pos + _int(s))))
            pos += _int(s)
        return out

    def slice(self, x, sl):
        sym = x.sym
        if isinstance(sym, self.G.FlatTensor) or sym.hw is None:  # net[:, a:b] on a flattened tensor
            s = sl[1]
            src = sym.sources[0] if isinstance(sym, self.G.FlatTensor) else sym
            assert (not isinstance(sym, self.G.FlatTensor)) or (len(sym.sources) == 1 and src.npix == 1)
            a, b, _ = s.indices(src.c)
            return self.wrap(src.slice_channels(a, b))
        h, w = sym.hw  # x[:, d:-d, d:-d, :]
        (a0, b0, _), (a1, b1, _) = sl[1].indices(h), sl[2].indices(w)
        assert a0 == a1 and h - b0 == a0 and w - b1 == a0 and sl[3] == slice(None)
        return self.wrap(sym.crop(a0))

    def gather(self, x, idx, axis):
        return self.wrap(self.G.ChanMap(x.sym, np.asarray(idx, np.int32)))

    def repeat(self, x, axis, repeats):
        return self.wrap(self.G.ChanMap(x.sym, np.arange(x.sym.c * repeats, dtype=np.int32) // repeats))

    def lrn(self, x, **kw):
        return self.wrap(self.G.local_response_normalization(x.sym, **kw))


# ------------------------------------------------------------------------------------------------ tf / tf_slim surface
ENGINE = [None]
_ARG_STACK = [{}]


def _relu(x):
    return ActDesc("relu") if isinstance(x, ActProbe) else None


def _xavier(rng, shape):
    fan_in = int(np.prod(shape[:-1]))
    fan_out = int(shape[-1]) * int(np.prod(shape[:-2])) if len(shape) > 2 else int(shape[-1])
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape)


class _Init:
    def __init__(self, desc, fn):
        self.desc, self.fn = desc, fn

    def __call__(self, rng, shape):
        return self.fn(rng, shape)


def variance_scaling(scale=1.0, **kw):
    def fn(rng, shape):
        fan_in = int(np.prod(shape[:-1]))
        std = math.sqrt(scale / fan_in) / 0.87962566103423978
        return np.clip(rng.standard_normal(shape), -2.0, 2.0) * std
    return _Init(("variance_scaling", float(scale)), fn)


def zeros_initializer(**kw):
    return _Init(("zeros",), lambda rng, shape: np.zeros(shape))


class arg_scope:
    def __init__(self, funcs, **kwargs):
        self.funcs, self.kwargs = [getattr(f, "__name__", str(f)) for f in funcs], kwargs

    def __enter__(self):
        top = {k: dict(v) for k, v in _ARG_STACK[-1].items()}
        for f in self.funcs:
            top.setdefault(f, {}).update(self.kwargs)
        _ARG_STACK.append(top)
        return self

    def __exit__(self, *exc):
        _ARG_STACK.pop()
        return False


def _opts(fname, kwargs):
    o = dict(_ARG_STACK[-1].get(fname, {}))
    o.update(kwargs)
    return o


def conv2d(inputs, num_outputs, kernel_size, scope=None, **kw):
    o = _opts("conv2d", kw)
    assert o.pop("data_format", None) in (None, "NHWC")
    return ENGINE[0].conv2d(inputs, num_outputs, kernel_size, scope, o)


def fully_connected(inputs, num_outputs, scope=None, **kw):
    return ENGINE[0].fully_connected(inputs, num_outputs, scope, _opts("fully_connected", kw))


def convolution1d(inputs, num_outputs, kernel_size, scope=None, padding="SAME", **kw):
    o = _opts("convolution1d", kw)
    o.pop("data_format", None)
    return ENGINE[0].convolution1d(inputs, num_outputs, kernel_size, scope, padding, o)


def dropout(inputs, keep_prob=0.5, is_training=True, **kw):
    return ENGINE[0].dropout(inputs, keep_prob, is_training)


def flatten(inputs, **kw):
    return ENGINE[0].flatten(inputs)


def batch_norm(*a, **k):
    raise RuntimeError("batch_norm is only a normalizer_fn marker on this path")


def l2_regularizer(scale):
    return float(scale)


def separable_conv2d(*a, **k):
    raise RuntimeError("not on the path")


conv2d_transpose = separable_conv2d


def leaky_relu(inp, alpha=0.2, **kw):
    if isinstance(inp, ActProbe):
        return ActDesc("leaky_relu", alpha)
    raise RuntimeError("leaky_relu is only used as an activation_fn on this path")


def sigmoid(inp, **kw):
    if isinstance(inp, ActProbe):
        return ActDesc("sigmoid")
    raise RuntimeError("sigmoid is only used as an activation_fn on this path")


def tanh(inp, **kw):
    if isinstance(inp, ActProbe):
        return ActDesc("tanh")
    raise RuntimeError("tanh is only used as an activation_fn on this path")


class _Ctx:
    def __init__(self, *a, **k):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def concat(values=None, axis=None, **kw):
    return ENGINE[0].concat(list(values), axis)


def split(value=None, num_or_size_splits=None, axis=0, **kw):
    return ENGINE[0].split(value, list(num_or_size_splits), axis)


def gather(params, indices, axis=None, **kw):
    return ENGINE[0].gather(params, indices, axis)


def repeat(input=None, repeats=None, axis=None, **kw):  # noqa: A002 (TensorFlow's own keyword)
    return ENGINE[0].repeat(input, axis, repeats)


def local_response_normalization(x, depth_radius=5, bias=1.0, alpha=1.0, beta=0.5, **kw):
    return ENGINE[0].lrn(x, depth_radius=depth_radius, bias=bias, alpha=alpha, beta=beta)


def squeeze(x, axis=None, **kw):
    e = ENGINE[0]
    v = x.var.v
    shape = tuple(s for i, s in enumerate(v.shape) if i not in axis)
    return e.reshape(x, shape, "squeeze")


def expand_dims(x, axis=None, **kw):
    e = ENGINE[0]
    shape = list(x.var.v.shape)
    shape.insert(axis, 1)
    return e.reshape(x, tuple(shape), "expand_dims")


def l2_normalize(x, **kw):
    return ENGINE[0].l2_normalize(x)


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in STUB_ROOTS:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _Anything(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        n = module.__name__
        if n == "tensorflow":
            module.device = _Ctx
            module.concat, module.split, module.gather, module.repeat = concat, split, gather, repeat
            module.sigmoid, module.squeeze, module.expand_dims = sigmoid, squeeze, expand_dims
            module.transpose = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("NCHW is not on the path"))
        elif n == "tensorflow.nn":
            module.local_response_normalization = local_response_normalization
        elif n == "tensorflow.math":
            module.l2_normalize = l2_normalize
        elif n in ("tensorflow.initializers", "tensorflow.compat.v1.initializers"):
            module.variance_scaling = variance_scaling
            module.zeros = zeros_initializer
        elif n == "tensorflow.compat.v1":
            module.name_scope = _Ctx
        elif n == "tensorflow.python.ops.gen_nn_ops":
            module.leaky_relu = leaky_relu
        elif n == "tensorflow.python.keras.activations":
            module.tanh = tanh
        elif n == "tensorflow.python.ops.initializers_ns":
            module.variance_scaling = variance_scaling
        elif n == "tf_slim":
            for f in (conv2d, fully_connected, convolution1d, dropout, flatten, batch_norm, l2_regularizer,
                      separable_conv2d, arg_scope):
                setattr(module, f.__name__, f)
            module.conv2d_transpose = conv2d_transpose
        elif n == "numba":
            module.jit = lambda *a, **k: (lambda f: f)


_INSTALLED = [False]


def install():
    """Serve tensorflow / tf_slim (and the other absent imports of the reference) from this module; put the reference on
    sys.path.  Submodules are pre-imported so that `tf.nn.x` / `tf.compat.v1.y` resolve to the functional pieces."""
    if _INSTALLED[0]:
        return
    if not os.path.isdir(REF):
        raise RuntimeError("the reference is only present in the build container")
    sys.meta_path.insert(0, _Finder())
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import importlib
    tf = importlib.import_module("tensorflow")
    for sub in ("nn", "math", "initializers", "compat", "compat.v1", "compat.v1.initializers", "python",
                "python.ops", "python.ops.gen_nn_ops", "python.keras", "python.keras.activations",
                "python.ops.initializers_ns"):
        m = importlib.import_module("tensorflow." + sub)
        parent = tf
        parts = sub.split(".")
        for p in parts[:-1]:
            parent = getattr(parent, p)
        setattr(parent, parts[-1], m)
    importlib.import_module("tf_slim")
    importlib.import_module("numba")
    np.int = int  # the reference's own shim (common/common_nn_ops.py:21) for numpy >= 1.24
    _INSTALLED[0] = True


class use_engine:
    def __init__(self, engine):
        self.engine = engine

    def __enter__(self):
        ENGINE[0] = self.engine
        _ARG_STACK[:] = [{}]
        return self.engine

    def __exit__(self, *exc):
        ENGINE[0] = None
        return False
