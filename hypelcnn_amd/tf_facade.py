"""`tensorflow` / `tf_slim` facade onto `hypelcnn_amd.graph`: the reference's UNCHANGED model plugin files record a product Tower.

The reference's plugins (`nnmodel/HYPELCNNModel.py:3-6`, `DUALCNNModel.py`, `CONCNNModel.py`, `gan/shadow_data_models.py`) import
`tensorflow` and `tf_slim`.  `install()` puts a meta-path finder in front of the import system that serves those names (only
where the real packages are absent) with the few dozen functions the plugin files touch -- `tf_slim.conv2d / fully_connected /
convolution1d / dropout / flatten / batch_norm / l2_regularizer / arg_scope`, `tf.concat / split / gather / repeat / squeeze /
expand_dims / nn.local_response_normalization / math.l2_normalize`, the initialisers, tensors with `get_shape()[i].value`, `+`
and slicing -- each forwarding to the ENGINE in force.  `GraphEngine` is the product's engine: every call lands in
`hypelcnn_amd.graph` (conv2d, fully_connected, arg_scope semantics, batch norm, dropout, ChanMap, crops, LRN).

    from hypelcnn_amd import tf_facade
    model = tf_facade.reference_model("HYPELCNNModel", "/path/to/hypelcnn")     # the reference checkout, unchanged
    # ... use `model` wherever common_nn_ops.get_model_from_name("HYPELCNNModel") is used (create_graph / optimize_nn)

`tests/test_reference_wiring.py` holds the Towers recorded this way to the product's own plugins node for node, and runs a
training step through both (kernel emulation) to identical numbers.  Test infrastructure adds a second engine on the same
surface (`tests/golden/tf_standin.py::OracleEngine`: records and evaluates with the float64 oracle); nothing here imports it.
"""
import importlib.abc
import importlib.machinery
import math
import os
import sys
import types

import numpy as np

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



# ------------------------------------------------------------------------------------------------ the product's engine
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


def install(reference_root=None):
    """Serve tensorflow / tf_slim (and the other absent imports of the reference) from this module; put the reference
    checkout on sys.path when given.  Submodules are pre-imported so that `tf.nn.x` / `tf.compat.v1.y` resolve to the
    functional pieces."""
    if reference_root is not None:
        if not os.path.isdir(reference_root):
            raise RuntimeError(f"no reference checkout at {reference_root}")
        if reference_root not in sys.path:
            sys.path.insert(0, reference_root)
    if _INSTALLED[0]:
        return
    sys.meta_path.insert(0, _Finder())
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


class ReferenceModel:
    """The NNModel contract (`nnmodel/NNModel.py`) served by the REFERENCE's plugin file: `create_tensor_graph` executes the
    reference's text through the facade into the Tower its input tensor belongs to; `get_loss_func` is the product plugin's
    (the reference's builds its loss from `tf.nn` / `tf.losses` calls, which are not part of the facade)."""

    def __init__(self, model_name, reference_root):
        install(reference_root)
        import importlib
        from .common import common_nn_ops as P
        mod = importlib.import_module("nnmodel." + model_name)
        if not getattr(mod, "__file__", "").startswith(os.path.abspath(reference_root)):
            raise RuntimeError(f"nnmodel.{model_name} resolved to {getattr(mod, '__file__', None)}, not to the reference checkout")
        self.reference = getattr(mod, model_name)()
        self.product = P.get_model_from_name(model_name)
        self.name = model_name

    def create_tensor_graph(self, model_input_params, class_count, algorithm_params):
        from .common import common_nn_ops as P
        import importlib
        ref_ops = importlib.import_module("common.common_nn_ops")  # the reference's own value objects
        x = model_input_params.x
        eng = GraphEngine(x.tower)
        with use_engine(eng):
            out = self.reference.create_tensor_graph(
                ref_ops.ModelInputParams(x=eng.wrap(x), y=None, device_id=model_input_params.device_id,
                                         is_training=model_input_params.is_training), class_count, algorithm_params)
        sym = lambda t: None if t is None else t.sym  # noqa: E731
        return P.ModelOutputTensors(y_conv=out.y_conv.sym, image_output=sym(getattr(out, "image_output", None)),
                                    image_original=sym(getattr(out, "image_original", None)), histogram_tensors=[])

    def get_loss_func(self, *a, **k):
        return self.product.get_loss_func(*a, **k)


def reference_model(model_name, reference_root):
    return ReferenceModel(model_name, reference_root)
