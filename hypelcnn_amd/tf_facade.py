"""`tensorflow` / `tf_slim` facade onto `hypelcnn_amd.graph`: the reference's UNCHANGED model plugin files record a product Tower.

The reference's plugins (`nnmodel/HYPELCNNModel.py:3-6`, `DUALCNNModel.py`, `CONCNNModel.py`, `gan/shadow_data_models.py`) import
`tensorflow` and `tf_slim`.  `install()` puts a meta-path finder in front of the import system that serves those names (only
where the real packages are absent) with the few dozen functions the plugin files touch -- `tf_slim.conv2d / fully_connected /
convolution1d / dropout / flatten / batch_norm / l2_regularizer / arg_scope`, `tf.concat / split / gather / repeat / squeeze /
expand_dims / nn.local_response_normalization / math.l2_normalize`, the initialisers, tensors with `get_shape()[i].value`, `+`
and slicing -- each forwarding to the ENGINE in force.  `GraphEngine` is the product's engine: every call lands in
`hypelcnn_amd.graph` (conv2d, fully_connected, arg_scope semantics, batch norm, dropout, ChanMap, crops, LRN, and the
`tf.nn.softmax_cross_entropy_with_logits` / `tf.reshape` / `tf.square` / `tf.reduce_mean` calls of `get_loss_func`).
The hijack is scoped: `installed(checkout)` is a context manager, `reference_model` enters it only while the reference's
code is imported or executed, and only names that no installed package serves are stubbed.

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

STUB_ROOTS = ("tensorflow", "tf_slim", "tensorflow_gan", "tifffile", "tqdm", "numba", "sklearn", "optuna", "matplotlib")  # only those that are ABSENT get stubbed


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

        def __sub__(self, other):  # image_output - reshape(image_original, [-1, F])  (HYPELCNNModel.py:108)
            if not isinstance(other, GraphEngine.Reshaped):
                raise RuntimeError("tensor subtraction is only the reconstruction error's on this path")
            return GraphEngine.Diff(self, other)

    class Reshaped:
        """tf.reshape(patch tensor, [-1, H * W * C]) -- the flattened original of the reconstruction loss."""

        def __init__(self, t):
            self.t = t

    class Diff:
        def __init__(self, a, b):
            self.a, self.b = a, b

    class Squared:
        def __init__(self, d):
            self.d = d

    def wrap(self, sym):
        return GraphEngine.T(self, sym)

    # -- the classifier plugins' loss (get_loss_func): the product's loss expressions
    def softmax_xent(self, labels, logits):
        return self.G.softmax_cross_entropy_with_logits(labels=labels, logits=logits.sym)

    def loss_reshape(self, tensor, shape):
        feats = tensor.sym.npix * tensor.sym.c
        if len(shape) != 2 or _int(shape[0]) != -1 or _int(shape[1]) != feats:
            raise RuntimeError(f"tf.reshape to {shape}: only the flattening [-1, {feats}] of the reconstruction loss is on the path")
        return GraphEngine.Reshaped(tensor)

    def loss_square(self, x):
        if not isinstance(x, GraphEngine.Diff):
            raise RuntimeError("tf.square is only the reconstruction error's on this path")
        return GraphEngine.Squared(x)

    def loss_reduce_mean(self, x):
        if not isinstance(x, GraphEngine.Squared):
            raise RuntimeError("tf.reduce_mean is only the reconstruction error's on this path")
        return self.G.mean_squared_reconstruction(x.d.a.sym, x.d.b.t.sym)

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


# The four tf.* calls of the classifier plugins' get_loss_func (nnmodel/HYPELCNNModel.py:101-112, DUALCNNModel.py:87-89,
# CONCNNModel.py:66-68): softmax cross entropy per sample, + the scalar mean of squares of (image_output -
# reshape(image_original, [-1, F])) broadcast onto every sample.  Served through the engine like the layers.
def loss_softmax_xent(labels=None, logits=None, **kw):
    return ENGINE[0].softmax_xent(labels, logits)


def loss_reshape(tensor, shape, **kw):
    return ENGINE[0].loss_reshape(tensor, shape)


def loss_square(x, **kw):
    return ENGINE[0].loss_square(x)


def loss_reduce_mean(input_tensor=None, axis=None, **kw):
    assert axis is None
    return ENGINE[0].loss_reduce_mean(input_tensor)


def _absent(roots):
    """The names among `roots` that no real installed package serves (checked with our finder out of the way)."""
    import importlib.util
    out = []
    for r in roots:
        if r in sys.modules and not isinstance(sys.modules[r], _Anything):
            continue
        try:
            spec = importlib.util.find_spec(r)
        except (ImportError, ValueError):
            spec = None
        if spec is None or isinstance(spec.loader, _Finder):
            out.append(r)
    return tuple(out)


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """Serves the root packages in `roots` (and everything below them).  EXTRA_SETUP: callables(module) that test
    infrastructure may add to give more stub modules a functional surface (tests/golden/tfgan_standin.py)."""
    EXTRA_SETUP = []

    def __init__(self, roots):
        self.roots = tuple(roots)

    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in self.roots:
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
            module.reshape, module.square, module.reduce_mean = loss_reshape, loss_square, loss_reduce_mean
        elif n == "tensorflow.nn":
            module.local_response_normalization = local_response_normalization
            module.softmax_cross_entropy_with_logits = loss_softmax_xent
        elif n == "tensorflow.math":
            module.l2_normalize = l2_normalize
        elif n in ("tensorflow.initializers", "tensorflow.compat.v1.initializers"):
            module.variance_scaling = variance_scaling
            module.zeros = zeros_initializer
        elif n == "tensorflow.compat.v1":
            module.name_scope = _Ctx
        elif n == "tensorflow.compat.v1.losses":
            module.add_loss = lambda *a, **k: None  # the classifier's train op differentiates the returned loss only
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
        for fn in _Finder.EXTRA_SETUP:
            fn(module)


_TF_SUBMODULES = ("nn", "math", "initializers", "compat", "compat.v1", "compat.v1.initializers", "compat.v1.losses", "python",
                  "python.ops", "python.ops.gen_nn_ops", "python.keras", "python.keras.activations",
                  "python.ops.initializers_ns")


class installed:
    """Context manager around IMPORTING and EXECUTING the reference's files: inside it `tensorflow`, `tf_slim` and the other
    imports of the reference that no installed package serves resolve to this module's facade, the reference checkout is
    importable, and `np.int` exists (the reference's own shim for numpy >= 1.24, common/common_nn_ops.py:21).  On exit
    everything is undone -- the finder leaves sys.meta_path, the checkout leaves sys.path, the stub modules AND the modules
    imported from the checkout (`common`, `nnmodel`, `gan`, ...) leave sys.modules (they are kept inside this object and come
    back on the next entry), modules they had shadowed are restored.  A process that builds a model through
    `reference_model()` can therefore go on importing sklearn / tqdm / its own `common` package as if nothing had happened
    (round-5 advisor finding: the hijack used to be permanent and stubbed installed packages)."""

    def __init__(self, reference_root=None):
        if reference_root is not None and not os.path.isdir(reference_root):
            raise RuntimeError(f"no reference checkout at {reference_root}")
        self.root = None if reference_root is None else os.path.abspath(reference_root)
        self.modules = {}     # ours: stub modules + modules loaded from the checkout
        self.finder = None
        self.depth = 0

    def _ours(self, m):
        if isinstance(m, _Anything):
            return True
        f = getattr(m, "__file__", None) or ""
        if self.root is not None and f and os.path.abspath(f).startswith(self.root + os.sep):
            return True
        # namespace packages of the checkout (directories without __init__.py) carry only __path__
        paths = [os.path.abspath(q) for q in list(getattr(m, "__path__", []) or [])] if f == "" else []
        return self.root is not None and any(q.startswith(self.root) for q in paths)

    def __enter__(self):
        self.depth += 1
        if self.depth > 1:
            return self
        import importlib
        if self.finder is None:
            self.finder = _Finder(_absent(STUB_ROOTS))
        self.shadowed = {k: sys.modules.pop(k) for k in list(self.modules) if k in sys.modules}
        sys.modules.update(self.modules)
        sys.meta_path.insert(0, self.finder)
        self.path_added = self.root is not None and self.root not in sys.path
        if self.path_added:
            sys.path.insert(0, self.root)
        self.np_int_added = not hasattr(np, "int")
        if self.np_int_added:
            np.int = int
        if "tensorflow" in self.finder.roots and "tensorflow" not in self.modules:
            # submodules are pre-imported so that `tf.nn.x` / `tf.compat.v1.y` resolve to the functional pieces
            tf = importlib.import_module("tensorflow")
            for sub in _TF_SUBMODULES:
                m = importlib.import_module("tensorflow." + sub)
                parent = tf
                parts = sub.split(".")
                for q in parts[:-1]:
                    parent = getattr(parent, q)
                setattr(parent, parts[-1], m)
            for r in ("tf_slim", "numba"):
                if r in self.finder.roots:
                    importlib.import_module(r)
        return self

    def __exit__(self, *exc):
        self.depth -= 1
        if self.depth > 0:
            return False
        for k, m in list(sys.modules.items()):
            if m is not None and self._ours(m):
                self.modules[k] = sys.modules.pop(k)
        sys.modules.update(self.shadowed)
        self.shadowed = {}
        if self.finder in sys.meta_path:
            sys.meta_path.remove(self.finder)
        if self.path_added and self.root in sys.path:
            sys.path.remove(self.root)
        if self.np_int_added and hasattr(np, "int"):
            del np.int
        return False


_PERMANENT = []


def install(reference_root=None):
    """Process-wide, never undone: for the fixture generators and subprocess tests under tests/golden (a fresh interpreter
    per run).  Product code uses `installed(...)` / `reference_model(...)`, which clean up after themselves."""
    ctx = installed(reference_root)
    ctx.__enter__()
    _PERMANENT.append(ctx)
    return ctx


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
    """The NNModel contract (`nnmodel/NNModel.py`) served by the REFERENCE's plugin file, whole: `create_tensor_graph` executes
    the reference's text through the facade into the Tower its input tensor belongs to, and `get_loss_func` executes the
    reference's own (`nnmodel/HYPELCNNModel.py:101-112`, `DUALCNNModel.py:87-89`, `CONCNNModel.py:66-68`: its `tf.nn` /
    `tf.reshape` / `tf.square` / `tf.reduce_mean` calls land on the product's loss expressions).  The import hijack lives only
    while the reference's code is being imported or executed (`installed`)."""

    def __init__(self, model_name, reference_root):
        import importlib
        from .common import common_nn_ops as P
        self._ctx = installed(reference_root)
        with self._ctx:
            mod = importlib.import_module("nnmodel." + model_name)
            if not os.path.abspath(getattr(mod, "__file__", "") or "").startswith(os.path.abspath(reference_root) + os.sep):
                raise RuntimeError(f"nnmodel.{model_name} resolved to {getattr(mod, '__file__', None)}, not to the reference checkout")
            self.reference = getattr(mod, model_name)()
            self.ref_ops = importlib.import_module("common.common_nn_ops")  # the reference's own value objects
        self.product_ops = P
        self.name = model_name

    def create_tensor_graph(self, model_input_params, class_count, algorithm_params):
        x = model_input_params.x
        eng = GraphEngine(x.tower)
        with self._ctx, use_engine(eng):
            out = self.reference.create_tensor_graph(
                self.ref_ops.ModelInputParams(x=eng.wrap(x), y=None, device_id=model_input_params.device_id,
                                              is_training=model_input_params.is_training), class_count, algorithm_params)
        sym = lambda t: None if t is None else t.sym  # noqa: E731
        return self.product_ops.ModelOutputTensors(
            y_conv=out.y_conv.sym, image_output=sym(getattr(out, "image_output", None)),
            image_original=sym(getattr(out, "image_original", None)), histogram_tensors=[])

    def get_loss_func(self, tensor_output, label):
        eng = GraphEngine(tensor_output.y_conv.tower)
        wrap = lambda t: None if t is None else eng.wrap(t)  # noqa: E731
        with self._ctx, use_engine(eng):
            out = self.ref_ops.ModelOutputTensors(y_conv=wrap(tensor_output.y_conv), image_output=wrap(tensor_output.image_output),
                                                  image_original=wrap(tensor_output.image_original), histogram_tensors=[])
            return self.reference.get_loss_func(out, label)


def reference_model(model_name, reference_root):
    return ReferenceModel(model_name, reference_root)
