"""Graph assembly, optimiser wiring, metrics, iterators, patch extraction -- the build's mirror of
the reference's common/common_nn_ops.py (same public names; file:line cited per function).

What changes underneath: "tensors" are symbolic nodes of hypelcnn_amd.graph, a training step is a
pre-planned list of HIP launches (hypelcnn_amd.plan / runtime) and datasets stay resident in HBM.
"""
import numpy
import torch

from hypelcnn_amd import graph as G
from hypelcnn_amd.common.common_ops import get_class, is_integer_num

INVALID_TARGET_VALUE = 255


# ----------------------------------------------------------------------------- data sets
class DataSet:
    """reference :23-42"""

    def get_data_shape(self):
        raise NotImplementedError

    def get_casi_band_count(self):
        raise NotImplementedError

    def get_scene_shape(self):
        raise NotImplementedError

    def get_unnormalized_casi_dtype(self):
        raise NotImplementedError

    def get_data_point(self, point_x, point_y):
        raise NotImplementedError


def get_data_point_func(casi, lidar, neighborhood, point_x, point_y):
    """reference :169-176 -- the padded scene makes the window start at (x, y)."""
    side = 2 * neighborhood + 1
    win = (slice(point_y, point_y + side), slice(point_x, point_x + side))
    return numpy.concatenate((casi[win], lidar[win]), axis=2)


def get_data_point_func_hsi(casi, lidar, neighborhood, point_x, point_y):
    """reference :180-185"""
    side = 2 * neighborhood + 1
    return casi[point_y:point_y + side, point_x:point_x + side, :]


class BasicDataSet(DataSet):
    """Symmetric padding + min/max normalisation + patch slicing (reference :45-106)."""

    def __init__(self, shadow_creator_dict, casi, lidar, neighborhood, normalize, casi_min=None, casi_max=None,
                 lidar_min=None, lidar_max=None):
        self.neighborhood = neighborhood
        self.shadow_creator_dict = shadow_creator_dict
        self.casi_unnormalized_dtype = casi.dtype
        pad = ((neighborhood, neighborhood), (neighborhood, neighborhood), (0, 0))
        self.lidar = None if lidar is None else numpy.pad(lidar, pad, mode="symmetric")
        self.casi = None if casi is None else numpy.pad(casi, pad, mode="symmetric")
        self.casi_min, self.casi_max, self.lidar_min, self.lidar_max = 0, 1, 0, 1
        if normalize:
            if self.lidar is not None:
                self.lidar_min = numpy.min(self.lidar) if lidar_min is None else lidar_min
                self.lidar = self.lidar - self.lidar_min
                self.lidar_max = numpy.max(self.lidar) if lidar_max is None else lidar_max
                self.lidar = self.lidar / self.lidar_max
            if self.casi is not None:
                self.casi_min = numpy.min(self.casi, axis=(0, 1)) if casi_min is None else casi_min
                self.casi = self.casi - self.casi_min
                self.casi_max = numpy.max(self.casi, axis=(0, 1)) if casi_max is None else casi_max
                self.casi = self.casi / self.casi_max.astype(numpy.float32)
        self._get_data_point_func = get_data_point_func if self.lidar is not None else get_data_point_func_hsi

    def get_data_shape(self):
        side = self.neighborhood * 2 + 1
        return [side, side, self.casi.shape[2] + (1 if self.lidar is not None else 0)]

    def get_casi_band_count(self):
        return self.casi.shape[2]

    def get_scene_shape(self):
        ref = self.lidar if self.lidar is not None else self.casi
        return [ref.shape[0] - 2 * self.neighborhood, ref.shape[1] - 2 * self.neighborhood]

    def get_unnormalized_casi_dtype(self):
        return self.casi_unnormalized_dtype

    def get_data_point(self, point_x, point_y):
        return self._get_data_point_func(self.casi, self.lidar, self.neighborhood, point_x, point_y)


# ----------------------------------------------------------------------------- value objects (reference :109-165)
class NNParams:
    def __init__(self, input_iterator, data_with_labels, metrics, predict_tensor):
        self.predict_tensor = predict_tensor
        self.metrics = metrics
        self.data_with_labels = data_with_labels
        self.input_iterator = input_iterator


class ModelInputParams:
    def __init__(self, x, y, device_id, is_training):
        self.is_training = is_training
        self.device_id = device_id
        self.y = y
        self.x = x


class HistogramTensorPair:
    def __init__(self, tensor, name):
        self.name = name
        self.tensor = tensor


class ModelOutputTensors:
    def __init__(self, y_conv, image_output, image_original, histogram_tensors):
        self.image_original = image_original
        self.image_output = image_output
        self.y_conv = y_conv
        self.histogram_tensors = histogram_tensors


class TrainingResult:
    def __init__(self, validation_accuracy, test_accuracy, loss):
        self.loss = loss
        self.test_accuracy = test_accuracy
        self.validation_accuracy = validation_accuracy


class AugmentationInfo:
    def __init__(self, shadow_struct, perform_shadow_augmentation, perform_rotation_augmentation,
                 perform_spectral_augmentation, perform_reflection_augmentation, augmentation_random_threshold):
        self.perform_reflection_augmentation = perform_reflection_augmentation
        self.perform_rotation_augmentation = perform_rotation_augmentation
        self.perform_shadow_augmentation = perform_shadow_augmentation
        self.perform_spectral_augmentation = perform_spectral_augmentation
        self.shadow_struct = shadow_struct
        self.augmentation_random_threshold = augmentation_random_threshold


NO_AUGMENTATION = AugmentationInfo(None, False, False, False, False, 0.0)


# ----------------------------------------------------------------------------- residual channel map
def scale_in_to_out(input_data, output_data, axis_no):
    """Weight-free residual shortcut with channel-count matching (reference :546-564): identity,
    integer repeat, or gather with min(round(o*Cin/Cout), Cin-1) (Python round = half to even).
    Returns a ChanMap; `tensor + chan_map` fuses the gather into the producer's epilogue."""
    cin, cout = input_data.c, output_data.c
    scale_ratio = cin / cout
    inv_scale_ratio = 1 / scale_ratio
    if is_integer_num(inv_scale_ratio):
        rep = int(inv_scale_ratio)
        idx = None if rep == 1 else numpy.arange(cout, dtype=numpy.int32) // rep
    else:
        idx = numpy.asarray([min(round(o * scale_ratio), cin - 1) for o in range(cout)], dtype=numpy.int32)
    return G.ChanMap(input_data, idx)


# ----------------------------------------------------------------------------- template / placeholders
class Placeholder:
    """A batch-shaped input fed by an iterator: [None, H, W, C] images or [None, classes] labels."""

    def __init__(self, name, hw, c):
        self.name, self.hw, self.c = name, hw, c
        self._bound = {}

    def bind(self, tower):
        t = self._bound.get(id(tower))
        if t is None:
            t = tower.placeholder(self.name, self.hw, self.c)
            self._bound[id(tower)] = t
        return t

    def get_shape(self):
        return [None, self.c] if self.hw is None else [None, self.hw[0], self.hw[1], self.c]


class Template:
    """tf.compat.v1.make_template("nn_core", model.create_tensor_graph, class_count=...) (reference :333):
    every call records a new tower that shares the variables of the first one."""

    def __init__(self, name, fn, **bound):
        self.store = G.VariableStore(name)
        self.fn = fn
        self.bound = bound
        self.towers = []

    def __call__(self, model_input_params, algorithm_params):
        tower = G.Tower(self.store, model_input_params.is_training, name=f"tower{len(self.towers)}")
        self.towers.append(tower)
        x = model_input_params.x
        if isinstance(x, Placeholder):
            x = x.bind(tower)
        mip = ModelInputParams(x=x, y=model_input_params.y, device_id=model_input_params.device_id,
                               is_training=model_input_params.is_training)
        out = self.fn(mip, algorithm_params=algorithm_params, **self.bound)
        out.tower = tower
        return out


def _bind_labels(labels, tower, classes):
    if isinstance(labels, Placeholder):
        return labels.bind(tower)
    return labels


# ----------------------------------------------------------------------------- iterators (reference :188-205)
class DeviceArrays:
    """A dataset resident on the compute device: float32 [N,P,P,C] patches + uint8 [N] labels."""

    def __init__(self):
        self.data = None
        self.labels = None

    def feed(self, data, labels, device):
        self.data = torch.as_tensor(numpy.ascontiguousarray(data), dtype=torch.float32).to(device)
        self.labels = torch.as_tensor(numpy.ascontiguousarray(labels)).to(torch.int64).to(device)

    def __len__(self):
        return 0 if self.data is None else self.data.shape[0]


class SceneArrays:
    """A data set that is never materialised: the padded, normalised scene resident in HBM plus the target list.
    A batch is cut by ONE hypel_gather_patches_f32 launch (reference: GeneratorImporter._iterator_function, one
    Python `get_data_point` per sample; importer/GeneratorImporter.py:19-21)."""

    def __init__(self):
        self.casi = self.lidar = self.points = self.labels = None
        self.shape = None

    def feed(self, data_set, targets, backend):
        dev = backend.device
        self.backend = backend
        casi = numpy.ascontiguousarray(data_set.casi, dtype=numpy.float32)
        self.casi = torch.from_numpy(casi).to(dev)
        self.lidar = None
        if data_set.lidar is not None:
            self.lidar = torch.from_numpy(numpy.ascontiguousarray(data_set.lidar, dtype=numpy.float32)).to(dev)
        t = numpy.asarray(targets)
        self.points = torch.from_numpy(numpy.ascontiguousarray(t[:, :2], dtype=numpy.int32)).to(dev)
        self.labels = torch.from_numpy(numpy.ascontiguousarray(t[:, 2]).astype(numpy.int64)).to(dev)
        self.shape = tuple(data_set.get_data_shape())
        self.casi_scale = int(getattr(data_set, "casi_scale", 1))  # 2: GRSS2018 (HSI at half the LiDAR resolution)
        self.neighborhood = int(data_set.neighborhood)

    def __len__(self):
        return 0 if self.points is None else self.points.shape[0]

    def gather(self, idx):
        from hypelcnn_amd.backend import Ref
        b = int(idx.shape[0])
        p, _, c = self.shape
        pts = self.points.index_select(0, idx).contiguous()
        out = torch.empty((b, p, p, c), dtype=torch.float32, device=self.casi.device)
        hp, wp, cc = self.casi.shape
        cl = 0 if self.lidar is None else int(self.lidar.shape[2])
        if self.casi_scale == 2:
            self.backend.call("gather_patches_2x_f32", Ref(self.casi.reshape(-1)), Ref(self.lidar.reshape(-1)), int(wp),
                              int(self.lidar.shape[1]), int(cc), cl, self.neighborhood, Ref(pts.reshape(-1)), b, int(p),
                              Ref(out.reshape(-1)))
            return out, pts
        self.backend.call("gather_patches_f32", Ref(self.casi.reshape(-1)),
                          None if self.lidar is None else Ref(self.lidar.reshape(-1)), int(hp), int(wp), int(cc), cl,
                          Ref(pts.reshape(-1)), b, int(p), Ref(out.reshape(-1)))
        return out, pts


class BatchIterator:
    """make_initializable_iterator over shuffle_and_repeat / map / batch (training) or batch (eval).

    Deviation (documented in DESIGN.md): TF's 10 000-element shuffle buffer and op-level RNG streams
    are not reproducible; each epoch is a fresh permutation drawn from a generator seeded 1234
    (monitored_session_runner.set_run_seed), batches are cut from the concatenated epochs and the
    final short batch is kept, as tf.data.batch(drop_remainder=False) does."""

    def __init__(self, element_shape, class_count, batch_size, shuffle, num_epochs, augmentation_info=None, seed=1234):
        self.element_shape = tuple(element_shape)
        self.class_count = class_count
        self.batch_size = int(batch_size)
        self.shuffle = shuffle
        self.num_epochs = num_epochs
        self.augmentation_info = augmentation_info
        self.arrays = DeviceArrays()
        self.seed = seed
        self.images = Placeholder("x", (element_shape[0], element_shape[1]), element_shape[2])
        self.labels = Placeholder("labels", None, class_count)
        self._order = None
        self._pos = 0
        self._epoch = 0
        self._gen = None
        self.collective = False  # True: every batch ends in a data-parallel collective (training_nn_iterator)
        self.last_global_count = None  # samples of the global batch the last next_batch() was a shard of

    def get_next(self):
        return self.images, self.labels

    # -- session side --
    def initializer(self, data, labels, backend):
        self.backend = backend
        self.arrays = DeviceArrays()
        self.arrays.feed(data, labels, backend.device)
        self._reset()

    def initializer_scene(self, data_set, targets, backend):
        """Generator-style data set (importer/GeneratorImporter.py): patches are cut from the resident scene."""
        self.backend = backend
        self.arrays = SceneArrays()
        self.arrays.feed(data_set, targets, backend)
        self._reset()

    def _reset(self):
        self._gen = torch.Generator(device="cpu")  # epoch permutations: identical on every rank (see next_batch)
        self._gen.manual_seed(self.seed)
        # augmentation draws: a stream of their own, per rank -- drawn ON THE DEVICE the batch lives on: on a 256-thread
        # host a handful of tiny CPU torch.rand calls costs 10 ms per batch (measured), more than the whole train step
        dev = getattr(getattr(self, "backend", None), "device", None)
        self._aug_gen = torch.Generator(device=dev if dev is not None and dev.type == "cuda" else "cpu")
        self._aug_gen.manual_seed(self.seed + 7919 * (self._dp()[1] + 1))
        self._epoch = 0
        self._pos = 0
        self._order = self._new_epoch()

    @staticmethod
    def _dp():
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return dist.get_world_size(), dist.get_rank()
        return 1, 0

    def _new_epoch(self):
        n = len(self.arrays)
        if self.shuffle:
            return torch.randperm(n, generator=self._gen).to(self.backend.device)
        return torch.arange(n, device=self.backend.device)

    def batch_shapes(self):
        """[(samples on this rank, samples in the global batch)] of every batch this iterator can produce: the full batch
        and, for an epoch-limited run, the ragged tail -- what a training loop needs to plan and capture BEFORE its
        first step (under data parallelism a HIP-graph capture next to live collectives costs a 0.5 s pause)."""
        world, rank = self._dp()
        shapes = [(self.batch_size, self.batch_size * world)]
        n = len(self.arrays) if self.arrays is not None else 0
        if n and self.num_epochs is not None:
            total = n * (self.num_epochs if self.shuffle else 1)
            tail = total % (self.batch_size * world)
            if tail and not (world > 1 and self.collective and tail < world):
                mine = len(range(rank, tail, world))
                if mine:
                    shapes.append((mine, tail))
        return shapes

    def next_batch(self):
        """Returns (x [b,P,P,C] float32, onehot [b,classes] float32, labels int64) or None when exhausted."""
        n = len(self.arrays)
        if n == 0:
            return None
        # data parallel: every rank walks the SAME seeded permutation, draws the global batch
        # (batch_size per rank x world) and keeps the samples rank::world of it -- disjoint shards, no communication
        world, rank = self._dp()
        idx_parts = []
        need = self.batch_size * world
        while need > 0:
            if self._pos >= n:
                self._epoch += 1
                if self.num_epochs is not None and self._epoch >= self.num_epochs:
                    break
                if not self.shuffle and self.num_epochs is not None:
                    break
                self._order = self._new_epoch()
                self._pos = 0
            take = min(need, n - self._pos)
            idx_parts.append(self._order[self._pos:self._pos + take])
            self._pos += take
            need -= take
        if not idx_parts:
            return None
        idx = torch.cat(idx_parts) if len(idx_parts) > 1 else idx_parts[0]
        self.last_global_count = int(idx.numel())
        if world > 1:
            if self.collective and idx.numel() < world:
                # Training steps end in collectives, so every rank must take part in every step or none: the ragged
                # tail of an epoch-limited run that is shorter than the world size (at most world - 1 samples, once,
                # at the very end) is dropped on ALL ranks -- each rank walks the same permutation and reaches this
                # decision without communication.  Longer tails are sharded unevenly and weighted by
                # nb_rank / nb_global in the loss gradient (TowerPlan._emit_loss).
                return None
            idx = idx[rank::world]
            if idx.numel() == 0:  # evaluation (no per-batch collective): this rank sits the batch out
                return None
        lab = self.arrays.labels.index_select(0, idx)
        augment = self.augmentation_info is not None and self.augmentation_info is not NO_AUGMENTATION
        self.last_points = None
        if isinstance(self.arrays, SceneArrays):
            x, self.last_points = self.arrays.gather(idx)
            if augment:
                x = apply_augmentations(self.backend, x, torch.arange(x.shape[0], device=x.device),
                                        self.augmentation_info, self._aug_gen)
        elif augment:
            x = apply_augmentations(self.backend, self.arrays.data, idx, self.augmentation_info, self._aug_gen)
        else:
            x = self.arrays.data.index_select(0, idx)
        onehot = torch.nn.functional.one_hot(lab, self.class_count).to(torch.float32)
        return x, onehot, lab


def training_nn_iterator(data_set, augmentation_info, batch_size, num_epochs, device, prefetch_size):
    """reference :188-201"""
    it = BatchIterator(data_set.element_shape, data_set.class_count, batch_size, True, num_epochs, augmentation_info)
    it.collective = True  # its batches feed train steps (gradient all-reduce under data parallelism)
    return it


def simple_nn_iterator(data_set, batch_size):
    """reference :204-205"""
    return BatchIterator(data_set.element_shape, data_set.class_count, batch_size, False, 1, None)


def draw_augmentations(b, c, info, gen):
    """Host half of the map stage (reference :376-440): the per-sample random decisions, drawn in the order the
    reference applies the maps -- rot90 by k in {0,1,2} (never 270 degrees, :402), shadow op with probability
    `augmentation_random_threshold`, left-right / up-down flips with p = 0.5, per-channel U(-s, 0) shift."""
    d = {}
    dev = gen.device  # the draws are made where the generator lives (the batch's device in the product iterators)
    if info.perform_rotation_augmentation:
        d["rot_k"] = torch.randint(0, 3, (b,), generator=gen, device=dev).to(torch.int32)
    if info.perform_shadow_augmentation and info.shadow_struct is not None:
        d["shadow_pick"] = (torch.rand(b, generator=gen, device=dev) < info.augmentation_random_threshold).to(torch.uint8)
    if info.perform_reflection_augmentation:
        d["flip_lr"] = (torch.rand(b, generator=gen, device=dev) < 0.5).to(torch.uint8)
        d["flip_ud"] = (torch.rand(b, generator=gen, device=dev) < 0.5).to(torch.uint8)
    if info.perform_spectral_augmentation:
        s = float(info.perform_spectral_augmentation)
        d["delta"] = (torch.rand(b, c, generator=gen, device=dev) * s - s).contiguous()
    return d


def apply_augmentations(backend, data, idx, info, gen):
    """Device half: ONE hypel_augment_patches_f32 launch gathers the batch `data[idx]` from the resident data set
    and applies every drawn map on the way.  data: [N,P,P,C] device tensor, idx: int64 device."""
    from hypelcnn_amd.backend import Ref
    b = int(idx.shape[0])
    _, p, _, c = data.shape
    d = draw_augmentations(b, c, info, gen)
    if not d:
        return data.index_select(0, idx)
    dev = {k: v.to(data.device) for k, v in d.items()}
    ratio = alt = None
    if "shadow_pick" in dev:
        r = getattr(info.shadow_struct, "ratio", None)
        if r is not None:
            ratio = torch.as_tensor(numpy.asarray(r, numpy.float32)).to(data.device)
        elif bool(d["shadow_pick"].any()):
            alt = info.shadow_struct.shadow_op(data.index_select(0, idx)).contiguous()
        else:
            del dev["shadow_pick"]
    out = torch.empty((b, p, p, c), dtype=torch.float32, device=data.device)

    def ref(t):
        return None if t is None else Ref(t.reshape(-1))

    backend.call("augment_patches_f32", ref(data), ref(idx), b, int(p), int(c), ref(dev.get("rot_k")),
                 ref(dev.get("shadow_pick")), ref(ratio), ref(alt), ref(dev.get("flip_lr")), ref(dev.get("flip_ud")),
                 ref(dev.get("delta")), ref(out))
    return out


# ----------------------------------------------------------------------------- optimiser wiring (reference :208-240)
class LearningRate:
    """exponential_decay(staircase=True): lr0 * rate ** (step // decay_steps) (reference :217-221)."""

    def __init__(self, base, decay_step, decay_factor):
        self.base, self.decay_step, self.decay_factor = base, decay_step, decay_factor

    def eval(self, step):
        return self.base * self.decay_factor ** (step // self.decay_step)


class GraphContext:
    """What tf.Graph + MonitoredTrainingSession hold in the reference: the variable store, the
    recorded towers and -- once `session()` is called -- the device session."""

    def __init__(self, template, backend=None):
        self.template = template
        self.backend = backend
        self._session = None
        self.external_masks = False
        self.seed = 1234
        self.capture_graphs = True

    def session(self):
        if self._session is None:
            from hypelcnn_amd.runtime import Session
            if self.backend is None:
                from hypelcnn_amd.backend import HipBackend
                self.backend = HipBackend()  # raises without a GPU / library: no CPU fallback
            self._session = Session(self.template.store, self.backend, seed=self.seed)
            self._session.finalize_variables()
            self._session.init_data_parallel()
        return self._session


class LossFetch:
    """The `cross_entropy` tensor of the reference: value of the last executed training step."""

    def __init__(self):
        self._compiled = None

    def eval(self):
        return float("nan") if self._compiled is None else self._compiled.loss_value()


class TrainOp:
    """tf_slim.learning.create_train_op(cross_entropy, optimizer, global_step) (reference :232):
    one call = next batch -> forward -> backward -> (DP all-reduce) -> optimiser -> global_step += 1."""

    def __init__(self, ctx, tower, loss, iterator, algorithm_params, learning_rate, loss_fetch):
        self.ctx = ctx
        self.tower = tower
        self.loss = loss
        self.iterator = iterator
        self.algorithm_params = algorithm_params
        self.learning_rate = learning_rate
        self.loss_fetch = loss_fetch
        opt = algorithm_params["optimizer"]
        self.momentum = opt[1] if isinstance(opt, (tuple, list)) and opt[0] == "MomentumOptimizer" else None
        if self.momentum is None and opt != "AdamOptimizer":
            raise ValueError(f"unknown optimizer {opt!r}")
        if self.momentum is not None:
            # known from the configuration, not from the first step: a step-0 checkpoint of a MomentumOptimizer run is
            # exported under <var>/nn_core/Momentum as well (tf_checkpoint.session_to_variables)
            ctx.session().optimizer_kind = "momentum"

    def compiled(self, nb, global_nb=None):
        sess = self.ctx.session()
        ct = sess.compile(self.tower, nb, loss=self.loss, external_masks=self.ctx.external_masks, global_nb=global_nb)
        if self.ctx.capture_graphs and getattr(sess.backend, "name", "") == "hip" and ct._graph_all is None:
            ct.capture()
        return ct

    def precapture(self):
        """Plan and capture every batch shape the iterator will deliver, in one burst (one watchdog pause under data
        parallelism: HipBackend.settle_before_capture), instead of paying for a new shape in the middle of the run."""
        shapes = self.iterator.batch_shapes() if hasattr(self.iterator, "batch_shapes") else []
        for nb, global_nb in shapes:
            self.compiled(nb, global_nb)

    def run(self, batch=None):
        sess = self.ctx.session()
        if batch is None:
            batch = self.iterator.next_batch()
            if batch is None:
                raise StopIteration
        x, onehot, _ = batch
        ct = self.compiled(x.shape[0], getattr(self.iterator, "last_global_count", None))
        ct.set_input("x", x)
        ct.set_input("labels", onehot)
        sess.train_step_exchange(ct)  # forward + backward (+ overlapped data-parallel gradient all-reduce)
        lr = self.learning_rate.eval(sess.global_step)
        if self.momentum is None:
            sess.adam_step(lr)
        else:
            sess.momentum_step(lr, self.momentum)
        self.loss_fetch._compiled = ct
        return ct


def optimize_nn(deep_nn_template, images, labels, device_id, name_prefix, algorithm_params, loss_func, ctx=None,
                iterator=None):
    """reference :208-240.  Gradients are those of the mean loss only -- the L2 regulariser declared on
    the HYPELCNN convolutions never reaches create_train_op in the reference."""
    tensor_outputs = deep_nn_template(
        model_input_params=ModelInputParams(x=images, y=labels, device_id=device_id, is_training=True),
        algorithm_params=algorithm_params)
    tower = tensor_outputs.tower
    bound_labels = _bind_labels(labels, tower, None)
    cross_entropy_expr = G.reduce_mean(loss_func(tensor_outputs, bound_labels))
    learning_rate = LearningRate(algorithm_params["learning_rate"], algorithm_params["learning_rate_decay_step"],
                                 algorithm_params["learning_rate_decay_factor"])
    cross_entropy = LossFetch()
    ctx = ctx or GraphContext(deep_nn_template)
    train_step = TrainOp(ctx, tower, cross_entropy_expr, iterator, algorithm_params, learning_rate, cross_entropy)
    return tensor_outputs.y_conv, cross_entropy, learning_rate, train_step


# ----------------------------------------------------------------------------- metrics (reference :243-310)
def _capture_forward(ctx, sess, ct):
    """Evaluation / inference towers replay one HIP graph per batch instead of ~100 launches (the host would
    otherwise be the bottleneck of full-scene inference)."""
    if (ctx is None or ctx.capture_graphs) and getattr(sess.backend, "name", "") == "hip" and not ct.bwd \
            and ct._graph_fwd is None:
        ct.capture()


class MetricOpsHolder:
    """Streaming OA / mean-per-class accuracy / Cohen kappa / confusion matrix.  The confusion matrix is
    accumulated on the device by hypel_argmax_confusion; the scalar metrics are its host-side functions
    (tf.metrics.accuracy, mean_per_class_accuracy (mean over ALL classes, div_no_nan), tf_slim cohen_kappa)."""

    def __init__(self, ctx, tower, y_conv, class_range, name_prefix):
        self.ctx = ctx
        self.tower = tower
        self.y_conv = y_conv
        self.num_classes = class_range.stop
        self.name_prefix = name_prefix
        self._confusion_dev = None
        self._labels_dev = {}

    def metric_variables_reset_op(self):
        be = self.ctx.session().backend
        if self._confusion_dev is None:
            self._confusion_dev = be.zeros(self.num_classes * self.num_classes, torch.int32)
        self._confusion_dev.zero_()

    def combined_metric_update_op(self, batch):
        from hypelcnn_amd.backend import Ref
        sess = self.ctx.session()
        x, _, lab = batch
        nb = x.shape[0]
        ct = sess.compile(self.tower, nb)
        _capture_forward(self.ctx, sess, ct)
        ct.set_input("x", x)
        ct.forward()
        st = ct.plan.storage_of(self.y_conv)
        lab32 = lab.to(torch.int32).contiguous()
        sess.backend.call("argmax_confusion", Ref(ct.plan.buffers[st.buf], st.ch_off), st.ld, nb, self.y_conv.c,
                          Ref(lab32), None, Ref(self._confusion_dev))
        self._keep = lab32
        return ct

    def allreduce(self):
        """Sum the int32 confusion matrices of the ranks (one small collective at the end of an evaluation)."""
        import torch.distributed as dist
        if self._confusion_dev is not None and dist.is_available() and dist.is_initialized() and \
                dist.get_world_size() > 1:
            from hypelcnn_amd.backend import note_collective
            note_collective()
            dist.all_reduce(self._confusion_dev, op=dist.ReduceOp.SUM)

    @property
    def confusion(self):
        k = self.num_classes
        return self._confusion_dev.detach().cpu().numpy().reshape(k, k).astype(numpy.int32)

    @property
    def accuracy(self):
        return confusion_metrics(self.confusion)[0]

    @property
    def mean_per_class_accuracy(self):
        return confusion_metrics(self.confusion)[1]

    @property
    def kappa(self):
        return confusion_metrics(self.confusion)[2]


def confusion_metrics(conf):
    conf = conf.astype(numpy.float64)
    total = conf.sum()
    if total == 0:
        return 0.0, 0.0, 0.0
    oa = numpy.trace(conf) / total
    rows = conf.sum(1)
    per_class = numpy.divide(numpy.diag(conf), rows, out=numpy.zeros_like(rows), where=rows > 0)
    aa = per_class.mean()
    pe = float((rows * conf.sum(0)).sum()) / (total * total)
    kappa = (oa - pe) / (1 - pe) if pe != 1 else 0.0
    return float(oa), float(aa), float(kappa)


def create_metric_tensors(labels, y_conv, class_range, name_prefix, ctx=None, tower=None):
    return MetricOpsHolder(ctx, tower if tower is not None else y_conv.tower, y_conv, class_range, name_prefix)


def calculate_class_accuracies_using_confusion(confusion_matrix, class_range):
    """reference :280-292: per-class recall and precision."""
    k = class_range.stop
    class_precisions, class_recall = numpy.zeros(k), numpy.zeros(k)
    for index in class_range:
        truths = numpy.sum(confusion_matrix[index, :])
        if truths != 0:
            class_recall[index] = confusion_matrix[index, index] / truths
        predictions = numpy.sum(confusion_matrix[:, index])
        if predictions != 0:
            class_precisions[index] = confusion_matrix[index, index] / predictions
    return class_recall[class_range], class_precisions[class_range]


def calculate_accuracy(sess, nn_params, class_range):
    """reference :295-310: reset the streaming metrics, drain the iterator, read the metrics."""
    m = nn_params.metrics
    m.metric_variables_reset_op()
    while True:
        batch = nn_params.input_iterator.next_batch()
        if batch is None:
            break
        m.combined_metric_update_op(batch)
    m.allreduce()  # data parallel: every rank evaluated its shard of the batches
    confusion_matrix = m.confusion
    overall_accuracy, mean_per_class_accuracy, kappa = confusion_metrics(confusion_matrix)
    class_recall, class_precisions = calculate_class_accuracies_using_confusion(confusion_matrix, class_range)
    return overall_accuracy, class_recall, class_precisions, kappa, mean_per_class_accuracy


def perform_prediction(sess, nn_params, prediction_result, margin_result=None):
    """reference :313-327: drain the iterator, argmax the logits, write the class of every target into the
    [H, W] uint8 raster at (row = y, col = x).  The per-sample Python loop of the reference is ONE
    hypel_argmax_scatter launch per batch into a device-resident raster, copied back once at the end.
    `margin_result` (optional float32 [H, W], not in the reference): receives the gap between the two largest
    logits of every classified pixel -- the confidence map the label-parity tests use to tell a genuine label
    difference from a tie within fp32 rounding."""
    from hypelcnn_amd.backend import Ref
    it = nn_params.input_iterator
    y_conv = nn_params.predict_tensor
    h, w = prediction_result.shape
    raster = torch.from_numpy(numpy.ascontiguousarray(prediction_result, dtype=numpy.uint8)).to(sess.backend.device)
    flat = raster.reshape(-1)
    targets = numpy.asarray(nn_params.data_with_labels.targets)
    done = 0
    while True:
        batch = it.next_batch()
        if batch is None:
            break
        x = batch[0]
        nb = x.shape[0]
        ct = sess.compile(y_conv.tower, nb)
        _capture_forward(None, sess, ct)
        ct.set_input("x", x)
        ct.forward()
        st = ct.plan.storage_of(y_conv)
        pts = it.last_points
        if pts is None:  # resident-patch data set: the i-th sample of the epoch is the i-th target
            pts = torch.from_numpy(numpy.ascontiguousarray(targets[done:done + nb, :2], dtype=numpy.int32)).to(
                sess.backend.device)
        sess.backend.call("argmax_scatter", Ref(ct.plan.buffers[st.buf], st.ch_off), st.ld, nb, y_conv.c,
                          Ref(pts.reshape(-1)), Ref(flat), int(w))
        if margin_result is not None:
            top2 = torch.topk(ct.value(y_conv).reshape(nb, -1), 2, dim=1).values
            p = pts.reshape(-1, 2).cpu().numpy()
            margin_result[p[:, 1], p[:, 0]] = (top2[:, 0] - top2[:, 1]).cpu().numpy()
        done += nb
    prediction_result[...] = raster.cpu().numpy()
    return done


def create_colored_image(target_image, color_list):
    """reference :455-462 (vectorised): classes beyond the colour list stay black."""
    colors = numpy.asarray(color_list, dtype=numpy.uint8).reshape(-1, 3)
    out = numpy.zeros([target_image.shape[0], target_image.shape[1], 3], dtype=numpy.uint8)
    valid = target_image < len(colors)
    out[valid] = colors[target_image[valid]]
    return out


# ----------------------------------------------------------------------------- graph assembly (reference :330-373)
def create_graph(training_data_set, testing_data_set, validation_data_set, class_range, batch_size, prefetch_size,
                 device_id, num_epochs, algorithm_params, model, augmentation_info, create_separate_validation_branch,
                 backend=None):
    deep_nn_template = Template("nn_core", model.create_tensor_graph, class_count=class_range.stop)
    ctx = GraphContext(deep_nn_template, backend)

    training_input_iter = training_nn_iterator(training_data_set, augmentation_info, batch_size, num_epochs,
                                               device_id, prefetch_size)
    images, labels = training_input_iter.get_next()
    training_y_conv, cross_entropy, learning_rate, train_step = optimize_nn(
        deep_nn_template, images, labels, device_id=device_id, name_prefix="training",
        algorithm_params=algorithm_params, loss_func=model.get_loss_func, ctx=ctx, iterator=training_input_iter)
    train_nn_params = NNParams(input_iterator=training_input_iter, data_with_labels=None, metrics=None,
                               predict_tensor=None)

    def eval_branch(data_set, prefix):
        it = simple_nn_iterator(data_set, batch_size)
        imgs, labs = it.get_next()
        outs = deep_nn_template(ModelInputParams(x=imgs, y=None, device_id=device_id, is_training=False),
                                algorithm_params=algorithm_params)
        holder = create_metric_tensors(labs, outs.y_conv, class_range, prefix, ctx=ctx, tower=outs.tower)
        return NNParams(input_iterator=it, data_with_labels=None, metrics=holder, predict_tensor=outs.y_conv)

    testing_nn_params = eval_branch(testing_data_set, "testing")
    validation_nn_params = testing_nn_params
    if create_separate_validation_branch:
        validation_nn_params = eval_branch(validation_data_set, "validation")
    return cross_entropy, learning_rate, testing_nn_params, train_nn_params, validation_nn_params, train_step


# ----------------------------------------------------------------------------- plugin lookup (reference :443-452)
def get_model_from_name(model_name):
    return get_class("nnmodel." + model_name + "." + model_name)()


def get_importer_from_name(importer_name):
    return get_class("importer." + importer_name + "." + importer_name)()


def get_loader_from_name(loader_name, path):
    return get_class("loader." + loader_name + "." + loader_name)(path)


# ----------------------------------------------------------------------------- target helpers (reference :465-494)
def create_target_image_via_samples(sample_set, scene_shape):
    image = numpy.full([scene_shape[0], scene_shape[1]], INVALID_TARGET_VALUE, dtype=numpy.uint8)
    targets = numpy.vstack([sample_set.training_targets, sample_set.test_targets, sample_set.validation_targets])
    for point in targets.astype(int):
        image[point[1], point[0]] = point[2]
    return image


def read_targets_from_image(targets, class_range):
    result = numpy.zeros((0, 3), dtype=int)
    for target_index in class_range:
        ys, xs = numpy.where(targets == target_index)
        rows = numpy.stack([xs.astype(int), ys.astype(int), numpy.full(len(xs), target_index, dtype=int)], axis=1)
        result = numpy.vstack([result, rows])
    return result


# ----------------------------------------------------------------------------- sample splits (reference :497-543)
def _stratified_split(rows, labels, **split_args):
    from sklearn.model_selection import StratifiedShuffleSplit
    first, second = next(StratifiedShuffleSplit(n_splits=1, **split_args).split(rows, labels))
    return first, second


def shuffle_training_data_using_ratio(result, train_data_ratio):
    """Stratified (by class column 2) split into (train, validation); unseeded, like the reference."""
    tr, va = _stratified_split(result[:, 0:1], result[:, 2], train_size=train_data_ratio)
    return result[tr], result[va]


def shuffle_training_data_using_size(class_count, result, train_data_size, validation_size):
    """Per class: `train_data_size` random samples for training (90 % of the class when it is smaller), the rest --
    or a random `validation_size` of the rest -- for validation."""
    train_parts, val_parts = [], []
    for sample_class in class_count:
        ids = numpy.where(result[:, 2] == sample_class)[0]
        if ids.size == 0:
            continue
        take = (ids.size * 9) // 10 if ids.size < train_data_size else train_data_size
        chosen = numpy.random.choice(ids.size, take, replace=False)
        rest = numpy.setdiff1d(numpy.arange(ids.size), chosen)
        if validation_size is not None:
            rest = rest[numpy.random.choice(rest.size, min(validation_size, rest.size), replace=False)]
        train_parts.append(result[ids[chosen]])
        val_parts.append(result[ids[rest]])
    empty = numpy.empty([0, result.shape[1]], dtype=int)
    return (numpy.vstack(train_parts) if train_parts else empty), (numpy.vstack(val_parts) if val_parts else empty)


def shuffle_test_data_using_ratio(train_set, test_data_ratio):
    """Stratified split of the training rows into (test, remaining train), random_state 0; empty test set for ratio 0."""
    test_set = numpy.empty([0, train_set.shape[1]])
    if test_data_ratio > 0:
        tr, te = _stratified_split(train_set[:, 0:1], train_set[:, 2], test_size=test_data_ratio, random_state=0)
        test_set, train_set = train_set[te], train_set[tr]
    return test_set, train_set


def calculate_shadow_ratio(casi, shadow_map, shadow_map_inverse):
    """reference :473-483: per-band mean(non-shadow) / mean(shadow)."""
    in_shadow = numpy.ma.array(casi, mask=numpy.repeat((shadow_map == 0)[:, :, None], casi.shape[2], axis=2))
    in_light = numpy.ma.array(casi, mask=numpy.repeat((shadow_map_inverse == 0)[:, :, None], casi.shape[2], axis=2))
    ratio = in_light.mean(axis=(0, 1)) / in_shadow.mean(axis=(0, 1))
    return ratio.filled().astype(numpy.float32)
