"""A DataLoader that synthesises a scene instead of reading TIFFs.

The reference's loaders (GRSS2013/2018, GULFPORT, AVON) need the contest rasters and `tifffile`, neither of
which exists in the build or test environment, so every benchmark and end-to-end test runs on this plugin.
`path` selects the geometry, e.g. "grss2013" (144 HSI bands + LiDAR, 15 classes), "grss2018" (48 + LiDAR, 20),
"avon" (360 bands, no LiDAR, 2 classes), optionally followed by ":key=value" overrides
(h, w, bands, classes, lidar, seed, samples, gan_ckpt=<npz checkpoint of a shadow GAN for the generator-based
shadow augmenters>, base_dir=<directory that get_model_base_dir reports, e.g. where TFRecord exports live>).  Each class has its own smooth spectrum and height, pixels are
class spectrum + noise laid out in blobs, so that a classifier can actually learn the scene."""
import numpy

from hypelcnn_amd.common.common_nn_ops import BasicDataSet, read_targets_from_image
from hypelcnn_amd.loader.DataLoader import DataLoader, SampleSet

PRESETS = {
    "grss2013": dict(bands=144, classes=15, lidar=1, h=60, w=80, lo=380, hi=1050),
    "grss2018": dict(bands=48, classes=20, lidar=1, h=60, w=80, lo=380, hi=1050),
    "gulfport": dict(bands=64, classes=11, lidar=1, h=50, w=60, lo=368, hi=1043),
    "avon": dict(bands=360, classes=2, lidar=0, h=40, w=50, lo=400, hi=2500),
    # the real GRSS2018 geometry: HSI at half the resolution of the LiDAR grid the samples live on
    "grss2018hr": dict(bands=48, classes=20, lidar=1, h=60, w=80, lo=380, hi=1050, half_res=1),
}


class SyntheticDataLoader(DataLoader):

    def __init__(self, path):
        parts = str(path).split(":")
        cfg = dict(PRESETS.get(parts[0], PRESETS["grss2013"]))
        cfg.update(seed=1234, samples=0.5)
        for kv in parts[1:]:
            k, v = kv.split("=", 1)
            cfg[k] = v if k in ("gan_ckpt", "base_dir") else (float(v) if k == "samples" else int(v))
        self.cfg = cfg
        self._targets = None

    def _scene(self):
        c = self.cfg
        rng = numpy.random.RandomState(c["seed"])
        h, w, b, k = c["h"], c["w"], c["bands"], c["classes"]
        # blobby label map: nearest of a few random seeds per class
        seeds = rng.rand(k * 3, 2) * [h, w]
        seed_cls = numpy.arange(k * 3) % k
        yy, xx = numpy.mgrid[0:h, 0:w]
        d = (yy[..., None] - seeds[:, 0]) ** 2 + (xx[..., None] - seeds[:, 1]) ** 2
        labels = seed_cls[d.argmin(-1)].astype(numpy.uint8)
        t = numpy.linspace(0, 1, b)
        spectra = numpy.stack([0.5 + 0.4 * numpy.sin(2 * numpy.pi * (t * (1 + 0.35 * i) + rng.rand())) for i in range(k)])
        casi = (spectra[labels] * 2000 + rng.randn(h, w, b) * 60 + 2200).astype(numpy.float32)
        # a blobby shadow region whose pixels are attenuated band by band (what the GAN has to learn to map)
        sseed = rng.rand(3, 2) * [h, w]
        sd = ((yy[..., None] - sseed[:, 0]) ** 2 + (xx[..., None] - sseed[:, 1]) ** 2).min(-1)
        self._shadow_map = (sd < (min(h, w) / 4.5) ** 2).astype(numpy.uint8)
        atten = 0.35 + 0.3 * numpy.linspace(0, 1, b)
        casi = numpy.where(self._shadow_map[..., None] == 1, casi * atten, casi).astype(numpy.float32)
        lidar = None
        if c["lidar"]:
            heights = rng.rand(k) * 30
            lidar = (heights[labels] + rng.randn(h, w) * 0.5 + 5).astype(numpy.float32)[:, :, None]
        return casi, lidar, labels

    def load_data(self, neighborhood, normalize):
        casi, lidar, labels = self._scene()
        self._labels = labels
        if self.cfg.get("half_res"):
            from hypelcnn_amd.loader.GRSS2018DataLoader import GRSS2018DataSet
            h, w = casi.shape[:2]
            half = numpy.ascontiguousarray(casi[::2, ::2])
            half = numpy.pad(half, ((0, (h + 1) // 2 + 1 - half.shape[0]), (0, (w + 1) // 2 + 1 - half.shape[1]), (0, 0)),
                             mode="edge")
            data_set = GRSS2018DataSet(shadow_creator_dict=None, casi=half, lidar=lidar, neighborhood=neighborhood,
                                       normalize=normalize)
            data_set.shadow_creator_dict = {}
            return data_set
        data_set = BasicDataSet(shadow_creator_dict=None, casi=casi, lidar=lidar, neighborhood=neighborhood,
                                normalize=normalize)
        # the shadow augmenters the reference's loaders register (loader/GRSS2013DataLoader.py:24-34): the per-band
        # ratio struct always, the generator-based ones when a trained shadow GAN checkpoint is given
        from functools import partial
        from hypelcnn_amd.gan.gan_utilities import create_gan_struct, create_simple_shadow_struct
        _, shadow_ratio = self.load_shadow_map(neighborhood, data_set)
        creators = {"simple": create_simple_shadow_struct(shadow_ratio) if lidar is not None
                    else create_simple_shadow_struct(shadow_ratio, lidar_passthrough=False)}
        ckpt = self.cfg.get("gan_ckpt")
        if ckpt:
            from hypelcnn_amd.gan.shadow_data_models import shadowdata_generator_model
            from hypelcnn_amd.gan.wrappers.cycle_gan_wrapper import CycleGANInferenceWrapper
            generator_fn = partial(shadowdata_generator_model, create_only_encoder=False, is_training=False)
            for name in ("cycle_gan", "dcl_gan", "dcl_cycle_gan"):
                creators[name] = create_gan_struct(CycleGANInferenceWrapper(generator_fn), "", ckpt,
                                                   bands=casi.shape[2])
        data_set.shadow_creator_dict = creators
        return data_set

    def load_shadow_map(self, neighborhood, data_set):
        """load_shadow_map_common (reference common_nn_ops.py:567-571): padded map + per-band lit/shadow ratio."""
        from hypelcnn_amd.common.common_nn_ops import calculate_shadow_ratio
        if getattr(self, "_shadow_map", None) is None:
            self._scene()
        shadow_map = numpy.pad(self._shadow_map, neighborhood, mode="symmetric")
        ratio = None if data_set is None else calculate_shadow_ratio(
            data_set.casi, shadow_map, numpy.logical_not(shadow_map).astype(int))
        return shadow_map, ratio

    def load_samples(self, train_data_ratio, test_data_ratio):
        if getattr(self, "_labels", None) is None:
            self._labels = self._scene()[2]
        rng = numpy.random.RandomState(self.cfg["seed"] + 1)
        rows = read_targets_from_image(self._labels, self.get_class_count())
        rows = rows[rng.permutation(len(rows))]
        n_used = int(len(rows) * self.cfg["samples"])
        rows = rows[:n_used]
        n_val = max(1, int(n_used * train_data_ratio))
        n_test = max(1, int(n_used * test_data_ratio))
        return SampleSet(validation_targets=rows[:n_val], test_targets=rows[n_val:n_val + n_test],
                         training_targets=rows[n_val + n_test:])

    def get_class_count(self):
        return range(0, self.cfg["classes"])

    def get_model_base_dir(self):
        base = self.cfg.get("base_dir", "")
        return base + "/" if base and not base.endswith("/") else base

    def get_samples_color_list(self):
        rng = numpy.random.RandomState(7)
        return rng.randint(0, 255, size=(self.cfg["classes"], 3)).astype(numpy.uint8)

    def get_band_measurements(self):
        return numpy.linspace(self.cfg["lo"], self.cfg["hi"], num=self.cfg["bands"])
