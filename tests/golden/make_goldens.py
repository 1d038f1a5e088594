#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by importing the REFERENCE here.

Runs only in the build container (needs /root/reference). The reference's hot path
cannot be imported (tensorflow / tf_slim / tifffile are absent), so a meta-path finder
provides empty stand-in modules for those names *for import purposes only*; every
function called below is pure numpy / pure Python in the reference:

  common/common_nn_ops.py:45-106,169-185   BasicDataSet padding+normalisation, patches
  common/common_nn_ops.py:546-564          scale_in_to_out (tf.gather / tf.repeat args captured)
  common/common_nn_ops.py:280-292          calculate_class_accuracies_using_confusion
  utilities/stat_extractor.py:24-62,91-110 calc_kappa, extract_accuracy_metrics
  gan/gan_sampling_methods.py:191-201      DummySampler
  common/cmd_parser.py                     flag defaults
  loader/GRSS2018DataLoader.py:10-44       GRSS2018DataSet.get_data_point (half-resolution HSI under full-resolution
                                           LiDAR; numba.jit replaced by a pass-through decorator)

Output: tests/golden/reference_numpy_side.npz + reference_numpy_side.json, reference_grss2018.npz (data only).
"""
import importlib.abc
import importlib.machinery
import json
import os
import sys
import types

import numpy

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

STUB_ROOTS = ("tensorflow", "tf_slim", "tifffile", "tqdm", "tensorflow_gan", "numba")


class _Anything(types.ModuleType):
    """Module whose every attribute is another permissive stand-in."""

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
        pass


def main():
    sys.meta_path.insert(0, _Finder())
    sys.path.insert(0, REF)
    import tensorflow as tf  # the stand-in
    import numba
    numba.jit = lambda *a, **k: (lambda f: f)  # the loops are plain Python; numba only compiles them

    captured = {}

    def fake_gather(t, idx, axis):
        captured["op"] = ("gather", [int(i) for i in idx])
        return None

    def fake_repeat(input, axis, repeats):
        captured["op"] = ("repeat", int(repeats))
        return None

    tf.gather = fake_gather
    tf.repeat = fake_repeat

    from common import common_nn_ops as ref_ops

    class _Dim:
        def __init__(self, v):
            self.value = v

    class _FakeTensor:
        def __init__(self, c):
            self.c = c

        def get_shape(self):
            return [_Dim(1), _Dim(1), _Dim(1), _Dim(self.c)]

    arrays = {}
    meta = {}

    # ---- scale_in_to_out for every (Cin, Cout) pair any config can produce -------
    pairs = set()
    for cin in (145, 144, 49, 48, 64, 65, 360, 361, 120, 240, 480, 60, 30, 15, 1200, 2400, 600, 300, 150, 75,
                7, 10, 100, 33):
        for cout in (120, 240, 480, 60, 30, 15, 145, 1200, 2400, 600, 300, 150, 75, 7, 10, 100, 33, 90, 180, 45):
            pairs.add((cin, cout))
    maps = {}
    for cin, cout in sorted(pairs):
        captured.clear()
        res = ref_ops.scale_in_to_out(_FakeTensor(cin), _FakeTensor(cout), axis_no=3)
        if "op" not in captured:
            maps[f"{cin}->{cout}"] = ["identity"]
        elif captured["op"][0] == "repeat":
            maps[f"{cin}->{cout}"] = ["repeat", captured["op"][1]]
        else:
            maps[f"{cin}->{cout}"] = ["gather"]
            arrays[f"map_{cin}_{cout}"] = numpy.asarray(captured["op"][1], dtype=numpy.int16)
    meta["scale_in_to_out"] = maps

    # ---- BasicDataSet: symmetric pad + per-band min/max normalise + patches -------
    rng = numpy.random.RandomState(1234)
    for tag, (h, w, c, nb, dtype) in {"u16": (13, 11, 6, 3, numpy.uint16), "f32": (9, 10, 5, 2, numpy.float32)}.items():
        if dtype == numpy.uint16:
            casi = rng.randint(0, 4000, size=(h, w, c)).astype(numpy.float32)
        else:
            casi = rng.rand(h, w, c).astype(numpy.float32) * 7.0 - 1.0
        lidar = (rng.rand(h, w, 1).astype(numpy.float32) * 30.0 + 5.0)
        arrays[f"ds_{tag}_casi"] = casi.copy()
        arrays[f"ds_{tag}_lidar"] = lidar.copy()
        ds = ref_ops.BasicDataSet(None, casi.copy(), lidar.copy(), nb, True)
        arrays[f"ds_{tag}_casi_min"] = numpy.asarray(ds.casi_min)
        arrays[f"ds_{tag}_casi_max"] = numpy.asarray(ds.casi_max)
        arrays[f"ds_{tag}_lidar_min"] = numpy.asarray(ds.lidar_min)
        arrays[f"ds_{tag}_lidar_max"] = numpy.asarray(ds.lidar_max)
        pts = [(0, 0), (w - 1, h - 1), (3, 5), (w - 1, 0), (0, h - 1), (w // 2, h // 2)]
        arrays[f"ds_{tag}_points"] = numpy.asarray(pts, dtype=numpy.int32)
        arrays[f"ds_{tag}_patches"] = numpy.stack([ds.get_data_point(px, py) for px, py in pts]).astype(numpy.float32)
        meta[f"ds_{tag}"] = {"neighborhood": nb, "data_shape": [int(v) for v in ds.get_data_shape()],
                             "scene_shape": [int(v) for v in ds.get_scene_shape()],
                             "band_count": int(ds.get_casi_band_count())}
        # HSI-only dataset (AVON style: lidar None)
        ds2 = ref_ops.BasicDataSet(None, casi.copy(), None, nb, True)
        arrays[f"ds_{tag}_hsi_patches"] = numpy.stack([ds2.get_data_point(px, py) for px, py in pts]).astype(
            numpy.float32)

    # ---- metrics --------------------------------------------------------------
    from utilities import stat_extractor
    confs = []
    for k, n in ((15, 400), (20, 1000), (2, 50), (5, 64)):
        lab = rng.randint(0, k, size=n)
        prd = numpy.where(rng.rand(n) < 0.7, lab, rng.randint(0, k, size=n))
        conf = numpy.zeros((k, k), dtype=numpy.int64)
        for a, b in zip(lab, prd):
            conf[a, b] += 1
        confs.append(conf)
    for i, conf in enumerate(confs):
        arrays[f"conf_{i}"] = conf
        rec, prec = ref_ops.calculate_class_accuracies_using_confusion(conf, range(0, conf.shape[0]))
        arrays[f"conf_{i}_recall"] = rec
        arrays[f"conf_{i}_precision"] = prec
        arrays[f"conf_{i}_kappa"] = numpy.asarray(stat_extractor.calc_kappa(conf))
        with numpy.errstate(invalid="ignore", divide="ignore"):
            oa, ca, kp, cnt = stat_extractor.extract_accuracy_metrics(conf)
        arrays[f"conf_{i}_oa"] = numpy.asarray(oa)
        arrays[f"conf_{i}_class_acc"] = numpy.asarray(ca)
    meta["n_conf"] = len(confs)

    # ---- read_targets_from_image / create_target_image_via_samples -----------------
    tgt = rng.randint(0, 4, size=(6, 7)).astype(numpy.uint8)
    arrays["targets_img"] = tgt
    arrays["targets_rows"] = ref_ops.read_targets_from_image(tgt, range(1, 4)).astype(numpy.int64)

    # ---- DummySampler -------------------------------------------------------------
    from gan import gan_sampling_methods as gsm
    s = gsm.DummySampler(element_count=5, fill_value=0.5, coefficient=2.0)

    class _DS:
        @staticmethod
        def get_data_shape():
            return [1, 1, 3]

    x, y = s.get_sample_pairs(_DS(), None, None)
    arrays["dummy_x"] = numpy.asarray(x)
    arrays["dummy_y"] = numpy.asarray(y)

    # ---- flag defaults ----------------------------------------------------------------
    import argparse
    from common import cmd_parser
    p = argparse.ArgumentParser()
    for fn in ("add_parse_cmds_for_json_loader", "add_parse_cmds_for_trainers", "add_parse_cmds_for_loaders",
               "add_parse_cmds_for_models", "add_parse_cmds_for_importers", "add_parse_cmds_for_opt"):
        getattr(cmd_parser, fn)(p)
    meta["flag_defaults"] = {k: v for k, v in vars(p.parse_args([])).items()}

    # ---- is_integer_num ----------------------------------------------------------------
    from common import common_ops
    meta["is_integer_num"] = [[repr(v), bool(common_ops.is_integer_num(v))] for v in (1, 2.0, 2.5, 1 / (145 / 120), "3")]

    # ---- GRSS2018DataSet: HSI at half the LiDAR resolution (own fixture file, own RNG stream) ----------------
    from loader.GRSS2018DataLoader import GRSS2018DataSet
    rng18 = numpy.random.RandomState(2018)
    g18 = {}
    for tag, (h, w, c, nb) in {"a": (14, 18, 5, 2), "b": (10, 12, 3, 5)}.items():
        lidar = (rng18.rand(h, w, 1).astype(numpy.float32) * 30.0 + 5.0)
        casi = rng18.rand((h + 1) // 2 + 1, (w + 1) // 2 + 1, c).astype(numpy.float32) * 3000.0
        ds = GRSS2018DataSet(shadow_creator_dict=None, casi=casi.copy(), lidar=lidar.copy(), neighborhood=nb,
                             normalize=True)
        pts = [(0, 0), (w - 1, h - 1), (3, 5), (w - 1, 0), (0, h - 1), (w // 2, h // 2), (1, 1), (2, 7)]
        g18[f"{tag}_casi"] = casi
        g18[f"{tag}_lidar"] = lidar
        g18[f"{tag}_nb"] = numpy.asarray(nb)
        g18[f"{tag}_points"] = numpy.asarray(pts, dtype=numpy.int32)
        g18[f"{tag}_patches"] = numpy.stack([ds.get_data_point(px, py) for px, py in pts]).astype(numpy.float32)
        g18[f"{tag}_data_shape"] = numpy.asarray(ds.get_data_shape())
        g18[f"{tag}_scene_shape"] = numpy.asarray(ds.get_scene_shape())
    numpy.savez_compressed(os.path.join(OUT, "reference_grss2018.npz"), **g18)

    numpy.savez_compressed(os.path.join(OUT, "reference_numpy_side.npz"), **arrays)
    with open(os.path.join(OUT, "reference_numpy_side.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote", len(arrays), "arrays;", len(maps), "channel maps")


if __name__ == "__main__":
    main()
