#!/usr/bin/env python3
"""Pin the classifier input's AUGMENTATION stage by executing the reference's own text (build container only; needs /root/reference).

`common/common_nn_ops.py:376-440` -- `add_augmentation_graph` and the four map functions `perform_rotation_augmentation_random`,
`perform_shadow_augmentation_random`, `perform_reflection_augmentation_random`, `perform_spectral_augmentation_random` -- and
`gan/gan_utilities.py:17-27` (`create_simple_shadow_struct`) run UNCHANGED on numpy patches, with a stand-in for the handful of
`tf.*` calls they make, restated after the published TensorFlow semantics:

    tf.random.uniform(shape, minval, maxval[, dtype])   minval + u (maxval - minval), u SCRIPTED in [0, 1); floor for int32
    random_ops.random_uniform(shape, minval, maxval)    the same
    tf.image.rot90(image, k)                            k quarter turns counter-clockwise of [H, W, C]
    tf.image.random_flip_left_right / _up_down(image)   one uniform draw u in [0, 1); flipped iff u < 0.5
    tf.less, tf.cond, tf.device, tf.compat.v1.name_scope

Every random number the reference asks for is scripted and RECORDED (shape, range, dtype, in call order), so the fixture holds:
the order in which `add_augmentation_graph` applies the maps, the draws each map makes (rot90 takes k in {0, 1, 2}, never 3;
the shadow map fires iff u < augmentation_random_threshold; the per-channel shift is U(-s, 0)), and float32 inputs / outputs of
seeded cases.  `tests/test_reference_augmentation.py` holds the product's host half (`draw_augmentations`: which decisions, which
ranges) and the specification of `hypel_augment_patches_f32` (tests/emu_backend.py; the HIP kernel is compared with it bit for
bit in tests/test_data_side.py) to this fixture -- including the product's REORDERING (per-band shadow ratio applied before the
rotation: it commutes with every spatial permutation; the fixture proves it bit for bit).  Only data is written."""
import contextlib
import json
import os
import sys
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import tf_standin as S  # noqa: E402


class Dim:
    def __init__(self, v):
        self.value = int(v)


class T(np.ndarray):
    """numpy array with the two TensorShape accessors the map functions use"""

    def get_shape(self):
        return [Dim(d) for d in self.shape]


def tensor(a):
    return np.asarray(a, np.float32).view(T)


class Script:
    """the uniform numbers of one sample, in the order the reference asks for them"""

    def __init__(self, us):
        self.us, self.calls = list(us), []

    def draw(self, kind, shape, minval, maxval, dtype):
        n = int(np.prod(shape)) if len(shape) else 1
        u = np.asarray([self.us.pop(0) for _ in range(n)], np.float64)
        self.calls.append({"fn": kind, "shape": [int(s) for s in shape], "minval": float(minval), "maxval": float(maxval),
                           "dtype": str(dtype), "u": [float(x) for x in u]})
        if str(dtype) == "int32":
            v = np.floor(minval + u * (maxval - minval)).astype(np.int32)
        else:
            v = (np.float32(minval) + u.astype(np.float32) * np.float32(maxval - minval)).astype(np.float32)
        return v.reshape(shape) if len(shape) else v.reshape(())


SCRIPT = [None]


def install_tf_surface(ref_ops):
    tf = ref_ops.tf
    tf.device = lambda name: contextlib.nullcontext()
    tf.compat.v1.name_scope = lambda name: contextlib.nullcontext()
    tf.less = lambda a, b: a < b
    tf.cond = lambda pred, true_fn, false_fn: true_fn() if bool(pred) else false_fn()

    def uniform(shape, minval=0, maxval=None, dtype="float32", seed=None, name=None):
        return SCRIPT[0].draw("tf.random.uniform", list(shape), minval, maxval, dtype)

    def random_uniform(shape, minval=0, maxval=None, dtype="float32", seed=None, name=None):
        return SCRIPT[0].draw("random_ops.random_uniform", list(shape), minval, maxval, dtype)

    def rot90(image, k=1, name=None):
        return tensor(np.rot90(np.asarray(image), int(k), axes=(0, 1)))

    def random_flip(axis):
        def f(image, seed=None):
            u = SCRIPT[0].draw("tf.image.random_flip_" + ("left_right" if axis == 1 else "up_down"), [], 0.0, 1.0, "float32")
            return tensor(np.flip(np.asarray(image), axis)) if float(u) < 0.5 else image
        return f

    tf.random = SimpleNamespace(uniform=uniform)
    tf.image = SimpleNamespace(rot90=rot90, random_flip_left_right=random_flip(1), random_flip_up_down=random_flip(0))
    ref_ops.random_ops.random_uniform = random_uniform


class FakeDataset:
    """tf.data.Dataset of ONE element: `.map(fn)` applies fn to it (the maps run per sample, before batching) and records it"""

    def __init__(self, x, y, log):
        self.x, self.y, self.log = x, y, log

    def map(self, fn, num_parallel_calls=None):
        before = len(SCRIPT[0].calls)
        x, y = fn(self.x, self.y)
        self.log.append({"num_parallel_calls": num_parallel_calls, "draws": [c["fn"] for c in SCRIPT[0].calls[before:]]})
        return FakeDataset(tensor(x), y, self.log)


# (case, patch, channels, flags (shadow, rotation, spectral, reflection), threshold, scripted u's: rot, shadow, lr, ud, C x delta)
def cases():
    rng = np.random.default_rng(77)
    out = []
    for name, p, c, flags, thr, picks in [
        ("all_maps", 7, 9, (True, True, 0.05, True), 0.5, dict(rot=0.40, sh=0.20, lr=0.10, ud=0.90)),     # k = 1, shadow, lr flip
        ("rot2_ud", 7, 9, (True, True, 0.05, True), 0.5, dict(rot=0.70, sh=0.80, lr=0.60, ud=0.30)),      # k = 2, no shadow, ud flip
        ("rot0_both", 5, 6, (True, True, 0.1, True), 0.3, dict(rot=0.05, sh=0.29, lr=0.49, ud=0.01)),     # k = 0, shadow, both flips
        ("threshold_edge", 5, 6, (True, True, 0.1, True), 0.3, dict(rot=0.999, sh=0.30, lr=0.5, ud=0.5)), # k = 2 (never 3), u == thr: no
        ("no_rotation", 3, 4, (True, False, 0.2, True), 1.0, dict(sh=0.99, lr=0.2, ud=0.2)),
        ("spectral_only", 3, 4, (False, False, 0.2, False), 0.5, dict()),
        ("no_spectral", 7, 9, (True, True, False, True), 0.5, dict(rot=0.34, sh=0.1, lr=0.7, ud=0.7)),
    ]:
        out.append((name, p, c, flags, thr, picks, rng.random((p, p, c)).astype(np.float32),
                    (0.5 + rng.random(c - 1)).astype(np.float32), rng.random(c)))
    return out


def main():
    S.install()
    import importlib
    ref_ops = importlib.import_module("common.common_nn_ops")
    gan_utils = importlib.import_module("gan.gan_utilities")
    install_tf_surface(ref_ops)
    meta, arrays = {}, {}
    for name, p, c, (f_sh, f_rot, f_spec, f_refl), thr, picks, x, ratio, du in cases():
        us = []
        if f_rot:
            us.append(picks["rot"])
        if f_sh:
            us.append(picks["sh"])
        if f_refl:
            us += [picks["lr"], picks["ud"]]
        if f_spec:
            us += list(du)
        SCRIPT[0] = Script(us)
        info = ref_ops.AugmentationInfo(shadow_struct=gan_utils.create_simple_shadow_struct(ratio) if f_sh else None,
                                        perform_shadow_augmentation=f_sh, perform_rotation_augmentation=f_rot,
                                        perform_spectral_augmentation=f_spec, perform_reflection_augmentation=f_refl,
                                        augmentation_random_threshold=thr)
        log = []
        ds = ref_ops.add_augmentation_graph(FakeDataset(tensor(x), 3, log), info,
                                            ref_ops.perform_rotation_augmentation_random,
                                            ref_ops.perform_shadow_augmentation_random,
                                            ref_ops.perform_reflection_augmentation_random,
                                            ref_ops.perform_spectral_augmentation_random)
        assert not SCRIPT[0].us, "the reference asked for fewer random numbers than scripted"
        assert ds.y == 3
        y = np.asarray(ds.x)
        meta[name] = {"patch": p, "channels": c, "threshold": thr,
                      "flags": {"shadow": f_sh, "rotation": f_rot, "spectral": f_spec, "reflection": f_refl},
                      "maps": log, "draws": SCRIPT[0].calls, "out_dtype": str(y.dtype)}
        arrays[f"{name}/x"] = x
        arrays[f"{name}/ratio"] = ratio
        arrays[f"{name}/y"] = y
        print(f"{name}: maps {[m['draws'] for m in log]}, out {y.dtype} {y.shape}")
    with open(os.path.join(HERE, "reference_augmentation.json"), "w") as f:
        json.dump(meta, f, sort_keys=True, indent=0, separators=(",", ":"))
    np.savez_compressed(os.path.join(HERE, "reference_augmentation.npz"), **arrays)
    print("wrote reference_augmentation.json / .npz")


if __name__ == "__main__":
    main()
