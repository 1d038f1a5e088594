"""ORACLE (test infrastructure) -- restatement of the reference's numpy-side host functions.

These ARE pinned: tests/test_oracle_golden.py checks every function here against
vectors captured from the reference itself (tests/golden/make_goldens.py).
"""
import numpy as np


def scale_in_to_out_index(cin, cout):
    """Index vector of length cout equivalent to common/common_nn_ops.py:546-564.

    inv = cout/cin integer -> identity (1) or tf.repeat (each input channel `inv` times,
    i.e. idx[o] = o // inv); otherwise tf.gather with min(round(o*cin/cout), cin-1) where
    round is Python's round-half-to-even."""
    scale_ratio = cin / cout
    inv = 1 / scale_ratio
    if float(inv).is_integer():
        r = int(inv)
        return np.arange(cout, dtype=np.int64) // r
    return np.asarray([min(round(o * scale_ratio), cin - 1) for o in range(cout)], dtype=np.int64)


def scale_in_to_out_kind(cin, cout):
    inv = 1 / (cin / cout)
    if float(inv).is_integer():
        return "identity" if int(inv) == 1 else "repeat"
    return "gather"


class PatchDataSet:
    """common/common_nn_ops.py:45-106 BasicDataSet: symmetric pad by `neighborhood`, per-band
    (casi) / global (lidar) min-max normalisation computed on the PADDED arrays, patches by
    slicing [y:y+2n+1, x:x+2n+1] with lidar appended as the last channel (:169-185)."""

    def __init__(self, casi, lidar, neighborhood, normalize=True):
        self.n = neighborhood
        pad = ((neighborhood, neighborhood), (neighborhood, neighborhood), (0, 0))
        self.lidar = None if lidar is None else np.pad(lidar, pad, mode="symmetric")
        self.casi = np.pad(casi, pad, mode="symmetric")
        self.casi_min, self.casi_max, self.lidar_min, self.lidar_max = 0, 1, 0, 1
        if normalize:
            if self.lidar is not None:
                self.lidar_min = np.min(self.lidar)
                self.lidar = self.lidar - self.lidar_min
                self.lidar_max = np.max(self.lidar)
                self.lidar = self.lidar / self.lidar_max
            self.casi_min = np.min(self.casi, axis=(0, 1))
            self.casi = self.casi - self.casi_min
            self.casi_max = np.max(self.casi, axis=(0, 1))
            self.casi = self.casi / self.casi_max.astype(np.float32)

    def get_data_shape(self):
        d = 2 * self.n + 1
        return [d, d, self.casi.shape[2] + (0 if self.lidar is None else 1)]

    def get_scene_shape(self):
        base = self.casi if self.lidar is None else self.lidar
        return [base.shape[0] - 2 * self.n, base.shape[1] - 2 * self.n]

    def get_data_point(self, px, py):
        d = 2 * self.n + 1
        c = self.casi[py:py + d, px:px + d, :]
        if self.lidar is None:
            return c
        return np.concatenate((c, self.lidar[py:py + d, px:px + d, :]), axis=2)


def class_accuracies_from_confusion(conf, class_range):
    """common/common_nn_ops.py:280-292: per-class recall (rows) and precision (columns)."""
    k = class_range.stop
    prec, rec = np.zeros(k), np.zeros(k)
    for i in class_range:
        gt = conf[i, :].sum()
        if gt != 0:
            rec[i] = conf[i, i] / gt
        pr = conf[:, i].sum()
        if pr != 0:
            prec[i] = conf[i, i] / pr
    return rec[class_range], prec[class_range]


def streaming_metrics_from_confusion(conf):
    """What tf.metrics.accuracy / mean_per_class_accuracy / tf_slim cohen_kappa report after
    accumulating `conf` (rows = labels, cols = predictions) -- Appendix A.15;
    common/common_nn_ops.py:243-277."""
    conf = conf.astype(np.float64)
    total = conf.sum()
    oa = np.trace(conf) / total if total else 0.0
    row = conf.sum(1)
    with np.errstate(invalid="ignore", divide="ignore"):
        per = np.where(row > 0, np.diag(conf) / np.where(row > 0, row, 1), 0.0)
    aa = per.mean()  # mean over ALL classes, div_no_nan -> 0 for absent classes
    pe = (row * conf.sum(0)).sum() / (total * total) if total else 0.0
    kappa = (oa - pe) / (1 - pe) if pe != 1 else 0.0
    return oa, aa, kappa


def read_targets_from_image(targets, class_range):
    """common/common_nn_ops.py:486-494: rows [x, y, class], class-major, row-major within class."""
    out = np.zeros((0, 3), dtype=np.int64)
    for t in class_range:
        ys, xs = np.where(targets == t)
        out = np.vstack([out, np.stack([xs, ys, np.full_like(xs, t)], 1)])
    return out


def exponential_decay_staircase(lr0, step, decay_steps, rate):
    """tf.compat.v1.train.exponential_decay(staircase=True) (common_nn_ops.py:217-221)."""
    return lr0 * rate ** (step // decay_steps)


def gan_lr(lr0, step, max_steps):
    """gan/wrappers/gan_common.py:222-244: constant for the first half, then
    polynomial_decay(power=1) from lr0 to 0 over the second half."""
    half = max_steps // 2
    if step < half:
        return lr0
    decay_steps = max_steps - half
    s = min(step - half, decay_steps)
    return lr0 * (1 - s / decay_steps)
