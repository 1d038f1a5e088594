"""Pairing of shadowed / non-shadowed pixels for GAN training (reference gan/gan_sampling_methods.py:16-201).
Vectorised numpy restatements (the reference walks the scene with Python double loops); pixel order is the
reference's row-major scan order, so results are identical (tests/test_gan_host.py checks against its goldens)."""
from abc import ABC, abstractmethod

import numpy
from scipy import ndimage


class Sampler(ABC):
    @abstractmethod
    def get_sample_pairs(self, data_set, loader, shadow_map):
        pass


def _gather(data_set, mask):
    """Patches of all pixels where mask == 1, row-major (x_index outer = rows, y_index inner = columns)."""
    rows, cols = numpy.nonzero(mask == 1)
    shape = data_set.get_data_shape()
    out = numpy.zeros([len(rows)] + list(shape), dtype=numpy.float32)
    for i, (r, c) in enumerate(zip(rows, cols)):
        out[i] = data_set.get_data_point(c, r)
    return out


class NeighborhoodBasedSampler(Sampler):
    """Normal samples come from a ring around the shadows: dilation(neighborhood_size) minus dilation(margin)."""

    def __init__(self, neighborhood_size, margin):
        self._margin = margin
        self._neighborhood_size = neighborhood_size

    def get_sample_pairs(self, data_set, loader, shadow_map):
        ring = ndimage.binary_dilation(shadow_map, iterations=self._neighborhood_size).astype(shadow_map.dtype) - \
            ndimage.binary_dilation(shadow_map, iterations=self._margin).astype(shadow_map.dtype)
        shadow = _gather(data_set, shadow_map)
        normal = _gather(data_set, numpy.where(shadow_map == 1, 0, ring))
        return normal[0:shadow.shape[0]], shadow


class RandomBasedSampler(Sampler):
    def __init__(self, multiply_shadowed_data):
        self._multiply_shadowed_data = multiply_shadowed_data

    def get_sample_pairs(self, data_set, loader, shadow_map):
        shadow = _gather(data_set, shadow_map)
        normal = _gather(data_set, numpy.where(shadow_map == 1, 0, 1))
        if self._multiply_shadowed_data:
            shadow = numpy.repeat(shadow, repeats=(normal.shape[0] // shadow.shape[0]), axis=0)
        return normal[0:shadow.shape[0]], shadow


class TargetBasedSampler(Sampler):
    """Pairs shadowed and lit pixels of the same class (needs the loader's class raster)."""

    def __init__(self, margin):
        self._margin = margin

    def get_sample_pairs(self, data_set, loader, shadow_map):
        targets = loader.read_targets("shadow_gen_model/class_result.tif").copy()
        h, w = data_set.get_scene_shape()
        m = self._margin
        ok = (targets[:, 1] > m) & (targets[:, 1] < h - m) & (targets[:, 0] > m) & (targets[:, 0] < w - m)
        targets[~ok, 2] = -1
        normal_all, shadow_all = [], []
        for cls in range(loader.get_class_count().stop):
            rows = targets[targets[:, 2] == cls]
            if not len(rows):
                continue
            in_shadow = shadow_map[rows[:, 1], rows[:, 0]] == 1
            sh = [data_set.get_data_point(x, y) for x, y, _ in rows[in_shadow]]
            no = [data_set.get_data_point(x, y) for x, y, _ in rows[~in_shadow]]
            if not sh or not no:
                continue
            sh, no = numpy.asarray(sh, numpy.float32), numpy.asarray(no, numpy.float32)
            mult, rem = len(no) // len(sh), len(no) % len(sh)
            shadow_all.append(numpy.vstack([numpy.repeat(sh, mult, axis=0), sh[0:rem]]))
            normal_all.append(no)
        return numpy.vstack(normal_all), numpy.vstack(shadow_all)


class DummySampler(Sampler):
    """Known-answer pair source: y == fill_value, x == fill_value * coefficient (ideal generator = x / coefficient)."""

    def __init__(self, element_count, fill_value, coefficient):
        self._element_count = element_count
        self._fill_value = fill_value
        self._coefficient = coefficient

    def get_sample_pairs(self, data_set, loader, shadow_map):
        shape = data_set.get_data_shape()
        shadow = numpy.full(numpy.concatenate([[self._element_count], shape]), fill_value=self._fill_value,
                            dtype=numpy.float32)
        return shadow * self._coefficient, shadow
