"""Data set cut on the fly from the scene (reference importer/GeneratorImporter.py:16-103).

The reference wraps a Python generator (`data_set.get_data_point` per sample) in tf.data.from_generator; here the
padded scene lives in HBM and `hypel_gather_patches_f32` cuts a whole batch per launch, so full-scene inference
(every pixel -> one patch) never materialises the [H*W, P, P, C] array on host or device."""
import time
from collections import namedtuple

import numpy

from hypelcnn_amd.common.common_nn_ops import get_loader_from_name
from hypelcnn_amd.importer.DataImporter import DataImporter

GeneratorDataTensor = namedtuple("GeneratorDataTensor", ["dataset"])
GeneratorDataInfo = namedtuple("GeneratorDataInfo", ["data", "targets", "loader", "dataset"])
GeneratorSpecialData = namedtuple("GeneratorSpecialData", ["shape", "size"])


class SceneDataSetSpec:
    """Stands where the reference has tf.data.Dataset.from_generator(...): static element shape, class count, and
    the (scene data set, targets) pair the iterator cuts batches from."""

    def __init__(self, info, class_count):
        self.element_shape = tuple(int(v) for v in info.dataset.get_data_shape())
        self.class_count = int(class_count)
        self.info = info


class GeneratorImporter(DataImporter):

    def read_data_set(self, loader_name, path, train_data_ratio, test_data_ratio, neighborhood, normalize):
        start = time.time()
        loader = get_loader_from_name(loader_name, path)
        data_set = loader.load_data(neighborhood, normalize)
        sample_set = loader.load_samples(train_data_ratio, test_data_ratio)

        def info(targets):
            shape = numpy.concatenate(([targets.shape[0]], data_set.get_data_shape()))
            return GeneratorDataInfo(data=GeneratorSpecialData(shape=shape, size=numpy.prod(shape)), targets=targets,
                                     loader=loader, dataset=data_set)

        print(f"Loaded dataset({time.time() - start:.3f} sec)")
        return info(sample_set.training_targets), info(sample_set.test_targets), info(sample_set.validation_targets), \
            data_set.shadow_creator_dict, loader.get_class_count(), data_set.get_scene_shape(), \
            loader.get_samples_color_list()

    def convert_data_to_tensor(self, test_data_with_labels, training_data_with_labels, validation_data_with_labels,
                               class_range):
        k = class_range.stop
        return GeneratorDataTensor(dataset=SceneDataSetSpec(test_data_with_labels, k)), \
            GeneratorDataTensor(dataset=SceneDataSetSpec(training_data_with_labels, k)), \
            GeneratorDataTensor(dataset=SceneDataSetSpec(validation_data_with_labels, k))

    def init_tensors(self, session, tensor, nn_params):
        info = nn_params.data_with_labels if nn_params.data_with_labels is not None else tensor.dataset.info
        nn_params.input_iterator.initializer_scene(info.dataset, info.targets, session.backend)

    def requires_separate_validation_branch(self):
        return True
