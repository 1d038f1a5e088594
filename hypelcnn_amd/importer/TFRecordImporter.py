"""Data set read from TFRecord files (reference importer/TFRecordImporter.py:15-72): `<base>/metadata.tfrecord`
carries the three data shapes, `training.tfrecord` / `test.tfrecord` / `validation.tfrecord` the examples
{"label": int64, "image": float32[P*P*C]}.  The reference streams them through tf.data.TFRecordDataset; here a file
is decoded once at iterator initialisation and kept resident in HBM like the in-memory importer's arrays."""
from collections import namedtuple

import numpy

from hypelcnn_amd.common.common_nn_ops import get_loader_from_name
from hypelcnn_amd.common.tfrecord_io import decode_example, read_records
from hypelcnn_amd.importer.DataImporter import DataImporter
from hypelcnn_amd.importer.InMemoryImporter import DataSetSpec

TFRecordDataInfo = namedtuple("TFRecordDataInfo", ["data", "path"])
TFRecordDataTensor = namedtuple("TFRecordDataTensor", ["dataset", "path_placeholder"])


class TFRecordSpecialData(namedtuple("TFRecordSpecialData", ["shape"])):
    """Shape-only stand-in for the data array.  `size` is added here: the reference's test hook reads
    `data_with_labels.data.size` (monitored_session_runner.py:117), which its own shape-only tuple lacks."""

    @property
    def size(self):
        return int(numpy.prod(self.shape))


class TFRecordImporter(DataImporter):

    def read_data_set(self, loader_name, path, train_data_ratio, test_data_ratio, neighborhood, normalize):
        loader = get_loader_from_name(loader_name, path)
        base = loader.get_model_base_dir()
        shapes = {}
        for record in read_records(base + "metadata.tfrecord"):
            ex = decode_example(record)
            shapes = {k: numpy.asarray(ex[k + "_data_shape"]) for k in ("training", "testing", "validation")}
        return TFRecordDataInfo(TFRecordSpecialData(shapes["training"]), base + "training.tfrecord"), \
            TFRecordDataInfo(TFRecordSpecialData(shapes["testing"]), base + "test.tfrecord"), \
            TFRecordDataInfo(TFRecordSpecialData(shapes["validation"]), base + "validation.tfrecord"), None, \
            loader.get_class_count(), None, loader.get_samples_color_list()

    def convert_data_to_tensor(self, test_data_with_labels, training_data_with_labels, validation_data_with_labels,
                               class_range):
        def spec(info):
            return DataSetSpec(tuple(int(v) for v in info.data.shape[1:4]), class_range.stop)

        testing = TFRecordDataTensor(dataset=spec(test_data_with_labels), path_placeholder="testing_path_placeholder")
        training = TFRecordDataTensor(dataset=spec(training_data_with_labels),
                                      path_placeholder="training_path_placeholder")
        return testing, training, testing  # the reference reuses the testing data set for validation (:62-64)

    @staticmethod
    def load_file(path, shape):
        images, labels = [], []
        for record in read_records(path):
            ex = decode_example(record)
            images.append(numpy.asarray(ex["image"], numpy.float32).reshape(shape))
            labels.append(int(ex["label"][0]))
        data = numpy.stack(images) if images else numpy.zeros((0,) + tuple(shape), numpy.float32)
        return data, numpy.asarray(labels, numpy.uint8)

    def init_tensors(self, session, tensor, nn_params):
        info = nn_params.data_with_labels
        data, labels = self.load_file(info.path, tuple(int(v) for v in info.data.shape[1:4]))
        nn_params.input_iterator.initializer(data, labels, session.backend)

    def requires_separate_validation_branch(self):
        return False
