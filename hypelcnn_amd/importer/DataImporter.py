"""Importer plugin contract (reference importer/DataImporter.py:4-20).

Interface file: the abstract method names and signatures below ARE the reference's plugin surface (upstream is MIT-licensed); they are
reproduced on purpose -- a plugin written for the reference must subclass exactly this -- and contain no behaviour."""
from abc import ABC, abstractmethod


class DataImporter(ABC):
    @abstractmethod
    def read_data_set(self, loader_name, path, train_data_ratio, test_data_ratio, neighborhood, normalize):
        """-> (train, test, validation Targets, shadow_dict, class_range, scene_shape, color_list)"""

    @abstractmethod
    def convert_data_to_tensor(self, test_data_with_labels, training_data_with_labels, validation_data_with_labels,
                               class_range):
        """-> (testing, training, validation) dataset descriptors"""

    @abstractmethod
    def init_tensors(self, session, tensor, nn_params):
        """(Re)initialise nn_params.input_iterator with the arrays in nn_params.data_with_labels."""

    @abstractmethod
    def requires_separate_validation_branch(self):
        pass
