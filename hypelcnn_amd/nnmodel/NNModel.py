"""Model plugin contract -- same two methods as the reference (nnmodel/NNModel.py:4-12)."""
from abc import ABC, abstractmethod


class NNModel(ABC):

    @abstractmethod
    def get_loss_func(self, tensor_output, label):
        """Per-sample loss expression for `tensor_output` (a ModelOutputTensors) and one-hot labels."""

    @abstractmethod
    def create_tensor_graph(self, model_input_params, class_count, algorithm_params):
        """Record the network on model_input_params.x's tower; return ModelOutputTensors."""
