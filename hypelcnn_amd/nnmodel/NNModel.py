"""Model plugin contract -- same two methods as the reference (nnmodel/NNModel.py:4-12).

Interface file: the abstract method names and signatures below ARE the reference's plugin surface (upstream is MIT-licensed); they are
reproduced on purpose -- a plugin written for the reference must subclass exactly this -- and contain no behaviour."""
from abc import ABC, abstractmethod


class NNModel(ABC):

    @abstractmethod
    def get_loss_func(self, tensor_output, label):
        """Per-sample loss expression for `tensor_output` (a ModelOutputTensors) and one-hot labels."""

    @abstractmethod
    def create_tensor_graph(self, model_input_params, class_count, algorithm_params):
        """Record the network on model_input_params.x's tower; return ModelOutputTensors."""
