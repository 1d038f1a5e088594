"""GAN method plugin contracts (reference gan/wrappers/wrapper.py:4-38).

Interface file: the abstract method names and signatures below ARE the reference's plugin surface (upstream is MIT-licensed); they are
reproduced on purpose -- a plugin written for the reference must subclass exactly this -- and contain no behaviour."""
from abc import ABC, abstractmethod


class Wrapper(ABC):
    @abstractmethod
    def define_model(self, images_x, images_y):
        pass

    @abstractmethod
    def define_loss(self, model):
        pass

    @abstractmethod
    def define_train_ops(self, model, loss, max_number_of_steps, **kwargs):
        pass

    @abstractmethod
    def get_train_hooks_fn(self):
        pass


class InferenceWrapper(ABC):
    @abstractmethod
    def construct_inference_graph(self, input_tensor, is_shadow_graph, clip_invalid_values):
        pass

    @abstractmethod
    def make_inference_graph(self, data_set, is_shadow_graph, clip_invalid_values):
        pass

    @abstractmethod
    def create_generator_restorer(self):
        pass
