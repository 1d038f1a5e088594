"""Context CNN on the MI355X graph builder (reference nnmodel/CONCNNModel.py:23-64): 1x1/3x3/5x5
inception head + LRN, residual 1x1 stacks, two dropouts, FC.  tf_slim defaults: ReLU, Xavier, biases."""
from hypelcnn_amd import graph as g
from hypelcnn_amd.common.common_nn_ops import ModelOutputTensors
from hypelcnn_amd.nnmodel.NNModel import NNModel


class CONCNNModel(NNModel):

    def create_tensor_graph(self, model_input_params, class_count, algorithm_params):
        p = algorithm_params
        training = model_input_params.is_training
        x = model_input_params.x
        f0 = p["filter_count"]
        head = g.concat([g.conv2d(x, f0, [k, k], scope=f"conv0_{k}x{k}") for k in (1, 3, 5)], axis=3)
        head = g.local_response_normalization(head)
        f1 = f0 * 3
        net11 = g.local_response_normalization(g.conv2d(head, f1, [1, 1], scope="conv11"))
        net12 = g.conv2d(net11, f1, [1, 1], scope="conv12")
        net13 = g.conv2d(net12, f1, [1, 1], scope="conv13") + net11
        net21 = g.conv2d(net13, f1, [1, 1], scope="conv21")
        net22 = g.conv2d(net21, f1, [1, 1], scope="conv22") + net13
        net31 = g.dropout(g.conv2d(net22, f1, [1, 1], scope="conv31"), p["drop_out_ratio"], is_training=training)
        net32 = g.dropout(g.conv2d(net31, f1, [1, 1], scope="conv32"), p["drop_out_ratio"], is_training=training)
        net33 = g.conv2d(net32, f1, [1, 1], scope="conv33")
        logits = g.fully_connected(g.flatten(net33), class_count, activation_fn=None, scope="fc")
        return ModelOutputTensors(y_conv=logits, image_output=None, image_original=None, histogram_tensors=[])

    def get_loss_func(self, tensor_output, label):
        return g.softmax_cross_entropy_with_logits(labels=label, logits=tensor_output.y_conv)
