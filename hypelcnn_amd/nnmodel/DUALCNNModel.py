"""Dual-branch HSI + LiDAR CNN on the MI355X graph builder (reference nnmodel/DUALCNNModel.py:11-104).

The HSI bands and the LiDAR channel are zero-copy views of one pixel-major input buffer; the HSI
view is additionally cropped by `hs_lidar_diff`.  Each branch is a chain of multi-kernel levels
(all odd kernels up to the map size, biases, leaky-ReLU, no normaliser) with 1x1 connectors; the
two flattened branch outputs feed four FC layers (the fusion concat is a list of GEMM segments).
"""
from hypelcnn_amd import graph as g
from hypelcnn_amd.common.common_nn_ops import ModelOutputTensors
from hypelcnn_amd.nnmodel.NNModel import NNModel


class DUALCNNModel(NNModel):
    HS_WIDTH_DIVISORS = (4, 2, 1, 2, 4, 8, 16, 32)
    LIDAR_WIDTHS = (2, 4, 8)

    def create_tensor_graph(self, model_input_params, class_count, algorithm_params):
        p = algorithm_params
        training = model_input_params.is_training
        x = model_input_params.x
        with g.arg_scope([g.conv2d, g.fully_connected], activation_fn=g.leaky_relu(p["lrelu_alpha"])):
            hs = x.slice_channels(0, x.c - 1)
            lidar = x.slice_channels(x.c - 1, x.c)
            if x.hw[0] > 1 or x.hw[1] > 1:
                hs = hs.crop(p["hs_lidar_diff"])

            for i, div in enumerate(self.HS_WIDTH_DIVISORS, start=1):
                hs = self._level(hs, p["filter_count"] // div, f"level{i}")
                hs = g.conv2d(hs, hs.c, [1, 1], scope=f"connector_conv{i}")
            for i, width in enumerate(self.LIDAR_WIDTHS, start=1):
                lidar = self._level(lidar, width, f"lidar_level{i}")
                lidar = g.conv2d(lidar, lidar.c, [1, 1], scope=f"lidar_connector_conv{i}")

            net = g.concat([g.flatten(hs), g.flatten(lidar)], axis=1)
            keep = p["drop_out_ratio"]  # the reference passes the ratio itself as keep_prob (:49)
            for i, mult in enumerate((9, 6, 3), start=1):
                net = g.fully_connected(net, class_count * mult, scope=f"fc{i}")
                net = g.dropout(net, keep, is_training=training)
            net = g.fully_connected(net, class_count, activation_fn=None, scope="fc4")
        return ModelOutputTensors(y_conv=net, image_output=None, image_original=None, histogram_tensors=[])

    def get_loss_func(self, tensor_output, label):
        return g.softmax_cross_entropy_with_logits(labels=label, logits=tensor_output.y_conv)

    @staticmethod
    def _level(net, width, name):
        return g.concat([g.conv2d(net, width, [k, k], scope=f"{name}_conv{k}x{k}")
                         for k in range(1, net.hw[0] + 1, 2)], axis=3)
