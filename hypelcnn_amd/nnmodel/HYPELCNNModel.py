"""HypeLCNN on the MI355X graph builder.

Same network as the reference plugin (nnmodel/HYPELCNNModel.py:34-183), recorded on
hypelcnn_amd.graph instead of tf_slim: a spectral 1x1 encoder/decoder pyramid, multi-kernel
spatial levels with 1x1 connectors, an FC pyramid whose depth follows
floor(log_d(flat/classes)), batch-normed logits and -- in the training tower only -- a four
layer image-reconstruction head.  Every conv / FC is GEMM -> batch-norm -> leaky-ReLU, and every
residual is a weight-free channel re-indexing (`scale_in_to_out`).
"""
import math

from hypelcnn_amd import graph as g
from hypelcnn_amd.common.common_nn_ops import HistogramTensorPair, ModelOutputTensors, scale_in_to_out
from hypelcnn_amd.nnmodel.NNModel import NNModel


class HYPELCNNModel(NNModel):

    def create_tensor_graph(self, model_input_params, class_count, algorithm_params):
        p = algorithm_params
        training = model_input_params.is_training
        residual = p["use_residual"]
        bn_params = {"is_training": training, "decay": p["bn_decay"]}
        with g.arg_scope([g.conv2d, g.fully_connected],
                         weights_initializer=g.variance_scaling_init(scale=2.0),
                         # recorded on the convolution weights as the reference does (:42); the classifier's loss never
                         # collects regularisation losses (common_nn_ops.py:214), the dense layers pass None (:80-93,121)
                         weights_regularizer=g.l2_regularizer(p["l2regularizer_scale"]),
                         normalizer_fn=g.batch_norm, normalizer_params=bn_params,
                         activation_fn=g.leaky_relu(p["lrelu_alpha"])):
            net0 = model_input_params.x
            width = p["filter_count"]
            depth = p["spectral_hierarchy_level"]

            net1 = self._spectral_stack(net0, [width >> (depth - 1 - i) for i in range(depth)], "conv_enc_", residual)
            if residual:
                net1 = net1 + scale_in_to_out(net0, net1, axis_no=3)
            net2 = self._spectral_stack(net1, [width >> i for i in range(depth)], "conv_dec_", residual)
            if residual:
                net2 = net2 + scale_in_to_out(net1, net2, axis_no=3)

            net3 = self._spatial_levels(net2, net2.c // 2, p["spatial_hierarchy_level"], residual)
            if residual:
                net3 = net3 + scale_in_to_out(net2, net3, axis_no=3)

            net5 = self._fc_pyramid(g.flatten(net3), class_count, p["degradation_coeff"], 1 - p["drop_out_ratio"],
                                    training)
            logits = g.fully_connected(net5, class_count, weights_regularizer=None, activation_fn=None, scope="fc_final")

            image_out = None
            if training:
                patch_elems = net0.hw[0] * net0.hw[1] * net0.c
                head = logits
                for i, mult in enumerate((3, 9, 27), start=1):
                    head = g.fully_connected(head, class_count * mult, weights_regularizer=None, scope=f"image_gen_net_{i}")
                image_out = g.fully_connected(head, patch_elems, weights_regularizer=None, activation_fn=g.sigmoid,
                                              scope="image_gen_net_4")
        return ModelOutputTensors(y_conv=logits, image_output=image_out, image_original=net0,
                                  histogram_tensors=[HistogramTensorPair(net1, "spectral_expansion"),
                                                     HistogramTensorPair(net2, "spectral_reduction"),
                                                     HistogramTensorPair(net3, "spatial"),
                                                     HistogramTensorPair(net5, "classification")])

    def get_loss_func(self, tensor_output, label):
        loss = g.softmax_cross_entropy_with_logits(labels=label, logits=tensor_output.y_conv)
        if tensor_output.image_output is not None:
            # scalar reconstruction error broadcast onto every per-sample CE (reference :106-109)
            loss = loss + g.mean_squared_reconstruction(tensor_output.image_output, tensor_output.image_original)
        return loss

    # ------------------------------------------------------------------
    @staticmethod
    def _spectral_stack(net, widths, prefix, residual):
        for i, cout in enumerate(widths):
            nxt = g.conv2d(net, cout, [1, 1], scope=f"{prefix}{i}")
            if residual:
                nxt = nxt + scale_in_to_out(net, nxt, axis_no=3)
            net = nxt
        return net

    @staticmethod
    def _spatial_levels(net, first_width, levels, residual):
        side = net.hw[0]
        kernels = [k for k in range(1, side + 1, 2)]  # odd, square kernels only (reference :174)
        for lvl in range(levels):
            cout = first_width >> lvl
            merged = g.concat([g.conv2d(net, cout, [k, k], scope=f"connector_{lvl}_conv{k}x{k}") for k in kernels],
                              axis=3)
            if residual:
                merged = merged + scale_in_to_out(net, merged, axis_no=3)
            joined = g.conv2d(merged, merged.c, [1, 1], scope=f"connector_conv_{lvl}")
            if residual:
                joined = joined + merged
            net = joined
        return net

    @staticmethod
    def _fc_pyramid(flat, class_count, shrink, keep_prob, training):
        size = flat.features
        stages = math.floor(math.log(size / class_count, shrink))
        net = flat
        for i in range(stages - 1):
            size //= shrink
            net = g.fully_connected(net, size, weights_regularizer=None, scope=f"fc_{i}")
            net = g.dropout(net, keep_prob=keep_prob, is_training=training)
        return net
