"""The shadow GAN networks on the MI355X graph builder (reference gan/shadow_data_models.py:43-149).

Tensors are [N, B] (the reference carries [N,1,1,B]; 1x1 "patches" only, gan_train_for_shadow.py:249)."""
from hypelcnn_amd import graph as g


def shadowdata_generator_model(netinput, create_only_encoder, is_training=True):
    """Seven 1-channel SAME 1-D convolutions over the band axis with dense skip sums, tanh output; the
    encoder-only variant (CUT features) returns n4.  Recorded as ONE fused node -> one wavefront per sample."""
    return g.shadow_generator(netinput, create_only_encoder, is_training)


def shadowdata_discriminator_model(generated_data, generator_input, is_training, scale):
    """flatten -> FC B->B -> FC B->B -> FC B->B/2 (linear); leaky-ReLU 0.1, He init, L2(scale) on the first two."""
    with g.arg_scope([g.fully_connected], weights_initializer=g.he_truncated_init(),
                     weights_regularizer=g.l2_regularizer(scale), activation_fn=g.leaky_relu(0.1)):
        band_size = generated_data.c
        net1 = g.fully_connected(generated_data, band_size)
        net2 = g.fully_connected(net1, band_size)
        net3 = g.fully_connected(net2, band_size // 2, weights_regularizer=None, activation_fn=None)
    # at narrow band counts (<= 128: Gulfport) the three layers of one application are one launch per direction
    return g.fuse_dense_stack(net3)


def shadowdata_feature_discriminator_model(generated_data, patch_count, embedded_feature_size, is_training, scale):
    """Band slices of width B // patch_count (the last one may be ragged), each through its own 4-layer MLP
    w -> ps -> ps/4 -> ps/2 -> E (leaky-ReLU on all four), whole-tensor l2 normalisation, stacked [N, P*E]."""
    with g.arg_scope([g.fully_connected], weights_initializer=g.he_truncated_init(),
                     weights_regularizer=g.l2_regularizer(scale), activation_fn=g.leaky_relu(0.1)):
        band_size = generated_data.c
        patch_size = band_size // patch_count
        outs = []
        for start in range(0, band_size, patch_size):
            cur = generated_data.slice_channels(start, min(start + patch_size, band_size))
            cur = g.fully_connected(cur, patch_size)
            cur = g.fully_connected(cur, patch_size // 4)
            cur = g.fully_connected(cur, patch_size // 2)
            cur = g.fully_connected(cur, embedded_feature_size)
            outs.append(cur)
    return g.feature_stack(outs)
