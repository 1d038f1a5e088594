"""Shadow augmenters handed to the classifier's input pipeline (reference gan/gan_utilities.py:7-43)."""
import numpy
import torch


class ShadowOpHolder:
    def __init__(self, shadow_op, deshadow_op, shadow_op_creater, shadow_op_initializer):
        self.shadow_op_initializer = shadow_op_initializer
        self.shadow_op_creater = shadow_op_creater
        self.shadow_op = shadow_op
        self.deshadow_op = deshadow_op


def create_simple_shadow_struct(shadow_ratio, lidar_passthrough=True):
    """Per-band ratio shadowing; the LiDAR channel (last) passes through (ratio 1).  The reference always appends
    the 1 (gan_utilities.py:19), which only fits data sets with a LiDAR channel; HSI-only scenes switch it off."""
    ratio = numpy.asarray(shadow_ratio, numpy.float32)
    if lidar_passthrough:
        ratio = numpy.append(ratio, 1).astype(numpy.float32)

    def _r(inp):
        return torch.as_tensor(ratio, device=inp.device)

    holder = ShadowOpHolder(shadow_op=lambda inp: inp / _r(inp), deshadow_op=lambda inp: inp * _r(inp),
                            shadow_op_creater=lambda: None, shadow_op_initializer=lambda restorer, session: None)
    holder.ratio = ratio  # lets the fused augmentation kernel divide in flight instead of materialising shadow_op(x)
    return holder


class GeneratorAugmenter:
    """create_gan_struct (reference :30-43) + create_inference_for_matrix_input (gan_common.py:282-304): the trained
    generator applied to EVERY pixel of a patch batch, LiDAR channel passed through.  The reference builds P*P copies
    of the generator graph on /cpu:0; here all N*P*P pixel spectra go through ONE fused generator launch."""

    def __init__(self, inference_wrapper, is_shadow_graph, bands, backend=None):
        from hypelcnn_amd.gan.wrappers import gan_common as C
        self.bands = bands
        tower, x, _ = C.new_gan_tower(bands)
        self.out = inference_wrapper.construct_inference_graph(x, is_shadow_graph, False)
        self.ctx = C.GanContext(tower, backend)
        self.tower = tower

    def load(self, variables):
        sess = self.ctx.session()
        for name, value in variables.items():
            if name in sess.store.vars:
                sess.set_variable(name, value)

    def __call__(self, patches):
        """patches: [N, P, P, C] device tensor with C = bands (+1 LiDAR)."""
        sess = self.ctx.session()
        n, p1, p2, c = patches.shape
        spectra = patches[..., :self.bands].reshape(-1, self.bands).contiguous()
        ct = sess.compile_phase(self.tower, spectra.shape[0], outputs=[self.out], key="augment")
        ct.set_input("x", spectra)
        ct.forward()
        # with a LiDAR channel the concat below copies; without one the caller gets its own tensor (not a plan-buffer view)
        conv = ct.value(self.out, copy=c <= self.bands).reshape(n, p1, p2, self.bands)
        if c > self.bands:
            conv = torch.cat([conv, patches[..., self.bands:]], dim=3)
        return conv


def create_gan_struct(gan_inference_wrapper, model_base_dir, ckpt_relative_path, bands=None, backend=None):
    """Lazily built generator augmenters; `shadow_op_initializer` loads an .npz checkpoint keyed by TF names."""
    holders = {}
    state = {"backend": backend}

    def _get(is_shadow):
        if is_shadow not in holders:
            holders[is_shadow] = GeneratorAugmenter(gan_inference_wrapper, is_shadow, bands, state["backend"])
        return holders[is_shadow]

    def _initializer(restorer, session):
        if state["backend"] is None and session is not None:
            state["backend"] = session.backend  # the augmenter runs on the classifier session's device
        path = model_base_dir + ckpt_relative_path
        from hypelcnn_amd.common import tf_checkpoint
        if not path.endswith(".npz") and tf_checkpoint.is_checkpoint(path):  # a TensorFlow bundle, e.g. the published
            variables = tf_checkpoint.read_checkpoint(path)                  # shadow_gen_model/cycle_gan/model.ckpt-5000
        else:
            with numpy.load(path if path.endswith(".npz") else path + ".npz") as z:
                variables = {k.replace("|", "/"): z[k] for k in z.files}
        for flag in (True, False):
            _get(flag).load(variables)

    return ShadowOpHolder(shadow_op=lambda inp: _get(True)(inp), deshadow_op=lambda inp: _get(False)(inp),
                          shadow_op_creater=gan_inference_wrapper.create_generator_restorer,
                          shadow_op_initializer=_initializer)
