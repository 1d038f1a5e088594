"""Full-scene / sample / ground-truth inference (reference classify/infer_for_classification.py:18-134).

`--domain all` classifies EVERY pixel of the scene: the reference materialises one patch per pixel through a Python
generator and writes the label raster sample by sample; here the padded scene is resident in HBM, each batch of
patches is cut by hypel_gather_patches_f32, runs through the inference tower (moving BN statistics, no dropout, no
reconstruction head) and hypel_argmax_scatter writes the classes straight into the device-resident raster."""
import argparse
import json
import os
import time

import numpy

from hypelcnn_amd.classify.monitored_session_runner import latest_checkpoint, restore_checkpoint
from hypelcnn_amd.common.cmd_parser import add_parse_cmds_for_importers, add_parse_cmds_for_loaders, \
    add_parse_cmds_for_loggers, add_parse_cmds_for_models, add_parse_cmds_for_trainers
from hypelcnn_amd.common.common_nn_ops import GraphContext, ModelInputParams, NNParams, Template, \
    create_colored_image, create_target_image_via_samples, get_loader_from_name, get_model_from_name, \
    perform_prediction, simple_nn_iterator
from hypelcnn_amd.common.tiff_io import imwrite
from hypelcnn_amd.importer.GeneratorImporter import GeneratorDataInfo, GeneratorImporter


def add_parse_cmds_for_app(parser):
    parser.add_argument("--domain", nargs="?", type=str, default="all",
                        help="Conversion domain for inferencing. It can be all(all scene inference), "
                             "sample(sample based inference) or gt(ground truth)")


def create_all_scene_data(scene_shape, data_with_labels_to_copy):
    """reference :24-35: one target [x, y, 0] per pixel, row-major over the scene."""
    ys, xs = numpy.meshgrid(numpy.arange(scene_shape[0]), numpy.arange(scene_shape[1]), indexing="ij")
    targets = numpy.stack([xs.reshape(-1), ys.reshape(-1), numpy.zeros(xs.size, dtype=int)], axis=1).astype(int)
    return GeneratorDataInfo(data=None, targets=targets, loader=data_with_labels_to_copy.loader,
                             dataset=data_with_labels_to_copy.dataset)


def create_sample_data(test_data_with_labels, training_data_with_labels, validation_data_with_labels):
    """reference :38-47"""
    targets = numpy.vstack([test_data_with_labels.targets.astype(numpy.int32),
                            training_data_with_labels.targets.astype(numpy.int32),
                            validation_data_with_labels.targets.astype(numpy.int32)])
    return GeneratorDataInfo(data=None, targets=targets, loader=test_data_with_labels.loader,
                             dataset=test_data_with_labels.dataset)


def gt_process(flags):
    """reference :73-80"""
    loader = get_loader_from_name(flags.loader_name, flags.path)
    sample_set = loader.load_samples(0.1, 0)
    data_set = loader.load_data(0, False)
    scene_as_image = create_target_image_via_samples(sample_set, data_set.get_scene_shape())
    return scene_as_image, loader.get_samples_color_list()


def prediction_process(flags, backend=None):
    """reference :83-134"""
    data_importer = GeneratorImporter()
    training, test, validation, shadow_dict, class_range, scene_shape, color_list = \
        data_importer.read_data_set(flags.loader_name, flags.path, 0.1, 0, flags.neighborhood, True)
    if flags.domain == "all":
        validation = create_all_scene_data(scene_shape, validation)
    elif flags.domain == "sample":
        validation = create_sample_data(training, test, validation)
    scene_as_image = numpy.full(shape=scene_shape, dtype=numpy.uint8, fill_value=255)
    if flags.algorithm_param_path is None:
        raise IOError("Algorithm parameter file is not given")
    algorithm_params = json.load(open(flags.algorithm_param_path, "r"))
    algorithm_params["batch_size"] = flags.batch_size
    nn_model = get_model_from_name(flags.model_name)

    testing_tensor, training_tensor, validation_tensor = data_importer.convert_data_to_tensor(
        test, training, validation, class_range)
    template = Template("nn_core", nn_model.create_tensor_graph, class_count=class_range.stop)
    ctx = GraphContext(template, backend)
    validation_input_iter = simple_nn_iterator(validation_tensor.dataset, flags.batch_size)
    images, _ = validation_input_iter.get_next()
    outputs = template(ModelInputParams(x=images, y=None, device_id="/gpu:0", is_training=False),
                       algorithm_params=algorithm_params)
    nn_params = NNParams(input_iterator=validation_input_iter, data_with_labels=validation, metrics=None,
                         predict_tensor=outputs.y_conv)
    session = ctx.session()
    # variables of nn_core except the training-only reconstruction head ("image_gen_net_", :117-118) -- the
    # inference tower never creates those, and load_state_dict ignores names the session does not hold
    ckpt = flags.base_log_path
    if os.path.isdir(ckpt):
        ckpt = latest_checkpoint(ckpt)
    if ckpt is None or not os.path.exists(ckpt):
        raise IOError(f"No checkpoint found at {flags.base_log_path}")
    restore_checkpoint(session, ckpt)
    data_importer.init_tensors(session, validation_tensor, nn_params)
    perform_prediction(session, nn_params, scene_as_image)
    return scene_as_image, color_list


def build_parser():
    parser = argparse.ArgumentParser()
    add_parse_cmds_for_loaders(parser)
    add_parse_cmds_for_loggers(parser)
    add_parse_cmds_for_trainers(parser)
    add_parse_cmds_for_models(parser)
    add_parse_cmds_for_importers(parser)
    add_parse_cmds_for_app(parser)
    return parser


def main(argv=None, backend=None):
    flags, _ = build_parser().parse_known_args(argv)
    start_time = time.time()
    if flags.domain in ("all", "sample"):
        scene_as_image, color_list = prediction_process(flags, backend)
    elif flags.domain == "gt":
        scene_as_image, color_list = gt_process(flags)
    else:
        raise ValueError(f"Domain flags does not support value:{flags.domain}")
    os.makedirs(flags.output_path, exist_ok=True)
    imwrite(os.path.join(flags.output_path, "result_raw.tif"), scene_as_image)
    imwrite(os.path.join(flags.output_path, "result_colorized.tif"), create_colored_image(scene_as_image, color_list))
    print(f"Done evaluation({time.time() - start_time:.3f} sec)")
    return scene_as_image


if __name__ == "__main__":
    main()
