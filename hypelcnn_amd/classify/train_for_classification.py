"""Training CLI with the reference's flags (classify/train_for_classification.py:20-226).

    python -m hypelcnn_amd.classify.train_for_classification --loader_name SyntheticDataLoader --path grss2013 \
        --neighborhood 3 --model_name HYPELCNNModel --algorithm_param_path <json> --batch_size 64 --step 300

Multi-GPU: launch with torch.distributed.run (one rank per GPU); gradients are all-reduced over RCCL.
The optuna mode of the reference (--flag_config_file_opt) is not on the hot path and is not provided."""
import argparse
import json
import os
import time

from numpy import mean, std

from hypelcnn_amd.classify.monitored_session_runner import add_classification_summaries, run_monitored_session, \
    set_run_seed
from hypelcnn_amd.common.cmd_parser import add_parse_cmds_for_importers, add_parse_cmds_for_loaders, \
    add_parse_cmds_for_loggers, add_parse_cmds_for_models, add_parse_cmds_for_opt, add_parse_cmds_for_trainers, \
    type_ensure_strtobool
from hypelcnn_amd.common.common_nn_ops import AugmentationInfo, TrainingResult, create_graph, \
    get_importer_from_name, get_model_from_name
from hypelcnn_amd.common.common_ops import path_leaf, replace_abbrs


def perform_an_episode(flags, algorithm_params, model, base_log_path, backend=None):
    print("Args:", json.dumps(vars(flags), indent=3))
    prefetch_size = 1000
    data_importer = get_importer_from_name(flags.importer_name)
    train_data, test_data, val_data, shadow_dict, class_range, scene_shape, color_list = \
        data_importer.read_data_set(flags.loader_name, flags.path, flags.train_ratio, flags.test_ratio,
                                    flags.neighborhood, True)
    shadow_struct = None
    if flags.augment_data_with_shadow is not None and shadow_dict is not None:
        shadow_struct = shadow_dict[flags.augment_data_with_shadow]
    augmentation_info = AugmentationInfo(shadow_struct=shadow_struct,
                                         perform_shadow_augmentation=flags.augment_data_with_shadow is not None,
                                         perform_rotation_augmentation=flags.augment_data_with_rotation,
                                         perform_reflection_augmentation=flags.augment_data_with_reflection,
                                         perform_spectral_augmentation=flags.augment_data_with_spectral,
                                         augmentation_random_threshold=flags.augmentation_random_threshold)
    batch_size = algorithm_params["batch_size"]
    epoch = flags.epoch
    required_steps = flags.step if epoch is None else (train_data.data.shape[0] * epoch) // batch_size
    print(f"Steps: {required_steps:d}, Algorithm Params: {algorithm_params}")
    device_id = "/cpu:0" if flags.device == "cpu" else "/gpu:0"

    set_run_seed()
    testing_tensor, training_tensor, validation_tensor = data_importer.convert_data_to_tensor(
        test_data, train_data, val_data, class_range)
    cross_entropy, learning_rate, testing_nn_params, training_nn_params, validation_nn_params, train_step = \
        create_graph(training_tensor.dataset, testing_tensor.dataset, validation_tensor.dataset, class_range,
                     batch_size, prefetch_size, device_id, epoch, augmentation_info=augmentation_info,
                     algorithm_params=algorithm_params, model=model,
                     create_separate_validation_branch=data_importer.requires_separate_validation_branch,
                     backend=backend)
    training_nn_params.data_with_labels = train_data
    testing_nn_params.data_with_labels = test_data
    validation_nn_params.data_with_labels = val_data
    if not flags.perform_validation:
        validation_nn_params = None
    summary_fn = add_classification_summaries(cross_entropy, learning_rate, flags.log_model_params, testing_nn_params,
                                              validation_nn_params)
    start = time.time()
    result = run_monitored_session(cross_entropy, base_log_path, class_range, flags.save_checkpoint_steps,
                                   flags.validation_steps, train_step, required_steps, augmentation_info,
                                   training_nn_params, training_tensor, testing_nn_params, testing_tensor,
                                   validation_nn_params, validation_tensor, data_importer,
                                   json.dumps(vars(flags), indent=3), json.dumps(algorithm_params, indent=3),
                                   summary_fn=summary_fn)
    print(f"Done training for {time.time() - start:.3f} sec")
    if flags.perform_validation:
        print(f"Validation accuracy={result.validation_accuracy:g}, Testing accuracy={result.test_accuracy:g}, "
              f"loss={result.loss:.2f}")
    else:
        print(f"Testing accuracy={result.test_accuracy:g}, loss={result.loss:.2f}")
    return TrainingResult(validation_accuracy=result.validation_accuracy if flags.perform_validation else None,
                          test_accuracy=result.test_accuracy, loss=result.loss)


def add_parse_cmds_for_app(parser):
    b = type_ensure_strtobool
    parser.add_argument("--perform_validation", nargs="?", const=True, type=b, default=False)
    parser.add_argument("--augment_data_with_rotation", nargs="?", const=True, type=b, default=False)
    parser.add_argument("--augment_data_with_spectral", nargs="?", const=True, type=float, default=None)
    parser.add_argument("--augment_data_with_shadow", nargs="?", const=True, type=str, default=None)
    parser.add_argument("--augment_data_with_reflection", nargs="?", const=True, type=b, default=False)
    parser.add_argument("--augmentation_random_threshold", nargs="?", type=float, default=0.5)
    parser.add_argument("--device", nargs="?", type=str, default="gpu")
    parser.add_argument("--save_checkpoint_steps", nargs="?", type=int, default=2000)
    parser.add_argument("--validation_steps", nargs="?", type=int, default=40000)
    parser.add_argument("--all_data_shuffle_ratio", nargs="?", type=float, default=None)
    parser.add_argument("--log_model_params", nargs="?", const=True, type=b, default=False)


def get_log_suffix(flags):
    abbreviations = {"model": "mdl", "dataloader": "ldr", "alg_param_": "p"}
    trn = f"{int(flags.train_ratio):d}" if flags.train_ratio > 1.0 else f"{flags.train_ratio:.2f}".replace(".", "")
    patch = flags.neighborhood * 2 + 1
    suffix = f"{flags.loader_name.lower():s}_{flags.model_name.lower():s}_trn{trn:s}_" \
             f"{os.path.splitext(path_leaf(flags.algorithm_param_path))[0].lower()}_{patch:d}x{patch:d}"
    if flags.augment_data_with_shadow is not None:
        suffix += f"_{flags.augment_data_with_shadow}" + f"_aug{flags.augmentation_random_threshold:.2f}".replace(".", "")
    if flags.augment_data_with_spectral is not None:
        suffix += f"_spectral{flags.augment_data_with_spectral:.3f}".replace(".", "")
    return replace_abbrs(suffix, abbreviations)


def build_parser():
    parser = argparse.ArgumentParser()
    for add in (add_parse_cmds_for_loaders, add_parse_cmds_for_loggers, add_parse_cmds_for_trainers,
                add_parse_cmds_for_models, add_parse_cmds_for_importers, add_parse_cmds_for_app, add_parse_cmds_for_opt):
        add(parser)
    return parser


def main(argv=None):
    flags, _unknown = build_parser().parse_known_args(argv)  # unknown flags are ignored, as in the reference
    import torch
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl")
    nn_model = get_model_from_name(flags.model_name)
    if flags.algorithm_param_path is None:
        raise IOError("Algorithm parameter file is not given")
    algorithm_params = json.load(open(flags.algorithm_param_path, "r"))
    algorithm_params["batch_size"] = flags.batch_size  # the CLI overrides the JSON (reference :225)
    return perform_an_episode(flags, algorithm_params, nn_model, os.path.join(flags.base_log_path, get_log_suffix(flags)))


if __name__ == "__main__":
    main()
