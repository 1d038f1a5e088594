"""Exports a data set to TFRecord files (reference utilities/tfrecord_writer.py:10-85): metadata.tfrecord with the
three data shapes, training / test / validation records of {"label": int64, "image": float32[P*P*C]}."""
import argparse
import os

import numpy

from hypelcnn_amd.common.cmd_parser import add_parse_cmds_for_loaders, add_parse_cmds_for_loggers, \
    type_ensure_strtobool
from hypelcnn_amd.common.common_nn_ops import get_importer_from_name
from hypelcnn_amd.common.tfrecord_io import encode_example, write_records


def add_parse_cmds_for_apps(parser):
    parser.add_argument("--compressed", nargs="?", const=True, type=type_ensure_strtobool, default=False,
                        help="If true, performs compression")
    parser.add_argument("--target_path", nargs="?", type=str, default=os.getcwd(), help="Directory of the record files")


def write_to_tfrecord(filename, data, labels, compressed):
    return write_records(filename, (encode_example({"label": [int(labels[i])], "image": data[i].reshape(-1)})
                                    for i in range(len(data))), compressed)


def write_metadata_record(filename, training_data, testing_data, validation_data):
    write_records(filename, [encode_example({"training_data_shape": numpy.asarray(training_data.shape),
                                             "testing_data_shape": numpy.asarray(testing_data.shape),
                                             "validation_data_shape": numpy.asarray(validation_data.shape)})])


def export(loader_name, path, train_ratio, neighborhood, target_path, compressed=False):
    importer = get_importer_from_name("InMemoryImporter")
    train, test, val, *_ = importer.read_data_set(loader_name, path, train_ratio, 0.05, neighborhood, True)
    os.makedirs(target_path, exist_ok=True)
    write_metadata_record(os.path.join(target_path, "metadata.tfrecord"), train.data, test.data, val.data)
    for name, t in (("training", train), ("test", test), ("validation", val)):
        write_to_tfrecord(os.path.join(target_path, name + ".tfrecord"), t.data, t.labels, compressed)


def main(argv=None):
    parser = argparse.ArgumentParser()
    add_parse_cmds_for_loaders(parser)
    add_parse_cmds_for_loggers(parser)
    add_parse_cmds_for_apps(parser)
    flags, _ = parser.parse_known_args(argv)
    export(flags.loader_name, flags.path, flags.train_ratio, flags.neighborhood, flags.target_path, flags.compressed)


if __name__ == "__main__":
    main()
