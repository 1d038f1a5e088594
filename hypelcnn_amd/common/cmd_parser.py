"""argparse flag groups with the reference's names and defaults (common/cmd_parser.py:5-79);
tests/test_host_api.py checks the defaults against values captured from the reference."""
import os


def type_ensure_strtobool(val):
    s = str(val).strip().lower()
    if s in ("y", "yes", "t", "true", "on", "1"):
        return True
    if s in ("n", "no", "f", "false", "off", "0"):
        return False
    raise ValueError(f"invalid truth value {val!r}")


def add_parse_cmds_for_json_loader(parser):
    parser.add_argument("--flag_config_file", nargs="?", type=str, default=None, help="Flags as json")


def add_parse_cmds_for_trainers(parser):
    parser.add_argument("--batch_size", nargs="?", type=int, default=20, help="Batch size")
    parser.add_argument("--step", nargs="?", const=True, type=int, default=50000,
                        help="Training steps; use either this or --epoch")
    parser.add_argument("--epoch", nargs="?", const=True, type=int, default=None,
                        help="Epochs over the training data; use either this or --step")


def add_parse_cmds_for_loggers(parser):
    parser.add_argument("--base_log_path", nargs="?", const=True, type=str, default=os.getcwd(),
                        help="Base path for logs / checkpoints")
    parser.add_argument("--output_path", nargs="?", const=True, type=str, default=os.getcwd(),
                        help="Path for output logs and images")


def add_parse_cmds_for_loaders(parser):
    parser.add_argument("--path", nargs="?", const=True, type=str, default="/data/2013_DFTC/2013_DFTC",
                        help="Input data path")
    parser.add_argument("--loader_name", nargs="?", const=True, type=str, default="GRSS2013DataLoader",
                        help="Data set loader name")
    parser.add_argument("--neighborhood", nargs="?", type=int, default=0,
                        help="Neighborhood for data extraction, e.g. 1 means 3x3 patches")
    parser.add_argument("--test_ratio", nargs="?", type=float, default=0.05,
                        help="Ratio of training data to use in testing")
    parser.add_argument("--train_ratio", nargs="?", type=float, default=0.10,
                        help="Ratio of training data to use in validation")


def add_parse_cmds_for_models(parser):
    parser.add_argument("--algorithm_param_path", nargs="?", const=True, type=str, default=None,
                        help="Algorithm parameter (json) file")
    parser.add_argument("--model_name", nargs="?", const=True, type=str, default="HYPELCNNModel",
                        help="CONCNNModel, DUALCNNModel or HYPELCNNModel")


def add_parse_cmds_for_importers(parser):
    parser.add_argument("--importer_name", nargs="?", const=True, type=str, default="InMemoryImporter",
                        help="Importer name")


def add_parse_cmds_for_opt(parser):
    parser.add_argument("--flag_config_file_opt", nargs="?", type=str, default=None,
                        help="Flag config file for hyper parameter optimization")
    parser.add_argument("--opt_trial_count", nargs="?", type=int, default=10, help="Trial count")
    parser.add_argument("--opt_run_count", nargs="?", type=int, default=3, help="Runs per trial")
