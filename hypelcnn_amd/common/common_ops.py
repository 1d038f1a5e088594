"""Name-based plugin loading and small helpers (reference common/common_ops.py:4-29)."""
import importlib
import ntpath

PACKAGE = "hypelcnn_amd"


def get_class(kls):
    """Resolve "nnmodel.HYPELCNNModel.HYPELCNNModel"-style names.  The reference imports the top-level
    packages of its own tree; here the same dotted names resolve inside the hypelcnn_amd package first
    and fall back to an absolute import (so out-of-tree plugins keep working)."""
    parts = kls.split(".")
    module_name, attr = ".".join(parts[:-1]), parts[-1]
    last_err = None
    for candidate in (f"{PACKAGE}.{module_name}", module_name):
        try:
            return getattr(importlib.import_module(candidate), attr)
        except (ImportError, AttributeError) as e:
            last_err = e
    raise ImportError(f"cannot resolve plugin {kls!r}: {last_err}")


def is_integer_num(n):
    if isinstance(n, int):
        return True
    if isinstance(n, float):
        return n.is_integer()
    return False


def replace_abbrs(txt, abbrs_dict):
    for word, abbr in abbrs_dict.items():
        txt = txt.replace(word, abbr)
    return txt


def path_leaf(path):
    head, tail = ntpath.split(path)
    return tail or ntpath.basename(head)
