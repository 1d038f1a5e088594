"""CPU: the host-side mirror of the reference interface (names, defaults, numpy-side behaviour) against the
vectors captured from the reference (tests/golden/)."""
import argparse
import json
import os

import numpy as np
import pytest

from hypelcnn_amd import graph as G
from hypelcnn_amd.common import cmd_parser, common_nn_ops as cno, common_ops


def test_flag_defaults_match_reference(golden):
    _, meta = golden
    p = argparse.ArgumentParser()
    for fn in ("add_parse_cmds_for_json_loader", "add_parse_cmds_for_trainers", "add_parse_cmds_for_loaders",
               "add_parse_cmds_for_models", "add_parse_cmds_for_importers", "add_parse_cmds_for_opt"):
        getattr(cmd_parser, fn)(p)
    assert vars(p.parse_args([])) == meta["flag_defaults"]


def test_scale_in_to_out_matches_reference(golden):
    arrs, meta = golden
    tower = G.Tower(G.VariableStore("nn_core"), True)
    for key, desc in meta["scale_in_to_out"].items():
        cin, cout = (int(v) for v in key.split("->"))
        cm = cno.scale_in_to_out(G.SymTensor(tower, (1, 1), cin), G.SymTensor(tower, (1, 1), cout), axis_no=3)
        if desc[0] == "identity":
            assert cm.idx is None
        elif desc[0] == "repeat":
            assert (cm.idx == np.repeat(np.arange(cin), desc[1])).all()
        else:
            assert (cm.idx == arrs[f"map_{cin}_{cout}"]).all(), key


def test_basic_dataset_matches_reference(golden):
    arrs, meta = golden
    for tag in ("u16", "f32"):
        nb = meta[f"ds_{tag}"]["neighborhood"]
        ds = cno.BasicDataSet(None, arrs[f"ds_{tag}_casi"].copy(), arrs[f"ds_{tag}_lidar"].copy(), nb, True)
        assert ds.get_data_shape() == meta[f"ds_{tag}"]["data_shape"]
        assert ds.get_scene_shape() == meta[f"ds_{tag}"]["scene_shape"]
        assert ds.get_casi_band_count() == meta[f"ds_{tag}"]["band_count"]
        got = np.stack([ds.get_data_point(px, py) for px, py in arrs[f"ds_{tag}_points"]]).astype(np.float32)
        np.testing.assert_array_equal(got, arrs[f"ds_{tag}_patches"])
        ds2 = cno.BasicDataSet(None, arrs[f"ds_{tag}_casi"].copy(), None, nb, True)
        got2 = np.stack([ds2.get_data_point(px, py) for px, py in arrs[f"ds_{tag}_points"]]).astype(np.float32)
        np.testing.assert_array_equal(got2, arrs[f"ds_{tag}_hsi_patches"])


def test_metrics_match_reference(golden):
    arrs, meta = golden
    for i in range(meta["n_conf"]):
        conf = arrs[f"conf_{i}"]
        rec, prec = cno.calculate_class_accuracies_using_confusion(conf, range(0, conf.shape[0]))
        np.testing.assert_array_equal(rec, arrs[f"conf_{i}_recall"])
        np.testing.assert_array_equal(prec, arrs[f"conf_{i}_precision"])
        oa, aa, kappa = cno.confusion_metrics(conf)
        assert abs(kappa - float(arrs[f"conf_{i}_kappa"])) < 1e-12   # == utilities/stat_extractor.calc_kappa
        assert abs(oa - float(arrs[f"conf_{i}_oa"])) < 1e-15
    np.testing.assert_array_equal(cno.read_targets_from_image(arrs["targets_img"], range(1, 4)), arrs["targets_rows"])


def test_plugin_lookup_and_helpers(golden):
    _, meta = golden
    for name in ("HYPELCNNModel", "DUALCNNModel", "CONCNNModel"):
        m = cno.get_model_from_name(name)
        assert type(m).__name__ == name and hasattr(m, "create_tensor_graph") and hasattr(m, "get_loss_func")
    with pytest.raises(ImportError):
        cno.get_model_from_name("NoSuchModel")
    for rep, want in meta["is_integer_num"]:
        assert common_ops.is_integer_num(eval(rep)) == want
    assert common_ops.path_leaf("/a/b/c.json") == "c.json" and common_ops.path_leaf("/a/b/") == "b"


def test_graph_fusion_structure():
    """HYPELCNN records 6 spectral + 3x(level + connector) + FC nodes; levels are merged 4-branch nodes with
    residual epilogues, exactly one LinearNode per reference conv2d/fully_connected group."""
    alg = json.load(open(os.path.join(os.path.dirname(cno.__file__), "..", "nnmodel", "modelconfigs",
                                      "alg_param_hypelcnn.json")))
    model = cno.get_model_from_name("HYPELCNNModel")
    t = cno.Template("nn_core", model.create_tensor_graph, class_count=15)
    out = t(cno.ModelInputParams(cno.Placeholder("x", (7, 7), 145), None, "/gpu:0", True), algorithm_params=alg)
    nodes = out.tower.nodes
    assert all(isinstance(n, G.LinearNode) for n in nodes) and len(nodes) == 6 + 6 + 3 + 1 + 4
    levels = [n for n in nodes if len(n.branches) == 4]
    assert [n.cout for n in levels] == [240, 120, 60] and all(len(n.residuals) == 1 for n in levels)
    assert len(nodes[2].residuals) == 2 and len(nodes[5].residuals) == 2      # end of encoder / decoder stacks
    assert [n.dropout_keep for n in nodes if n.dropout_keep is not None] == [pytest.approx(0.3)] * 3
    names = sorted(v.name for v in t.store.order)
    assert "nn_core/conv_enc_0/weights" in names and "nn_core/connector_1_conv5x5/BatchNorm/moving_variance" in names
    assert "nn_core/fc_final/BatchNorm/beta" in names and "nn_core/image_gen_net_4/weights" in names
    # inference tower: shares variables, has no reconstruction head
    out2 = t(cno.ModelInputParams(cno.Placeholder("x", (7, 7), 145), None, "/gpu:0", False), algorithm_params=alg)
    assert out2.image_output is None and len(out2.tower.nodes) == len(nodes) - 4
    assert len(t.store.order) == len(names)
