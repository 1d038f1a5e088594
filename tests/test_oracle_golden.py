"""The oracle's host-side restatements vs vectors captured from the reference itself
(tests/golden/make_goldens.py ran the reference's numpy code in the build container)."""
import numpy as np

from oracle import host


def test_scale_in_to_out_maps(golden):
    arrs, meta = golden
    for key, desc in meta["scale_in_to_out"].items():
        cin, cout = (int(v) for v in key.split("->"))
        idx = host.scale_in_to_out_index(cin, cout)
        assert len(idx) == cout
        assert host.scale_in_to_out_kind(cin, cout) == desc[0], key
        if desc[0] == "identity":
            assert (idx == np.arange(cout)).all()
        elif desc[0] == "repeat":
            assert (idx == np.repeat(np.arange(cin), desc[1])).all(), key
        else:
            assert (idx == arrs[f"map_{cin}_{cout}"].astype(np.int64)).all(), key


def test_appendix_b2_maps():
    # SURVEY Appendix B.2 (K2)
    assert list(host.scale_in_to_out_index(145, 120)[:8]) == [0, 1, 2, 4, 5, 6, 7, 8]
    assert list(host.scale_in_to_out_index(145, 120)[-4:]) == [140, 141, 143, 144]
    assert list(host.scale_in_to_out_index(145, 480)[:8]) == [0, 0, 1, 1, 1, 2, 2, 2]
    assert (host.scale_in_to_out_index(480, 120) == 4 * np.arange(120)).all()
    assert (host.scale_in_to_out_index(120, 240) == np.arange(240) // 2).all()


def test_patch_dataset(golden):
    arrs, meta = golden
    for tag in ("u16", "f32"):
        nb = meta[f"ds_{tag}"]["neighborhood"]
        ds = host.PatchDataSet(arrs[f"ds_{tag}_casi"].copy(), arrs[f"ds_{tag}_lidar"].copy(), nb, True)
        assert ds.get_data_shape() == meta[f"ds_{tag}"]["data_shape"]
        assert ds.get_scene_shape() == meta[f"ds_{tag}"]["scene_shape"]
        np.testing.assert_array_equal(np.asarray(ds.casi_min), arrs[f"ds_{tag}_casi_min"])
        np.testing.assert_array_equal(np.asarray(ds.casi_max), arrs[f"ds_{tag}_casi_max"])
        got = np.stack([ds.get_data_point(px, py) for px, py in arrs[f"ds_{tag}_points"]]).astype(np.float32)
        np.testing.assert_array_equal(got, arrs[f"ds_{tag}_patches"])
        ds2 = host.PatchDataSet(arrs[f"ds_{tag}_casi"].copy(), None, nb, True)
        got2 = np.stack([ds2.get_data_point(px, py) for px, py in arrs[f"ds_{tag}_points"]]).astype(np.float32)
        np.testing.assert_array_equal(got2, arrs[f"ds_{tag}_hsi_patches"])


def test_metrics(golden):
    arrs, meta = golden
    for i in range(meta["n_conf"]):
        conf = arrs[f"conf_{i}"]
        rec, prec = host.class_accuracies_from_confusion(conf, range(0, conf.shape[0]))
        np.testing.assert_array_equal(rec, arrs[f"conf_{i}_recall"])
        np.testing.assert_array_equal(prec, arrs[f"conf_{i}_precision"])
        oa, aa, kappa = host.streaming_metrics_from_confusion(conf)
        assert abs(kappa - float(arrs[f"conf_{i}_kappa"])) < 1e-12
        assert abs(oa - float(arrs[f"conf_{i}_oa"])) < 1e-15
        ca = arrs[f"conf_{i}_class_acc"]
        assert abs(aa - np.nan_to_num(ca).mean()) < 1e-12


def test_read_targets(golden):
    arrs, _ = golden
    np.testing.assert_array_equal(host.read_targets_from_image(arrs["targets_img"], range(1, 4)), arrs["targets_rows"])


def test_lr_schedules():
    # K5: staircase LR at steps 0, 349, 350, 700
    f = host.exponential_decay_staircase
    assert f(3e-4, 0, 350, 0.96) == 3e-4 and f(3e-4, 349, 350, 0.96) == 3e-4
    assert abs(f(3e-4, 350, 350, 0.96) - 3e-4 * 0.96) < 1e-18
    assert abs(f(3e-4, 700, 350, 0.96) - 3e-4 * 0.96 ** 2) < 1e-18
    assert host.gan_lr(2e-4, 0, 1000) == 2e-4 and host.gan_lr(2e-4, 499, 1000) == 2e-4
    assert abs(host.gan_lr(2e-4, 750, 1000) - 1e-4) < 1e-18 and host.gan_lr(2e-4, 1000, 1000) == 0.0
