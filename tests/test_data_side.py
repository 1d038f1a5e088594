"""Data side of the path: patch gather from the resident scene, the fused augmentation map,
prediction scatter, GeneratorImporter and full-scene inference.

CPU tests run the host logic on the numpy kernel emulation and pin it to (a) the patches captured from the
reference's BasicDataSet (tests/golden), (b) an independent torch composition of the augmentation maps;
`-m gpu` tests compare the HIP kernels with the emulation bit for bit (pure data movement + one division + one add).
"""
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from hypelcnn_amd.backend import Ref
from hypelcnn_amd.common import common_nn_ops as cno
from hypelcnn_amd.common.tiff_io import imread, imwrite
from tests.emu_backend import EmuBackend

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    arrs = np.load(os.path.join(HERE, "golden", "reference_numpy_side.npz"))
    meta = json.load(open(os.path.join(HERE, "golden", "reference_numpy_side.json")))
    return arrs, meta


# ------------------------------------------------------------------------------------------------ references
def torch_augment(x, d, ratio=None, alt=None):
    """Independent composition in the reference's order (common_nn_ops.py:376-440): rot90 -> shadow -> flip lr ->
    flip ud -> spectral shift.  x: [b,P,P,C] (already gathered)."""
    b = x.shape[0]
    if "rot_k" in d:
        kk = d["rot_k"].view(b, 1, 1, 1)
        out = x
        for k in (1, 2, 3):
            out = torch.where(kk == k, torch.rot90(x, k, dims=(1, 2)), out)
        x = out
    if "shadow_pick" in d:
        pick = d["shadow_pick"].bool().view(b, 1, 1, 1)
        if alt is not None:  # the generator output was computed on the un-rotated batch: rotate it the same way
            sh = alt
            if "rot_k" in d:
                kk = d["rot_k"].view(b, 1, 1, 1)
                o = alt
                for k in (1, 2, 3):
                    o = torch.where(kk == k, torch.rot90(alt, k, dims=(1, 2)), o)
                sh = o
        else:
            sh = x / ratio
        x = torch.where(pick, sh, x)
    if "flip_lr" in d:
        x = torch.where(d["flip_lr"].bool().view(b, 1, 1, 1), torch.flip(x, dims=(2,)), x)
    if "flip_ud" in d:
        x = torch.where(d["flip_ud"].bool().view(b, 1, 1, 1), torch.flip(x, dims=(1,)), x)
    if "delta" in d:
        x = x + d["delta"].view(b, 1, 1, -1)
    return x


def _draws(b, c, seed, rot3=False):
    g = torch.Generator().manual_seed(seed)
    return {"rot_k": torch.randint(0, 4 if rot3 else 3, (b,), generator=g).to(torch.int32),
            "shadow_pick": (torch.rand(b, generator=g) < 0.5).to(torch.uint8),
            "flip_lr": (torch.rand(b, generator=g) < 0.5).to(torch.uint8),
            "flip_ud": (torch.rand(b, generator=g) < 0.5).to(torch.uint8),
            "delta": torch.rand(b, c, generator=g) * 0.05 - 0.05}


def _run_augment(be, x, idx, d, ratio, alt, n, p, c):
    up = lambda t: None if t is None else Ref(be.upload(t.numpy() if isinstance(t, torch.Tensor) else t))
    out = be.zeros(n * p * p * c)
    be.call("augment_patches_f32", up(x), up(idx), n, p, c, up(d.get("rot_k")), up(d.get("shadow_pick")), up(ratio),
            up(alt), up(d.get("flip_lr")), up(d.get("flip_ud")), up(d.get("delta")), Ref(out))
    be.synchronize()
    return out.cpu().numpy().reshape(n, p, p, c)


AUG_CASES = [(37, 7, 145, "ratio"), (16, 11, 49, "alt"), (5, 1, 64, "ratio"), (9, 3, 1, None), (1, 5, 360, "alt")]


@pytest.mark.parametrize("n,p,c,shadow", AUG_CASES)
def test_augment_spec_matches_torch_composition(n, p, c, shadow):
    rng = np.random.default_rng(n * 100 + p)
    pool = torch.tensor(rng.random((n + 11, p, p, c), dtype=np.float32))
    idx = torch.tensor(rng.integers(0, n + 11, n))
    d = _draws(n, c, seed=n + p + c, rot3=True)
    ratio = alt = None
    if shadow == "ratio":
        ratio = torch.tensor(rng.uniform(1.2, 3.0, c).astype(np.float32))
    elif shadow == "alt":
        alt = torch.tensor(rng.random((n, p, p, c), dtype=np.float32))
    else:
        d.pop("shadow_pick")
    got = _run_augment(EmuBackend(), pool, idx, d, ratio, alt, n, p, c)
    want = torch_augment(pool.index_select(0, idx), d, ratio, alt).numpy()
    np.testing.assert_array_equal(got, want)


def test_draw_order_and_product_iterator_use_the_kernel():
    info = cno.AugmentationInfo(SimpleNamespace(ratio=np.full(6, 2.0, np.float32), shadow_op=None), True, True, 0.05,
                                True, 0.5)
    g = torch.Generator().manual_seed(1234)
    d = cno.draw_augmentations(8, 6, info, g)
    assert list(d) == ["rot_k", "shadow_pick", "flip_lr", "flip_ud", "delta"]
    assert int(d["rot_k"].max()) <= 2 and float(d["delta"].max()) <= 0 and float(d["delta"].min()) >= -0.05
    be = EmuBackend()
    it = cno.BatchIterator((3, 3, 6), 4, 8, True, None, info)
    data = np.random.default_rng(0).random((20, 3, 3, 6), dtype=np.float32)
    it.initializer(data, np.arange(20) % 4, be)
    be.launch_log = []
    real_call = be.call
    seen = []
    be.call = lambda name, *a: (seen.append(name), real_call(name, *a))[1]
    x, onehot, lab = it.next_batch()
    assert seen == ["augment_patches_f32"] and x.shape == (8, 3, 3, 6) and onehot.shape == (8, 4)


# ------------------------------------------------------------------------------------------------ gather
def test_scene_gather_matches_reference_patches(golden):
    arrs, meta = golden
    be = EmuBackend()
    for tag in ("u16", "f32"):
        nb = meta[f"ds_{tag}"]["neighborhood"]
        for lidar_key, patches_key in ((f"ds_{tag}_lidar", f"ds_{tag}_patches"), (None, f"ds_{tag}_hsi_patches")):
            lidar = None if lidar_key is None else arrs[lidar_key].copy()
            ds = cno.BasicDataSet(None, arrs[f"ds_{tag}_casi"].copy(), lidar, nb, True)
            pts = arrs[f"ds_{tag}_points"]
            targets = np.concatenate([pts, np.zeros((len(pts), 1), pts.dtype)], axis=1)
            sa = cno.SceneArrays()
            sa.feed(ds, targets, be)
            got, _ = sa.gather(torch.arange(len(pts)))
            np.testing.assert_array_equal(got.numpy(), arrs[patches_key])


# ------------------------------------------------------------------------------------------------ importer + inference
ALG = {"batch_size": 32, "drop_out_ratio": 0.3, "filter_count": 32, "learning_rate": 3e-3,
       "learning_rate_decay_factor": 0.96, "learning_rate_decay_step": 350, "lrelu_alpha": 0.18,
       "optimizer": "AdamOptimizer", "bn_decay": 0.9, "l2regularizer_scale": 1e-5, "spectral_hierarchy_level": 1,
       "spatial_hierarchy_level": 1, "degradation_coeff": 3, "use_residual": True}
SCENE = "grss2013:h=20:w=26:bands=10:classes=3:samples=0.6"


def _train(tmp_path, importer, backend, steps=40):
    from hypelcnn_amd.classify import train_for_classification as T
    os.makedirs(tmp_path, exist_ok=True)
    p = tmp_path / "alg.json"
    p.write_text(json.dumps(ALG))
    argv = ["--loader_name", "SyntheticDataLoader", "--path", SCENE, "--neighborhood", "1", "--model_name",
            "HYPELCNNModel", "--algorithm_param_path", str(p), "--batch_size", "32", "--step", str(steps),
            "--base_log_path", str(tmp_path / "log"), "--importer_name", importer, "--save_checkpoint_steps", "1000",
            "--perform_validation", "false"]
    flags, _ = T.build_parser().parse_known_args(argv)
    log_dir = os.path.join(flags.base_log_path, T.get_log_suffix(flags))
    res = T.perform_an_episode(flags, dict(ALG), T.get_model_from_name(flags.model_name), log_dir, backend=backend)
    return res, log_dir, str(p)


def _infer(tmp_path, log_dir, alg_path, backend, domain="all", batch=64):
    from hypelcnn_amd.classify import infer_for_classification as I
    out = tmp_path / f"out_{domain}"
    argv = ["--loader_name", "SyntheticDataLoader", "--path", SCENE, "--neighborhood", "1", "--model_name",
            "HYPELCNNModel", "--algorithm_param_path", alg_path, "--batch_size", str(batch), "--base_log_path", log_dir,
            "--output_path", str(out), "--domain", domain]
    return I.main(argv, backend=backend), str(out)


def test_generator_importer_trains_like_in_memory_importer(tmp_path):
    """Same seed, same shuffles: cutting batches from the resident scene must give the very same training run as
    the materialised [N,P,P,C] data set."""
    r1, _, _ = _train(tmp_path / "a", "InMemoryImporter", EmuBackend(), steps=12)
    r2, _, _ = _train(tmp_path / "b", "GeneratorImporter", EmuBackend(), steps=12)
    assert r1.loss == r2.loss and r1.test_accuracy == r2.test_accuracy


def test_full_scene_inference_on_emulation(tmp_path):
    res, log_dir, alg_path = _train(tmp_path, "InMemoryImporter", EmuBackend(), steps=60)
    raster, out = _infer(tmp_path, log_dir, alg_path, EmuBackend(), "all", batch=64)
    assert raster.shape == (20, 26) and raster.dtype == np.uint8 and raster.max() < 3
    # against the ground truth the synthetic loader painted: a trained model labels most of the scene correctly
    from hypelcnn_amd.classify.infer_for_classification import gt_process
    gt, colors = gt_process(SimpleNamespace(loader_name="SyntheticDataLoader", path=SCENE))
    known = gt != 255
    assert (raster[known] == gt[known]).mean() > 0.85
    # a different batch size (ragged last batch) gives the same raster; the rasters round-trip through TIFF
    raster2, _ = _infer(tmp_path, log_dir, alg_path, EmuBackend(), "all", batch=37)
    np.testing.assert_array_equal(raster, raster2)
    np.testing.assert_array_equal(imread(os.path.join(out, "result_raw.tif")), raster)
    col = imread(os.path.join(out, "result_colorized.tif"))
    np.testing.assert_array_equal(col, cno.create_colored_image(raster, colors))
    # sample domain: only sampled pixels are written, the rest keeps the 255 fill
    raster3, _ = _infer(tmp_path, log_dir, alg_path, EmuBackend(), "sample")
    assert (raster3[~known] == 255).all() and (raster3[known] == raster[known]).all()


def test_tiff_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    for shape in ((5, 7), (4, 3, 3), (1, 1)):
        a = rng.integers(0, 255, shape).astype(np.uint8)
        imwrite(str(tmp_path / "t.tif"), a)
        np.testing.assert_array_equal(imread(str(tmp_path / "t.tif")), a)


# ------------------------------------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def hip():
    from hypelcnn_amd.backend import HipBackend
    return HipBackend()


@pytest.mark.gpu
@pytest.mark.parametrize("n,p,c,shadow", AUG_CASES + [(1024, 7, 145, "ratio")])
def test_gpu_augment_bit_exact(hip, n, p, c, shadow):
    rng = np.random.default_rng(n + p)
    pool = torch.tensor(rng.random((n + 11, p, p, c), dtype=np.float32))
    idx = torch.tensor(rng.integers(0, n + 11, n))
    d = _draws(n, c, seed=7 * n + c, rot3=True)
    ratio = alt = None
    if shadow == "ratio":
        ratio = torch.tensor(rng.uniform(1.2, 3.0, c).astype(np.float32))
    elif shadow == "alt":
        alt = torch.tensor(rng.random((n, p, p, c), dtype=np.float32))
    else:
        d.pop("shadow_pick")
    got = _run_augment(hip, pool, idx, d, ratio, alt, n, p, c)
    want = _run_augment(EmuBackend(), pool, idx, d, ratio, alt, n, p, c)
    np.testing.assert_array_equal(got, want)
    # every selector NULL = plain gather
    got = _run_augment(hip, pool, idx, {}, None, None, n, p, c)
    np.testing.assert_array_equal(got, pool.index_select(0, idx).numpy())


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,cc,cl,p,n", [(30, 41, 144, 1, 7, 500), (16, 16, 48, 1, 11, 36), (9, 9, 5, 0, 1, 81),
                                           (12, 20, 360, 0, 3, 7)])
def test_gpu_gather_and_scatter_bit_exact(hip, h, w, cc, cl, p, n):
    rng = np.random.default_rng(h * w)
    nb = p // 2
    casi = rng.random((h + 2 * nb, w + 2 * nb, cc), dtype=np.float32)
    lidar = rng.random((h + 2 * nb, w + 2 * nb, cl), dtype=np.float32) if cl else None
    pts = np.stack([rng.integers(0, w, n), rng.integers(0, h, n)], axis=1).astype(np.int32)
    outs = []
    for be in (hip, EmuBackend()):
        o = be.zeros(n * p * p * (cc + cl))
        be.call("gather_patches_f32", Ref(be.upload(casi)), None if lidar is None else Ref(be.upload(lidar)),
                h + 2 * nb, w + 2 * nb, cc, cl, Ref(be.upload(pts)), n, p, Ref(o))
        be.synchronize()
        outs.append(o.cpu().numpy())
    np.testing.assert_array_equal(outs[0], outs[1])
    # argmax + scatter, with ties (first maximum wins) and a strided logits matrix
    k, ld = 15, 24
    logits = rng.integers(-3, 4, (n, ld)).astype(np.float32)
    uniq = np.unique(pts[:, 1].astype(np.int64) * w + pts[:, 0], return_index=True)[1]  # duplicates race: drop them
    pts_u = np.ascontiguousarray(pts[uniq])
    lg_u = np.ascontiguousarray(logits[uniq])
    rasters = []
    for be in (hip, EmuBackend()):
        r = be.upload(np.full(h * w, 255, np.uint8))
        be.call("argmax_scatter", Ref(be.upload(lg_u)), ld, len(uniq), k, Ref(be.upload(pts_u)), Ref(r), w)
        be.synchronize()
        rasters.append(r.cpu().numpy())
    np.testing.assert_array_equal(rasters[0], rasters[1])
    assert (rasters[0] != 255).sum() == len(uniq)


@pytest.mark.gpu
def test_gpu_full_scene_inference_matches_emulation(hip, tmp_path):
    res, log_dir, alg_path = _train(tmp_path, "GeneratorImporter", hip, steps=60)
    from hypelcnn_amd.classify import infer_for_classification as I
    raster_gpu, _ = _infer(tmp_path, log_dir, alg_path, hip, "all", batch=128)
    margin = np.full(raster_gpu.shape, np.inf, np.float32)
    real = I.perform_prediction
    I.perform_prediction = lambda s, p, r: real(s, p, r, margin_result=margin)
    try:
        raster_emu, _ = _infer(tmp_path / "emu", log_dir, alg_path, EmuBackend(), "all", batch=128)
    finally:
        I.perform_prediction = real
    # same checkpoint; the fp32 GEMM order differs from the float64-accumulating emulation.  north_star: per-pixel
    # labels bit-exact -- asserted for every pixel whose two best logits (emulation) are more than 1e-4 apart, i.e.
    # clear of fp32 rounding; the remaining near-ties are counted
    assert np.isfinite(margin).all()
    clear = margin > 1e-4
    np.testing.assert_array_equal(raster_gpu[clear], raster_emu[clear])
    n_tie = int((~clear).sum())
    print(f"\nfull-scene labels: {int(clear.sum())} pixels exact, {n_tie} near-ties (top-2 gap <= 1e-4), of which "
          f"{int((raster_gpu != raster_emu).sum())} differ")
    assert n_tie <= 0.005 * margin.size


@pytest.mark.gpu
def test_gpu_full_scene_inference_matches_the_oracle_logits(hip, tmp_path):
    """Row f1 against the ORACLE itself (not only the emulation): the label raster the HIP inference path writes for the
    whole scene equals argmax of the float64 oracle's logits (oracle/train.py forward, inference mode, same checkpoint)
    wherever the oracle's two best logits are more than 1e-4 apart; near-ties are counted."""
    import glob
    from hypelcnn_amd.loader.SyntheticDataLoader import SyntheticDataLoader
    from oracle import train as OT
    res, log_dir, alg_path = _train(tmp_path, "GeneratorImporter", hip, steps=60)
    raster_gpu, _ = _infer(tmp_path, log_dir, alg_path, hip, "all", batch=128)
    ckpt = sorted(glob.glob(os.path.join(log_dir, "model.ckpt-*.npz")))[-1]
    with np.load(ckpt) as z:
        state = {k.replace("|", "/"): z[k] for k in z.files}
    params = {k[len("nn_core/"):]: v.astype(np.float64) for k, v in state.items() if k.startswith("nn_core/")
              and "/Adam" not in k}
    loader = SyntheticDataLoader(SCENE)
    data_set = loader.load_data(1, True)
    h, w = raster_gpu.shape
    patches = np.stack([data_set.get_data_point(x, y) for y in range(h) for x in range(w)]).astype(np.float64)
    cc = loader.get_class_count()
    classes = len(cc) if hasattr(cc, "__len__") else int(cc)
    ref = OT.forward_backward("HYPELCNNModel", params, patches, None, classes, ALG, False)["logits"]
    top2 = np.sort(ref, axis=1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 1e-4
    want = ref.argmax(1).astype(raster_gpu.dtype).reshape(h, w)
    np.testing.assert_array_equal(raster_gpu[clear.reshape(h, w)], want[clear.reshape(h, w)])
    n_tie = int((~clear).sum())
    print(f"\nfull-scene labels vs fp64 oracle: {int(clear.sum())} pixels exact, {n_tie} near-ties")
    assert n_tie <= 0.005 * clear.size


# ------------------------------------------------------------------------------------------------ GRSS2018 (2x)
@pytest.fixture(scope="module")
def golden18():
    return np.load(os.path.join(HERE, "golden", "reference_grss2018.npz"))


def _grss2018(g, tag):
    from hypelcnn_amd.loader.GRSS2018DataLoader import GRSS2018DataSet
    return GRSS2018DataSet(shadow_creator_dict=None, casi=g[f"{tag}_casi"].copy(), lidar=g[f"{tag}_lidar"].copy(),
                           neighborhood=int(g[f"{tag}_nb"]), normalize=True)


def test_grss2018_dataset_and_gather_match_reference_patches(golden18):
    """Host get_data_point and the device gather (emulated) against patches captured from the reference's
    GRSS2018DataSet (half-resolution HSI under the LiDAR grid)."""
    g = golden18
    be = EmuBackend()
    for tag in ("a", "b"):
        ds = _grss2018(g, tag)
        assert list(ds.get_data_shape()) == list(g[f"{tag}_data_shape"])
        assert list(ds.get_scene_shape()) == list(g[f"{tag}_scene_shape"])
        pts = g[f"{tag}_points"]
        host = np.stack([ds.get_data_point(int(px), int(py)) for px, py in pts]).astype(np.float32)
        np.testing.assert_array_equal(host, g[f"{tag}_patches"])
        sa = cno.SceneArrays()
        sa.feed(ds, np.concatenate([pts, np.zeros((len(pts), 1), pts.dtype)], axis=1), be)
        got, _ = sa.gather(torch.arange(len(pts)))
        np.testing.assert_array_equal(got.numpy(), g[f"{tag}_patches"])


def test_two_resolution_scene_trains_through_both_importers(tmp_path):
    global SCENE
    old = SCENE
    SCENE = "grss2018hr:h=20:w=26:bands=6:classes=3:samples=0.6"
    try:
        r1, _, _ = _train(tmp_path / "a", "InMemoryImporter", EmuBackend(), steps=8)
        r2, _, _ = _train(tmp_path / "b", "GeneratorImporter", EmuBackend(), steps=8)
    finally:
        SCENE = old
    assert np.isfinite(r1.loss) and r1.loss == r2.loss


@pytest.mark.gpu
def test_gpu_gather_2x_bit_exact(hip, golden18):
    g = golden18
    for tag in ("a", "b"):
        ds = _grss2018(g, tag)
        pts = g[f"{tag}_points"]
        sa = cno.SceneArrays()
        sa.feed(ds, np.concatenate([pts, np.zeros((len(pts), 1), pts.dtype)], axis=1), hip)
        got, _ = sa.gather(torch.arange(len(pts), device=hip.device))
        hip.synchronize()
        np.testing.assert_array_equal(got.cpu().numpy(), g[f"{tag}_patches"])
