"""The model WIRING, pinned by execution of the reference's own plugin files (tests/golden/make_reference_graphs.py ran
/root/reference/nnmodel/*.py and gan/shadow_data_models.py UNCHANGED under the functional tf_slim stand-in of
tests/golden/tf_standin.py and committed what they recorded: tests/golden/reference_graphs.json / .npz).

  * the PRODUCT's plugins record, node for node, the Tower the reference's files record through the tf_slim facade;
  * oracle/models.py creates exactly the variables the reference creates, and its float64 forward pass reproduces the
    values the reference's wiring produces on the same parameters / inputs / dropout masks;
  * the GAN stacks: variable tables, kernel lists, widths, the ragged last slice, and float64 outputs.

Scope of the pin: layer order, scopes, widths, kernels, normaliser / activation / regulariser / keep-prob arguments,
residual channel maps.  NOT the operator semantics -- the stand-in evaluates with oracle/ops.py (SURVEY Appendix A stays
unpinned at the TensorFlow boundary)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
ROOT = os.path.dirname(HERE)
sys.path.insert(0, GOLD)

from graph_dump import dump_output, dump_store, dump_tower  # noqa: E402
from oracle import models as OM, ops as O  # noqa: E402

FIX = json.load(open(os.path.join(GOLD, "reference_graphs.json")))
ARR = np.load(os.path.join(GOLD, "reference_graphs.npz"))
HAVE_REF = os.path.isdir("/root/reference")


def _product_tower(model_name, alg, patch, channels, classes, is_training):
    from hypelcnn_amd import graph as G
    from hypelcnn_amd.common import common_nn_ops as P
    store = G.VariableStore("nn_core")
    tower = G.Tower(store, is_training)
    x = tower.placeholder("x", (patch, patch), channels)
    out = P.get_model_from_name(model_name).create_tensor_graph(
        P.ModelInputParams(x=x, y=None, device_id="/gpu:0", is_training=is_training), classes, alg)
    return {"tower": dump_tower(tower), "variables": dump_store(store), "y_conv": dump_output(out.y_conv),
            "image_output": None if out.image_output is None else dump_output(out.image_output)}


def _first_difference(a, b, path=""):
    if type(a) != type(b):
        return f"{path}: {type(a).__name__} vs {type(b).__name__}"
    if isinstance(a, dict):
        for k in sorted(set(a) | set(b)):
            if k not in a or k not in b:
                return f"{path}/{k}: only on one side"
            d = _first_difference(a[k], b[k], f"{path}/{k}")
            if d:
                return d
    elif isinstance(a, list):
        if len(a) != len(b):
            return f"{path}: {len(a)} vs {len(b)} entries"
        for i, (p, q) in enumerate(zip(a, b)):
            d = _first_difference(p, q, f"{path}[{i}]")
            if d:
                return d
    elif a != b:
        return f"{path}: {a!r} vs {b!r}"
    return None


@pytest.mark.parametrize("case", sorted(FIX["classifiers"]))
def test_product_plugin_records_the_graph_the_reference_file_records(case):
    """Every shipped modelconfigs/*.json at its BASELINE shape, training and inference towers: the product's
    nnmodel/<Model>.py and the reference's unchanged nnmodel/<Model>.py (through the tf_slim facade) give the same Tower --
    nodes, sources, branches (scope, kernel, width, bias / batch norm), activation, decay, dropout keep-prob, residual
    channel maps, views -- and the same variable table (names, shapes, trainable, regulariser scale, initialiser)."""
    c = FIX["classifiers"][case]
    for mode, training in (("train", True), ("eval", False)):
        got = _product_tower(c["model"], c["alg"], c["patch"], c["channels"], c["classes"], training)
        assert json.loads(json.dumps(got)) == c[mode]["tower"], _first_difference(json.loads(json.dumps(got)), c[mode]["tower"])


def _oracle_variable_table(model, patch, channels, classes, alg):
    if model == "HYPELCNNModel":
        table = OM.hypelcnn_layer_table(patch, channels, classes, alg)
        out = {}
        for scope, kind, k, cin, cout in table:
            out[scope + "/weights"] = [k, k, cin, cout] if kind == "conv" else [cin, cout]
            for nm in ("beta", "moving_mean", "moving_variance"):
                out[f"{scope}/BatchNorm/{nm}"] = [cout]
        return out
    table = (OM.dualcnn_layer_table if model == "DUALCNNModel" else OM.concnn_layer_table)(patch, channels, classes, alg)
    out = {}
    for scope, kind, k, cin, cout in table:
        out[scope + "/weights"] = [k, k, cin, cout] if kind == "conv" else [cin, cout]
        out[scope + "/biases"] = [cout]
    return out


@pytest.mark.parametrize("case", sorted(FIX["classifiers"]))
def test_oracle_layer_tables_are_the_reference_variable_tables(case):
    """oracle/models.py's layer tables (from which every parity test's parameters are drawn) name exactly the variables,
    with exactly the shapes, that the reference's create_tensor_graph creates -- incl. fc_stage_count, the batch norm on
    fc_final, the image head in the training tower only."""
    c = FIX["classifiers"][case]
    want = _oracle_variable_table(c["model"], c["patch"], c["channels"], c["classes"], c["alg"])
    assert c["train"]["variables"] == want, _first_difference(c["train"]["variables"], want)
    assert set(c["eval"]["variables"]) <= set(want)
    if c["model"] == "HYPELCNNModel":
        assert not any(k.startswith("image_gen_net") for k in c["eval"]["variables"])
    # facts the records carry that a table cannot: the keep-prob conventions and the normaliser arguments
    drops = [r["keep_prob"] for r in c["train"]["records"] if r["op"] == "dropout"]
    if c["model"] == "HYPELCNNModel":
        assert drops and all(abs(d - (1 - c["alg"]["drop_out_ratio"])) < 1e-12 for d in drops)  # keep = 1 - ratio
        last = [r for r in c["train"]["records"] if r.get("scope") == "fc_final"][0]
        assert last["normalizer"] == "batch_norm" and last["activation"] is None and last["bn_decay"] == c["alg"]["bn_decay"]
    else:
        assert drops and all(abs(d - c["alg"]["drop_out_ratio"]) < 1e-12 for d in drops)  # keep = ratio itself


@pytest.mark.parametrize("case", sorted(FIX["values"]))
def test_oracle_forward_reproduces_the_values_of_the_reference_wiring(case):
    """Small configurations, float64: the reference's create_tensor_graph evaluated op by op with oracle/ops.py (fixture)
    against oracle/models.py's hand-written forward pass on the same parameters, inputs and dropout masks -- training
    (batch statistics, dropout, image head) and inference (moving statistics)."""
    c = FIX["values"][case]
    pre = f"{case}/param/"
    params = {k[len(pre):]: ARR[k] for k in ARR.files if k.startswith(pre)}
    x = ARR[f"{case}/x"]
    fwd = {"HYPELCNNModel": OM.hypelcnn_forward, "DUALCNNModel": OM.dualcnn_forward, "CONCNNModel": OM.concnn_forward}[c["model"]]
    for mode, training in (("train", True), ("eval", False)):
        mk = f"{case}/{mode}/dropout_"
        masks = {k[len(f"{case}/{mode}/"):]: ARR[k] for k in ARR.files if k.startswith(mk)}
        ctx = OM.Ctx(params, training, masks)
        out = fwd(ctx, O.Var(x.astype(np.float64)), c["classes"], c["alg"])
        want = ARR[f"{case}/{mode}/y_conv"]
        np.testing.assert_allclose(out["y_conv"].v, want, rtol=1e-11, atol=1e-12, err_msg=f"{case}/{mode} logits")
        if f"{case}/{mode}/image_output" in ARR.files:
            np.testing.assert_allclose(out["image_output"].v, ARR[f"{case}/{mode}/image_output"], rtol=1e-11, atol=1e-12)
        else:
            assert out["image_output"] is None


@pytest.mark.parametrize("bands", [24, 64, 144, 360])
def test_gan_stacks_variable_tables_and_structure(bands):
    """shadowdata_{generator, discriminator, feature_discriminator}_model as the reference's file builds them: the
    oracle's tables and the product's variable store name the same variables with the same shapes; kernel list B, B/2,
    B/4, B/8 (, B/4, B/2, B), encoder = the first four; 0.1 leaky-ReLU everywhere but the last discriminator layer and
    tanh on net7; L2 on the first two discriminator layers only; the slices of the feature discriminator incl. the
    ragged last one."""
    g = FIX["gan"]
    gen, enc = g[f"generator_{bands}"], g[f"encoder_{bands}"]
    ks = OM.generator_kernel_sizes(bands)
    assert gen["variables"] == {f"net{i}/{nm}": ([k, 1, 1] if nm == "weights" else [1])
                                for i, k in enumerate(ks, 1) for nm in ("weights", "biases")}
    assert enc["variables"] == {k: v for k, v in gen["variables"].items() if int(k[3]) <= 4}
    convs = [r for r in gen["records"] if r["op"] == "convolution1d"]
    assert [r["kernel"][0] for r in convs] == ks and all(r["padding"] == "SAME" and r["biases"] for r in convs)
    assert [r["activation"] for r in convs] == [["leaky_relu", 0.1]] * 6 + [["tanh"]]
    assert all(r["initializer"] == ["zeros"] for r in convs)
    # skip structure: net_l = conv(net_{l-1}) + net_{l-1} + net_{l-2} (net1: + net0 only; net7: no skips)
    adds = [r for r in gen["records"] if r["op"] == "add"]
    assert len(adds) == 1 + 2 * 5
    dis = g[f"discriminator_{bands}"]
    assert dis["variables"] == {f"{s}/{nm}": ([cin, cout] if nm == "weights" else [cout])
                                for s, cin, cout in OM.discriminator_layer_table(bands) for nm in ("weights", "biases")}
    fcs = [r for r in dis["records"] if r["op"] == "fully_connected"]
    assert [r["activation"] for r in fcs] == [["leaky_relu", 0.1], ["leaky_relu", 0.1], None]
    assert [r["regularizer"] for r in fcs] == [1e-4, 1e-4, None]
    feat = g[f"feature_discriminator_{bands}"]
    table = OM.feature_discriminator_layer_table(bands, feat["patches"], feat["embed"])
    assert feat["variables"] == {f"{s}/{nm}": ([cin, cout] if nm == "weights" else [cout])
                                 for s, cin, cout in table for nm in ("weights", "biases")}
    slices, ps = OM.feature_discriminator_slices(bands, feat["patches"])
    got_slices = [tuple(r["ranges"][0]) for r in feat["records"] if r["op"] == "slice"]
    assert got_slices == slices and feat["out_shape"] == [len(slices), feat["embed"]]
    if bands == 64:
        assert slices[-1] == (60, 64) and len(slices) == 7  # 64 // 6 = 10: six full slices and a ragged seventh
    # the product's builders create the same variables (names relative to the scope they are built in)
    from hypelcnn_amd import graph as G
    from hypelcnn_amd.gan import shadow_data_models as PG
    for key, build in (("generator", lambda x: PG.shadowdata_generator_model(x, False, True)),
                       ("discriminator", lambda x: PG.shadowdata_discriminator_model(x, x, True, 1e-4)),
                       ("feature_discriminator", lambda x: PG.shadowdata_feature_discriminator_model(
                           x, feat["patches"], feat["embed"], True, 1e-3))):
        store = G.VariableStore("")
        tower = G.Tower(store, True)
        with G.variable_scope("probe_" + key):
            build(tower.placeholder("x", None, bands))
        got = {v.name.split("/", 1)[1]: list(v.shape) for v in store.order}
        assert got == g[f"{key}_{bands}"]["variables"], key
        regs = {v.name.split("/", 1)[1]: v.l2_scale for v in store.order if v.name.endswith("weights")}
        want_regs = {r["scope"] + "/weights": (r["regularizer"] or 0.0) for r in g[f"{key}_{bands}"]["records"]
                     if r["op"] in ("fully_connected", "convolution1d")}
        assert regs == want_regs, key


@pytest.mark.parametrize("bands", [24, 64])
def test_oracle_gan_forward_reproduces_the_values_of_the_reference_wiring(bands):
    for key, fwd in (("generator", lambda ctx, x: OM.generator_forward(ctx, x)),
                     ("encoder", lambda ctx, x: OM.generator_forward(ctx, x, only_encoder=True)),
                     ("discriminator", lambda ctx, x: OM.discriminator_forward(ctx, x)),
                     ("feature_discriminator", lambda ctx, x: OM.feature_discriminator_forward(ctx, x, 6, 2))):
        k = f"{key}_{bands}"
        pre = f"gan/{k}/param/"
        params = {n[len(pre):]: ARR[n] for n in ARR.files if n.startswith(pre)}
        if key == "encoder":  # the oracle's generator context holds all seven layers; the encoder reads the first four
            for i, kk in enumerate(OM.generator_kernel_sizes(bands), 1):
                params.setdefault(f"net{i}/weights", np.zeros((kk, 1, 1)))
                params.setdefault(f"net{i}/biases", np.zeros((1,)))
        out = fwd(OM.Ctx(params, True), O.Var(ARR[f"gan/{k}/x"]))
        want = ARR[f"gan/{k}/out"]
        np.testing.assert_allclose(out.v.reshape(want.shape), want, rtol=1e-11, atol=1e-12, err_msg=k)


@pytest.mark.skipif(not HAVE_REF, reason="the reference is only present in the build container")
def test_fixture_is_what_the_reference_records_today():
    """In the build container: re-run the reference's HYPELCNNModel / DUALCNNModel / CONCNNModel through both engines
    and compare with the committed fixture (a stale fixture would pin nothing)."""
    import subprocess
    code = r"""
import json, os, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import tf_standin as S, make_reference_graphs as M
S.install()
fix = json.load(open(os.path.join(%r, "reference_graphs.json")))
for name, model, cfg, patch, ch, classes in M.CLASSIFIER_CASES:
    alg = json.load(open(os.path.join(M.CFG, cfg)))
    for mode, training in (("train", True), ("eval", False)):
        eng, _ = M.record_classifier(model, alg, patch, ch, classes, training)
        assert json.loads(json.dumps(eng.records)) == fix["classifiers"][name][mode]["records"], (name, mode, "records")
        assert eng.variables == fix["classifiers"][name][mode]["variables"], (name, mode, "variables")
        t = M.tower_via_facade(model, alg, patch, ch, classes, training)
        assert json.loads(json.dumps(t)) == fix["classifiers"][name][mode]["tower"], (name, mode, "tower")
print("ok")
""" % (os.path.dirname(HERE), GOLD, GOLD)
    # a fresh interpreter: the stand-in installs a meta-path finder and shims numpy attributes
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-2000:]


# ------------------------------------------------------------------------------------------------ the facade, end to end
@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference checkout exists in the build container only")
@pytest.mark.parametrize("model_name,patch,ch,classes,alg,nb", [
    ("HYPELCNNModel", 5, 11, 4, {"drop_out_ratio": 0.7, "filter_count": 48, "learning_rate": 3e-4,
                                 "learning_rate_decay_factor": 0.96, "learning_rate_decay_step": 350, "lrelu_alpha": 0.18,
                                 "optimizer": "AdamOptimizer", "bn_decay": 0.95, "l2regularizer_scale": 1e-5,
                                 "spectral_hierarchy_level": 3, "spatial_hierarchy_level": 3, "degradation_coeff": 3,
                                 "use_residual": True, "batch_size": 6}, 6),
    ("DUALCNNModel", 5, 7, 3, {"drop_out_ratio": 0.7, "lrelu_alpha": 0.18, "filter_count": 32, "hs_lidar_diff": 1,
                               "optimizer": "AdamOptimizer", "learning_rate": 3e-4, "learning_rate_decay_factor": 0.96,
                               "learning_rate_decay_step": 350}, 4),
    ("CONCNNModel", 5, 9, 3, {"drop_out_ratio": 0.5, "filter_count": 6, "optimizer": ["MomentumOptimizer", 0.9],
                              "learning_rate": 1e-3, "learning_rate_decay_factor": 0.01,
                              "learning_rate_decay_step": 33333}, 4),
])
def test_reference_plugin_file_trains_through_the_facade(model_name, patch, ch, classes, alg, nb):
    """`hypelcnn_amd.tf_facade.reference_model`: the reference's UNCHANGED plugin file in the place of the product's plugin
    -- optimize_nn, planning, one training step on the kernel emulation -- gives the product plugin's numbers bit for bit
    (the two record the same Tower) and the oracle's to the usual tolerance.  Runs in a subprocess: the facade's finder
    and the reference's `common` / `nnmodel` packages must not leak into this test process."""
    code = """
import sys, json
import numpy as np
sys.path.insert(0, %r)
from hypelcnn_amd import tf_facade
from tests import parity_util as U
from tests.emu_backend import EmuBackend
model_name, patch, ch, classes, alg, nb = json.loads(%r)
def run(model):
    rng = np.random.default_rng(3)
    built = U.build(model_name, patch, ch, classes, alg, EmuBackend(), model=model)
    sess = built.ctx.session()
    params = U.make_params(model_name, patch, ch, classes, alg, rng)
    U.inject(sess, params)
    x = rng.random((nb, patch, patch, ch)).astype(np.float32)
    onehot = np.eye(classes, dtype=np.float32)[rng.integers(0, classes, nb)]
    masks = U.make_masks(built, nb, rng)
    ct = U.run_train_step(built, x, onehot, masks)
    ref, err, worst = U.compare_step(built, ct, params, x, onehot, masks, model_name, classes, alg, tol_logit=2e-5, tol_grad=2e-4)
    return np.concatenate([np.asarray(sess.get_gradient("nn_core/" + k), np.float32).ravel() for k in sorted(params)])
a = run(None)
b = run(tf_facade.reference_model(model_name, "/root/reference"))
assert a.shape == b.shape and np.array_equal(a, b), float(np.abs(a - b).max())
print("FACADE_OK", a.size)
""" % (ROOT, json.dumps([model_name, patch, ch, classes, alg, nb]))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert "FACADE_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference checkout exists in the build container only")
def test_facade_hijack_is_scoped_and_leaves_the_process_clean():
    """Round-5 advisor finding: `reference_model()` used to leave a permissive stub finder at sys.meta_path[0], the checkout on
    sys.path and stub `sklearn` / `tqdm` modules behind.  Now: after building a model (and its loss, through the REFERENCE's
    own get_loss_func) nothing of it is left in the import system, installed packages import as themselves, and the product's
    data split (which imports sklearn lazily) works in the same process."""
    code = """
import sys, importlib
sys.path.insert(0, %r)
before_path, before_meta = list(sys.path), list(sys.meta_path)
from hypelcnn_amd import tf_facade, graph as G
from hypelcnn_amd.common import common_nn_ops as P
m = tf_facade.reference_model("HYPELCNNModel", "/root/reference")
alg = {"drop_out_ratio": 0.7, "filter_count": 48, "lrelu_alpha": 0.18, "bn_decay": 0.95, "l2regularizer_scale": 1e-5,
       "spectral_hierarchy_level": 3, "spatial_hierarchy_level": 3, "degradation_coeff": 3, "use_residual": True}
tower = G.Tower(G.VariableStore("nn_core"), True)
x = tower.placeholder("x", (5, 5), 11)
out = m.create_tensor_graph(P.ModelInputParams(x=x, y=None, device_id="/gpu:0", is_training=True), 4, alg)
labels = tower.placeholder("labels", None, 4)
loss = m.get_loss_func(out, labels)                      # the reference's own text (HYPELCNNModel.py:101-112)
ref = P.get_model_from_name("HYPELCNNModel").get_loss_func(out, labels)
assert type(loss) is type(ref) is G.PerSampleXent and loss.extra_mse is not None and ref.extra_mse is not None
assert loss.logits is ref.logits and loss.labels is ref.labels
assert loss.extra_mse.a is ref.extra_mse.a and [s for s in loss.extra_mse.b.sources] == [s for s in ref.extra_mse.b.sources]
clean = lambda: sys.path == before_path and not any(isinstance(f, tf_facade._Finder) for f in sys.meta_path)
assert clean(), "finder / checkout left behind"
for name in ("tensorflow", "tf_slim", "common", "nnmodel", "common.common_nn_ops"):
    assert name not in sys.modules, name
assert not hasattr(__import__("numpy"), "int")
import sklearn.model_selection, tqdm
assert "site-packages" in sklearn.__file__ or "dist-packages" in sklearn.__file__
import numpy as np
pts = np.stack([np.arange(40), np.arange(40), np.arange(40) %% 2], axis=1)
tr, va = P.shuffle_training_data_using_ratio(pts, 0.25)     # the product's data split imports sklearn lazily: the REAL one
assert len(tr) == 10 and len(va) == 30
m.create_tensor_graph(P.ModelInputParams(x=G.Tower(G.VariableStore("nn_core"), False).placeholder("x", (5, 5), 11), y=None,
                                         device_id="/gpu:0", is_training=False), 4, alg)   # and the model still works afterwards
assert clean()
print("SCOPED_OK")
""" % (ROOT,)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert "SCOPED_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
