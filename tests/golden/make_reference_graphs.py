#!/usr/bin/env python3
"""Pin the model WIRING by executing the reference's own plugin files (build container only; needs /root/reference).

`tests/golden/tf_standin.py` serves a functional `tensorflow` / `tf_slim` under which
    /root/reference/nnmodel/{HYPELCNNModel,DUALCNNModel,CONCNNModel}.py :: create_tensor_graph
    /root/reference/gan/shadow_data_models.py :: shadowdata_{generator,discriminator,feature_discriminator}_model
run UNCHANGED.  For every shipped modelconfigs/*.json at the shapes BASELINE.json names (and the GAN stacks at 64 / 144 /
360 bands, incl. the ragged last slice at 64 bands and 6 patches) this script writes to tests/golden/reference_graphs.json

  records    every layer call the reference made, in order: scope, op, kernel, num_outputs, normaliser (+ decay, is_training),
             activation (+ alpha), keep_prob, regulariser scale, initialiser, biases, input ids, output shape -- and the
             variable table (name -> shape, creation order).  The big configurations are recorded in shape-only mode (no
             arithmetic); the small ones are also EVALUATED in float64 with oracle/ops.py and their outputs stored in
             reference_graphs.npz together with the inputs, so that tests hold oracle/models.py against values produced
             by the reference's wiring;
  tower      the product Tower (tests/golden/graph_dump.py) the same reference file records through the facade engine
             (classifier models), which tests compare node for node with the Tower of the product's own plugin.

What this pins: layer order, scopes, widths, kernel lists, the batch norm on the logits, both dropout keep-prob
conventions, fc_stage_count, residual channel maps, the ragged last feature-discriminator slice.  What it does NOT pin:
operator semantics (the stand-in ops are oracle/ops.py: SURVEY Appendix A stays unpinned at the TensorFlow boundary).
Only data is written; no reference text leaves the container."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import tf_standin as S  # noqa: E402
from graph_dump import dump_output, dump_store, dump_tower  # noqa: E402

CFG = os.path.join(S.REF, "nnmodel", "modelconfigs")

# (case name, model, config file, patch, channels, classes) at the shapes the benchmarks / BASELINE configs use
CLASSIFIER_CASES = [
    ("hypelcnn_grss2013", "HYPELCNNModel", "alg_param_hypelcnn.json", 7, 145, 15),
    ("hypelcnn_avon", "HYPELCNNModel", "alg_param_hypelcnn.json", 7, 360, 2),
    ("hypelcnn_low", "HYPELCNNModel", "alg_param_hypelcnn_low.json", 7, 145, 15),
    ("hypelcnn_very_low", "HYPELCNNModel", "alg_param_hypelcnn_very_low.json", 7, 145, 15),
    ("hypelcnn_med", "HYPELCNNModel", "alg_param_hypelcnn_med.json", 7, 145, 15),
    ("hypelcnn_very_high", "HYPELCNNModel", "alg_param_hypelcnn_very_high.json", 7, 145, 15),
    ("hypelcnn_filt4800", "HYPELCNNModel", "alg_param_hypelcnn_filt4800.json", 7, 145, 15),
    ("hypelcnn_nonres", "HYPELCNNModel", "alg_param_hypelcnn_nonres.json", 7, 145, 15),
    ("dualcnn_grss2018", "DUALCNNModel", "alg_param_dualcnn.json", 11, 49, 20),
    ("concnn_grss2013", "CONCNNModel", "alg_param_concnn.json", 5, 145, 15),
]
# small cases that are also evaluated in float64: (name, model, config overrides, patch, channels, classes, batch)
VALUE_CASES = [
    ("hypelcnn_small", "HYPELCNNModel", "alg_param_hypelcnn.json", {"filter_count": 48}, 5, 21, 5, 6),
    ("hypelcnn_small_nonres", "HYPELCNNModel", "alg_param_hypelcnn_nonres.json", {"filter_count": 96}, 3, 10, 4, 5),
    ("dualcnn_small", "DUALCNNModel", "alg_param_dualcnn.json", {"filter_count": 32}, 5, 9, 4, 4),
    ("concnn_small", "CONCNNModel", "alg_param_concnn.json", {"filter_count": 8}, 5, 12, 6, 4),
]


def _model(name):
    import importlib
    return getattr(importlib.import_module("nnmodel." + name), name)()


def _mip(x, is_training):
    from common.common_nn_ops import ModelInputParams
    return ModelInputParams(x=x, y=None, device_id="/cpu:0", is_training=is_training)


class ShapeOnly(S.OracleEngine):
    """Records without arithmetic: every layer output is a zero array of the right shape (batch 1)."""

    def variable(self, name, shape, init):
        shape = tuple(int(s) for s in shape)
        self.variables.setdefault(name, list(shape))
        return None

    def _zero(self, shape):
        return self.O.Var(np.zeros(shape, np.float32))

    def conv2d(self, x, num_outputs, kernel_size, scope, opts):
        kh, kw = [S._int(k) for k in kernel_size]
        cin, cout = x.var.v.shape[3], S._int(num_outputs)
        scope = scope or self.unique("Conv")
        self.variable(scope + "/weights", (kh, kw, cin, cout), None)
        act = self._vars_and_act(scope, opts, cout)
        return self.new(self._zero(x.var.v.shape[:3] + (cout,)), "conv2d", [x],
                        **self._layer_attrs(scope, opts, act, kernel=[kh, kw], num_outputs=cout,
                                            biases=opts.get("normalizer_fn") is None))

    def fully_connected(self, x, num_outputs, scope, opts):
        cin, cout = x.var.v.shape[1], S._int(num_outputs)
        scope = scope or self.unique("fully_connected")
        self.variable(scope + "/weights", (cin, cout), None)
        act = self._vars_and_act(scope, opts, cout)
        return self.new(self._zero((x.var.v.shape[0], cout)), "fully_connected", [x],
                        **self._layer_attrs(scope, opts, act, num_outputs=cout, biases=opts.get("normalizer_fn") is None))

    def convolution1d(self, x, num_outputs, kernel_size, scope, padding, opts):
        k, cin, cout = S._int(kernel_size), x.var.v.shape[2], S._int(num_outputs)
        scope = scope or self.unique("Conv")
        self.variable(scope + "/weights", (k, cin, cout), None)
        act = self._vars_and_act(scope, opts, cout)
        return self.new(self._zero(x.var.v.shape[:2] + (cout,)), "convolution1d", [x],
                        **self._layer_attrs(scope, opts, act, kernel=[k], num_outputs=cout, padding=padding,
                                            biases=opts.get("normalizer_fn") is None))

    def _vars_and_act(self, scope, opts, cout):
        if opts.get("normalizer_fn") is not None:
            for nm in ("beta", "moving_mean", "moving_variance"):
                self.variable(f"{scope}/BatchNorm/{nm}", (cout,), None)
        else:
            self.variable(scope + "/biases", (cout,), None)
        return S.describe_activation(opts.get("activation_fn", S._relu))

    def lrn(self, x, **kw):
        return self.new(x.var, "local_response_normalization", [x], **kw)

    def l2_normalize(self, x):
        return self.new(x.var, "l2_normalize", [x])


def record_classifier(model_name, alg, patch, channels, classes, is_training, engine_cls=ShapeOnly, x_value=None, **ekw):
    eng = engine_cls(is_training=is_training, **ekw)
    with S.use_engine(eng):
        x = eng.placeholder(np.zeros((1, patch, patch, channels)) if x_value is None else x_value, "x")
        out = _model(model_name).create_tensor_graph(_mip(x, is_training), classes, alg)
    return eng, out


def tower_via_facade(model_name, alg, patch, channels, classes, is_training):
    """The product Tower the REFERENCE's plugin file records through the tf_slim facade."""
    from hypelcnn_amd import graph as G
    store = G.VariableStore("nn_core")
    tower = G.Tower(store, is_training)
    eng = S.GraphEngine(tower)
    with S.use_engine(eng):
        x = eng.wrap(tower.placeholder("x", (patch, patch), channels))
        out = _model(model_name).create_tensor_graph(_mip(x, is_training), classes, alg)
    return {"tower": dump_tower(tower), "variables": dump_store(store), "y_conv": dump_output(out.y_conv.sym),
            "image_output": None if out.image_output is None else dump_output(out.image_output.sym)}


def tower_via_product(model_name, alg, patch, channels, classes, is_training):
    from hypelcnn_amd import graph as G
    from hypelcnn_amd.common import common_nn_ops as P
    store = G.VariableStore("nn_core")
    tower = G.Tower(store, is_training)
    x = tower.placeholder("x", (patch, patch), channels)
    out = P.get_model_from_name(model_name).create_tensor_graph(
        P.ModelInputParams(x=x, y=None, device_id="/gpu:0", is_training=is_training), classes, alg)
    return {"tower": dump_tower(tower), "variables": dump_store(store), "y_conv": dump_output(out.y_conv),
            "image_output": None if out.image_output is None else dump_output(out.image_output)}


def main():
    S.install()
    out, arrays = {"classifiers": {}, "values": {}, "gan": {}}, {}
    for name, model, cfg, patch, ch, classes in CLASSIFIER_CASES:
        alg = json.load(open(os.path.join(CFG, cfg)))
        case = {"model": model, "config": cfg, "alg": alg, "patch": patch, "channels": ch, "classes": classes}
        for mode, training in (("train", True), ("eval", False)):
            eng, _ = record_classifier(model, alg, patch, ch, classes, training)
            case[mode] = {"records": eng.records, "variables": eng.variables}
            ref_tower = tower_via_facade(model, alg, patch, ch, classes, training)
            prod_tower = tower_via_product(model, alg, patch, ch, classes, training)
            assert ref_tower == prod_tower, f"{name}/{mode}: the product plugin records a different graph than the reference"
            case[mode]["tower"] = ref_tower
        out["classifiers"][name] = case
        print(f"{name}: {len(case['train']['records'])} calls, {len(case['train']['variables'])} variables, "
              f"{len(case['train']['tower']['tower']['nodes'])} tower nodes")

    # ---- small cases, evaluated in float64 --------------------------------------------------------------------
    from oracle import models as OM
    for name, model, cfg, over, patch, ch, classes, nb in VALUE_CASES:
        alg = dict(json.load(open(os.path.join(CFG, cfg))), **over)
        rng = np.random.default_rng(abs(hash(name)) % (2 ** 31))
        if model == "HYPELCNNModel":
            params = OM.hypelcnn_init_params(patch, ch, classes, alg, rng, np.float64)
            for k in params:
                if k.endswith("beta") or k.endswith("moving_mean"):
                    params[k] = rng.standard_normal(params[k].shape) * 0.1
                if k.endswith("moving_variance"):
                    params[k] = rng.random(params[k].shape) + 0.5
        else:
            table = (OM.dualcnn_layer_table if model == "DUALCNNModel" else OM.concnn_layer_table)(patch, ch, classes, alg)
            params = OM.xavier_init_params(table, rng, np.float64)
            for k in params:
                if k.endswith("biases"):
                    params[k] = rng.standard_normal(params[k].shape) * 0.1
        x = rng.random((nb, patch, patch, ch))
        case = {"model": model, "config": cfg, "alg": alg, "patch": patch, "channels": ch, "classes": classes, "batch": nb}
        for mode, training in (("train", True), ("eval", False)):
            probe, _ = record_classifier(model, alg, patch, ch, classes, training)  # dropout shapes for the masks
            masks, di = {}, 0
            for r in probe.records:
                if r["op"] == "dropout" and r["is_training"]:
                    keep = r["keep_prob"]
                    m = (rng.random((nb,) + tuple(r["shape"])) < keep) / keep
                    masks[f"dropout_{di}"] = m
                    di += 1
            eng, res = record_classifier(model, alg, patch, ch, classes, training, engine_cls=S.OracleEngine, x_value=x,
                                         params=params, dropout_masks=masks)
            # the training tower creates every variable oracle/models.py's table names; the inference tower a subset
            assert set(eng.variables) <= set(params) and (not training or set(eng.variables) == set(params)), \
                sorted(set(eng.variables) ^ set(params))
            arrays[f"{name}/{mode}/y_conv"] = res.y_conv.var.v
            if res.image_output is not None:
                arrays[f"{name}/{mode}/image_output"] = res.image_output.var.v
            for k, m in masks.items():
                arrays[f"{name}/{mode}/{k}"] = m
            case[mode] = {"records": eng.records, "variables": eng.variables}
        arrays[f"{name}/x"] = x
        for k, v in params.items():
            arrays[f"{name}/param/{k}"] = v
        out["values"][name] = case
        print(f"{name}: evaluated, logits {arrays[name + '/train/y_conv'].shape}")

    # ---- GAN stacks -------------------------------------------------------------------------------------------------
    import argparse
    from common import cmd_parser
    ap = argparse.ArgumentParser()
    for fn in ("add_parse_cmds_for_json_loader", "add_parse_cmds_for_trainers", "add_parse_cmds_for_loaders",
               "add_parse_cmds_for_models", "add_parse_cmds_for_importers", "add_parse_cmds_for_opt"):
        getattr(cmd_parser, fn)(ap)
    try:
        from gan import gan_train_for_shadow  # noqa: F401  (its flags, if the module imports under the stand-in)
    except Exception:
        pass
    import gan.shadow_data_models as ref_gan
    patches, embed = 6, 2
    for bands in (24, 64, 144, 360):
        for nm, call in (
                ("generator", lambda x: ref_gan.shadowdata_generator_model(x, create_only_encoder=False, is_training=True)),
                ("encoder", lambda x: ref_gan.shadowdata_generator_model(x, create_only_encoder=True, is_training=True)),
                ("discriminator", lambda x: ref_gan.shadowdata_discriminator_model(x, x, is_training=True, scale=1e-4)),
                ("feature_discriminator", lambda x: ref_gan.shadowdata_feature_discriminator_model(
                    x, patch_count=patches, embedded_feature_size=embed, is_training=True, scale=1e-3))):
            evaluate = bands <= 64
            rng = np.random.default_rng(bands * 7 + len(nm))
            nb = 5
            xv = rng.random((nb, 1, 1, bands)) if evaluate else np.zeros((1, 1, 1, bands))
            if evaluate:
                # pass 1 (shape-only) names the variables, then random fp64 values for them
                probe = ShapeOnly()
                with S.use_engine(probe):
                    call(probe.placeholder(np.zeros((1, 1, 1, bands)), "x"))
                params = {k: rng.standard_normal(shp) * (0.3 / np.sqrt(max(1, np.prod(shp[:-1]))))
                          for k, shp in probe.variables.items()}
                eng = S.OracleEngine(params=params)
            else:
                eng = ShapeOnly()
            with S.use_engine(eng):
                res = call(eng.placeholder(xv, "x"))
            key = f"{nm}_{bands}"
            out["gan"][key] = {"bands": bands, "patches": patches, "embed": embed, "records": eng.records,
                               "variables": eng.variables, "out_shape": [int(s) for s in res.var.v.shape[1:]]}
            if evaluate:
                arrays[f"gan/{key}/x"] = xv
                arrays[f"gan/{key}/out"] = res.var.v
                for k, v in params.items():
                    arrays[f"gan/{key}/param/{k}"] = v
        print(f"gan stacks at {bands} bands recorded")

    with open(os.path.join(HERE, "reference_graphs.json"), "w") as f:
        json.dump(out, f, sort_keys=True, separators=(",", ":"))
    np.savez_compressed(os.path.join(HERE, "reference_graphs.npz"), **arrays)
    print("wrote reference_graphs.json", os.path.getsize(os.path.join(HERE, "reference_graphs.json")) // 1024, "KB;",
          "reference_graphs.npz", os.path.getsize(os.path.join(HERE, "reference_graphs.npz")) // 1024, "KB")


if __name__ == "__main__":
    main()
