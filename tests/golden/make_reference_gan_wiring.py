#!/usr/bin/env python3
"""Pin the GAN WRAPPERS' wiring by executing the reference's own files (build container only; needs /root/reference).

Under `tfgan_standin.py` (a recording `tensorflow_gan` restated after its published source + the `tf.*` calls the wrappers
make) the reference's UNCHANGED

    gan/wrapper_registry.py :: get_wrapper_dict(flags)            (which wrapper class gets which network functions)
    gan/wrappers/{gan,cycle_gan,cut,dcl_gan,dcl_cycle_gan}_wrapper.py :: define_model -> define_loss -> define_train_ops
    gan/wrappers/gan_common.py :: _get_lr, define_standard_train_ops
    gan/shadow_data_models.py :: the three network builders

run for every `--gan_type` (cycle_gan with and without the identity loss, gan_x2y, gan_y2x, cut_x2y, cut_y2x, dcl_gan,
dcl_cycle_gan) on a seeded small case, in float64.  Written to tests/golden/reference_gan_wiring.json / .npz, per case:

  hooks      the ordered RunTrainOpsHooks `get_train_hooks_fn()(train_ops)` returns (= the sequential phases of one global
             step); per train op of a hook: the loss as a weighted list of primitive terms -- kind (mean_sq / mean / mean_abs /
             nce / l2), operand tensors BY PROVENANCE (`D[Model/ModelX2Y/Discriminator](pool(G[...](x)))`), target / tau /
             regulariser scale, weight, value --, the variables the optimiser updates, Adam's beta1 and the LR schedule
             sampled at eight steps, and (npz) the float64 loss and the gradient of every trained variable;
  variables  every variable the run created, in creation order, with its shape; which of them carry an L2 regulariser;
  networks   every application of a network function: kind, variable scope, input by provenance;
  npz        the inputs x / y and every parameter value (the stand-in's draw: the reference zero-initialises its generator).

`tests/test_reference_gan_wiring.py` holds `oracle/gan.py` (phase order, trained variables, loss values, gradients) and the
product's `GANLoss.phases` (terms, operands by provenance, train groups, LR schedule, tensor-pool use) to this fixture.  Scope
of the pin: the reference's wrapper text on a restated tensorflow_gan (SURVEY Appendix A.12 stays a restatement).  Only data
is written."""
import json
import os
import re
import sys
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import tfgan_standin as W  # noqa: E402

LR_STEPS = [0, 1, 9, 10, 11, 15, 19, 20, 25]
MAX_STEPS = 20

# (case, gan_type, bands, batch, flag overrides)
CASES = [
    ("cycle_gan", "cycle_gan", 16, 6, {}),
    ("cycle_gan_no_identity", "cycle_gan", 16, 6, {"use_identity_loss": False}),
    ("gan_x2y", "gan_x2y", 16, 6, {}),
    ("gan_y2x", "gan_y2x", 24, 6, {}),
    ("cut_x2y", "cut_x2y", 24, 6, {}),
    ("cut_y2x", "cut_y2x", 24, 5, {"patches": 4}),
    ("cut_x2y_no_identity", "cut_x2y", 24, 6, {"use_identity_loss": False}),
    ("cut_x2y_ragged_64", "cut_x2y", 64, 4, {}),          # 64 bands / 6 patches: ten slices of 6 + one of 4
    ("dcl_gan", "dcl_gan", 16, 6, {"patches": 4}),
    ("dcl_cycle_gan", "dcl_cycle_gan", 16, 6, {"patches": 4}),
]


def default_flags(batch, **over):
    """The defaults of gan_train_for_shadow.py:add_parse_cmds_for_app (:31-70)."""
    f = dict(use_identity_loss=True, identity_loss_weight=0.5, cycle_consistency_loss_weight=10.0, nce_loss_weight=10.0,
             tau=0.07, patches=6, embedded_feat_size=2, generator_lr=0.0002, discriminator_lr=0.0001,
             gen_discriminator_lr=0.0001, discriminator_reg_scale=0.00001, gen_disc_reg_scale=0.0001, batch_size=batch)
    f.update(over)
    return SimpleNamespace(**f)


def canonical_term(term):
    """("mean", label, reduction) / ("l2", variable, scale) -> a dict with the operands parsed out of the provenance label."""
    if term[0] == "l2":
        return {"kind": "l2", "variable": term[1], "scale": term[2]}
    label = term[1]
    m = re.fullmatch(r"sqdiff_half\((.*),([-0-9.e]+)\)", label)
    if m:
        return {"kind": "mean_sq", "a": m.group(1), "target": float(m.group(2)), "factor": 0.5, "reduction": term[2]}
    m = re.fullmatch(r"neg\((.*)\)", label)
    if m:
        return {"kind": "mean", "a": m.group(1), "factor": -1.0, "reduction": term[2]}
    m = re.fullmatch(r"xent_eye\(div\(matmul_nt\((.*)\),([-0-9.e]+)\)\)", label)
    if m:
        a, b = split_top(m.group(1))
        return {"kind": "nce", "a": a, "b": b, "tau": float(m.group(2)), "factor": 1.0, "reduction": term[2]}
    m = re.fullmatch(r"abs_diff\((.*)\)", label)
    if m:
        a, b = split_top(m.group(1))
        return {"kind": "mean_abs", "a": a, "b": b, "factor": 1.0, "reduction": term[2]}
    return {"kind": "mean", "a": label, "factor": 1.0, "reduction": term[2]}


def split_top(s):
    depth = 0
    for i, ch in enumerate(s):
        depth += ch in "(["
        depth -= ch in ")]"
        if ch == "," and depth == 0:
            return s[:i], s[i + 1:]
    raise ValueError(s)


def run_case(gan_type, bands, batch, over, seed):
    import gan.wrapper_registry as registry
    import gan.shadow_data_models as nets
    flags = default_flags(batch, **over)
    # the registry binds the reference's network functions; give them provenance labels without touching the registry's text
    orig = (nets.shadowdata_generator_model, nets.shadowdata_discriminator_model, nets.shadowdata_feature_discriminator_model)
    registry.shadowdata_generator_model = W.labelled("G", orig[0])
    registry.shadowdata_discriminator_model = W.labelled("D", orig[1])
    registry.shadowdata_feature_discriminator_model = W.labelled("F", orig[2])
    try:
        wrapper = registry.get_wrapper_dict(flags)[gan_type]
    finally:
        registry.shadowdata_generator_model, registry.shadowdata_discriminator_model, \
            registry.shadowdata_feature_discriminator_model = orig
    rng = np.random.default_rng(seed)
    x = rng.random((batch, 1, 1, bands))
    y = rng.random((batch, 1, 1, bands)) * 0.5
    eng = W.WiringEngine(rng=np.random.default_rng(seed + 1))
    with W.S.use_engine(eng):
        tx, ty = eng.placeholder(x, "x"), eng.placeholder(y, "y")
        model = wrapper.define_model(tx, ty)
        loss = wrapper.define_loss(model)
        train_ops = wrapper.define_train_ops(model, loss, max_number_of_steps=MAX_STEPS, generator_lr=flags.generator_lr,
                                             discriminator_lr=flags.discriminator_lr,
                                             gen_discriminator_lr=flags.gen_discriminator_lr)
        hooks = wrapper.get_train_hooks_fn()(train_ops)
    case = {"gan_type": gan_type, "bands": bands, "batch": batch, "flags": vars(flags), "max_steps": MAX_STEPS,
            "lr_steps": LR_STEPS, "variables": eng.variables, "regularised": [[n, s] for n, s in eng.reg_losses],
            "networks": [list(a) for a in eng.applications], "pooled": sorted(set(eng.pools)), "hooks": []}
    arrays = {"x": x, "y": y}
    for k, v in eng.params.items():
        arrays["param/" + k] = np.asarray(v, np.float64)
    for hi, hook in enumerate(hooks):
        assert isinstance(hook, W.RunTrainOpsHook), hook
        h = {"train_steps": int(hook.train_steps), "ops": []}
        for oi, op in enumerate(hook.train_ops):
            # gradients: a fresh backward pass per train op over the shared tape
            _reset_grads(op.loss.var)
            eng.O.backward(op.loss.var)
            lr = op.optimizer.learning_rate
            terms = []
            for w, t in op.loss.lin:
                d = canonical_term(t)
                factor = d.pop("factor", 1.0)
                d["weight"] = float(w) * factor                       # loss = sum over terms of weight * value
                d["value"] = float(_term_value(eng, t)) / factor      # (mean_sq: mean((a - target)^2), mean: mean(a), ...)
                terms.append(d)
            total = sum(t["weight"] * t["value"] for t in terms)
            assert abs(total - float(op.loss.var.v)) <= 1e-12 * max(1.0, abs(total)), (total, float(op.loss.var.v))
            h["ops"].append({"variables": op.variables, "beta1": op.optimizer.beta1, "beta2": op.optimizer.beta2,
                             "epsilon": op.optimizer.epsilon, "lr": [float(lr(s) if callable(lr) else lr) for s in LR_STEPS],
                             "terms": terms, "loss": float(op.loss.var.v)})
            arrays[f"hook{hi}/op{oi}/loss"] = np.asarray(float(op.loss.var.v))
            for vn in op.variables:
                g = eng.var_objs[vn].g
                arrays[f"hook{hi}/op{oi}/grad/{vn}"] = np.zeros_like(eng.var_objs[vn].v) if g is None else np.asarray(g, np.float64)
        case["hooks"].append(h)
    return case, arrays


def _term_value(eng, term):
    """Value of a primitive term: the record that produced it carries the value in the engine's tape -- recompute from records."""
    for rec, t in zip(eng.records, eng._tensors):
        if rec.get("term") == list(term):
            return float(t.var.v)
    raise KeyError(term)


def _reset_grads(root):
    seen, stack = set(), [root]
    while stack:
        n = stack.pop()
        if id(n) in seen:
            continue
        seen.add(id(n))
        n.g = None
        stack.extend(n.parents)


def main():
    W.install()
    # keep every tensor the engine creates (values of the primitive terms are read back from them)
    orig_new = W.WiringEngine.new

    def new(self, var, op, inputs=(), **attrs):
        t = orig_new(self, var, op, inputs, **attrs)
        self.__dict__.setdefault("_tensors", []).append(t)
        return t
    W.WiringEngine.new = new
    out, arrays = {}, {}
    for i, (name, gan_type, bands, batch, over) in enumerate(CASES):
        case, arr = run_case(gan_type, bands, batch, over, seed=100 + i)
        out[name] = case
        for k, v in arr.items():
            arrays[f"{name}/{k}"] = v
        print(f"{name}: {len(case['variables'])} variables, {len(case['networks'])} network applications, "
              f"{len(case['hooks'])} hooks x {[len(h['ops']) for h in case['hooks']]} ops, "
              f"terms per op {[len(o['terms']) for h in case['hooks'] for o in h['ops']]}")
    with open(os.path.join(HERE, "reference_gan_wiring.json"), "w") as f:
        json.dump(out, f, sort_keys=True, indent=0, separators=(",", ":"))
    np.savez_compressed(os.path.join(HERE, "reference_gan_wiring.npz"), **arrays)
    print("wrote reference_gan_wiring.json", os.path.getsize(os.path.join(HERE, "reference_gan_wiring.json")) // 1024, "KB;",
          "reference_gan_wiring.npz", os.path.getsize(os.path.join(HERE, "reference_gan_wiring.npz")) // 1024, "KB")


if __name__ == "__main__":
    main()
