"""Host-side logic of the product (graph recording, fusion, planning, GEMM tables, hand-derived backward,
optimiser) executed on the numpy emulation of the kernel ABI and compared with the oracle.  CPU only."""
import numpy as np
import pytest

from oracle import train as OT
from tests import parity_util as U
from tests.emu_backend import EmuBackend

ALG_H = {"drop_out_ratio": 0.7, "filter_count": 48, "learning_rate": 3e-4, "learning_rate_decay_factor": 0.96,
         "learning_rate_decay_step": 350, "lrelu_alpha": 0.18, "optimizer": "AdamOptimizer", "bn_decay": 0.95,
         "l2regularizer_scale": 1e-5, "spectral_hierarchy_level": 3, "spatial_hierarchy_level": 3,
         "degradation_coeff": 3, "use_residual": True, "batch_size": 6}
ALG_D = {"drop_out_ratio": 0.7, "lrelu_alpha": 0.18, "filter_count": 32, "hs_lidar_diff": 1,
         "optimizer": "AdamOptimizer", "learning_rate": 3e-4, "learning_rate_decay_factor": 0.96,
         "learning_rate_decay_step": 350}
ALG_C = {"drop_out_ratio": 0.5, "filter_count": 6, "optimizer": ["MomentumOptimizer", 0.9], "learning_rate": 1e-3,
         "learning_rate_decay_factor": 0.01, "learning_rate_decay_step": 33333}


def _case(model_name, patch, ch, classes, alg, nb, seed):
    rng = np.random.default_rng(seed)
    built = U.build(model_name, patch, ch, classes, alg, EmuBackend())
    sess = built.ctx.session()
    params = U.make_params(model_name, patch, ch, classes, alg, rng)
    U.inject(sess, params)
    x = rng.random((nb, patch, patch, ch)).astype(np.float32)
    onehot = np.eye(classes, dtype=np.float32)[rng.integers(0, classes, nb)]
    masks = U.make_masks(built, nb, rng)
    return built, sess, params, x, onehot, masks


@pytest.mark.parametrize("model_name,patch,ch,classes,alg,nb", [
    ("HYPELCNNModel", 5, 11, 4, ALG_H, 6),
    ("HYPELCNNModel", 7, 9, 4, dict(ALG_H, filter_count=96), 5),  # fc_0 has K=588: exercises the split-K path
    ("HYPELCNNModel", 3, 7, 3, dict(ALG_H, use_residual=False, spectral_hierarchy_level=2), 5),
    ("DUALCNNModel", 5, 7, 3, ALG_D, 4),
    ("CONCNNModel", 5, 9, 3, ALG_C, 4),
])
def test_train_step_matches_oracle(model_name, patch, ch, classes, alg, nb):
    built, sess, params, x, onehot, masks = _case(model_name, patch, ch, classes, alg, nb, 11)
    ct = U.run_train_step(built, x, onehot, masks)
    ref, err, worst = U.compare_step(built, ct, params, x, onehot, masks, model_name, classes, alg,
                                     tol_logit=2e-5, tol_grad=2e-4)
    # inference tower shares the (now updated) moving statistics
    p2 = {k: sess.get_variable("nn_core/" + k).astype(np.float64) for k in params}
    li = U.run_eval(built, x)
    ri = OT.forward_backward(model_name, p2, x.astype(np.float64), None, classes, alg, False)
    assert np.abs(li - ri["logits"]).max() < 2e-6 * max(1.0, np.abs(ri["logits"]).max())


def test_three_adam_steps_track_oracle_trainer():
    model_name, patch, ch, classes, nb = "HYPELCNNModel", 5, 11, 4, 6
    built, sess, params, x, onehot, masks = _case(model_name, patch, ch, classes, ALG_H, nb, 5)
    trainer = OT.ClassifierTrainer(model_name, {k: v.copy() for k, v in params.items()}, classes, ALG_H)
    for step in range(3):
        ct = U.run_train_step(built, x, onehot, masks)
        sess.adam_step(built.lr.eval(sess.global_step))
        trainer.train_step(x.astype(np.float64), onehot.astype(np.float64), masks)
    for k, v in trainer.params.items():
        got = sess.get_variable("nn_core/" + k)
        assert np.abs(got - v).max() < 5e-5 * max(1.0, np.abs(v).max()), k
    assert sess.global_step == 3


def test_exact_tap_flop_count_matches_survey():
    """Planned GEMM work for the full GRSS2013 HYPELCNN = SURVEY Appendix B.1's exact-tap MAC count."""
    import json, os
    alg = json.load(open(os.path.join(os.path.dirname(U.cno.__file__), "..", "nnmodel", "modelconfigs",
                                      "alg_param_hypelcnn.json")))
    built = U.build("HYPELCNNModel", 7, 145, 15, alg, EmuBackend(), with_eval=False)
    ct = built.train_step.compiled(2)
    fwd, bwd = ct.flops()
    assert fwd == 2 * 2 * 78579058
    n_params = built.ctx.session().params.numel()
    assert n_params == 8160297


def test_tap_split_forward_matches_oracle(monkeypatch):
    """Heavy branches (k >= 5) of a level have their tap list split into chunks that write partial outputs, summed
    by the strided reduce; force the path at a tiny batch."""
    from hypelcnn_amd import plan
    monkeypatch.setattr(plan, "TAP_SPLIT_MIN_BATCH", 1)
    monkeypatch.setattr(plan, "MAX_TAPS_PER_TILE", 4)
    built, sess, params, x, onehot, masks = _case("HYPELCNNModel", 7, 9, 4, ALG_H, 5, 21)
    ct = U.run_train_step(built, x, onehot, masks)
    assert any(l.tag == "tap-split-reduce" for l in ct.plan.fwd)
    U.compare_step(built, ct, params, x, onehot, masks, "HYPELCNNModel", 4, ALG_H, tol_logit=2e-5, tol_grad=2e-4)


# ---------------------------------------------------------------------------------------------------------------
# Large-batch planner branches (plan.py: biased tap / channel-part split, data-gradient segment split, 64-row
# filter-gradient ranges).  They are gated on nb >= 64 / a device-filling launch in production; here each one is
# forced at a size the float64 oracle finishes in seconds, the tagged launch is asserted to be present, and the
# whole training step is compared with the oracle.
def _tags(ct):
    return [l.tag for l in ct.plan.fwd + ct.plan.bwd]


def test_biased_tap_and_channel_part_split_matches_oracle(monkeypatch):
    """DUALCNN (biases, no BN): branches with > MAX_TAPS taps go into a bias-free launch whose partial copies are
    summed (+ bias) by the reduce; the reduction dimension is additionally cut into channel parts."""
    from hypelcnn_amd import plan
    monkeypatch.setattr(plan, "TAP_SPLIT_MIN_BATCH", 1)
    monkeypatch.setattr(plan, "MAX_TAPS_PER_TILE", 4)
    monkeypatch.setattr(plan, "L2_CHUNK_BYTES", 4096)  # every multi-tap level input "exceeds the L2": channel parts
    built, sess, params, x, onehot, masks = _case("DUALCNNModel", 5, 7, 3, ALG_D, 4, 31)
    ct = U.run_train_step(built, x, onehot, masks)
    tags = _tags(ct)
    assert any(t.endswith("/split") and t.startswith("fwd:") for t in tags), "biased split launch missing"
    assert "tap-split-reduce" in tags
    # channel parts: at least one forward launch carries groups with a non-zero sub-key (second channel part)
    assert any(getattr(l, "kparts", 1) > 1 for l in ct.plan.fwd), "no level was cut into channel parts"
    U.compare_step(built, ct, params, x, onehot, masks, "DUALCNNModel", 3, ALG_D, tol_logit=2e-5, tol_grad=2e-4)


def test_channel_part_split_with_batch_norm_matches_oracle(monkeypatch):
    """The same channel-part cut for the un-biased (batch-normed) levels of HYPELCNN."""
    from hypelcnn_amd import plan
    monkeypatch.setattr(plan, "TAP_SPLIT_MIN_BATCH", 1)
    monkeypatch.setattr(plan, "MAX_TAPS_PER_TILE", 4)
    monkeypatch.setattr(plan, "L2_CHUNK_BYTES", 4096)
    alg = dict(ALG_H, filter_count=96)
    built, sess, params, x, onehot, masks = _case("HYPELCNNModel", 5, 9, 4, alg, 5, 23)
    ct = U.run_train_step(built, x, onehot, masks)
    assert "tap-split-reduce" in _tags(ct)
    assert any(getattr(l, "kparts", 1) > 1 for l in ct.plan.fwd)
    U.compare_step(built, ct, params, x, onehot, masks, "HYPELCNNModel", 4, alg, tol_logit=2e-5, tol_grad=2e-4)


def test_split6_plan_matches_oracle(monkeypatch):
    """HYPEL_GEMM_SPLIT=6 (include/hypel.h HYPEL_GEMM_SPLIT6): with the size threshold forced down the planner marks the
    eligible forward, data-gradient and merged filter-gradient launches of a toy HYPELCNN; the emulation checks the
    eligibility rules of the flag, and the whole step still equals the oracle."""
    from hypelcnn_amd import plan
    monkeypatch.setattr(plan, "GEMM_SPLIT", 6)
    monkeypatch.setattr(plan, "GEMM_SPLIT_MIN_FLOPS", 0.0)
    alg = dict(ALG_H, filter_count=96)
    built, sess, params, x, onehot, masks = _case("HYPELCNNModel", 5, 40, 4, alg, 70, 29)
    ct = U.run_train_step(built, x, onehot, masks)
    launches = ct.plan.fwd + ct.plan.bwd
    split = [l for l in launches if l.name.startswith("seg_gemm") and l.name != "seg_gemm_multi_f32" and l.args[14] & 0x8000]
    assert any(l.tag.startswith("fwd:") for l in split) and any(l.tag.startswith("dgrad:") for l in split), \
        [l.tag for l in split]
    assert all((l.args[14] >> 8) & 3 for l in split)
    multi = [l for l in launches if l.name == "seg_gemm_multi_f32"]
    assert any(l.args[3] & 0x100 for l in multi), [l.tag for l in multi]
    # launches the flag does not apply to stay on the fp32 kernel: n <= 16, paired short segments
    assert any(l.name.startswith("seg_gemm") and l.name != "seg_gemm_multi_f32" and not (l.args[14] & 0x8000)
               for l in launches)
    U.compare_step(built, ct, params, x, onehot, masks, "HYPELCNNModel", 4, alg, tol_logit=2e-5, tol_grad=2e-4)


@pytest.mark.parametrize("merged", [False, True])
def test_dgrad_segment_split_matches_oracle(monkeypatch, merged):
    """Data gradient of an unfolded multi-kernel level: the per-pixel segment list is cut into chunks that write
    partial copies of dX, summed by one reduce (DUALCNN levels; HYPELCNN levels fold their shortcut instead).  merged
    (round 6): a BIASED level's data and filter gradients may take the merged per-offset form too (the bias only concerns
    the forward pass, which stays unmerged): merged segments over the packed weight image, cut into the same chunks."""
    from hypelcnn_amd import plan
    monkeypatch.setattr(plan, "TAP_SPLIT_MIN_BATCH", 1)
    monkeypatch.setattr(plan, "DGRAD_MAX_SEGS", 5)
    if not merged:
        monkeypatch.setattr(plan, "MERGE_LEVELS", set())
    built, sess, params, x, onehot, masks = _case("DUALCNNModel", 5, 7, 3, ALG_D, 4, 37)
    ct = U.run_train_step(built, x, onehot, masks)
    tags = _tags(ct)
    want = "/split/merged" if merged else "/split"
    assert "dgrad-split-reduce" in tags and any(t.startswith("dgrad:") and t.endswith(want) for t in tags), tags
    assert ("level-pack" in tags) == merged and not any(t.startswith("fwd:") and t.endswith("/merged") for t in tags)
    if merged:
        prods = [p_ for l in ct.plan.bwd if l.name == "seg_gemm_multi_f32" for p_ in l.meta["products"]]
        assert any(p_.endswith("/merged") for p_ in prods) and "level-unpack" in tags
    U.compare_step(built, ct, params, x, onehot, masks, "DUALCNNModel", 3, ALG_D, tol_logit=2e-5, tol_grad=2e-4)


@pytest.mark.parametrize("model_name,patch,ch,classes,alg", [
    ("DUALCNNModel", 5, 7, 3, ALG_D),
    ("HYPELCNNModel", 5, 11, 4, ALG_H),
])
def test_wgrad_row_ranges_of_device_filling_launch_match_oracle(monkeypatch, model_name, patch, ch, classes, alg):
    """Filter gradients of a launch that already fills the device are still cut into 64-row batch ranges
    (split-major tile order); forced by declaring one block 'device filling' at batch 192 (3 ranges, the last
    boundary not a multiple of the 128-row tile)."""
    from hypelcnn_amd import plan
    monkeypatch.setattr(plan, "TARGET_BLOCKS", 1)
    built, sess, params, x, onehot, masks = _case(model_name, patch, ch, classes, alg, 192, 41)
    ct = U.run_train_step(built, x, onehot, masks)
    assert "wgrad-reduce" in _tags(ct)
    n_split = [S for l in ct.plan.bwd if l.tag == "wgrad-reduce" for S in (l.meta.get("splits") or [l.args[2]])]
    assert max(n_split) >= 3, n_split
    U.compare_step(built, ct, params, x, onehot, masks, model_name, classes, alg, tol_logit=2e-5, tol_grad=2e-4)


def test_all_large_batch_branches_together_match_oracle(monkeypatch):
    """Every split policy at once, at the batch size where production enables them (nb = 64) on a small DUALCNN."""
    from hypelcnn_amd import plan
    monkeypatch.setattr(plan, "MAX_TAPS_PER_TILE", 4)
    monkeypatch.setattr(plan, "DGRAD_MAX_SEGS", 5)
    monkeypatch.setattr(plan, "L2_CHUNK_BYTES", 1 << 16)
    monkeypatch.setattr(plan, "TARGET_BLOCKS", 4)
    built, sess, params, x, onehot, masks = _case("DUALCNNModel", 5, 7, 3, ALG_D, 64, 43)
    ct = U.run_train_step(built, x, onehot, masks)
    tags = _tags(ct)
    for want in ("tap-split-reduce", "dgrad-split-reduce", "wgrad-reduce"):
        assert want in tags, want
    U.compare_step(built, ct, params, x, onehot, masks, "DUALCNNModel", 3, ALG_D, tol_logit=2e-5, tol_grad=2e-4)


def test_nonfinite_loss_is_flagged_on_the_device_and_the_update_refused():
    """NanTensorHook + check_numerics semantics (monitored_session_runner.py:151, common_nn_ops.py:232): a step
    whose loss is NaN sets the flag behind the gradient buffer, the guarded optimiser leaves parameters and slots
    untouched, and the session reports the step -- on the next poll without `sync`, immediately with it."""
    built, sess, params, x, onehot, masks = _case("HYPELCNNModel", 5, 11, 4, ALG_H, 6, 3)
    U.run_train_step(built, x, onehot, masks)
    sess.adam_step(1e-3)
    assert float(sess.grads[sess.n_train]) == 0.0 and sess.nonfinite_step(sync=True) is None
    p1, m1 = sess.params.clone(), sess.slot_m.clone()
    bad = x.copy()
    bad[0, 0, 0, 0] = np.nan
    U.run_train_step(built, bad, onehot, masks)
    assert float(sess.grads[sess.n_train]) == 1.0
    sess.adam_step(1e-3)
    import torch
    assert torch.equal(sess.params, p1) and torch.equal(sess.slot_m, m1), "a non-finite step must not update anything"
    assert sess.nonfinite_step() is None, "the newest flag copy is not inspected without sync (pipelined loop)"
    assert sess.nonfinite_step(sync=True) == 2
    assert sess.nonfinite_step() == 2


def test_filter_gradients_go_out_as_merged_launches(monkeypatch):
    """All filter gradients of a step leave as ONE multi-product launch per tile width plus ONE merged reduce; the
    un-merged plan (HYPEL_MERGE_WGRAD=0) gives the same gradients."""
    from hypelcnn_amd import plan
    built, sess, params, x, onehot, masks = _case("HYPELCNNModel", 7, 9, 4, dict(ALG_H, filter_count=96), 5, 21)
    ct = U.run_train_step(built, x, onehot, masks)
    names = [l.name for l in ct.plan.bwd]
    n_multi = names.count("seg_gemm_multi_f32")
    assert 1 <= n_multi <= 4 and names.count("reduce_splits_multi_f32") == 1  # (split 128 / 64, fp32 32 / 16)
    assert not any(l.tag.startswith("wgrad:") for l in ct.plan.bwd), "a filter gradient was launched on its own"
    merged = [l for l in ct.plan.bwd if l.name == "seg_gemm_multi_f32"]
    n_layers = sum(1 for n in built.train_tower.nodes if hasattr(n, "branches"))
    assert sum(len(l.meta["products"]) for l in merged) >= n_layers
    for l in merged:  # XCD work lists are balanced (heaviest-first dealing)
        w = l.meta["xcd_work"]
        assert max(w) <= 1.5 * (sum(w) / 8) + max(w) / max(1, l.meta["blocks"] // 8)
    g_merged = sess.grads.clone()
    U.compare_step(built, ct, params, x, onehot, masks, "HYPELCNNModel", 4, dict(ALG_H, filter_count=96),
                   tol_logit=2e-5, tol_grad=2e-4)
    monkeypatch.setattr(plan, "MERGE_WGRAD", False)
    built2, sess2, _, _, _, _ = _case("HYPELCNNModel", 7, 9, 4, dict(ALG_H, filter_count=96), 5, 21)
    ct2 = U.run_train_step(built2, x, onehot, masks)
    assert any(l.tag.startswith("wgrad:") for l in ct2.plan.bwd)
    import torch
    torch.testing.assert_close(sess2.grads, g_merged, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("passes,patch,fc", [("fwd", 7, 48), ("fwd,dgrad", 7, 48), ("fwd,dgrad", 5, 96), ("dgrad", 5, 48),
                                             ("wgrad", 7, 96), ("fwd,dgrad,wgrad", 5, 480), ("fwd,dgrad,wgrad", 7, 48)])
def test_merged_levels_match_oracle(monkeypatch, passes, patch, fc):
    """Merged multi-kernel levels (include/hypel.h, HYPEL_GEMM_VAR_N): the nested branches of a level share one packed
    weight image; per output pixel and ring ONE product on the column range of the branches that contain the ring
    (forward), merged reduction segments (data gradient).  Forced at a tiny batch, every pass alone and together, incl.
    channel parts, against the float64 oracle."""
    from hypelcnn_amd import plan
    monkeypatch.setattr(plan, "TAP_SPLIT_MIN_BATCH", 1)
    monkeypatch.setattr(plan, "MAX_TAPS_PER_TILE", 5)
    monkeypatch.setattr(plan, "MERGE_LEVELS", set(passes.split(",")))
    monkeypatch.setattr(plan, "MERGE_LEVELS_MAX_COUT", 1 << 20)
    monkeypatch.setattr(plan, "MERGE_WGRAD_MIN_COUT", 1)
    monkeypatch.setattr(plan, "MERGE_WGRAD_MAX_COUT", 1 << 20)
    monkeypatch.setattr(plan, "MERGE_SPLIT_KPARTS", True)
    if patch == 5:
        monkeypatch.setattr(plan, "L2_CHUNK_BYTES", 4096)  # channel parts in the merged forward
    alg = dict(ALG_H, filter_count=fc)
    built, sess, params, x, onehot, masks = _case("HYPELCNNModel", patch, 9, 4, alg, 5, 29)
    ct = U.run_train_step(built, x, onehot, masks)
    tags = _tags(ct)
    names = [l.name for l in ct.plan.fwd + ct.plan.bwd]
    for what in passes.split(","):
        if what == "wgrad":  # the packed filter gradients travel in the merged multi-product launches + one scatter
            prods = [p_ for l in ct.plan.bwd if l.name == "seg_gemm_multi_f32" for p_ in l.meta["products"]]
            assert any(p_.endswith("/merged") for p_ in prods) and "level-unpack" in tags, (prods, tags)
            continue
        assert any(t.startswith(what + ":") and t.endswith("/merged") for t in tags), (what, tags)
    assert ("level-pack" in tags or passes == "wgrad") and "copy_blocks_f32" in names
    U.compare_step(built, ct, params, x, onehot, masks, "HYPELCNNModel", 4, alg, tol_logit=2e-5, tol_grad=2e-4)




@pytest.mark.parametrize("frac_min", [0.25, 0.75])
def test_kslice_records_match_oracle(monkeypatch, frac_min):
    """K-slice records (include/hypel.h HYPEL_TILE_PLAIN) forced onto the data gradients of a toy HYPELCNN: heavy tiles of the
    multi-kernel levels cut into slices and the tail tiles of the 1x1 stack's launches, slice 0 with the folded shortcut
    gather / accumulate bit, the others as plain partials in scratch, one reduce launch adding them in slice order; the
    whole step still equals the oracle."""
    from hypelcnn_amd import plan
    monkeypatch.setattr(plan, "GEMM_SPLIT", 6)
    monkeypatch.setattr(plan, "GEMM_SPLIT_MIN_FLOPS", 0.0)
    monkeypatch.setattr(plan, "KSLICE_MIN_GAIN", -10.0)   # take the best candidate whatever it is worth here
    monkeypatch.setattr(plan, "KSLICE_MIN_K", 4)
    monkeypatch.setattr(plan, "KSLICE_OVERHEAD_K", 1)
    monkeypatch.setattr(plan, "KSLICE_REDUCE_COST_K", 0)
    monkeypatch.setattr(plan, "KSLICE_FRAC_MIN", frac_min)
    monkeypatch.setattr(plan, "KSLICE_SLOTS", {1: 8, 2: 8, 3: 8})
    alg = dict(ALG_H, filter_count=400)
    built, sess, params, x, onehot, masks = _case("HYPELCNNModel", 5, 40, 4, alg, 150, 31)
    ct = U.run_train_step(built, x, onehot, masks)
    tags = _tags(ct)
    sliced = [l for l in ct.plan.bwd if l.meta.get("kslices") and l.name.startswith("seg_gemm")]
    assert "kslice-reduce" in tags and sliced, tags
    assert any(l.name == "seg_gemm_res_f32" for l in sliced), "no sliced launch with a folded shortcut gradient"
    from hypelcnn_amd.backend import TILE_DTYPE
    for l in sliced:  # every sliced launch carries plain records and padding to eight equal shares
        t = l.args[11].t.numpy()[l.args[11].off:].view(TILE_DTYPE)[:l.args[12]]
        assert (t["flags"] & 1).any() and len(t) % 8 == 0
    U.compare_step(built, ct, params, x, onehot, masks, "HYPELCNNModel", 4, alg, tol_logit=2e-5, tol_grad=2e-4)
