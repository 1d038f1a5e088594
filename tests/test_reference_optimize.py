"""`optimize_nn` and the classifiers' `get_loss_func`, pinned by EXECUTING the reference's text.

`tests/golden/make_reference_optimize.py` ran `common/common_nn_ops.py:208-240` and the three plugin files' `get_loss_func`
unchanged under the float64 recording engine and committed: the optimiser's class / name / hyper-parameters, the LR at ten
steps, the loss value and every gradient of the train op's loss.  Held to it: `oracle/train.py::forward_backward` (loss and
gradients at 1e-10, the set of trained variables), `oracle/host.py::exponential_decay_staircase` and the product's
`LearningRate`, the product's optimiser settings, and the product's numbers on the kernel emulation."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import train as OT
from oracle.host import exponential_decay_staircase

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(HERE, "golden")
FIX = json.load(open(os.path.join(GOLD, "reference_optimize.json")))
ARR = np.load(os.path.join(GOLD, "reference_optimize.npz"))
CASES = sorted(FIX)
HAVE_REF = os.path.isdir("/root/reference")


def _arrays(case, prefix):
    pre = f"{case}/{prefix}"
    return {k[len(pre):]: ARR[k] for k in ARR.files if k.startswith(pre)}


@pytest.mark.parametrize("case", CASES)
def test_oracle_step_equals_the_executed_reference(case):
    c = FIX[case]
    params, masks = _arrays(case, "param/"), _arrays(case, "mask/")
    r = OT.forward_backward(c["model"], {k: v.copy() for k, v in params.items()}, ARR[f"{case}/x"], ARR[f"{case}/labels"],
                            c["classes"], c["alg"], True, masks)
    assert abs(r["loss"] - c["loss"]) <= 1e-11 * max(1.0, abs(c["loss"])), (r["loss"], c["loss"])
    np.testing.assert_allclose(r["logits"], ARR[f"{case}/logits"], rtol=1e-11, atol=1e-12)
    assert sorted(r["grads"]) == c["trained"] == sorted(OT.trainable_names(params))
    for k in c["trained"]:
        want = ARR[f"{case}/grad/{k}"]
        assert np.abs(r["grads"][k] - want).max() <= 1e-10 * max(1e-6, float(np.abs(want).max())), k
    # what the reference's loss text is made of: per-sample cross entropy (+ the scalar reconstruction MSE), then the mean
    tail = c["loss_ops"]
    assert tail[-1] == "reduce_mean" and "softmax_cross_entropy_with_logits" in tail
    assert (tail[-6:] == ["reshape", "sub", "square", "reduce_mean", "add", "reduce_mean"]) == (c["model"] == "HYPELCNNModel")
    assert c["create_train_op_kwargs"] == ["global_step"] or c["create_train_op_kwargs"] == []
    assert c["global_step_is_the_shared_one"]


@pytest.mark.parametrize("case", CASES)
def test_lr_schedule_and_optimiser(case):
    from hypelcnn_amd.common.common_nn_ops import LearningRate
    import inspect
    from hypelcnn_amd.runtime import Session
    c = FIX[case]
    alg = c["alg"]
    lr = LearningRate(alg["learning_rate"], alg["learning_rate_decay_step"], alg["learning_rate_decay_factor"])
    for s, want in zip(c["lr_steps"], c["lr"]):
        assert abs(exponential_decay_staircase(alg["learning_rate"], s, alg["learning_rate_decay_step"],
                                               alg["learning_rate_decay_factor"]) - want) <= 1e-15 * max(1.0, want)
        assert abs(lr.eval(s) - want) <= 1e-15 * max(1.0, want)
    opt = c["optimizer"]
    if isinstance(alg["optimizer"], (list, tuple)):
        assert opt["class"] == "MomentumOptimizer" and opt["name"] == "nn_core/Momentum"
        assert opt["hyper"] == {"momentum": alg["optimizer"][1], "use_nesterov": False}
    else:
        assert opt["class"] == "AdamOptimizer" and opt["name"] == "nn_core/Adam"
        d = inspect.signature(Session.adam_step).parameters
        assert opt["hyper"] == {"beta1": d["beta1"].default, "beta2": d["beta2"].default, "epsilon": d["eps"].default}


@pytest.mark.parametrize("case", CASES)
def test_product_step_on_the_emulation_equals_the_executed_reference(case):
    from tests import parity_util as U
    from tests.emu_backend import EmuBackend
    c = FIX[case]
    params = {k: v.astype(np.float32).astype(np.float64) for k, v in _arrays(case, "param/").items()}
    masks = _arrays(case, "mask/")
    x = ARR[f"{case}/x"].astype(np.float32)
    onehot = ARR[f"{case}/labels"].astype(np.float32)
    built = U.build(c["model"], c["patch"], c["channels"], c["classes"], c["alg"], EmuBackend())
    sess = built.ctx.session()
    U.inject(sess, params)
    ct = U.run_train_step(built, x, onehot, masks)
    # the float64 reference values at the fp32-rounded parameters come from the oracle, which the first test ties to the fixture
    U.compare_step(built, ct, params, x, onehot, masks, c["model"], c["classes"], c["alg"], tol_logit=2e-5, tol_grad=2e-4)
    assert sorted(k for k in c["trained"]) == sorted(v.name[len("nn_core/"):] for v in sess.trainable)


@pytest.mark.skipif(not HAVE_REF, reason="the reference is only present in the build container")
def test_optimize_fixture_is_what_the_reference_produces_today():
    code = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import make_reference_optimize as M
M.F._Finder.EXTRA_SETUP.append(M._setup)
M.S.install()
import importlib
for name in ("tensorflow.compat.v1.train", "tf_slim.learning"):
    m = importlib.import_module(name)
    parent, _, attr = name.rpartition(".")
    setattr(importlib.import_module(parent), attr, m)
fix = json.load(open(os.path.join(%r, "reference_optimize.json")))
arr = np.load(os.path.join(%r, "reference_optimize.npz"))
for i, (name, model, cfg, over, patch, ch, classes, nb) in enumerate(M.CASES):
    case, a = M.run_case(model, cfg, over, patch, ch, classes, nb, seed=300 + i)
    assert json.loads(json.dumps(case)) == fix[name], name
    for k, v in a.items():
        assert np.array_equal(arr[name + "/" + k], v), (name, k)
print("ok")
""" % (ROOT, GOLD, GOLD, GOLD)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.skipif(not HAVE_REF, reason="the reference checkout exists in the build container only")
def test_shadow_ratio_augmentation_of_the_gan_input_equals_the_executed_reference():
    """`gan/gan_train_for_shadow.py:171-182` (perform_shadow_augmentation_random: with probability reg_support_rate the
    normal spectrum becomes shadow x ratio, then -- from the POSSIBLY REPLACED normal -- the shadow spectrum becomes
    normal / ratio) executed with its two uniform draws pinned, against the product's pair kernel specification
    (`hypel_gather_pairs_f32`, tests/emu_backend.py) for all four branch combinations."""
    code = r"""
import sys
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import tf_standin as S
from hypelcnn_amd import tf_facade as F
draws = []
def _setup(m):
    if m.__name__ == "tensorflow":
        m.less = lambda a, b: a < b
        m.cond = lambda pred=None, true_fn=None, false_fn=None, **k: true_fn() if pred else false_fn()
        m.device = F._Ctx
    elif m.__name__ == "tensorflow.random":
        m.uniform = lambda shape, lo, hi, **k: [draws.pop(0)]
    elif m.__name__ == "tensorflow.compat.v1":
        m.name_scope = F._Ctx
from hypelcnn_amd import tfgan_facade as TG
TG.enable()                                  # (the module imports the wrapper registry, which needs the tensorflow_gan surface)
F._Finder.EXTRA_SETUP.append(_setup)         # after it: this test's eager tf.cond wins
S.install()
TG.preload()
import importlib
tf = importlib.import_module("tensorflow"); tf.random = importlib.import_module("tensorflow.random")
ref = importlib.import_module("gan.gan_train_for_shadow")
from tests.emu_backend import EmuBackend
from hypelcnn_amd.backend import Ref
import torch
rng = np.random.default_rng(3)
bands, rate = 7, 0.4
normal = rng.random(bands).astype(np.float32); shadow = (rng.random(bands) * 0.5).astype(np.float32)
ratio = (1.5 + rng.random(bands)).astype(np.float32)
emu = EmuBackend()
for u1, u2 in ((0.1, 0.1), (0.1, 0.9), (0.9, 0.1), (0.9, 0.9), (0.4, 0.39999)):
    draws[:] = [u1, u2]
    x_ref, y_ref = ref.perform_shadow_augmentation_random(normal.copy(), shadow.copy(), ratio, rate)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    ox, oy = torch.zeros(bands), torch.zeros(bands)
    emu.k_gather_pairs_f32(Ref(t(normal)), Ref(t(shadow)), Ref(torch.zeros(1, dtype=torch.int64)), 1, bands, Ref(t(ratio)),
                           Ref(t(np.float32([u1]))), Ref(t(np.float32([u2]))), rate, Ref(ox), Ref(oy))
    assert np.array_equal(ox.numpy(), np.asarray(x_ref, np.float32)) and np.array_equal(oy.numpy(), np.asarray(y_ref, np.float32)), (u1, u2)
print("ok")
""" % (ROOT, GOLD)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.skipif(not HAVE_REF, reason="the reference checkout exists in the build container only")
def test_generator_as_augmenter_equals_the_executed_reference():
    """`gan/gan_utilities.py:30-43` (create_gan_struct) + `gan/wrappers/gan_common.py:282-304`
    (create_inference_for_matrix_input) + the reference's CycleGANInferenceWrapper and generator, executed in float64 on a
    [P, P, B + 1] patch -- P * P generator copies under Model/ModelX2Y|ModelY2X/Generator, the LiDAR channel passed through --
    against the product's GeneratorAugmenter (ONE fused launch over all pixel spectra) on the kernel emulation, with the
    variables the reference's run created, for the shadowing and the de-shadowing direction."""
    code = r"""
import sys
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import tfgan_standin as W
W.install()
import importlib, torch
util = importlib.import_module("gan.gan_utilities")
registry = importlib.import_module("gan.wrapper_registry")
P, B = 3, 16
rng = np.random.default_rng(11)
patch = rng.random((P, P, B + 1))
ref_wrapper = registry.get_infer_wrapper_dict()["cycle_gan"]
holder = util.create_gan_struct(ref_wrapper, "", "")
eng = W.WiringEngine(rng=np.random.default_rng(12))
outs = {}
with W.S.use_engine(eng):
    t = eng.placeholder(patch, "patch")
    outs[True] = holder.shadow_op(t).var.v
    outs[False] = holder.deshadow_op(t).var.v
assert sorted({n.rsplit("/", 2)[0] for n in eng.variables}) == ["Model/ModelX2Y/Generator", "Model/ModelY2X/Generator"]
assert len(eng.variables) == 28   # 2 x 7 layers x (weights, biases): P * P applications SHARE them
from hypelcnn_amd.gan.gan_utilities import GeneratorAugmenter
from hypelcnn_amd.gan.wrapper_registry import get_infer_wrapper_dict
from tests.emu_backend import EmuBackend
for is_shadow in (True, False):
    aug = GeneratorAugmenter(get_infer_wrapper_dict()["cycle_gan"], is_shadow, B, EmuBackend())
    aug.load({k: np.asarray(v, np.float32) for k, v in eng.params.items()})
    got = aug(torch.as_tensor(patch[None].astype(np.float32))).numpy()[0]
    want = outs[is_shadow]
    assert got.shape == want.shape == (P, P, B + 1)
    assert np.array_equal(got[..., -1], patch[..., -1].astype(np.float32)), "LiDAR channel must pass through"
    assert np.abs(got - want).max() < 2e-5 * max(1.0, np.abs(want).max()), (is_shadow, np.abs(got - want).max())
assert np.abs(outs[True] - outs[False]).max() > 1e-3   # two different generators
print("ok")
""" % (ROOT, GOLD)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-3000:]
