"""CPU (numpy kernel emulation): the whole classifier path through the reference-shaped entry points --
flags -> SyntheticDataLoader -> InMemoryImporter -> create_graph -> run_monitored_session (hooks, checkpoints,
resume, summaries) -- learns a small synthetic scene."""
import json
import os

import numpy as np

from hypelcnn_amd.classify import train_for_classification as T
from tests.emu_backend import EmuBackend

ALG = {"batch_size": 32, "drop_out_ratio": 0.3, "filter_count": 32, "learning_rate": 3e-3,
       "learning_rate_decay_factor": 0.96, "learning_rate_decay_step": 350, "lrelu_alpha": 0.18,
       "optimizer": "AdamOptimizer", "bn_decay": 0.9, "l2regularizer_scale": 1e-5, "spectral_hierarchy_level": 1,
       "spatial_hierarchy_level": 1, "degradation_coeff": 3, "use_residual": True}


def _flags(tmp_path, steps, extra=()):
    p = tmp_path / "alg.json"
    p.write_text(json.dumps(ALG))
    argv = ["--loader_name", "SyntheticDataLoader", "--path", "grss2013:h=24:w=30:bands=10:classes=3:samples=0.6",
            "--neighborhood", "1", "--model_name", "HYPELCNNModel", "--algorithm_param_path", str(p),
            "--batch_size", "32", "--step", str(steps), "--base_log_path", str(tmp_path / "log"),
            "--perform_validation", "true", "--validation_steps", "60", "--save_checkpoint_steps", "50",
            "--unknown_flag_is_ignored", "1"] + list(extra)
    flags, _ = T.build_parser().parse_known_args(argv)
    return flags


def test_end_to_end_training_learns_and_resumes(tmp_path):
    flags = _flags(tmp_path, 121, ["--augment_data_with_rotation", "true", "--augment_data_with_reflection", "true"])
    alg = dict(ALG)
    model = T.get_model_from_name(flags.model_name)
    log_dir = os.path.join(flags.base_log_path, T.get_log_suffix(flags))
    res = T.perform_an_episode(flags, alg, model, log_dir, backend=EmuBackend())
    assert np.isfinite(res.loss)
    assert res.test_accuracy > 0.85 and res.validation_accuracy > 0.85, (res.test_accuracy, res.validation_accuracy)
    ckpts = sorted(os.listdir(log_dir))
    assert "model.ckpt-50.npz" in ckpts and "model.ckpt-100.npz" in ckpts and "model.ckpt-120.npz" in ckpts
    recs = [json.loads(l) for l in open(os.path.join(log_dir, "summaries.jsonl"))]
    assert "flags" in recs[0] and any("validation_kappa" in r for r in recs)
    with np.load(os.path.join(log_dir, "model.ckpt-120.npz")) as z:
        keys = {k.replace("|", "/") for k in z.files}
    assert {"nn_core/conv_enc_0/weights", "nn_core/fc_final/BatchNorm/moving_mean", "global_step",
            "training_optimizer/m"} <= keys
    # resume: a second episode with a larger step budget continues from global_step 120
    flags2 = _flags(tmp_path, 131)
    res2 = T.perform_an_episode(flags2, alg, model, log_dir, backend=EmuBackend())
    assert "model.ckpt-130.npz" in os.listdir(log_dir)
    assert res2.test_accuracy > 0.85


def test_log_suffix_format(tmp_path):
    flags = _flags(tmp_path, 1, ["--augment_data_with_shadow", "simple", "--augment_data_with_spectral", "0.05"])
    s = T.get_log_suffix(flags)
    assert s == "syntheticldr_hypelcnnmdl_trn010_alg_3x3_simple_aug050_spectral0050", s
