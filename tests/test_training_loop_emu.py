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


def test_training_through_tfrecord_importer(tmp_path):
    """--importer_name TFRecordImporter on an export of the same scene trains like the in-memory importer."""
    from hypelcnn_amd.utilities import tfrecord_writer
    scene = "grss2013:h=24:w=30:bands=10:classes=3:samples=0.6"
    rec = str(tmp_path / "records")
    tfrecord_writer.export("SyntheticDataLoader", scene, 0.1, 1, rec)
    flags = _flags(tmp_path, 101, ["--importer_name", "TFRecordImporter", "--path", scene + ":base_dir=" + rec])
    model = T.get_model_from_name(flags.model_name)
    log_dir = os.path.join(flags.base_log_path, "tfrecord")
    res = T.perform_an_episode(flags, dict(ALG), model, log_dir, backend=EmuBackend())
    assert res.test_accuracy > 0.85 and res.validation_accuracy > 0.85, (res.test_accuracy, res.validation_accuracy)


def test_log_suffix_format(tmp_path):
    flags = _flags(tmp_path, 1, ["--augment_data_with_shadow", "simple", "--augment_data_with_spectral", "0.05"])
    s = T.get_log_suffix(flags)
    assert s == "syntheticldr_hypelcnnmdl_trn010_alg_3x3_simple_aug050_spectral0050", s


# ------------------------------------------------------------------------------------------------ cfg5: joint loop
GAN_SCENE = "gulfport:h=24:w=30:bands=16:classes=3:samples=0.6"


def _gan_params(tmp_path, gan_type, steps, scene=None, batch=32):
    from hypelcnn_amd.gan import gan_train_for_shadow as GT
    import argparse
    from hypelcnn_amd.common import cmd_parser as cp
    parser = argparse.ArgumentParser()
    for add in (cp.add_parse_cmds_for_loaders, cp.add_parse_cmds_for_loggers, cp.add_parse_cmds_for_trainers,
                cp.add_parse_cmds_for_json_loader, GT.add_parse_cmds_for_app, cp.add_parse_cmds_for_opt):
        add(parser)
    flags, _ = parser.parse_known_args(["--loader_name", "SyntheticDataLoader", "--path", scene or GAN_SCENE,
                                        "--gan_type", gan_type, "--batch_size", str(batch), "--step", str(steps),
                                        "--base_log_path",
                                        str(tmp_path / "gan"), "--validation_steps", "1000",
                                        "--validation_sample_count", "64"])
    return GT, dict(vars(flags))


def run_joint_loop(tmp_path, backend_factory, gan_steps=30, cls_steps=40, scene=None, gan_type="cycle_gan",
                   neighborhood=1, alg=None, batch=32, gan_batch=32):
    """SURVEY cfg5: train a shadow GAN on (lit, shadowed) spectra of the scene, then train the classifier with the
    trained generator applied to every pixel of a patch with probability 0.5 (augment_data_with_shadow=<gan_type>;
    the loaders register cycle_gan / dcl_gan / dcl_cycle_gan generators, loader/AVONDataLoader.py:38-45)."""
    scene = scene or GAN_SCENE
    alg = alg or ALG
    GT, params = _gan_params(tmp_path, gan_type, gan_steps, scene, gan_batch)
    div = GT.run_session(params, params["base_log_path"], backend=backend_factory())
    assert all(np.isfinite(d) for d in div)
    gan_dir = f"{params['base_log_path']}_{GT.get_log_suffix(type('F', (), params))}"
    ckpts = sorted(os.listdir(gan_dir), key=lambda f: int(f.split("-")[1].split(".")[0]))
    assert ckpts, "the GAN session must leave a checkpoint (it stops at the step budget or when the pairs run out)"
    ckpt = os.path.join(gan_dir, ckpts[-1])
    with np.load(ckpt) as z:
        names = {k.replace("|", "/") for k in z.files}
    assert any(n.startswith("Model/ModelX2Y/Generator/net1/weights") for n in names), sorted(names)[:6]

    p = tmp_path / "alg.json"
    p.write_text(json.dumps(alg))
    argv = ["--loader_name", "SyntheticDataLoader", "--path", scene + f":gan_ckpt={ckpt}", "--neighborhood",
            str(neighborhood), "--model_name", "HYPELCNNModel", "--algorithm_param_path", str(p), "--batch_size",
            str(batch), "--step", str(cls_steps), "--base_log_path", str(tmp_path / "log"), "--perform_validation",
            "false", "--save_checkpoint_steps", "1000", "--augment_data_with_shadow", gan_type,
            "--augment_data_with_rotation", "true", "--augmentation_random_threshold", "0.5"]
    flags, _ = T.build_parser().parse_known_args(argv)
    be = backend_factory()
    seen = []
    real_call = be.call
    be.call = lambda name, *a: (seen.append(name), real_call(name, *a))[1]
    log_dir = os.path.join(flags.base_log_path, T.get_log_suffix(flags))
    res = T.perform_an_episode(flags, dict(alg), T.get_model_from_name(flags.model_name), log_dir, backend=be)
    assert np.isfinite(res.loss)
    assert "augment_patches_f32" in seen, "the fused augmentation kernel must carry the generator output"
    return res, seen


def test_joint_gan_augmentation_and_classifier_loop(tmp_path):
    res, seen = run_joint_loop(tmp_path, EmuBackend)
    assert res.test_accuracy > 0.5


def test_joint_loop_with_cut_based_generator_on_hsi_only_scene(tmp_path):
    """BASELINE configs[4] in miniature: AVON geometry (no LiDAR channel, 2 classes), the CUT-based DCL-GAN (two
    cut models, NCE + feature discriminators) trains the shadow generator, which then augments the HYPELCNN input."""
    res, seen = run_joint_loop(tmp_path, EmuBackend, gan_steps=12, cls_steps=40, gan_type="dcl_gan",
                               scene="avon:h=24:w=30:bands=24:samples=0.6")
    assert res.test_accuracy > 0.5


def test_simple_shadow_struct_is_registered_by_the_loader():
    from hypelcnn_amd.loader.SyntheticDataLoader import SyntheticDataLoader
    ds = SyntheticDataLoader("grss2013:h=12:w=14:bands=6:classes=2").load_data(1, True)
    assert set(ds.shadow_creator_dict) == {"simple"}
    ratio = ds.shadow_creator_dict["simple"].ratio
    assert ratio.shape == (7,) and ratio[-1] == 1.0 and (ratio[:-1] > 1.0).all()   # lit / shadow > 1, LiDAR untouched
    ds2 = SyntheticDataLoader("avon:h=12:w=14:bands=8").load_data(0, True)
    assert ds2.shadow_creator_dict["simple"].ratio.shape == (8,)


def test_session_loop_stops_on_nan_loss_without_poisoned_checkpoint(tmp_path, monkeypatch):
    """A batch with a NaN input at step 7: the loop stops within two steps, no checkpoint is written after it and
    the parameters are still finite (the device-side guard refused the update)."""
    from hypelcnn_amd.common import common_nn_ops as cno
    flags = _flags(tmp_path, 60, ["--save_checkpoint_steps", "5", "--perform_validation", "false"])
    calls = {"n": 0}
    real = cno.BatchIterator.next_batch

    def poisoned(self):
        b = real(self)
        if b is not None and self.collective:
            calls["n"] += 1
            if calls["n"] == 7:
                x = b[0].clone()
                x[0, 0, 0, 0] = float("nan")
                b = (x,) + tuple(b[1:])
        return b

    monkeypatch.setattr(cno.BatchIterator, "next_batch", poisoned)
    model = T.get_model_from_name(flags.model_name)
    log_dir = os.path.join(flags.base_log_path, T.get_log_suffix(flags))
    T.perform_an_episode(flags, dict(ALG), model, log_dir, backend=EmuBackend())
    ckpts = sorted(int(f.split("-")[1].split(".")[0]) for f in os.listdir(log_dir) if f.startswith("model.ckpt-"))
    assert ckpts == [5], ckpts
    with np.load(os.path.join(log_dir, "model.ckpt-5.npz")) as z:
        assert all(np.isfinite(z[k]).all() for k in z.files)
