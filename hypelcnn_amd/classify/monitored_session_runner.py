"""The training step loop (reference classify/monitored_session_runner.py:31-188).

`run_monitored_session` keeps the reference's signature and hook schedule:
  * Init hook: feed the training arrays once (device resident afterwards).
  * Test hook: every 100 steps (global_step % 100 == 1) and at the end -- full test-set evaluation.
    The reference also re-evaluates `cross_entropy` there with an extra session.run that consumes a
    batch and updates BN statistics without a weight update; not reproduced: the
    reported loss is the one of the last real training step.
  * Validation hook: at the last step and whenever step % validation_steps == 1 (step != 1).
  * StopAtStepHook(last_step=required_steps - 1), NaN-loss guard (stop, do not raise).
  * Checkpoints every `save_checkpoint_steps` into log_dir (keep 20), keyed by the TF variable names
    (nn_core/*, global_step, training_optimizer/*); the newest one is restored on start = implicit resume.
  * Summaries: JSON lines in log_dir/summaries.jsonl instead of TensorBoard event files.
"""
import glob
import json
import os
import re

import numpy

from hypelcnn_amd.common.common_nn_ops import TrainingResult, calculate_accuracy

TEST_ITERATION_COUNT = 100
MAX_CHECKPOINTS = 20


def set_run_seed(seed=1234):
    """tf.compat.v1.set_random_seed(1234) (reference :11-13): seeds parameter init, shuffling, dropout."""
    import torch
    torch.manual_seed(seed)
    numpy.random.seed(seed)
    return seed


def add_classification_summaries(cross_entropy, learning_rate, log_all_model_variables, testing_nn_params,
                                 validation_nn_params):
    """Returns a callable producing the summary record the reference writes to TensorBoard (:16-28)."""

    def collect(sess, step):
        rec = {"step": int(step), "training_cross_entropy": cross_entropy.eval(),
               "training_learning_rate": learning_rate.eval(step)}
        if testing_nn_params is not None and testing_nn_params.metrics._confusion_dev is not None:
            rec["test_overall_accuracy"] = testing_nn_params.metrics.accuracy
            rec["test_confusion"] = testing_nn_params.metrics.confusion.tolist()
        if validation_nn_params is not None and validation_nn_params.metrics._confusion_dev is not None:
            m = validation_nn_params.metrics
            rec.update(validation_overall_accuracy=m.accuracy, validation_average_accuracy=m.mean_per_class_accuracy,
                       validation_kappa=m.kappa, validation_confusion=m.confusion.tolist())
        if log_all_model_variables:
            rec["variable_norms"] = {n: float(numpy.linalg.norm(sess.get_variable(n))) for n in sess.variable_names()}
        return rec

    return collect


class SummaryWriter:
    def __init__(self, log_dir):
        self.path = os.path.join(log_dir, "summaries.jsonl") if log_dir and is_chief() else None
        if self.path:
            os.makedirs(log_dir, exist_ok=True)

    def add(self, record):
        if self.path:
            with open(self.path, "a") as f:
                f.write(json.dumps(record) + "\n")


def latest_checkpoint(log_dir):
    """Newest model.ckpt-N in the directory: .npz files of this build or TensorFlow bundles (returned as prefix)."""
    cands = []
    if log_dir:
        for f in glob.glob(os.path.join(log_dir, "model.ckpt-*.npz")):
            cands.append((int(re.search(r"ckpt-(\d+)\.npz$", f).group(1)), 1, f))
        for f in glob.glob(os.path.join(log_dir, "model.ckpt-*.index")):
            cands.append((int(re.search(r"ckpt-(\d+)\.index$", f).group(1)), 0, f[:-len(".index")]))
    return max(cands)[2] if cands else None


def is_chief():
    """Rank 0 writes checkpoints and summaries (the reference's MonitoredTrainingSession is_chief)."""
    import torch.distributed as dist
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0


def save_checkpoint(sess, log_dir, step):
    os.makedirs(log_dir, exist_ok=True)
    sess.average_state()  # collective: every rank takes part, only the chief writes
    path = os.path.join(log_dir, f"model.ckpt-{int(step)}.npz")
    if not is_chief():
        return path
    numpy.savez(path, **{k.replace("/", "|"): v for k, v in sess.state_dict().items()})
    files = sorted(glob.glob(os.path.join(log_dir, "model.ckpt-*.npz")),
                   key=lambda p: int(re.search(r"ckpt-(\d+)\.npz$", p).group(1)))
    for old in files[:-MAX_CHECKPOINTS]:
        os.remove(old)
    return path


def restore_checkpoint(sess, path):
    """`path`: an .npz written by save_checkpoint, or the prefix of a TensorFlow bundle (model.ckpt-N with its .index /
    .data-00000-of-00001 files, what the reference's Saver writes) -- read without TensorFlow."""
    from hypelcnn_amd.common import tf_checkpoint
    if not path.endswith(".npz") and tf_checkpoint.is_checkpoint(path):
        tf_checkpoint.variables_to_session(sess, tf_checkpoint.read_checkpoint(path))
        return
    with numpy.load(path) as z:
        sess.load_state_dict({k.replace("|", "/"): z[k] for k in z.files})


def export_tf_checkpoint(sess, prefix):
    """Writes the session as a TensorFlow checkpoint bundle (variables, global_step, Adam slots under the TF1 names)
    plus the `checkpoint` state file tf.train.latest_checkpoint reads."""
    from hypelcnn_amd.common import tf_checkpoint
    sess.average_state()
    if not is_chief():
        return prefix
    # a MomentumOptimizer session is exported under <var>/nn_core/Momentum (one slot, no beta powers): the names the
    # reference's Saver restores for alg_param_concnn.json
    tf_checkpoint.write_checkpoint(prefix, tf_checkpoint.session_to_variables(
        sess, momentum=getattr(sess, "optimizer_kind", "adam") == "momentum"))
    with open(os.path.join(os.path.dirname(os.path.abspath(prefix)), "checkpoint"), "w") as f:
        base = os.path.basename(prefix)
        f.write(f'model_checkpoint_path: "{base}"\nall_model_checkpoint_paths: "{base}"\n')
    return prefix


class ValidationHook:
    def __init__(self, validation_nn_params, validation_tensor, class_range, required_steps, iteration, importer):
        self.nn_params, self.tensor, self.class_range = validation_nn_params, validation_tensor, class_range
        self.required_steps, self.iteration_count, self.importer = required_steps, iteration, importer
        self.validation_accuracy = 0

    def after_run(self, sess, iteration):
        if self.nn_params is None:
            return False
        if (iteration == self.required_steps - 1) or (iteration % self.iteration_count == 1 and iteration != 1):
            self.importer.init_tensors(sess, self.tensor, self.nn_params)
            self.validation_accuracy, _, _, kappa, mpca = calculate_accuracy(sess, self.nn_params, self.class_range)
            print("Validation metrics #%d : Overall accuracy=%g, Class based average accuracy=%g, Kappa=%g" % (
                iteration, self.validation_accuracy, mpca, kappa))
            return True
        return False


class TestHook:
    __test__ = False

    def __init__(self, testing_nn_params, testing_tensor, cross_entropy, test_iteration_count, class_range, importer):
        self.nn_params, self.tensor, self.cross_entropy = testing_nn_params, testing_tensor, cross_entropy
        self.count, self.class_range, self.importer = test_iteration_count, class_range, importer
        self.testing_accuracy = 0
        self.loss = 0

    def after_run(self, sess, iteration):
        if iteration % self.count == 1:
            self._perform(sess, iteration)
            return True
        return False

    def end(self, sess, iteration):
        self._perform(sess, iteration)

    def _perform(self, sess, iteration):
        self.loss = self.cross_entropy.eval()
        if self.nn_params.data_with_labels.data.size != 0:
            self.importer.init_tensors(sess, self.tensor, self.nn_params)
            self.testing_accuracy, _, _, _, _ = calculate_accuracy(sess, self.nn_params, self.class_range)
        print("Training step=%d, Testing accuracy=%g, loss=%.5f" % (iteration, self.testing_accuracy, self.loss))


def run_monitored_session(cross_entropy, log_dir, class_range, save_checkpoint_steps, validation_steps, train_step,
                          required_steps, augmentation_info, training_nn_params, training_tensor, testing_nn_params,
                          testing_tensor, validation_nn_params, validation_tensor, importer, flags_as_json_str,
                          alg_params_as_json_str, summary_fn=None):
    sess = train_step.ctx.session()
    writer = SummaryWriter(log_dir)
    ckpt = latest_checkpoint(log_dir)
    if ckpt is not None:
        restore_checkpoint(sess, ckpt)
        print(f"Restored {ckpt} (global_step={sess.global_step})")
    else:
        writer.add({"flags": flags_as_json_str, "algorithm_params": alg_params_as_json_str})
    if augmentation_info is not None and augmentation_info.perform_shadow_augmentation and \
            augmentation_info.shadow_struct is not None and \
            getattr(augmentation_info.shadow_struct, "shadow_op_initializer", None) is not None:
        augmentation_info.shadow_struct.shadow_op_initializer(None, sess)
    importer.init_tensors(sess, training_tensor, training_nn_params)  # InitHook (:40-45)
    if sess.dist is not None and hasattr(train_step, "precapture"):
        train_step.precapture()  # full batch + ragged tail now, not next to a live all-reduce later

    validation_hook = ValidationHook(validation_nn_params, validation_tensor, class_range, required_steps,
                                     validation_steps, importer)
    test_hook = TestHook(testing_nn_params, testing_tensor, cross_entropy, TEST_ITERATION_COUNT, class_range, importer)
    last_step = required_steps - 1  # StopAtStepHook(last_step=required_steps - 1)

    while sess.global_step < last_step:
        try:
            train_step.run()
        except StopIteration:  # the epoch-limited iterator is exhausted (tf.errors.OutOfRangeError)
            break
        iteration = sess.global_step
        # NanTensorHook(fail_on_nan_loss=False) (:151), every step: the device flags a non-finite loss, the guarded
        # optimiser has already refused that update; here the host only reads flag copies that are one step old, so
        # the loop stays asynchronous.  Under data parallelism the flag rode in the gradient all-reduce: every rank
        # stops at the same iteration.
        bad = sess.nonfinite_step()
        if bad is not None:
            print(f"NaN loss at step {bad}: stopping")
            break
        evaluated = validation_hook.after_run(sess, iteration)
        evaluated = test_hook.after_run(sess, iteration) or evaluated
        if evaluated or iteration % TEST_ITERATION_COUNT == 0:
            loss = cross_entropy.eval()
            # NanTensorHook(fail_on_nan_loss=False): stop, do not raise.  The verdict is the step's ALL-REDUCED flag
            # (sync=True also reads the newest copy), never this rank's local loss: under data parallelism only the
            # rank whose shard produced the NaN would leave the loop, and its test_hook.end / average_state
            # collectives would meet the other ranks' gradient all-reduce.
            bad = sess.nonfinite_step(sync=True)
            if bad is not None or (sess.dist is None and not numpy.isfinite(loss)):
                print(f"NaN loss at step {bad if bad is not None else iteration}: stopping")
                break
            if summary_fn is not None:
                writer.add(summary_fn(sess, iteration))
        if save_checkpoint_steps and iteration % save_checkpoint_steps == 0:
            if sess.nonfinite_step(sync=True) is not None:  # never write a checkpoint past a non-finite step
                print(f"NaN loss at step {sess.nonfinite_step()}: stopping")
                break
            save_checkpoint(sess, log_dir, iteration)
    test_hook.end(sess, sess.global_step)
    if log_dir and sess.nonfinite_step(sync=True) is None:
        save_checkpoint(sess, log_dir, sess.global_step)
    return TrainingResult(validation_accuracy=validation_hook.validation_accuracy,
                          test_accuracy=test_hook.testing_accuracy, loss=test_hook.loss)
