"""-m gpu: the RCCL path of the data-parallel step on the 1-GPU box."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_exchange_with_segmented_graphs_is_bit_identical_to_plain_step():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HYPEL_DP_SELFTEST="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dp_rccl_worker.py")]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DP_RCCL_SELFTEST_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("global_nb", [128, 512])
def test_two_ranks_with_sync_bn_equal_one_device_at_the_global_batch_on_the_gpu(tmp_path, global_nb):
    """The GPU twin of tests/test_dp_gloo.py's synchronised-batch-norm equivalence, at two row counts (round-5 verdict): two
    processes on the one MI355X, half of the global batch each, against one process with the whole batch -- GRSS2013
    HYPELCNN at its shipped configuration, on the HIP kernels.  What the emulation cannot see: both plans must pick the SAME
    kernel family for every product (the split-operand / fp32 choice is a function of the layer, priced at a nominal
    batch -- not of the rows a rank happens to hold), so that a sample meets the same arithmetic on 1 or N ranks."""
    import torch
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "ranks.pt")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dp_syncbn_gpu_worker.py"), str(global_nb), out]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "SYNCBN_GPU_WORKER_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    two = torch.load(out)
    sys.path.insert(0, ROOT)
    from hypelcnn_amd.backend import HipBackend
    from tests import dp_syncbn_gpu_worker as W
    from tests import parity_util as U
    alg, params, x, onehot, rng = W.case(global_nb)
    built = U.build("HYPELCNNModel", 7, 145, 15, alg, HipBackend(), with_eval=False)
    sess = built.ctx.session()
    assert sess.dist is None
    U.inject(sess, params)
    masks = U.make_masks(built, global_nb, rng)
    ct = U.run_train_step(built, x, onehot, masks)
    torch.cuda.synchronize()
    tags_one = W.split_tags(ct)
    assert tags_one and tags_one == two["split_tags"], (sorted(set(tags_one) ^ set(two["split_tags"])))
    g = sess.grads[:sess.n_train].cpu()
    scale = float(g.abs().max())
    logits = ct.value(built.y_conv).cpu()
    assert float((two["logits"] - logits).abs().max()) <= 2e-5 * max(1.0, float(logits.abs().max()))
    assert torch.equal(two["logits"].argmax(1), logits.argmax(1))
    # a different summation order (two partial statistics merged instead of one pass) may flip one or two leaky-ReLU kink
    # decisions among millions of activations (tests/parity_util.py): bound the bulk tightly, the maximum loosely
    d = (two["grads"] - g).abs()
    assert float(d.median()) < 1e-6 * scale and float(d.max()) < 2e-2 * scale, (float(d.median()) / scale, float(d.max()) / scale)
    torch.testing.assert_close(two["state"], sess.state.cpu(), rtol=1e-4, atol=1e-5)
