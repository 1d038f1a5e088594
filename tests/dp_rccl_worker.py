"""Worker of tests/test_gpu_dp.py: run under `python -m torch.distributed.run --nproc-per-node 1` with
HYPEL_DP_SELFTEST=1 so that the whole RCCL path (init, weight broadcast, step cut into HIP-graph segments at the
plan's sync point, asynchronous all-reduce of the finished gradient tail under the rest of the backward pass, head
all-reduce, guarded Adam) runs on a 1-rank communicator -- the only multi-process GPU check a 1-GPU box allows."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
    from hypelcnn_amd.backend import HipBackend
    from tests import parity_util as U
    alg = {"drop_out_ratio": 0.7, "filter_count": 96, "learning_rate": 3e-4, "learning_rate_decay_factor": 0.96,
           "learning_rate_decay_step": 350, "lrelu_alpha": 0.18, "optimizer": "AdamOptimizer", "bn_decay": 0.95,
           "l2regularizer_scale": 1e-5, "spectral_hierarchy_level": 3, "spatial_hierarchy_level": 3,
           "degradation_coeff": 3, "use_residual": True}
    rng = np.random.default_rng(5)
    patch, ch, classes, nb = 7, 33, 6, 256
    built = U.build("HYPELCNNModel", patch, ch, classes, alg, HipBackend(), with_eval=False)
    sess = built.ctx.session()
    assert sess.dist == (1, 0), "HYPEL_DP_SELFTEST=1 keeps the data-parallel path on a 1-rank communicator"
    params = U.make_params("HYPELCNNModel", patch, ch, classes, alg, rng)
    U.inject(sess, params)
    x = rng.random((nb, patch, patch, ch)).astype(np.float32)
    onehot = np.eye(classes, dtype=np.float32)[rng.integers(0, classes, nb)]
    masks = U.make_masks(built, nb, rng)
    ct = U.run_train_step(built, x, onehot, masks)  # plain eager step, no exchange
    torch.cuda.synchronize()
    g_plain = sess.grads.clone()
    assert ct.sync_points, "the data-parallel plan must carry a sync point"
    U.inject(sess, params)
    sess.train_step_exchange(ct)  # eager segments + async all-reduce
    torch.cuda.synchronize()
    assert torch.equal(sess.grads, g_plain), "eager exchange changed the gradients"
    U.inject(sess, params)
    ct.capture()  # one HIP graph per segment
    assert ct._segments is not None
    assert sum(1 for kind, _ in ct._segments if kind == "run") == len(ct.sync_points) + 1
    U.inject(sess, params)
    calls = []
    orig = dist.all_reduce
    dist.all_reduce = lambda t, **kw: (calls.append(int(t.numel())), orig(t, **kw))[1]
    sess.train_step_exchange(ct)
    dist.all_reduce = orig
    torch.cuda.synchronize()
    assert torch.equal(sess.grads, g_plain), "segmented HIP graphs + async all-reduce changed the gradients"
    assert len(calls) == 2 and sum(calls) == sess.grads.numel(), calls
    p0 = sess.params.clone()
    sess.adam_step(3e-4)
    torch.cuda.synchronize()
    assert not torch.equal(sess.params, p0) and sess.nonfinite_step(sync=True) is None
    # synchronised batch norm: the statistics travel through RCCL (all-gather of the per-rank records in the forward
    # pass, all-reduce of the two backward sums) between the graph segments; on one rank the global batch IS the local
    # one, so the step must reproduce the plain gradients up to the different merge arithmetic
    U.inject(sess, params)
    ct_s = sess.compile(built.train_tower, nb, loss=built.train_step.loss, external_masks=True, sync_bn=True)
    assert ct_s is not ct and ct_s.plan.sync_bn
    n_coll = sum(1 for l in ct_s.plan.fwd + ct_s.plan.bwd if l.name in ("_allgather", "_allreduce"))
    assert n_coll >= 2
    U.feed(ct_s, x, onehot, masks)
    sess.train_step_exchange(ct_s)
    torch.cuda.synchronize()
    g_eager = sess.grads.clone()
    # against the float64 oracle (the plain step is no yardstick here: the two paths round the statistics differently,
    # and one leaky-ReLU kink decision of 250 000 flipping moves a dense-layer gradient by percents)
    U.compare_step(built, ct_s, params, x, onehot, masks, "HYPELCNNModel", classes, alg, tol_logit=1e-4, tol_grad=5e-4)
    ct_s.capture()
    assert sum(1 for kind, _ in ct_s._segments if kind == "host") == n_coll
    sess.train_step_exchange(ct_s)
    torch.cuda.synchronize()
    assert torch.equal(sess.grads, g_eager), "graph segments around the BN collectives changed the gradients"
    print(f"DP_RCCL_SELFTEST_OK buckets={calls} sync_bn_collectives={n_coll}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
