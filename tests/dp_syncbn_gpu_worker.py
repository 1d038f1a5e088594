"""Worker of tests/test_gpu_dp.py::test_two_ranks_with_sync_bn_equal_one_device_at_the_global_batch_on_the_gpu: two processes
on ONE MI355X (gloo carries the collectives between them: a 1-GPU box has no second device for RCCL), each holding half of
the global batch of the GRSS2013 HYPELCNN configuration, synchronised batch norm.  Rank 0 writes what it computed."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def case(global_nb):
    from tests import parity_util as U
    alg = json.load(open(os.path.join(ROOT, "hypelcnn_amd", "nnmodel", "modelconfigs", "alg_param_hypelcnn.json")))
    rng = np.random.default_rng(4242)
    params = U.make_params("HYPELCNNModel", 7, 145, 15, alg, rng)
    x = rng.random((global_nb, 7, 7, 145)).astype(np.float32)
    onehot = np.eye(15, dtype=np.float32)[rng.integers(0, 15, global_nb)]
    return alg, params, x, onehot, rng


def split_tags(ct):
    launches = ct.plan.fwd + ct.plan.bwd
    return sorted({l.tag for l in launches if l.name.startswith("seg_gemm") and l.name != "seg_gemm_multi_f32"
                   and l.args[14] & 0x8000} |
                  {p for l in launches if l.name == "seg_gemm_multi_f32" and l.args[3] & 0x100 for p in l.meta["products"]})


def main():
    global_nb, out = int(sys.argv[1]), sys.argv[2]
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from hypelcnn_amd.backend import HipBackend
    from tests import parity_util as U
    alg, params, x, onehot, rng = case(global_nb)
    built = U.build("HYPELCNNModel", 7, 145, 15, alg, HipBackend(), with_eval=False)
    sess = built.ctx.session()
    assert sess.dist == (world, rank)
    U.inject(sess, params)
    masks = U.make_masks(built, global_nb, rng)
    per = global_nb // world
    lo, hi = rank * per, (rank + 1) * per
    ct = sess.compile(built.train_tower, per, loss=built.train_step.loss, external_masks=True, global_nb=global_nb,
                      sync_bn=True)
    assert ct.plan.sync_bn
    U.feed(ct, x[lo:hi], onehot[lo:hi], {k: m[lo:hi] for k, m in masks.items()})
    sess.train_step_exchange(ct)
    torch.cuda.synchronize()
    logits = ct.value(built.y_conv).cpu()
    gathered = [torch.zeros_like(logits) for _ in range(world)]
    dist.all_gather(gathered, logits)
    if rank == 0:
        torch.save({"grads": sess.grads[:sess.n_train].cpu(), "logits": torch.cat(gathered), "split_tags": split_tags(ct),
                    "state": sess.state.cpu()}, out)
    dist.barrier()
    dist.destroy_process_group()
    print("SYNCBN_GPU_WORKER_OK")


if __name__ == "__main__":
    main()
