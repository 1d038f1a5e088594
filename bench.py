#!/usr/bin/env python3
"""Headline benchmark: HSI+LiDAR patches/sec, forward+backward+optimiser, GRSS2013 HYPELCNN
(7x7x(144+1), 15 classes, alg_param_hypelcnn.json), batch 1024 per GPU, fp32 -- BASELINE.json configs[1].

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" = next resident batch -> nhwc->pixel-major -> forward -> loss -> backward -> (RCCL all-reduce)
-> TF1-Adam.  Inputs are synthetic (U[0,1) patches, uniform labels) and already resident in HBM.
Prints ONE JSON line on rank 0 (contract in the task statement) incl. `roofline` for the dominant kernel
(hypel_seg_gemm_f32: fp32 MFMA) and `cpu_baseline` (oracle restatement timed on the host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
MAC_FWD_PER_PATCH = 78579058  # SURVEY.md Appendix B.1 (exact taps, training graph)


def build_model(batch, backend):
    from hypelcnn_amd.common import common_nn_ops as cno
    alg = json.load(open(os.path.join(ROOT, "hypelcnn_amd", "nnmodel", "modelconfigs", "alg_param_hypelcnn.json")))
    alg["batch_size"] = batch
    model = cno.get_model_from_name("HYPELCNNModel")
    template = cno.Template("nn_core", model.create_tensor_graph, class_count=15)
    ctx = cno.GraphContext(template, backend)
    images = cno.Placeholder("x", (7, 7), 145)
    labels = cno.Placeholder("labels", None, 15)
    _, cross_entropy, lr, train_step = cno.optimize_nn(template, images, labels, "/gpu:0", "training", alg,
                                                       model.get_loss_func, ctx=ctx)
    return ctx, train_step, lr, alg


def measure_gemm_events(ct, sess, lr, steps):
    """Serial replay of the same step (every launch on the main stream, no graph) with a HIP event pair around
    every hypel_seg_gemm_f32 launch, recorded on the stream the kernels are launched on."""
    launches = ct.serial_launches()
    total_ms, total_flops, n_launch = 0.0, 0, 0
    for _ in range(steps):
        evs = []
        for l, f in launches:
            if l.name == "seg_gemm_f32":
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                f()
                b.record()
                evs.append((a, b, l.flops))
            else:
                f()
        sess.adam_step(lr.eval(sess.global_step))
        torch.cuda.synchronize()
        total_ms += sum(a.elapsed_time(b) for a, b, _ in evs)
        n_launch += len(evs)
        total_flops += sum(fl for _, _, fl in evs)
    return total_ms, total_flops, n_launch


def cpu_baseline(seconds_budget=20.0):
    """Oracle restatement (numpy, float32, OpenBLAS threads) of the identical train step on a bounded sample:
    batch 64 (BASELINE configs[0] shape), repeated until ~seconds_budget of CPU work."""
    from oracle import models as OM, train as OT
    alg = json.load(open(os.path.join(ROOT, "hypelcnn_amd", "nnmodel", "modelconfigs", "alg_param_hypelcnn.json")))
    rng = np.random.default_rng(1234)
    params = OM.hypelcnn_init_params(7, 145, 15, alg, rng, np.float32)
    tr = OT.ClassifierTrainer("HYPELCNNModel", params, 15, alg)
    nb = 64
    x = rng.random((nb, 7, 7, 145), dtype=np.float32)
    onehot = np.eye(15, dtype=np.float32)[rng.integers(0, 15, nb)]
    tr.train_step(x, onehot)  # warm-up
    t0 = time.perf_counter()
    n = 0
    while True:
        tr.train_step(x, onehot)
        n += 1
        if time.perf_counter() - t0 > seconds_budget or n >= 50:
            break
    dt = time.perf_counter() - t0
    return {"value": nb * n / dt, "unit": "patches/s", "cores": os.cpu_count(), "kind": "port",
            "sample": f"{n} train steps of batch 64 (7x7x145, fp32 numpy/OpenBLAS oracle restatement, "
                      f"{dt:.1f} s); not a TensorFlow number"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1024, help="patches per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of HIP-graph replay")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        print("bench.py: --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from hypelcnn_amd.backend import HipBackend
    be = HipBackend()
    ctx, train_step, lr, alg = build_model(args.batch, be)
    ctx.seed = 1234
    ctx.capture_graphs = not args.no_graph
    sess = ctx.session()  # broadcasts rank-0 weights when world > 1
    nb = args.batch
    gen = torch.Generator(device="cpu")
    gen.manual_seed(1234 + rank)
    x = torch.rand((nb, 7, 7, 145), generator=gen).cuda()
    lab = torch.randint(0, 15, (nb,), generator=gen)
    onehot = torch.nn.functional.one_hot(lab, 15).float().cuda()
    ct = train_step.compiled(nb)
    ct.set_input("x", x)
    ct.set_input("labels", onehot)

    def one_step():
        ct.forward_backward()
        sess.allreduce_gradients()
        sess.adam_step(lr.eval(sess.global_step))

    for _ in range(args.warmup):
        one_step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
    loss = ct.loss_value()
    assert np.isfinite(loss), "non-finite loss"

    roof = None
    cpu = None
    if rank == 0:
        ev_steps = max(2, min(5, args.steps))
        ms, flops, n_launch = measure_gemm_events(ct, sess, lr, ev_steps)
        achieved = flops / (ms * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": "hypel_seg_gemm_f32 (fp32 v_mfma_f32_32x32x2)", "achieved": achieved,
                "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_F32_MFMA_TFLOPS,
                "traffic": None, "launches_per_step": n_launch // ev_steps,
                "avg_launch_us": ms * 1e3 / n_launch,
                "gemm_ms_per_step": ms / ev_steps,
                "algorithmic_gflop_per_step": flops / ev_steps / 1e9}
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline()
    if world > 1:
        dist.barrier()
    if rank == 0:
        out = {"metric": "HSI+LiDAR patches/sec fwd+bwd (GRSS2013 7x7x145)", "value": nb * world * args.steps / dt,
               "unit": "patches/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic",
               "config": {"workload": "GRSS2013 HYPELCNNModel train step (fwd+bwd+TF1-Adam), 7x7 patch, 144 HSI + 1 "
                                      "LiDAR bands, 15 classes, alg_param_hypelcnn.json, random-init weights",
                          "batch_per_gpu": nb, "global_batch": nb * world,
                          "parallelism": f"dp{world}" if world > 1 else "single",
                          "hip_graph": ctx.capture_graphs, "loss": loss,
                          "mfma_ceiling_patches_per_s_per_gpu": PEAK_F32_MFMA_TFLOPS * 1e12 / (6 * MAC_FWD_PER_PATCH)},
               "roofline": roof, "cpu_baseline": cpu}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
