"""CPU, world_size 2, gloo: the data-parallel path -- identical weights after broadcast, one flat
gradient all-reduce, identical parameters on both ranks after the optimiser step, and the exchanged gradient
equals the mean of the two shards' gradients computed without DP."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ALG = {"drop_out_ratio": 0.7, "filter_count": 48, "learning_rate": 3e-4, "learning_rate_decay_factor": 0.96,
       "learning_rate_decay_step": 350, "lrelu_alpha": 0.18, "optimizer": "AdamOptimizer", "bn_decay": 0.95,
       "l2regularizer_scale": 1e-5, "spectral_hierarchy_level": 2, "spatial_hierarchy_level": 2,
       "degradation_coeff": 3, "use_residual": True}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data(rank):
    rng = np.random.default_rng(77)
    x = rng.random((8, 5, 5, 9)).astype(np.float32)
    onehot = np.eye(4, dtype=np.float32)[rng.integers(0, 4, 8)]
    return x[rank::2], onehot[rank::2]


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import parity_util as U
    from tests.emu_backend import EmuBackend
    built = U.build("HYPELCNNModel", 5, 9, 4, ALG, EmuBackend(), with_eval=False)
    built.ctx.seed = 100 + rank          # different init per rank: the broadcast must make them equal
    sess = built.ctx.session()
    p0 = sess.params.clone()
    x, onehot = _data(rank)
    ct = U.run_train_step(built, x, onehot, {})
    local = sess.grads.clone()
    sess.allreduce_gradients()
    plain = sess.grads.clone()
    # the overlapped exchange (bucket at the plan's sync point + remaining head) must give the same gradients
    assert ct.sync_points, "a data-parallel plan carries a sync point for the early all-reduce"
    i_sync, lo, hi = ct.sync_points[0]
    assert 0 < lo < hi == max(v.offset + v.size for v in sess.trainable if len(v.shape) > 1)
    assert hi - lo >= 0.6 * sum(v.size for v in sess.trainable if len(v.shape) > 1)
    calls = []
    orig = dist.all_reduce
    dist.all_reduce = lambda t, **kw: (calls.append(t.numel()), orig(t, **kw))[1]
    sess.train_step_exchange(ct)
    dist.all_reduce = orig
    # [tail of the weights + the non-finite-loss flag behind them] first, then the head
    assert calls == [hi - lo + 1, lo], calls
    torch.testing.assert_close(sess.grads, plain, rtol=0, atol=0)
    # a sync point that does not reach the end of the buffer (a trailing variable no LinearNode owns): the
    # exchange must still cover every element
    ct.sync_points[0] = (i_sync, lo, hi - 10)
    calls.clear()
    dist.all_reduce = lambda t, **kw: (calls.append(t.numel()), orig(t, **kw))[1]
    sess.train_step_exchange(ct)
    dist.all_reduce = orig
    assert sum(calls) == sess.grads.numel() and len(calls) == 3, calls
    torch.testing.assert_close(sess.grads, plain, rtol=0, atol=0)
    ct.sync_points[0] = (i_sync, lo, hi)
    assert float(sess.grads[sess.n_train]) == 0.0 and sess.nonfinite_step(sync=True) is None
    # ragged global batch (5 samples over 2 ranks: 3 + 2): each rank weights its local mean-loss gradient by
    # nb_rank / nb_global instead of 1 / world
    nb_r = 3 - rank
    xr, ohr = x[:nb_r], onehot[:nb_r]
    ct_eq = sess.compile(built.train_tower, nb_r, loss=built.train_step.loss, external_masks=True)
    ct_rg = sess.compile(built.train_tower, nb_r, loss=built.train_step.loss, external_masks=True, global_nb=5)
    assert ct_eq is not ct_rg
    outs = []
    for c in (ct_eq, ct_rg):
        c.set_input("x", torch.as_tensor(xr))
        c.set_input("labels", torch.as_tensor(ohr))
        c.forward_backward()
        outs.append(sess.grads[:sess.n_train].clone())
    # (batch norm over 2-3 rows amplifies fp32 rounding: compare at 1e-5 of the largest gradient)
    torch.testing.assert_close(outs[1], outs[0] * ((nb_r / 5) / 0.5), rtol=1e-4,
                               atol=1e-5 * float(outs[0].abs().max()))
    U.run_train_step(built, x, onehot, {})
    sess.allreduce_gradients()
    sess.adam_step(3e-4)
    torch.save({"p0": p0, "local": local, "avg": sess.grads.clone(), "p1": sess.params.clone(),
                "state": sess.state.clone()}, os.path.join(outdir, f"r{rank}.pt"))
    sess.average_state()
    torch.save(sess.state.clone(), os.path.join(outdir, f"s{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gradient_exchange(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert torch.equal(r0["p0"], r1["p0"]), "rank-0 weights must be broadcast"
    assert not torch.equal(r0["local"], r1["local"]), "shards differ, so local gradients differ"
    assert torch.equal(r0["avg"], r1["avg"])
    # the 1/world of the global mean is applied at the loss (no rescaling pass after the all-reduce)
    torch.testing.assert_close(r0["avg"], r0["local"] + r1["local"], rtol=1e-6, atol=1e-9)
    assert torch.equal(r0["p1"], r1["p1"]), "parameters stay in lock-step"
    assert not torch.equal(r0["state"], r1["state"]), "BN moving statistics are per-rank until averaged"
    assert torch.equal(torch.load(tmp_path / "s0.pt"), torch.load(tmp_path / "s1.pt"))


ALG_DUAL = {"drop_out_ratio": 0.7, "filter_count": 96, "learning_rate": 3e-4, "learning_rate_decay_factor": 0.96,
            "learning_rate_decay_step": 350, "lrelu_alpha": 0.18, "optimizer": "AdamOptimizer", "bn_decay": 0.95,
            "l2regularizer_scale": 1e-5, "hs_lidar_diff": 1}


def _dual_worker(rank, world, port, outdir):
    """DUALCNN (the 1.03 GB model of BASELINE configs[2], here at toy width) with the bucket thresholds forced down:
    the backward pass carries several sync points, each all-reduce covers what no earlier one did, and the result
    equals the flat all-reduce bit for bit."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hypelcnn_amd import plan
    plan.DP_TWO_BUCKET_BYTES = 0
    plan.DP_BUCKET_BYTES = 64 << 10
    from tests import parity_util as U
    from tests.emu_backend import EmuBackend
    built = U.build("DUALCNNModel", 7, 9, 4, ALG_DUAL, EmuBackend(), with_eval=False)
    sess = built.ctx.session()
    rng = np.random.default_rng(5)
    x = rng.random((4, 7, 7, 9)).astype(np.float32)[rank::2]
    onehot = np.eye(4, dtype=np.float32)[rng.integers(0, 4, 4)][rank::2]
    ct = U.run_train_step(built, x, onehot, U.make_masks(built, 2, np.random.default_rng(9)))
    sess.allreduce_gradients()
    plain = sess.grads.clone()
    pts = ct.sync_points
    assert len(pts) >= 3, pts
    los = [lo for _, lo, _ in pts]
    assert los == sorted(los, reverse=True) and len(set(los)) == len(los), "each point finishes a longer tail"
    calls = []
    orig = dist.all_reduce
    dist.all_reduce = lambda t, **kw: (calls.append(t.numel()), orig(t, **kw))[1]
    sess.train_step_exchange(ct)
    dist.all_reduce = orig
    assert len(calls) == len(pts) + 1 and sum(calls) == sess.grads.numel(), (calls, sess.grads.numel())
    torch.testing.assert_close(sess.grads, plain, rtol=0, atol=0)
    torch.save({"g": sess.grads.clone(), "n_points": len(pts)}, os.path.join(outdir, f"d{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_dualcnn_bucketed_exchange(tmp_path):
    port = _free_port()
    mp.spawn(_dual_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    d0, d1 = torch.load(tmp_path / "d0.pt"), torch.load(tmp_path / "d1.pt")
    assert torch.equal(d0["g"], d1["g"]) and d0["n_points"] == d1["n_points"] >= 3


def _iter_worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hypelcnn_amd.common import common_nn_ops as cno
    from tests.emu_backend import EmuBackend
    data = np.arange(23 * 1 * 1 * 2, dtype=np.float32).reshape(23, 1, 1, 2)
    labels = np.arange(23) % 3
    it = cno.BatchIterator((1, 1, 2), 3, 4, True, 1, None)
    it.initializer(data, labels, EmuBackend())
    got = []
    while True:
        b = it.next_batch()
        if b is None:
            break
        got.append(b[0][:, 0, 0, 0].numpy().copy())
    # evaluation: sharded batches + one all-reduce of the confusion matrix
    conf = torch.zeros(9, dtype=torch.int32)
    ev = cno.BatchIterator((1, 1, 2), 3, 5, False, 1, None)
    ev.initializer(data, labels, EmuBackend())
    while True:
        b = ev.next_batch()
        if b is None:
            break
        for lab in b[2].tolist():
            conf[lab * 3 + lab] += 1
    dist.all_reduce(conf)
    # training iterator (its batches end in collectives): a tail shorter than the world size is dropped on EVERY
    # rank (17 samples, global batch 8: 8 + 8 + 1 -> two batches per rank); a longer tail is sharded unevenly and
    # reports the global count the loss gradient is weighted with
    tails = {}
    for n in (17, 19):
        tr = cno.BatchIterator((1, 1, 2), 3, 4, True, 1, None)
        tr.collective = True
        tr.initializer(data[:n], labels[:n], EmuBackend())
        announced = tr.batch_shapes()  # what a training loop captures before its first step
        sizes = []
        while True:
            b = tr.next_batch()
            if b is None:
                break
            sizes.append((int(b[0].shape[0]), tr.last_global_count))
        assert set(sizes) == set(announced), (sizes, announced)
        tails[n] = sizes
    torch.save({"batches": got, "conf": conf, "tails": tails}, os.path.join(outdir, f"it{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_iterator_shards_are_disjoint_and_complete(tmp_path):
    port = _free_port()
    mp.spawn(_iter_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "it0.pt", weights_only=False), torch.load(tmp_path / "it1.pt", weights_only=False)
    a = np.concatenate(r0["batches"]) if r0["batches"] else np.zeros(0)
    b = np.concatenate(r1["batches"]) if r1["batches"] else np.zeros(0)
    assert len(set(a.tolist()) & set(b.tolist())) == 0, "shards must be disjoint"
    assert sorted(a.tolist() + b.tolist()) == [2.0 * i for i in range(23)], "one epoch covers every sample once"
    assert all(len(x) == 4 for x in r0["batches"][:-1]), "per-rank batch size is the configured batch size"
    assert torch.equal(r0["conf"], r1["conf"]) and int(r0["conf"].sum()) == 23
    assert r0["tails"][17] == r1["tails"][17] == [(4, 8), (4, 8)], "a 1-sample tail is dropped on both ranks"
    assert r0["tails"][19] == [(4, 8), (4, 8), (2, 3)] and r1["tails"][19] == [(4, 8), (4, 8), (1, 3)]


def _syncbn_case():
    """Global batch of 8 (HYPELCNN toy), fixed parameters and dropout masks: what one device computes."""
    from tests import parity_util as U
    rng = np.random.default_rng(2024)
    params = U.make_params("HYPELCNNModel", 5, 9, 4, ALG, rng)
    x = rng.random((8, 5, 5, 9)).astype(np.float32)
    onehot = np.eye(4, dtype=np.float32)[rng.integers(0, 4, 8)]
    return params, x, onehot, rng


def _syncbn_worker(rank, world, port, outdir, split):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import parity_util as U
    from tests.emu_backend import EmuBackend
    built = U.build("HYPELCNNModel", 5, 9, 4, ALG, EmuBackend(), with_eval=False)
    sess = built.ctx.session()
    params, x, onehot, rng = _syncbn_case()
    U.inject(sess, params)
    masks = U.make_masks(built, 8, rng)
    lo, hi = (0, split) if rank == 0 else (split, 8)
    ct = sess.compile(built.train_tower, hi - lo, loss=built.train_step.loss, external_masks=True, global_nb=8,
                      sync_bn=True)
    assert ct.plan.sync_bn and not any(l.name.startswith("bn_act_small") for l in ct.plan.fwd + ct.plan.bwd)
    U.feed(ct, x[lo:hi], onehot[lo:hi], {k: m[lo:hi] for k, m in masks.items()})
    state0 = sess.state.clone()
    sess.train_step_exchange(ct)
    out = {"grads": sess.grads.clone(), "state": sess.state.clone(), "logits": ct.value(built.y_conv).clone(),
           "loss": ct.loss_value()}
    # the same shard WITHOUT synchronisation normalises with local statistics: a different result
    sess.state.copy_(state0)
    ct_l = sess.compile(built.train_tower, hi - lo, loss=built.train_step.loss, external_masks=True, global_nb=8,
                        sync_bn=False)
    assert ct_l is not ct
    U.feed(ct_l, x[lo:hi], onehot[lo:hi], {k: m[lo:hi] for k, m in masks.items()})
    sess.train_step_exchange(ct_l)
    out["grads_local_bn"] = sess.grads.clone()
    torch.save(out, os.path.join(outdir, f"sbn{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("split", [4, 5])   # equal shards, ragged shards (5 + 3)
def test_two_rank_sync_batch_norm_equals_one_device_at_the_global_batch(tmp_path, split):
    """SURVEY 8e (optional row): with synchronised batch norm two ranks holding 4+4 (or 5+3) samples compute what the
    reference's single device computes on all 8 -- logits, loss, gradients and the moving averages."""
    port = _free_port()
    mp.spawn(_syncbn_worker, args=(2, port, str(tmp_path), split), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "sbn0.pt"), torch.load(tmp_path / "sbn1.pt")
    sys.path.insert(0, ROOT)
    from tests import parity_util as U
    from tests.emu_backend import EmuBackend
    built = U.build("HYPELCNNModel", 5, 9, 4, ALG, EmuBackend(), with_eval=False)
    sess = built.ctx.session()
    assert sess.dist is None
    params, x, onehot, rng = _syncbn_case()
    U.inject(sess, params)
    masks = U.make_masks(built, 8, rng)
    ct = U.run_train_step(built, x, onehot, masks)
    g = sess.grads[:sess.n_train]
    scale = float(g.abs().max())
    assert torch.equal(r0["grads"], r1["grads"]) and torch.equal(r0["state"], r1["state"])
    torch.testing.assert_close(r0["grads"][:sess.n_train], g, rtol=1e-4, atol=2e-6 * scale)
    torch.testing.assert_close(r0["state"], sess.state, rtol=1e-5, atol=1e-6)
    logits = ct.value(built.y_conv)
    torch.testing.assert_close(torch.cat([r0["logits"], r1["logits"]]), logits, rtol=1e-4, atol=1e-5)
    assert abs((r0["loss"] * split + r1["loss"] * (8 - split)) / 8 - ct.loss_value()) < 1e-5
    # local statistics on 3-5 samples are a different model: the check above is not vacuous
    assert float((r0["grads_local_bn"][:sess.n_train] - g).abs().max()) > 1e-2 * scale


def _episode_worker(rank, world, port, outdir, sync_bn):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    if sync_bn:
        os.environ["HYPEL_SYNC_BN"] = "1"
    import json
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hypelcnn_amd.classify import train_for_classification as T
    from tests.emu_backend import EmuBackend
    alg = {"batch_size": 25, "drop_out_ratio": 0.3, "filter_count": 32, "learning_rate": 3e-3,
           "learning_rate_decay_factor": 0.96, "learning_rate_decay_step": 350, "lrelu_alpha": 0.18,
           "optimizer": "AdamOptimizer", "bn_decay": 0.9, "l2regularizer_scale": 1e-5, "spectral_hierarchy_level": 1,
           "spatial_hierarchy_level": 1, "degradation_coeff": 3, "use_residual": True}
    p = os.path.join(outdir, f"alg{rank}.json")
    with open(p, "w") as f:
        json.dump(alg, f)
    # epoch-limited run (11 epochs of 320 samples, global batch 2 x 25): the last batch of the run is short (40 samples,
    # 20 per rank), so a second training plan is compiled mid-run and its collectives must line up as well (uneven
    # shards: test_two_rank_sync_batch_norm_equals_one_device_at_the_global_batch, _worker)
    argv = ["--loader_name", "SyntheticDataLoader", "--path", "grss2013:h=23:w=29:bands=10:classes=3:samples=0.6",
            "--neighborhood", "1", "--model_name", "HYPELCNNModel", "--algorithm_param_path", p,
            "--batch_size", "25", "--epoch", "11", "--base_log_path", os.path.join(outdir, f"log{rank}"),
            "--perform_validation", "true", "--validation_steps", "30", "--save_checkpoint_steps", "50"]
    flags, _ = T.build_parser().parse_known_args(argv)
    model = T.get_model_from_name(flags.model_name)
    log_dir = os.path.join(flags.base_log_path, "dp")
    import hypelcnn_amd.plan as P
    plans = []
    init = P.TowerPlan.__init__

    def recording_init(self, *a, **kw):
        init(self, *a, **kw)
        plans.append((self.training, self.nb, self.global_nb, self.sync_bn, self.world))

    P.TowerPlan.__init__ = recording_init
    res = T.perform_an_episode(flags, alg, model, log_dir, backend=EmuBackend())
    import torch
    torch.save({"loss": float(res.loss), "test": float(res.test_accuracy), "val": float(res.validation_accuracy),
                "plans": plans,
                "files": sorted(os.listdir(log_dir)) if os.path.isdir(log_dir) else []},
               os.path.join(outdir, f"ep{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("sync_bn", [False, True])
def test_two_rank_training_episode_runs_in_lock_step(tmp_path, sync_bn):
    """The reference-shaped entry point (flags -> loader -> importer -> graph -> monitored session with validation and
    checkpoints) under two gloo ranks: every collective of the loop -- iterator tail decisions, gradient buckets with
    the non-finite flag, confusion-matrix sums, and with HYPEL_SYNC_BN=1 one all-gather / all-reduce per batch-norm
    layer and direction -- has to line up on both ranks or the run dead-locks (the timeout fails the test)."""
    port = _free_port()
    mp.spawn(_episode_worker, args=(2, port, str(tmp_path), sync_bn), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "ep0.pt"), torch.load(tmp_path / "ep1.pt")
    assert np.isfinite(r0["loss"]) and np.isfinite(r1["loss"])
    # the evaluation confusion matrices are summed over the ranks: both report the same accuracies
    assert r0["test"] == r1["test"] and r0["val"] == r1["val"]
    assert r0["test"] > 0.8 and r0["val"] > 0.8, r0
    train0 = [p for p in r0["plans"] if p[0]]
    assert train0 and all(p[4] == 2 and p[3] == sync_bn for p in train0), r0["plans"]
    assert sorted(p[1] for p in train0) == [20, 25], train0  # the full batch and the short last one


def _gan_setup(n, bands, backend):
    from oracle import gan as OG
    from tests import gan_util as GU
    cfg = OG.GanConfig("cycle_gan", bands, max_steps=20, generator_lr=2e-3, discriminator_lr=1e-3)
    params = GU.fp32(OG.init_gan_params("cycle_gan", bands, np.random.default_rng(2), dtype=np.float64, zero_generator=False))
    wrapper, model, loss, ops = GU.build(cfg, n, backend)
    ops.use_pool = False  # the tensor pool draws per rank; the exchange is what is under test
    sess = ops.ctx.session()
    GU.inject(sess, params)
    return ops, sess


def _gan_batch(bands):
    rng = np.random.default_rng(11)
    return (torch.as_tensor(rng.random((8, bands)).astype(np.float32)),
            torch.as_tensor((rng.random((8, bands)) * 0.5).astype(np.float32)))


def _gan_worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.emu_backend import EmuBackend
    bands = 16
    ops, sess = _gan_setup(4, bands, EmuBackend())
    x, y = _gan_batch(bands)
    for _ in range(2):
        ops.run_step(x[rank::2].contiguous(), y[rank::2].contiguous())
    names = set()
    for ct in ops.last_losses.values():
        names |= {l.name for l in ct.plan.fwd + ct.plan.bwd}
    torch.save({"p": sess.params.clone(), "names": sorted(names), "step": sess.global_step}, os.path.join(outdir, f"g{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_cyclegan_steps_equal_one_device_at_the_global_batch(tmp_path):
    """The GAN train ops under data parallelism (gan_common.py: per-phase all-reduce of the phase's variable groups): two
    ranks on disjoint halves of 8 pairs take the same two CycleGAN steps as one device on all 8 (every loss is a batch
    mean, nothing in the CycleGAN stacks couples samples) -- with the fused generator and the fused discriminator stack."""
    port = _free_port()
    mp.spawn(_gan_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "g0.pt"), torch.load(tmp_path / "g1.pt")
    assert torch.equal(r0["p"], r1["p"]) and r0["step"] == 2, "ranks stay in lock step"
    fused = [n for n in r0["names"] if n.startswith(("dense_stack_bwd", "gan_generator_bwd"))]  # (plain or *_apps forms)
    assert any(n.startswith("dense_stack_bwd") for n in fused) and any(n.startswith("gan_generator_bwd") for n in fused), \
        r0["names"]
    sys.path.insert(0, ROOT)
    from tests.emu_backend import EmuBackend
    ops, sess = _gan_setup(8, 16, EmuBackend())
    x, y = _gan_batch(16)
    for _ in range(2):
        ops.run_step(x, y)
    scale = float(sess.params.abs().max())
    torch.testing.assert_close(r0["p"], sess.params, rtol=2e-4, atol=2e-6 * scale)
