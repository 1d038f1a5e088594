"""Session: parameter storage, compiled towers, optimiser step, data-parallel gradient exchange.

The TF1 reference runs `session.run(train_step)` in a loop (classify/monitored_session_runner.py:
182-184).  Here a "session" owns
  * one flat fp32 buffer per kind (parameters, gradients, optimiser slots, BN moving statistics) so
    that the optimiser is ONE launch and the data-parallel exchange is ONE RCCL all-reduce;
  * per (tower, batch size) a `CompiledTower`: pre-bound C-ABI launches, optionally captured in a
    HIP graph and replayed.
"""
import math

import numpy as np
import torch

from . import graph as G
from .backend import Ref, note_collective
from .plan import TowerPlan
from .plan_gan import PhasePlan


class CompiledTower:
    def __init__(self, plan, backend):
        self.plan = plan
        self.be = backend
        self.fwd = [backend.bind(l.name, l.args) for l in plan.fwd]
        self.bwd = [backend.bind(l.name, l.args) for l in plan.bwd]
        self._graph_fwd = None
        self._graph_all = None
        self._segments = None  # graph replay callables of [fwd + bwd up to sync point 0], [.. sync point 1], ...
        self.sync_points = list(getattr(plan, "sync_points", []))

    def serial_launches(self):
        """Every launch of the step, bound one by one (eager replay for per-kernel timing)."""
        return [(l, self.be.bind(l.name, l.args, 0)) for l in self.plan.fwd + self.plan.bwd]

    # ---- inputs / outputs ----
    def input(self, name):
        return self.plan.buffers["in:" + name]

    def set_input(self, name, tensor):
        dst = self.plan.buffers["in:" + name]
        dst.copy_(tensor.reshape(-1).to(dst.dtype), non_blocking=True)

    def set_inputs(self, items):
        """set_input for several inputs; two fp32 tensors that already live on the device go in ONE launch
        (hypel_copy_pair_f32): a GAN train op is fed with two batches."""
        items = list(items)
        while len(items) >= 2:
            (n0, t0), (n1, t1) = items[0], items[1]
            d0, d1 = self.plan.buffers["in:" + n0], self.plan.buffers["in:" + n1]
            def same_dev(a, b):  # torch.device("cuda") != torch.device("cuda:0"): compare type and resolved index
                return a.type == b.type and (a.index if a.index is not None else torch.cuda.current_device() if a.type == "cuda" else 0) == \
                    (b.index if b.index is not None else torch.cuda.current_device() if b.type == "cuda" else 0)
            ok = all(t.dtype == d.dtype == torch.float32 and same_dev(t.device, d.device) and t.is_contiguous() and
                     t.numel() == d.numel() for t, d in ((t0, d0), (t1, d1)))
            if not ok:
                break
            self.be.call("copy_pair_f32", Ref(d0), Ref(t0.reshape(-1)), d0.numel(), Ref(d1), Ref(t1.reshape(-1)),
                              d1.numel())
            items = items[2:]
        for name, t in items:
            self.set_input(name, t)

    def value(self, sym, nhwc=True, copy=True):
        """Fetch a tensor of the tower as [N, H, W, C] / [N, C].  copy=True (default): the caller owns the result.
        copy=False: an [N, C] tensor comes back as a VIEW of the plan buffer (no gather / permute / copy kernels) that the
        next forward() / forward_backward() at this batch size OVERWRITES -- for callers that consume it before then
        (the GAN tensor pool's pass-through fetches two such tensors per step)."""
        st = self.plan.storage_of(sym)
        buf = self.plan.buffers[st.buf]
        nb = self.plan.nb
        own = self.plan.storage[id(sym.owner)]
        if sym.hw is None and st.pixmap is None and sym.owner.npix == 1:
            # (an application of a row-concatenated batch is a row block of the batch's buffer: its offset holds whole rows)
            row0, col = divmod(st.ch_off, own.ld)
            v = buf[: (row0 + nb) * own.ld].view(row0 + nb, own.ld)[row0:, col:col + st.c]
            return v.clone() if copy else v
        full = buf[: sym.owner.npix * nb * own.ld].reshape(sym.owner.npix, nb, own.ld)
        pm = list(range(sym.npix)) if st.pixmap is None else st.pixmap
        v = full[pm][:, :, st.ch_off:st.ch_off + st.c]
        v = v.permute(1, 0, 2)
        if sym.hw is None:
            return v.reshape(nb, st.c)
        return v.reshape(nb, sym.hw[0], sym.hw[1], st.c)

    def grad_value(self, sym):
        st = self.plan.grad_storage_of(sym)
        buf = self.plan.buffers[st.buf]
        nb = self.plan.nb
        own = self.plan.storage[id(sym.owner)]
        if sym.hw is None and st.pixmap is None and sym.owner.npix == 1:
            row0, col = divmod(st.ch_off, own.ld)
            return buf[: (row0 + nb) * own.ld].view(row0 + nb, own.ld)[row0:, col:col + st.c].clone()
        full = buf[: sym.owner.npix * nb * own.ld].reshape(sym.owner.npix, nb, own.ld)
        pm = list(range(sym.npix)) if st.pixmap is None else st.pixmap
        v = full[pm][:, :, st.ch_off:st.ch_off + st.c].permute(1, 0, 2)
        return v.reshape(nb, st.c) if sym.hw is None else v.reshape(nb, sym.hw[0], sym.hw[1], st.c)

    def dropout_mask(self, index):
        return self.plan.buffers[self.plan.mask_bufs[index]]

    # ---- execution ----
    def _items(self, with_bwd):
        """The step as a list of ("run", [launches]) | ("host", call) | ("hook", k): runs are what a HIP graph can hold;
        host calls (collectives of synchronised batch norm) and the data-parallel sync points sit between them."""
        cuts = {i: k for k, (i, _, _) in enumerate(self.sync_points)} if with_bwd else {}
        items, cur = [], []

        def flush():
            if cur:
                items.append(("run", list(cur)))
                cur.clear()

        def add(f):
            if getattr(f, "host", False):
                flush()
                items.append(("host", f))
            else:
                cur.append(f)

        for f in self.fwd:
            add(f)
        if with_bwd:
            for j, f in enumerate(self.bwd):
                if j in cuts:
                    flush()
                    items.append(("hook", cuts[j]))
                add(f)
        flush()
        return items

    @staticmethod
    def _run_items(items, hook=None):
        for kind, v in items:
            if kind == "run":
                if callable(v):
                    v()
                else:
                    for f in v:
                        f()
            elif kind == "host":
                v()
            elif hook is not None:
                hook(v)

    def forward(self):
        if self._graph_fwd is not None:
            self._graph_fwd()
        else:
            self._run_items(self._items(False))

    def forward_backward(self, hook=None):
        """hook(k): called on the host right after the launches up to sync point k were issued (data-parallel
        overlap: the session starts the all-reduce of the gradient range that is final at that point)."""
        if self._segments is not None:
            self._run_items(self._segments, hook)
        elif self._graph_all is not None:
            self._graph_all()
        else:
            self._run_items(self._items(True), hook)

    def capture(self):
        """Capture forward (+ backward) into HIP graphs: one host call per step instead of ~300.  A step with sync
        points or host-side collectives becomes a chain of graphs with those calls in between."""
        # warm run (first-use allocations / lazy module loads) on whatever the input buffers hold: it must not move the
        # trained state -- batch-norm moving averages, the dropout step counter, the non-finite flag -- away from a run
        # without capture, so those are put back afterwards
        sess = self.plan.sess
        keep = [(t, t.clone()) for t in (getattr(sess, "state", None), self.plan.buffers.get("step_ctr")) if t is not None]
        if getattr(sess, "grads", None) is not None and getattr(sess, "n_train", None) is not None:
            flag = sess.grads[sess.n_train:sess.n_train + 1]
            keep.append((flag, flag.clone()))
        self.forward_backward() if self.bwd else self.forward()
        self.be.synchronize()
        for t, saved in keep:
            t.copy_(saved)
        items = self._items(bool(self.bwd))
        settle = getattr(self.be, "settle_before_capture", None)
        if settle is not None:
            settle()  # once for the whole chain of segments: no collective is issued between their captures
        graphs = [(kind, self.be.capture(v, True) if kind == "run" else v) for kind, v in items]
        if len(graphs) == 1 and graphs[0][0] == "run":
            if self.bwd:
                self._graph_all = graphs[0][1]
            else:
                self._graph_fwd = graphs[0][1]
        elif self.bwd:
            self._segments = graphs
            self._graph_all = graphs[0][1]  # "captured" marker for callers
        else:
            self._graph_fwd = lambda: self._run_items(graphs)

    def loss_value(self):
        b = self.plan.buffers
        if "loss" in b and "loss_ce" not in b:
            return float(b["loss"][0])
        v = float(b["loss_ce"][0])
        if self.plan.loss is not None and self.plan.loss.per_sample.extra_mse is not None:
            v += float(b["loss_mse"][0])
        return v

    def flops(self):
        return sum(l.flops for l in self.plan.fwd), sum(l.flops for l in self.plan.bwd)


class Session:
    def __init__(self, store, backend, seed=1234):
        self.store = store
        self.backend = backend
        self.seed = seed
        self.params = self.grads = self.state = None
        self.slot_m = self.slot_v = None
        self.trainable = []
        self.stateful = []
        self.global_step = 0
        self._compiled = {}
        self.shared_inputs = {}  # (name, rows, c) -> device buffer shared by every GAN phase plan
        self.dist = None  # (world_size, rank) once init_data_parallel() ran
        self.sync_bn = False

    # ---- variables ----
    def finalize_variables(self, rng=None, group_affinity=()):
        """Lay variables out in the flat buffers (per optimiser group: per-channel vectors first, then the weights,
        each in creation order so that the vectors of a merged level are contiguous and the weights of the layers
        that finish their backward first form one contiguous tail) and initialise them.
        group_affinity: lists of group names that one train op updates together (a GAN phase's generators, its critics):
        they are laid out next to each other, so that their optimiser update and their gradient all-reduce are one
        launch / one collective per train op (adam_step_groups, allreduce_group_gradients merge adjacent ranges)."""
        order = self.store.order
        present = []
        for v in order:
            if v.trainable and v.group not in present:
                present.append(v.group)
        groups = []
        for together in group_affinity:
            groups += [g for g in together if g in present and g not in groups]
        groups += [g for g in present if g not in groups]
        self.trainable = []
        self.group_ranges = {}
        off = 0
        for gname in groups:  # one contiguous range per optimiser group (GAN: generator / discriminator / ...)
            lo = off
            members = [v for v in order if v.trainable and v.group == gname]
            # creation order, except where the graph builder merged sibling layers and asked for their variables to
            # sit next to each other (Variable.order_key)
            members.sort(key=lambda v: v.order_key if v.order_key is not None else (order.index(v),))
            for v in [m for m in members if len(m.shape) == 1] + [m for m in members if len(m.shape) > 1]:
                v.offset = off
                off += v.size
                self.trainable.append(v)
            self.group_ranges[gname] = (lo, off)
        n_train = off
        mm = [v for v in order if not v.trainable and v.name.endswith("moving_mean")]
        mv = [v for v in order if not v.trainable and v.name.endswith("moving_variance")]
        other = [v for v in order if not v.trainable and v not in mm and v not in mv]
        self.stateful = mm + mv + other
        off = 0
        for v in self.stateful:
            v.offset = off
            off += v.size
        be = self.backend
        self.params = be.zeros(n_train)
        # one extra element BEHIND the gradients: the non-finite-loss flag of the step (hypel_loss_guard_f32).  It
        # rides in the gradient all-reduce, so every data-parallel rank sees the same verdict and skips the update
        self.grads = be.zeros(n_train + 1)
        self.n_train = n_train
        self.slot_m = be.zeros(n_train)
        self.slot_v = be.zeros(n_train)
        self.state = be.zeros(max(off, 1))
        rng = rng or np.random.default_rng(self.seed)
        host_p = np.zeros(n_train, np.float32)
        for v in self.trainable:
            host_p[v.offset:v.offset + v.size] = v.init(rng, v.shape).reshape(-1)
        host_s = np.zeros(max(off, 1), np.float32)
        for v in self.stateful:
            host_s[v.offset:v.offset + v.size] = v.init(rng, v.shape).reshape(-1)
        self.params.copy_(torch.from_numpy(host_p))
        self.state.copy_(torch.from_numpy(host_s))
        return n_train

    def _lookup(self, name):
        v = self.store.vars.get(name) or self.store.vars.get(f"{self.store.prefix}/{name}")
        if v is None:
            raise KeyError(name)
        return v

    def _buf_of(self, v):
        return self.params if v.trainable else self.state

    def get_variable(self, name):
        v = self._lookup(name)
        return self._buf_of(v)[v.offset:v.offset + v.size].detach().cpu().numpy().reshape(v.shape).copy()

    def set_variable(self, name, value):
        v = self._lookup(name)
        arr = np.ascontiguousarray(value, np.float32).reshape(-1)
        assert arr.size == v.size, (name, arr.size, v.size)
        self._buf_of(v)[v.offset:v.offset + v.size].copy_(torch.from_numpy(arr))

    def get_gradient(self, name):
        v = self._lookup(name)
        return self.grads[v.offset:v.offset + v.size].detach().cpu().numpy().reshape(v.shape).copy()

    def variable_names(self):
        return [v.name for v in self.store.order]

    def state_dict(self):
        """Checkpoint payload keyed by the TF variable names (monitored_session_runner.py:164-168)."""
        d = {v.name: self.get_variable(v.name) for v in self.store.order}
        d["global_step"] = np.asarray(self.global_step, np.int64)
        d["training_optimizer/m"] = self.slot_m.detach().cpu().numpy()
        d["training_optimizer/v"] = self.slot_v.detach().cpu().numpy()
        return d

    def load_state_dict(self, d):
        for v in self.store.order:
            if v.name in d:
                self.set_variable(v.name, d[v.name])
        if "global_step" in d:
            self.global_step = int(d["global_step"])
        if "training_optimizer/m" in d and d["training_optimizer/m"].size == self.slot_m.numel():
            self.slot_m.copy_(torch.from_numpy(np.asarray(d["training_optimizer/m"], np.float32)))
            self.slot_v.copy_(torch.from_numpy(np.asarray(d["training_optimizer/v"], np.float32)))

    # ---- towers ----
    def compile(self, tower, nb, loss=None, external_masks=False, global_nb=None, sync_bn=None):
        """global_nb: size of the global batch `nb` is this rank's shard of (data parallel; None = nb x world).
        sync_bn: batch-norm statistics over the global batch (None = the session's `sync_bn`, see init_data_parallel)."""
        if self.dist is None or (global_nb is not None and int(global_nb) == int(nb) * self.dist[0]):
            global_nb = None
        sync_bn = bool(self.sync_bn if sync_bn is None else sync_bn) and self.dist is not None
        key = (id(tower), int(nb), id(loss), external_masks, global_nb, sync_bn)
        ct = self._compiled.get(key)
        if ct is None:
            plan = TowerPlan(tower, nb, self, loss=loss, external_masks=external_masks, seed=self._rank_seed(),
                             global_nb=global_nb, sync_bn=sync_bn)
            ct = CompiledTower(plan, self.backend)
            self._compiled[key] = ct
        return ct

    def _rank_seed(self):
        """Philox key of the dropout masks: per rank under data parallelism (different samples, different masks)."""
        return self.seed + (1000003 * self.dist[1] if self.dist is not None else 0)

    def compile_phase(self, tower, nb, terms=(), train_groups=(), outputs=(), key=None):
        k = ("phase", id(tower), int(nb), key)
        ct = self._compiled.get(k)
        if ct is None:
            plan = PhasePlan(tower, nb, self, terms=terms, train_groups=train_groups, outputs=outputs,
                             seed=self._rank_seed())
            ct = CompiledTower(plan, self.backend)
            self._compiled[k] = ct
        return ct

    def adam_step_groups(self, groups, lr, t, beta1=0.5, beta2=0.999, eps=1e-8):
        """TF1 Adam on the flat ranges of the given optimiser groups (gan_common.py:264-265: beta1 = 0.5)."""
        lr_t = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
        for lo, hi in self._merged_ranges(groups):  # (the groups of one train op are neighbours: finalize_variables)
            self.backend.call("adam_tf1", Ref(self.params, lo), Ref(self.grads, lo), Ref(self.slot_m, lo),
                              Ref(self.slot_v, lo), hi - lo, float(lr_t), float(beta1), float(beta2), float(eps))

    def _merged_ranges(self, groups):
        """the flat ranges of the given groups, adjacent ones merged"""
        ranges = []
        for lo, hi in sorted(self.group_ranges[g] for g in groups):
            if hi > lo:
                if ranges and ranges[-1][1] == lo:
                    ranges[-1][1] = hi
                else:
                    ranges.append([lo, hi])
        return ranges

    def allreduce_group_gradients(self, groups):
        if self.dist is None:
            return
        import torch.distributed as dist
        # adjacent groups travel in one call.  ONE averaging path on every backend -- SUM, then scale by 1 / world -- so that
        # the 2-rank gloo tests (tests/test_dp_gloo.py) exercise exactly what runs over RCCL (round 3 averaged inside the
        # collective on RCCL only: two code paths, one of them never multi-rank-tested)
        for lo, hi in self._merged_ranges(groups):
            view = self.grads[lo:hi]
            note_collective()
            dist.all_reduce(view, op=dist.ReduceOp.SUM)
            view.mul_(1.0 / self.dist[0])

    # ---- optimiser (common_nn_ops.py:223-230) ----
    def guard_ref(self):
        """Device address of the step's non-finite-loss flag (the element behind the flat gradient buffer)."""
        return Ref(self.grads, self.n_train)

    def adam_step(self, lr, beta1=0.9, beta2=0.999, eps=1e-8):
        t = self.global_step + 1
        lr_t = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
        self.backend.call("adam_tf1_guarded", Ref(self.params), Ref(self.grads), Ref(self.slot_m), Ref(self.slot_v),
                          self.params.numel(), float(lr_t), float(beta1), float(beta2), float(eps), self.guard_ref())
        self.global_step += 1
        self._post_guard()

    def momentum_step(self, lr, mu):
        self.optimizer_kind = "momentum"  # checkpoint export: <var>/nn_core/Momentum slots, no beta powers
        self.backend.call("momentum_tf1_guarded", Ref(self.params), Ref(self.grads), Ref(self.slot_m),
                          self.params.numel(), float(lr), float(mu), self.guard_ref())
        self.global_step += 1
        self._post_guard()

    # ---- non-finite loss guard (NanTensorHook, monitored_session_runner.py:151; check_numerics, common_nn_ops.py:232)
    _GUARD_SLOTS = 4  # ring of pinned flag copies: the host looks at most this many optimiser steps behind the device

    def _post_guard(self):
        """After every optimiser launch: copy the step's flag into pinned host memory, asynchronously, and remember
        an event.  `nonfinite_step()` later looks at copies that are at least one step old, so the loop never drains
        the device queue to learn about a NaN -- and it does not have to: the guarded optimiser has already refused
        the update on the device.  The copies live in a fixed ring of pinned slots and events (allocated once): a
        caller that never polls (bench.py, library users of adam_step) pays O(1) per step -- the oldest copy is folded
        into the verdict when the ring is full -- instead of one hipHostMalloc per step."""
        if not hasattr(self, "_guard_q"):
            self._guard_q = []
            self._guard_bad = None
            self._guard_free = None
        if self.grads.device.type != "cuda":
            self._guard_q.append((self.global_step, float(self.grads[self.n_train]), None))
            if len(self._guard_q) > self._GUARD_SLOTS:
                self._guard_fold(self._guard_q.pop(0))
            return
        if self._guard_free is None:
            pinned = torch.empty(self._GUARD_SLOTS, dtype=torch.float32, pin_memory=True)
            self._guard_free = [(pinned[i:i + 1], torch.cuda.Event()) for i in range(self._GUARD_SLOTS)]
        if not self._guard_free:  # ring full: the oldest copy is several steps old, its event long since signalled
            self._guard_fold(self._guard_q.pop(0))
        host, ev = self._guard_free.pop()
        # the library's own copy kernel stores the flag straight into the pinned (device-mapped) slot: no runtime blit
        # kernel (__amd_rocclr_copyBuffer) in the step
        self.backend.call("copy_pair_f32", Ref(host), Ref(self.grads, self.n_train), 1, Ref(host), Ref(self.grads, self.n_train), 0)
        ev.record()
        self._guard_q.append((self.global_step, host, ev))

    def _guard_fold(self, entry):
        """Read one queued flag copy (waiting for its event), return its slot to the ring, record a bad step."""
        step, host, ev = entry
        if ev is not None:
            ev.synchronize()
            value = float(host[0])
            self._guard_free.append((host, ev))
        else:
            value = host
        if value != 0.0 and self._guard_bad is None:
            self._guard_bad = step
        return value != 0.0

    def nonfinite_step(self, sync=False):
        """global_step value after the first step whose loss was NaN/Inf, or None.  Without `sync` only flag copies
        older than the newest one are inspected (the host waits at most for the step before the one in flight; under
        data parallelism every rank therefore decides at the same iteration, on the same all-reduced flag)."""
        if getattr(self, "_guard_bad", None) is not None:
            return self._guard_bad
        q = getattr(self, "_guard_q", [])
        keep = 0 if sync else 1
        while len(q) > keep:
            if self._guard_fold(q.pop(0)):
                while q:  # later copies are moot: hand their slots back
                    self._guard_fold(q.pop(0))
                return self._guard_bad
        return None

    # ---- data parallel (new vs the reference: SURVEY §2.3 / §8e) ----
    def init_data_parallel(self, broadcast=True, sync_bn=None):
        """sync_bn (None = environment HYPEL_SYNC_BN=1): training towers compiled from now on normalise with the
        statistics of the GLOBAL batch, which makes N ranks x nb samples equal one device at N x nb samples (the
        reference's single-device semantics); costs one small collective per BN layer and direction."""
        import torch.distributed as dist
        import os
        self.sync_bn = (os.environ.get("HYPEL_SYNC_BN") == "1") if sync_bn is None else bool(sync_bn)
        selftest = os.environ.get("HYPEL_DP_SELFTEST") == "1"  # keep the collectives on a 1-rank communicator
        if not dist.is_initialized() or (dist.get_world_size() == 1 and not selftest):
            self.dist = None
            return
        self.dist = (dist.get_world_size(), dist.get_rank())
        if broadcast:
            note_collective()
            dist.broadcast(self.params, src=0)
            dist.broadcast(self.state, src=0)

    def allreduce_gradients(self):
        """One flat all-reduce (RCCL over xGMI on the GPU box, gloo in the CPU tests), then average."""
        if self.dist is None:
            return
        import torch.distributed as dist
        # the plan scales the loss gradient by 1/world at its source (TowerPlan._emit_loss), so the SUM is the mean
        note_collective()
        dist.all_reduce(self.grads, op=dist.ReduceOp.SUM)

    def train_step_exchange(self, ct, hook_ranges=True):
        """forward + backward with the gradient exchange overlapped: at each sync point of the compiled tower the
        finished tail of the flat gradient buffer goes out as an asynchronous all-reduce (RCCL runs it on its own
        stream under the rest of the backward pass); the remaining head follows after the last launch.
        Equivalent to forward_backward() + allreduce_gradients()."""
        if self.dist is None or not ct.sync_points:
            ct.forward_backward()
            self.allreduce_gradients()
            return
        import torch.distributed as dist
        works, covered = [], []  # reduced ranges [lo, hi) of the flat buffer (incl. the flag element at its end)
        total = self.grads.numel()

        def reduce_range(lo, hi):
            for clo, chi in covered:  # clip against what already went out
                if lo >= clo and hi <= chi:
                    return
                if lo < chi and hi > clo:
                    if lo < clo:
                        reduce_range(lo, clo)
                    if hi > chi:
                        reduce_range(chi, hi)
                    return
            if hi > lo:
                note_collective()
                works.append(dist.all_reduce(self.grads[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
                covered.append((lo, hi))

        def hook(k):
            _, lo, hi = ct.sync_points[k]
            if hi == self.n_train:  # the flag behind the gradients was written by the forward pass: final as well
                hi = total
            reduce_range(lo, hi)

        ct.forward_backward(hook=hook)
        reduce_range(0, total)  # whatever no sync point covered: the head, and any range behind the last LinearNode
        for w in works:
            w.wait()

    def average_state(self):
        """BN moving statistics are per-rank (local batch statistics); average them before
        a checkpoint / evaluation so every rank holds the same model."""
        if self.dist is None:
            return
        import torch.distributed as dist
        note_collective()
        dist.all_reduce(self.state, op=dist.ReduceOp.SUM)
        self.state.mul_(1.0 / self.dist[0])
