"""GAN train ops on the planner: `PhasePlan` lowers ONE tfgan train op (a set of loss terms minimised w.r.t. some variable
groups: gan/wrappers/*.py, gan_train_for_shadow.py:141-142) to a launch list -- which applications of which network run
as one row-concatenated or two-variable-set launch (`_schedule_units`), the fused generator / dense-stack nodes, the
encoder tap, loss slots, and ONE slab reduction + ONE optimiser launch per train op.  The per-node forward / backward
emission is inherited from plan.TowerPlan."""
import numpy as np

from . import graph as G
from .backend import REDUCE_ENTRY_DTYPE, Ref
from .gemm_tables import GemmTables, Launch
from .plan import GEN_KEEP, LOSS_SLOTS, Storage, TowerPlan, stat_chunk_rows


def gen_kernel_sizes(bands):
    return [bands, bands // 2, bands // 4, bands // 8, bands // 4, bands // 2, bands]


# Same-weight applications of one phase as ONE application on the row-concatenated batch (no GAN network has batch
# statistics, so D([real; fake]), enc([G(x); x; y; G(y)]), G([x; y]) and the feature-discriminator layers on 4N rows are
# exact; cut_wrapper.py:301-339): fewer, longer launches, >= 2 resident blocks per CU for the generator kernels.
BATCH_APPS = True
BATCH_APPS_MAX = 8
# the per-block gradient slabs of every fused generator / dense-stack application of a train op in ONE reduction launch
SLAB_REDUCE_MULTI = True
# An encoder-only generator application on a tensor that the FULL generator of the same train op (same variables) also
# consumes is that application's n_4 (cut_wrapper.py:301-339: gen(x) and gen(x, only_encoder), gen(y) and gen(y, only_encoder)):
# the full launch writes it too (hypel_gan_generator_fwd_tap) and its backward takes the gradient that reached it
# (hypel_gan_generator_bwd_tap) -- the encoder-only launches of those tensors disappear.
GEN_TAP = True
# Two same-shaped networks with DIFFERENT variables (CycleGAN's G_x2y / G_y2x and D_x / D_y, cycle_gan_wrapper.py:82-124) whose
# applications can run side by side share one launch: each takes its own share of the blocks (hypel_*_apps).  At the
# Gulfport sizes every one of those applications is a latency chain on a fraction of the chip.
BATCH_HETERO = True


class PhasePlan(TowerPlan):
    """One train op of a GAN step (tfgan RunTrainOpsHook = one session.run): the sub-graph that the phase's loss
    terms depend on, differentiated w.r.t. the variable groups the phase trains.  `outputs` (no loss terms) gives a
    forward-only plan, e.g. "generate fresh fake data for the tensor pool"."""

    def nominal_batch(self):
        from . import plan as _P
        return _P.SPLIT_NOMINAL_BATCH_GAN

    def __init__(self, tower, nb, session, terms=(), train_groups=(), outputs=(), seed=1234):
        self.terms = list(terms)
        self.train_groups = set(train_groups)
        self.outputs = list(outputs)
        self._defer_bias_sums = SLAB_REDUCE_MULTI
        super().__init__(tower, nb, session, loss=None, external_masks=True, seed=seed)

    # ---- analysis ----
    def _trains(self, variables):
        return all(v.group in self.train_groups for v in variables)

    def _needs_grad(self, t):
        return id(t.owner) in self._grad_needed

    def _node_inputs(self, node):
        if isinstance(node, G.LinearNode):
            return list(node.sources) + [r for r, _ in node.residuals]
        if isinstance(node, G.PostNode):
            return [node.src] + [r for r, _ in node.residuals]
        if isinstance(node, G.FeatStackNode):
            return list(node.srcs)
        return [node.src]

    def _node_vars(self, node):
        if isinstance(node, G.LinearNode):
            out = []
            for b in node.branches:
                out.append(b.w)
                if b.bias is not None:
                    out.append(b.bias)
            return out
        if isinstance(node, (G.GeneratorNode, G.DenseStackNode)):
            return list(node.weights) + list(node.biases)
        return []

    def _analyse(self):
        roots = []
        for t in self.terms:
            roots += [t.a] + ([t.b] if t.b is not None else [])
        roots += self.outputs
        needed = set()
        stack = [r.owner for r in roots]
        while stack:
            t = stack.pop()
            if t.node is None or id(t.node) in needed:
                continue
            needed.add(id(t.node))
            stack += [i.owner for i in self._node_inputs(t.node)]
        self.needed = [n for n in self.tower.nodes if id(n) in needed]
        self._taps, self._tapped = {}, {}  # id(full generator node) -> the encoder output it also produces; id(enc node) -> full node
        # (only with the dependency-driven schedule of _schedule_units: in tower order an encoder application may precede the
        # full one it would be read from)
        if GEN_TAP and BATCH_APPS and hasattr(self.be, "gan_generator_tap_supported"):
            fulls = {(id(n.weights[0]), id(n.src)): n for n in self.needed
                     if isinstance(n, G.GeneratorNode) and not n.only_encoder}
            for e in self.needed:
                if isinstance(e, G.GeneratorNode) and e.only_encoder:
                    f = fulls.get((id(e.weights[0]), id(e.src)))
                    if f is not None and id(f) not in self._taps and self.be.gan_generator_tap_supported(e.src.c):
                        self._taps[id(f)] = e.out
                        self._tapped[id(e)] = f
        self._grad_needed = set()
        for n in self.needed:
            vs = self._node_vars(n)
            own = bool(vs) and any(v.group in self.train_groups for v in vs)
            if own or any(id(i.owner) in self._grad_needed for i in self._node_inputs(n)):
                self._grad_needed.add(id(n.out))

    # ---- build ----
    def _build(self):
        self._analyse()
        nb = self.nb
        used_inputs = set()
        for n in self.needed:
            for i in self._node_inputs(n):
                if i.owner.node is None:
                    used_inputs.add(id(i.owner))
        for t in self.terms:
            for x in (t.a, t.b):
                if x is not None and x.owner.node is None:
                    used_inputs.add(id(x.owner))
        for name, t in self.tower.inputs.items():
            if id(t) not in used_inputs:
                continue
            assert t.hw is None, "GAN inputs are [N, B]"
            # the phases of one GAN step read the same batch: they share one device buffer per input, fed once per step.
            # All [N, c] inputs of the tower are row blocks of ONE slab, in the order the wrapper asks for
            # (tower.input_layout; default: creation order): applications that run as one row-concatenated batch on
            # neighbouring inputs then need no gather launch (_concat_inputs)
            layout = [n_ for n_ in (getattr(self.tower, "input_layout", None) or list(self.tower.inputs))
                      if self.tower.inputs[n_].hw is None and self.tower.inputs[n_].c == t.c]
            layout += [n_ for n_, t_ in self.tower.inputs.items() if t_.hw is None and t_.c == t.c and n_ not in layout]
            slab = self.sess.shared_inputs.setdefault(("slab", nb, t.c), self.be.zeros(len(layout) * nb * t.c))
            k = layout.index(name)
            self.buffers[f"in:slab{t.c}"] = slab
            self.buffers["in:" + name] = slab[k * nb * t.c:(k + 1) * nb * t.c]  # what set_input writes
            self.storage[id(t)] = Storage(f"in:slab{t.c}", nb, t.c, None, k * nb * t.c, t.c, 1)
        self._alloc("loss", 1)
        units = self._schedule_units()
        for unit in units:
            if len(unit) > 1:
                self._fwd_group(unit)
            else:
                self._fwd_node(*unit[0])
        for ti, term in enumerate(self.terms):
            self._emit_term(ti, term)
        self._flush_loss_terms()
        if self.terms:
            for unit in reversed(units):
                if len(unit) > 1:
                    self._bwd_group(unit)
                    continue
                idx, node = unit[0]
                if id(node.out) not in self._grad_needed:
                    continue
                tap = self._taps.get(id(node))
                tap_written = tap is not None and self.grad_written.get(id(tap), False)
                if not self.grad_written.get(id(node.out.owner), False):
                    if not tap_written:
                        continue  # this application does not feed the phase's loss
                    # only the encoder output read from this application carries a gradient: zero for its own output
                    gname = self._ensure_grad(node.out.owner)
                    st = self.storage[id(node.out.owner)]
                    self.bwd.append(Launch("fill_f32", (self._ref(gname, st.ch_off), st.rows * st.ld, 0.0),
                                           tag="zero-grad-rows"))
                    self.grad_written[id(node.out.owner)] = True
                self._bwd_node(idx, node)
            self._flush_wgrads()
            self._flush_slab_reduces()
            self._emit_regularisers()
            self._finish_loss_slots()
        self._finish_scratch()

    def _fwd_node(self, idx, node):
        if isinstance(node, G.LinearNode):
            self._fwd_linear(idx, node)
        elif isinstance(node, G.GeneratorNode):
            self._fwd_generator(idx, node)
        elif isinstance(node, G.DenseStackNode):
            self._fwd_densestack(idx, node)
        elif isinstance(node, G.FeatStackNode):
            self._fwd_featstack(idx, node)
        elif isinstance(node, G.PostNode):
            self._fwd_post(idx, node)
        else:
            raise TypeError(node)

    def _bwd_node(self, idx, node):
        if isinstance(node, G.LinearNode):
            self._bwd_linear(idx, node)
        elif isinstance(node, G.GeneratorNode):
            self._bwd_generator(idx, node)
        elif isinstance(node, G.DenseStackNode):
            self._bwd_densestack(idx, node)
        elif isinstance(node, G.FeatStackNode):
            self._bwd_featstack(idx, node)
        elif isinstance(node, G.PostNode):
            self._bwd_post(idx, node)

    # ---- same-weight applications as one row-concatenated application ----
    def _batch_signature(self, node):
        """Hashable identity of "the same network layer" (same variables, same hyper-parameters), or None when the node
        kind is not batched.  Only ops whose rows are independent qualify: no batch statistics, no per-tensor norms."""
        if isinstance(node, G.GeneratorNode):
            return ("gen", id(node.weights[0]), bool(node.only_encoder), node.src.c, id(node) in self._taps)
        if isinstance(node, G.DenseStackNode):
            return ("ds", id(node.weights[0]), node.src.c)
        if isinstance(node, G.LinearNode) and node.kind in ("dense", "blockdense"):
            src = node.sources[0] if len(node.sources) == 1 else None
            if (src is None or node.has_bn or node.residuals or node.dropout_keep is not None or src.hw is not None
                    or src.pixmap is not None or node.out.hw is not None):
                return None
            return ("lin", node.kind, tuple(id(b.w) for b in node.branches), src.c)
        if isinstance(node, G.FeatStackNode):
            # the stacked slices must be the adjacent equal-width column blocks of ONE tensor that they cover entirely: the
            # group then concatenates that tensor, and the kernel keeps one set of norms per application (row segment)
            srcs = node.srcs
            own = srcs[0].owner
            if (len(srcs) > 1 and all(t.owner is own and t.root is not None and t.c == srcs[0].c and t.hw is None
                                      for t in srcs)
                    and all(t.ch_off == i * srcs[0].c for i, t in enumerate(srcs)) and len(srcs) * srcs[0].c == own.c):
                return ("feat", len(srcs), srcs[0].c)
        return None

    def _shape_signature(self, node):
        """Hashable identity of "a network of this shape" for the fused kernels that take several variable sets per launch
        (None: not one of them, or this application cannot join such a launch)."""
        if not (BATCH_HETERO and SLAB_REDUCE_MULTI and hasattr(self.be, "gan_generator_blocks_apps")):
            return None
        if isinstance(node, G.GeneratorNode):
            if id(node) in self._taps or not self.be.gan_generator_tap_supported(node.src.c):  # (matrix-core shapes only)
                return None
            return ("gen", bool(node.only_encoder), node.src.c, self._trains(node.weights))
        if isinstance(node, G.DenseStackNode):
            return ("ds", tuple(node.widths), tuple(bool(l[2]) for l in node.layers), float(node.alpha),
                    self._trains(node.weights))
        return None

    @staticmethod
    def _app_runs(unit):
        """The members of a unit as runs of consecutive same-variable applications: [[(idx, node), ...], ...]."""
        runs = []
        for m in unit:
            w = m[1].weights[0] if isinstance(m[1], (G.GeneratorNode, G.DenseStackNode)) else None
            if runs and w is not None and runs[-1][0][1].weights[0] is w:
                runs[-1].append(m)
            else:
                runs.append([m])
        return runs

    @staticmethod
    def _node_src(node):
        if isinstance(node, G.FeatStackNode):
            return node.srcs[0].owner
        return node.sources[0] if isinstance(node, G.LinearNode) else node.src

    def _schedule_units(self):
        """[[(idx, node), ...]]: the needed nodes in an executable order, same-weight applications grouped.  Members of a
        group run when the last of their inputs is ready; a grouping that would make the unit graph cyclic (CycleGAN:
        G_xy(G_yx(y)) next to G_xy(x)) is split by depth."""
        order = [(i, n) for i, n in enumerate(self.tower.nodes) if n in self.needed and id(n) not in self._tapped]
        if not BATCH_APPS:
            return [[u] for u in order]
        prod = {id(n.out): k for k, (_, n) in enumerate(order)}  # tensor owner -> position of its producer
        for k, (_, n) in enumerate(order):  # an encoder tap is produced by its full generator's launch
            if id(n) in self._taps:
                prod[id(self._taps[id(n)])] = k
        deps = [sorted({prod[id(t.owner)] for t in self._node_inputs(n) if id(t.owner) in prod}) for _, n in order]
        depth = [0] * len(order)  # (with encoder taps a consumer may precede its producer in tower order: no single sweep)

        def depth_of(k):
            if depth[k] == 0:
                depth[k] = 1 + max([depth_of(d) for d in deps[k]], default=0)
            return depth[k]

        for k in range(len(order)):
            depth_of(k)
        sigs = [self._batch_signature(n) for _, n in order]

        def attempt(by_depth, hetero=False):
            groups = {}
            for k, sg in enumerate(sigs):
                key = (k,) if sg is None else ((sg, depth[k]) if by_depth else (sg,))
                groups.setdefault(key, []).append(k)
            units = []
            for key, ks in groups.items():
                srcs = [id(self._node_src(order[k][1])) for k in ks] if len(ks) > 1 else []
                if len(ks) > 1 and len(set(srcs)) != len(srcs):
                    units += [[k] for k in ks]  # two applications on the same tensor: their input gradients would race
                    continue
                for c0 in range(0, len(ks), BATCH_APPS_MAX):
                    units.append(ks[c0:c0 + BATCH_APPS_MAX])
            if hetero:  # pairs of same-shaped units with different variables -> one unit of two variable sets
                pools, merged, used = {}, [], set()
                for u, ks in enumerate(units):
                    sh = self._shape_signature(order[ks[0]][1])
                    if sh is not None:
                        pools.setdefault((sh, len(ks), depth[ks[0]] if by_depth else 0), []).append(u)
                for us in pools.values():
                    for a, b in zip(us[0::2], us[1::2]):
                        srcs = [id(self._node_src(order[k][1])) for k in units[a] + units[b]]
                        if len(set(srcs)) == len(srcs):
                            merged.append(units[a] + units[b])
                            used |= {a, b}
                units = [ks for u, ks in enumerate(units) if u not in used] + merged
                units.sort(key=lambda ks: ks[0])
            unit_of = {k: u for u, ks in enumerate(units) for k in ks}
            udeps = [sorted({unit_of[d] for k in ks for d in deps[k]} - {u}) for u, ks in enumerate(units)]
            if any(unit_of[d] == u for u, ks in enumerate(units) for k in ks for d in deps[k]):
                return None  # a member depends on another member
            done, out = set(), []
            while len(out) < len(units):  # Kahn, ties by first member (tower order)
                ready = [u for u in range(len(units)) if u not in done and all(d in done for d in udeps[u])]
                if not ready:
                    return None
                u = min(ready, key=lambda v: units[v][0])
                done.add(u)
                out.append([order[k] for k in units[u]])
            return out

        tries = [attempt(False), attempt(True)]
        if BATCH_HETERO:
            tries += [attempt(False, True), attempt(True, True)]
        tries = [t for t in tries if t is not None]
        return min(tries, key=len) if tries else [[u] for u in order]  # fewest units; ties: the plainer grouping

    def _address_order(self, unit):
        """Reorder the members of a unit IN PLACE (they are independent of each other) so that inputs which are row
        blocks of one buffer come in address order -- same-variable runs kept together, runs ordered by their first
        block: neighbouring inputs (the slab of the tower's placeholders, the outputs of an earlier batch) then
        concatenate without a copy."""
        sts = [self.storage_of(self._node_src(n)) for _, n in unit]
        if len({st.buf for st in sts}) != 1 or any(st.pixmap is not None for st in sts):
            return
        addr = {id(m[1]): st.ch_off for m, st in zip(unit, sts)}
        runs = [sorted(r, key=lambda m: addr[id(m[1])]) for r in self._app_runs(unit)]
        # (runs of _app_runs are consecutive same-variable members; members of one variable set may be split into several
        # runs only if the scheduler interleaved them, which it does not)
        runs.sort(key=lambda r: addr[id(r[0][1])])
        unit[:] = [m for r in runs for m in r]

    def _concat_inputs(self, tag, srcs):
        """Storage of the row-concatenated inputs [G * nb, c] of a group.  Zero copy when the inputs already are
        consecutive row blocks of one buffer (the outputs of the previous batched layer), else one hypel_copy_blocks_f32
        gathers them.  Returns (storage, gathered?)."""
        nb = self.nb
        sts = [self.storage_of(t) for t in srcs]
        s0 = sts[0]
        if all(st.buf == s0.buf and st.ld == s0.ld and st.c == s0.c and st.pixmap is None and
               st.ch_off == s0.ch_off + g * nb * s0.ld for g, st in enumerate(sts)):
            return Storage(s0.buf, nb * len(srcs), s0.ld, None, s0.ch_off, s0.c, 1), False
        from .backend import COPY_BLOCK_DTYPE
        c = srcs[0].c
        name = f"cat:{tag}"
        self._alloc(name, len(srcs) * nb * c)
        base = Ref(self.sess.params)

        def rel(ref):
            return (ref.ptr() - base.ptr()) // 4

        ents = [(rel(self._ref(st.buf, st.ch_off)), rel(self._ref(name, g * nb * c)), nb, c, st.ld, c, 0, 0)
                for g, st in enumerate(sts)]
        t = self.be.upload(np.array(ents, COPY_BLOCK_DTYPE))
        self.tables.append(t)
        self.fwd.append(Launch("copy_blocks_f32", (base, Ref(t), len(ents), nb * c), nbytes=8 * len(srcs) * nb * c,
                               tag="batch-gather"))
        return Storage(name, nb * len(srcs), c, None, 0, c, 1), True

    def _fwd_group(self, unit):
        """G same-weight applications as one application on G * nb rows: a representative copy of the node runs through
        the ordinary handler with self.nb = G * nb; the members' outputs are row-block views of its output buffer."""
        import copy
        nb0, G_ = self.nb, len(unit)
        self._address_order(unit)
        idx0, n0 = unit[0]
        srcs = [self._node_src(n) for _, n in unit]
        cat_st, gathered = self._concat_inputs(idx0, srcs)
        syn_src = G.SymTensor(self.tower, None, srcs[0].c, node=None)
        syn_src.needs_grad = any(self._needs_grad(t) for t in srcs)
        self.storage[id(syn_src)] = cat_st
        rep = copy.copy(n0)
        if isinstance(n0, G.LinearNode):
            rep.sources = [syn_src]
        elif isinstance(n0, G.FeatStackNode):
            w_ = n0.srcs[0].c
            rep.srcs = [syn_src.slice_channels(i * w_, (i + 1) * w_) for i in range(len(n0.srcs))]
            rep.segments = G_
        else:
            rep.src = syn_src
        rep.out = G.SymTensor(self.tower, None, n0.out.c, node=rep)
        runs = self._app_runs(unit)
        if len(runs) > 1 and isinstance(n0, (G.GeneratorNode, G.DenseStackNode)):
            assert len(runs) == 2 and len(runs[0]) == len(runs[1]), "two variable sets, equally many applications each"
            rep.app_nodes = [r[0][1] for r in runs]  # the handlers emit the *_apps form: one variable set per run
        taps = [self._taps.get(id(n)) for _, n in unit]
        syn_tap = None
        if taps[0] is not None:  # (the signature keeps tapped and untapped applications apart)
            syn_tap = G.SymTensor(self.tower, None, taps[0].c, node=rep)
            self._taps[id(rep)] = syn_tap
            if any(id(t) in self._grad_needed for t in taps):
                self._grad_needed.add(id(syn_tap))
        if syn_src.needs_grad:
            self._grad_needed.add(id(syn_src))
        if any(id(n.out) in self._grad_needed for _, n in unit):
            self._grad_needed.add(id(rep.out))
        self.nb = nb0 * G_
        try:
            self._fwd_node(idx0, rep)
        finally:
            self.nb = nb0
        out_st = self.storage[id(rep.out)]
        for g, (_, n) in enumerate(unit):
            self.storage[id(n.out)] = Storage(out_st.buf, nb0, out_st.ld, None, out_st.ch_off + g * nb0 * out_st.ld,
                                              n.out.c, 1)
        if id(rep.out) in self._grad_needed and self.terms and "g:" + out_st.buf not in self.buffers:
            self._alloc("g:" + out_st.buf, G_ * nb0 * out_st.ld)  # one gradient buffer, the members' are its row blocks
        if syn_tap is not None:
            t_st = self.storage[id(syn_tap)]
            for g, t in enumerate(taps):
                self.storage[id(t)] = Storage(t_st.buf, nb0, t_st.ld, None, t_st.ch_off + g * nb0 * t_st.ld, t.c, 1)
            if id(syn_tap) in self._grad_needed and self.terms and "g:" + t_st.buf not in self.buffers:
                self._alloc("g:" + t_st.buf, G_ * nb0 * t_st.ld)
        self._groups = getattr(self, "_groups", {})
        self._groups[idx0] = dict(rep=rep, syn_src=syn_src, srcs=srcs, gathered=gathered, cat=cat_st, taps=taps,
                                  syn_tap=syn_tap)

    def _bwd_group(self, unit):
        nb0, G_ = self.nb, len(unit)
        idx0, _ = unit[0]
        grp = self._groups[idx0]
        rep, syn_src, srcs = grp["rep"], grp["syn_src"], grp["srcs"]
        if id(rep.out) not in self._grad_needed:
            return
        written = [self.grad_written.get(id(n.out), False) for _, n in unit]
        taps, syn_tap = grp["taps"], grp["syn_tap"]
        tap_written = [t is not None and self.grad_written.get(id(t), False) for t in taps]
        if not any(written) and not any(tap_written):
            return  # none of these applications feeds the phase's loss
        out_st = self.storage[id(rep.out)]
        gname = "g:" + out_st.buf
        if gname not in self.buffers:
            self._alloc(gname, G_ * nb0 * out_st.ld)
        for g, ((_, n), wr) in enumerate(zip(unit, written)):
            if not wr:  # an application without a gradient contributes zero rows
                m_st = self.storage[id(n.out)]
                self.bwd.append(Launch("fill_f32", (self._ref(gname, m_st.ch_off), nb0 * m_st.ld, 0.0),
                                       tag="zero-grad-rows"))
        self.grad_written[id(rep.out)] = True
        if syn_tap is not None and any(tap_written):
            for t, wr in zip(taps, tap_written):
                if not wr:
                    m_st = self.storage[id(t)]
                    self.bwd.append(Launch("fill_f32", (self._ref("g:" + m_st.buf, m_st.ch_off), nb0 * m_st.ld, 0.0),
                                           tag="zero-grad-rows"))
            self.grad_written[id(syn_tap)] = True
        # input gradient: the inputs either are row blocks of ONE buffer (outputs of the previous batched layer) -- the
        # handler then writes straight into that buffer's gradient, blocks nobody wrote yet zeroed first so that one
        # accumulate state serves all rows -- or were gathered: the gradient of the concatenation is scattered (added)
        # into the members' gradient targets by one block-copy launch
        want = [self._needs_grad(t) for t in srcs]
        if any(want):
            if not grp["gathered"]:
                for t in srcs:
                    self._ensure_grad(t.owner)
                states = [self.grad_written.get(id(t.owner), False) for t in srcs]
                if any(states) and not all(states):
                    for t, wr in zip(srcs, states):
                        if not wr:
                            st = self.storage_of(t)
                            self.bwd.append(Launch("fill_f32", (self._ref("g:" + st.buf, st.ch_off), nb0 * st.ld, 0.0),
                                                   tag="zero-grad-rows"))
                self.grad_written[id(syn_src)] = any(states)
            else:
                self.grad_written.pop(id(syn_src), None)
        self.nb = nb0 * G_
        try:
            self._bwd_node(idx0, rep)
        finally:
            self.nb = nb0
        if any(want):
            if not grp["gathered"]:
                for t in srcs:
                    self.grad_written[id(t.owner)] = True
            else:
                from .backend import COPY_BLOCK_DTYPE
                base = Ref(self.sess.params)

                def rel(ref):
                    return (ref.ptr() - base.ptr()) // 4

                c = srcs[0].c
                scatter = "g:" + grp["cat"].buf
                ents = []
                for g, (t, w_) in enumerate(zip(srcs, want)):
                    if not w_:
                        continue
                    gst, acc = self._grad_target(t)
                    ents.append((rel(self._ref(scatter, g * nb0 * c)), rel(self._ref(gst.buf, gst.ch_off)), nb0, c, c,
                                 gst.ld, acc, 0))
                tbl = self.be.upload(np.array(ents, COPY_BLOCK_DTYPE))
                self.tables.append(tbl)
                self.bwd.append(Launch("copy_blocks_f32", (base, Ref(tbl), len(ents), nb0 * c), nbytes=8 * len(ents) * nb0 * c,
                                       tag="batch-scatter"))

    # ---- per-block gradient slabs of the fused kernels: reduced once per train op ----
    def _defer_slab_reduce(self, launch, pos_w, pos_b, blocks, w0, w_stride, w_count, b0, b_stride, b_count, acc):
        """The backward kernel `launch` leaves `blocks` filter / bias gradient slabs (arguments pos_w / pos_b).  Instead of
        one reduction launch per application, every application of one weight set appends its slabs to that set's
        region and ONE hypel_reduce_splits_wave_multi_f32 at the end of the backward pass sums each region (one entry per
        weight set and kind: two entries never write the same gradient)."""
        sets = self.__dict__.setdefault("_slab_sets", {})
        st = sets.setdefault(id(w0), dict(w0=w0, b0=b0, w=(w_stride, w_count), b=(b_stride, b_count), acc=acc, apps=[],
                                         blocks=0))
        st["apps"].append((launch, pos_w, pos_b, st["blocks"]))
        st["blocks"] += blocks

    def _defer_slab_reduce_apps(self, launch, pos_w, pos_b, pos_ws, pos_bs, bpa, nodes, w_stride, w_count, b_stride, b_count):
        """The *_apps form: `launch` leaves `bpa` slabs for each of its variable sets (`nodes`: one application node per
        set).  Every set's slabs still go to that set's own region; the launch reaches set g's first slab by a stride from
        set 0's (arguments pos_ws / pos_bs), known once the regions are laid out."""
        sets = self.__dict__.setdefault("_slab_sets", {})
        where = []
        for nd in nodes:
            w0, b0 = nd.weights[0], nd.biases[0]
            wacc = self._param_acc(w0)
            for v in nd.weights[1:] + nd.biases:
                self._param_acc(v)
            st = sets.setdefault(id(w0), dict(w0=w0, b0=b0, w=(w_stride, w_count), b=(b_stride, b_count), acc=wacc, apps=[],
                                             blocks=0))
            where.append((id(w0), st["blocks"]))
            st["blocks"] += bpa
        self.__dict__.setdefault("_slab_apps", []).append((launch, pos_w, pos_b, pos_ws, pos_bs, where))

    def _flush_slab_reduces(self):
        sets = self.__dict__.get("_slab_sets") or {}
        self._slab_sets = {}
        bias_entries = self.__dict__.get("_bias_sum_entries") or []
        self._bias_sum_entries = []
        if not sets and not bias_entries:
            return
        base = Ref(self.sess.params)

        def rel(ref):
            return (ref.ptr() - base.ptr()) // 4

        # Two entries of one launch must not write the same output (include/hypel.h): a BN-less layer applied twice as
        # separate units (no row-concatenated batch) leaves two chunk-sum entries for ONE bias gradient, acc = 0 then 1.
        # Entry k of an output goes to round k; every round is its own launch, in order.
        ents, total = [], 0
        later, seen = [], {}
        for part_ref, out_ref, stride, count, n_splits, acc in bias_entries:  # chunk sums [chunk][2][c]: plane 0 = sum(dY)
            e = (rel(part_ref), rel(out_ref), stride, count, n_splits, acc)
            r = seen.get(e[1], 0)
            seen[e[1]] = r + 1
            if r == 0:
                ents.append(e)
                total += count
            else:
                while len(later) < r:
                    later.append([])
                later[r - 1].append(e)
        for k, st in enumerate(sets.values()):
            fid = self.__dict__.setdefault("_slab_bufs", 0)
            self._slab_bufs = fid + 1
            names = st["names"] = {}
            for kind, var in (("w", st["w0"]), ("b", st["b0"])):
                stride, count = st[kind]
                names[kind] = f"slabs_{kind}:{fid}"
                self._alloc(names[kind], st["blocks"] * stride)
                ents.append((rel(self._ref(names[kind])), rel(self._g(var)), stride, count, st["blocks"], st["acc"]))
                total += count
            for launch, pos_w, pos_b, b0 in st["apps"]:
                args = list(launch.args)
                args[pos_w] = self._ref(names["w"], b0 * st["w"][0])
                args[pos_b] = self._ref(names["b"], b0 * st["b"][0])
                launch.args = tuple(args)
        for launch, pos_w, pos_b, pos_ws, pos_bs, where in self.__dict__.get("_slab_apps") or []:
            first = [(self._ref(sets[sid]["names"]["w"], b0 * sets[sid]["w"][0]),
                      self._ref(sets[sid]["names"]["b"], b0 * sets[sid]["b"][0])) for sid, b0 in where]
            args = list(launch.args)
            args[pos_w], args[pos_b] = first[0]
            args[pos_ws] = (first[1][0].ptr() - first[0][0].ptr()) // 4
            args[pos_bs] = (first[1][1].ptr() - first[0][1].ptr()) // 4
            launch.args = tuple(args)
        self._slab_apps = []
        e_t = self.be.upload(np.array(ents, REDUCE_ENTRY_DTYPE))
        self.tables.append(e_t)
        self.bwd.append(Launch("reduce_splits_wave_multi_f32", (base, Ref(e_t), len(ents), total), tag="slab-reduce"))
        for rnd in later:
            e_t = self.be.upload(np.array(rnd, REDUCE_ENTRY_DTYPE))
            self.tables.append(e_t)
            self.bwd.append(Launch("reduce_splits_wave_multi_f32", (base, Ref(e_t), len(rnd), sum(e[3] for e in rnd)),
                                   tag="slab-reduce"))

    @staticmethod
    def _rel(var0, var1):
        """element distance of var1 from var0 in the flat parameter buffer"""
        return int(var1.offset) - int(var0.offset)

    # ---- fused generator ----
    def _gen_refs(self, node):
        self._assert_contiguous(node.weights)
        self._assert_contiguous(node.biases)
        return node.weights[0], node.biases[0], sum(w.size for w in node.weights)

    def _fwd_generator(self, idx, node):
        src, out = node.src, node.out
        s_st = self.storage_of(src)
        st = self._new_value(out, f"z:{idx}")
        w0, b0, _ = self._gen_refs(node)
        tap = self._taps.get(id(node))
        apps = getattr(node, "app_nodes", None)
        if apps is not None:  # two generators of one shape, each on its half of the rows
            w1, b1, _ = self._gen_refs(apps[1])
            l = Launch("gan_generator_fwd_apps", (self._ref(s_st.buf, s_st.ch_off), s_st.ld, self.nb // len(apps), len(apps),
                                                  self._rel(w0, w1), self._rel(b0, b1), src.c, self._p(w0), self._p(b0),
                                                  int(node.only_encoder), self._ref(st.buf), st.ld, None),
                       nbytes=8 * self.nb * src.c, tag="gen-fwd-apps")
        elif tap is not None:
            # the encoder-only application on the same input is this launch's n_4
            t_st = self._new_value(tap, f"ztap:{idx}")
            l = Launch("gan_generator_fwd_tap", (self._ref(s_st.buf, s_st.ch_off), s_st.ld, self.nb, src.c, self._p(w0),
                                                 self._p(b0), self._ref(st.buf), st.ld, self._ref(t_st.buf), t_st.ld, None),
                       nbytes=12 * self.nb * src.c, tag="gen-fwd+enc")
        else:
            l = Launch("gan_generator_fwd", (self._ref(s_st.buf, s_st.ch_off), s_st.ld, self.nb, src.c, self._p(w0),
                                             self._p(b0), int(node.only_encoder), self._ref(st.buf), st.ld),
                       nbytes=8 * self.nb * src.c, tag="gen-fwd")
        self.fwd.append(l)
        self._gen_fwd = getattr(self, "_gen_fwd", {})
        self._gen_fwd[idx] = l  # _bwd_generator turns it into the activation-keeping form when a backward pass follows

    def _bwd_generator(self, idx, node):
        src, out = node.src, node.out
        s_st = self.storage_of(src)
        z_st = self.storage[id(out)]
        w0, b0, wtotal = self._gen_refs(node)
        apps = getattr(node, "app_nodes", None)
        n_apps = len(apps) if apps is not None else 1
        blocks = self.be.gan_generator_blocks(self.nb) if apps is None else \
            self.be.gan_generator_blocks_apps(self.nb // n_apps, n_apps)
        dx, lddx, acc = None, 0, 0
        if self._needs_grad(src):
            gst, acc = self._grad_target(src)
            dx, lddx = self._ref(gst.buf, gst.ch_off), gst.ld
        tap = self._taps.get(id(node))
        tap_grad = tap is not None and self.grad_written.get(id(tap), False)
        keep_n = n_apps * self.be.gan_generator_keep_floats(self.nb // n_apps, src.c, int(node.only_encoder)) if GEN_KEEP else 0
        kref = None
        if keep_n > 0:
            # the forward pass of this application leaves its activations for this launch (hypel.h: bit-identical to
            # recomputing them; 188 KB per 16 samples at 360 bands)
            self._alloc(f"gkeep:{idx}", keep_n)
            kref = self._ref(f"gkeep:{idx}")
            f = self._gen_fwd[idx]
            if f.name in ("gan_generator_fwd_tap", "gan_generator_fwd_apps"):
                f.args = tuple(f.args[:-1]) + (kref,)
            else:
                f.name, f.args = "gan_generator_fwd_keep", tuple(f.args) + (kref,)
            f.bytes += 4 * keep_n
        if apps is not None:
            w1, b1, _ = self._gen_refs(apps[1])
            l1 = Launch("gan_generator_bwd_apps", (self._ref(s_st.buf, s_st.ch_off), s_st.ld, self._ref("g:" + z_st.buf),
                                                   z_st.ld, self.nb // n_apps, n_apps, self._rel(w0, w1), self._rel(b0, b1), 0,
                                                   0, src.c, self._p(w0), self._p(b0), int(node.only_encoder), dx, lddx, acc,
                                                   None, None, kref),
                        nbytes=12 * self.nb * src.c + 4 * keep_n, tag="gen-bwd-apps")
            self.bwd.append(l1)
            if self._trains(node.weights):
                self._defer_slab_reduce_apps(l1, 17, 18, 8, 9, blocks // n_apps, apps, wtotal, wtotal, 8, 7)
            else:
                self._scratch(l1, 17, "scratch_gen_w", blocks * wtotal)
                self._scratch(l1, 18, "scratch_gen_b", blocks * 8)
            return
        if tap_grad:
            # one backward pass for the full application and the encoder-only one read from it: the gradient that reached
            # the encoder output joins dn_4
            t_st = self.storage[id(tap)]
            l1 = Launch("gan_generator_bwd_tap", (self._ref(s_st.buf, s_st.ch_off), s_st.ld, self._ref("g:" + z_st.buf),
                                                  z_st.ld, self._ref("g:" + t_st.buf, t_st.ch_off), t_st.ld, self.nb, src.c,
                                                  self._p(w0), self._p(b0), dx, lddx, acc, None, None, kref),
                        nbytes=16 * self.nb * src.c + 4 * keep_n, tag="gen-bwd+enc")
            pw_pos, pb_pos = 13, 14
        else:
            l1 = Launch("gan_generator_bwd", (self._ref(s_st.buf, s_st.ch_off), s_st.ld, self._ref("g:" + z_st.buf),
                                              z_st.ld, self.nb, src.c, self._p(w0), self._p(b0), int(node.only_encoder),
                                              dx, lddx, acc, None, None), nbytes=12 * self.nb * src.c, tag="gen-bwd")
            if kref is not None:
                l1.name, l1.args = "gan_generator_bwd_kept", tuple(l1.args) + (kref,)
                l1.bytes += 4 * keep_n
            pw_pos, pb_pos = 12, 13
        self.bwd.append(l1)
        if self._trains(node.weights):
            wacc = self._param_acc(w0)
            if SLAB_REDUCE_MULTI:
                self._defer_slab_reduce(l1, pw_pos, pb_pos, blocks, w0, wtotal, wtotal, b0, 8, 7, wacc)
                return
        self._scratch(l1, pw_pos, "scratch_gen_w", blocks * wtotal)
        self._scratch(l1, pb_pos, "scratch_gen_b", blocks * 8)
        if self._trains(node.weights):
            # filter and bias slabs in one launch
            l2 = Launch("reduce_splits_pair_f32", (None, wtotal, wtotal, self._g(w0), None, 8, 7, self._g(b0), blocks, wacc),
                        tag="gen-dw+db")
            self._scratch(l2, 0, "scratch_gen_w", blocks * wtotal)
            self._scratch(l2, 4, "scratch_gen_b", blocks * 8)
            self.bwd.append(l2)

    # ---- fused fully-connected stack (narrow discriminators) ----
    def _densestack_args(self, node):
        self._assert_contiguous(node.weights)
        self._assert_contiguous(node.biases)
        widths = node.widths
        if not self.be.dense_stack_supported(widths):
            raise RuntimeError(f"fused fully-connected stack {widths} is not supported by the library")
        mask = sum(1 << l for l, (_, _, leaky) in enumerate(node.layers) if leaky)
        return (len(node.layers), *(widths + [0] * (5 - len(widths))), mask, float(node.alpha))

    def _fwd_densestack(self, idx, node):
        src, out = node.src, node.out
        s_st = self.storage_of(src)
        st = self._new_value(out, f"z:{idx}")
        macs = sum(w.size for w in node.weights)
        apps = getattr(node, "app_nodes", None)
        if apps is not None:  # two critics of one shape, each on its half of the rows
            self._densestack_args(apps[1])
            self.fwd.append(Launch("dense_stack_fwd_apps", (self._ref(s_st.buf, s_st.ch_off), s_st.ld, self.nb // len(apps),
                                                            len(apps), self._rel(node.weights[0], apps[1].weights[0]),
                                                            self._rel(node.biases[0], apps[1].biases[0]),
                                                            *self._densestack_args(node), self._p(node.weights[0]),
                                                            self._p(node.biases[0]), self._ref(st.buf), st.ld),
                                   flops=2 * self.nb * macs, nbytes=4 * self.nb * (src.c + out.c),
                                   tag="dense-stack-fwd-apps"))
            return
        self.fwd.append(Launch("dense_stack_fwd", (self._ref(s_st.buf, s_st.ch_off), s_st.ld, self.nb,
                                                   *self._densestack_args(node), self._p(node.weights[0]),
                                                   self._p(node.biases[0]), self._ref(st.buf), st.ld),
                               flops=2 * self.nb * macs, nbytes=4 * self.nb * (src.c + out.c), tag="dense-stack-fwd"))

    def _bwd_densestack(self, idx, node):
        src, out = node.src, node.out
        s_st = self.storage_of(src)
        z_st = self.storage[id(out)]
        w0, b0 = node.weights[0], node.biases[0]
        wtotal, btotal = sum(w.size for w in node.weights), sum(b.size for b in node.biases)
        blocks = self.be.dense_stack_blocks(self.nb)
        dx, lddx, acc = None, 0, 0
        if self._needs_grad(src):
            gst, acc = self._grad_target(src)
            dx, lddx = self._ref(gst.buf, gst.ch_off), gst.ld
        apps = getattr(node, "app_nodes", None)
        if apps is not None:
            n_apps = len(apps)
            blocks = self.be.dense_stack_blocks_apps(self.nb // n_apps, n_apps)
            l1 = Launch("dense_stack_bwd_apps", (self._ref(s_st.buf, s_st.ch_off), s_st.ld, self._ref("g:" + z_st.buf),
                                                 z_st.ld, self.nb // n_apps, n_apps, self._rel(w0, apps[1].weights[0]),
                                                 self._rel(b0, apps[1].biases[0]), 0, 0, *self._densestack_args(node),
                                                 self._p(w0), self._p(b0), dx, lddx, acc, None, None),
                        flops=6 * self.nb * wtotal, nbytes=4 * self.nb * (2 * src.c + out.c), tag="dense-stack-bwd-apps")
            n_args = len(l1.args)
            self.bwd.append(l1)
            if self._trains(node.weights):
                self._defer_slab_reduce_apps(l1, n_args - 2, n_args - 1, 8, 9, blocks // n_apps, apps, wtotal, wtotal, btotal,
                                             btotal)
            else:
                self._scratch(l1, n_args - 2, "scratch_ds_w", blocks * wtotal)
                self._scratch(l1, n_args - 1, "scratch_ds_b", blocks * btotal)
            return
        l1 = Launch("dense_stack_bwd", (self._ref(s_st.buf, s_st.ch_off), s_st.ld, self._ref("g:" + z_st.buf), z_st.ld,
                                        self.nb, *self._densestack_args(node), self._p(w0), self._p(b0), dx, lddx, acc, None,
                                        None), flops=6 * self.nb * wtotal, nbytes=4 * self.nb * (2 * src.c + out.c),
                    tag="dense-stack-bwd")
        n_args = len(l1.args)
        self.bwd.append(l1)
        if self._trains(node.weights):
            wacc = self._param_acc(w0)
            for v in node.weights[1:] + node.biases:
                self._param_acc(v)
            if SLAB_REDUCE_MULTI:
                self._defer_slab_reduce(l1, n_args - 2, n_args - 1, blocks, w0, wtotal, wtotal, b0, btotal, btotal, wacc)
                return
        self._scratch(l1, n_args - 2, "scratch_ds_w", blocks * wtotal)
        self._scratch(l1, n_args - 1, "scratch_ds_b", blocks * btotal)
        if self._trains(node.weights):
            l2 = Launch("reduce_splits_pair_f32", (None, wtotal, wtotal, self._g(w0), None, btotal, btotal, self._g(b0),
                                                   blocks, wacc), tag="ds-dw+db")
            self._scratch(l2, 0, "scratch_ds_w", blocks * wtotal)
            self._scratch(l2, 4, "scratch_ds_b", blocks * btotal)
            self.bwd.append(l2)

    # ---- feature stack (global l2 normalise per slice, stacked) ----
    def _fwd_featstack(self, idx, node):
        out = node.out
        st = self._new_value(out, f"z:{idx}")
        segs = getattr(node, "segments", 1)  # applications of a row-concatenated batch: each keeps its own norms
        self._alloc(f"l2stat:{idx}", 2 * len(node.srcs) * segs)
        if self._adjacent_parts(node):
            s0 = self.storage_of(node.srcs[0])
            self.fwd.append(Launch("l2norm_segs_fwd", (self._ref(s0.buf, s0.ch_off), s0.ld, self.nb // segs,
                                                       node.srcs[0].c, len(node.srcs), segs, self._ref(st.buf), st.ld,
                                                       self._ref(f"l2stat:{idx}")), tag="l2norm"))
            return
        assert segs == 1
        off = 0
        for p, src in enumerate(node.srcs):
            s_st = self.storage_of(src)
            self.fwd.append(Launch("l2norm_fwd", (self._ref(s_st.buf, s_st.ch_off), s_st.ld, self.nb, src.c,
                                                  self._ref(st.buf, off), st.ld, self._ref(f"l2stat:{idx}", 2 * p)),
                                   tag="l2norm"))
            off += src.c

    def _adjacent_parts(self, node):
        """The stacked embeddings are equal-width neighbouring column blocks of one buffer (merged slice MLPs)."""
        srcs = node.srcs
        own = srcs[0].owner
        return (len(srcs) > 1 and all(s.owner is own and s.root is not None and s.c == srcs[0].c for s in srcs)
                and all(s.ch_off == srcs[0].ch_off + i * srcs[0].c for i, s in enumerate(srcs))
                and all(self._needs_grad(s) == self._needs_grad(srcs[0]) for s in srcs))

    def _bwd_featstack(self, idx, node):
        out = node.out
        st = self.storage[id(out)]
        if self._adjacent_parts(node):
            if self._needs_grad(node.srcs[0]):
                s0 = self.storage_of(node.srcs[0])
                own = node.srcs[0].owner
                if node.srcs[0].ch_off == 0 and len(node.srcs) * node.srcs[0].c == own.c:
                    # the parts cover their tensor: one gradient target, no zero fill in front of a partial first write
                    gst, acc = self._grad_target(own)
                    gst = Storage(gst.buf, gst.nb, gst.ld, None, gst.ch_off, node.srcs[0].c, 1)
                else:
                    accs = [self._grad_target(s) for s in node.srcs]
                    gst, acc = accs[0]
                    assert all(a[1] == acc for a in accs), "parts of one buffer share the accumulate state"
                segs = getattr(node, "segments", 1)
                self.bwd.append(Launch("l2norm_segs_bwd", (self._ref(s0.buf, s0.ch_off), s0.ld,
                                                           self._ref("g:" + st.buf), st.ld, self.nb // segs,
                                                           node.srcs[0].c, len(node.srcs), segs,
                                                           self._ref(f"l2stat:{idx}"),
                                                           self._ref(gst.buf, gst.ch_off), gst.ld, acc),
                                       tag="l2norm-bwd"))
            return
        off = 0
        for p, src in enumerate(node.srcs):
            if self._needs_grad(src):
                s_st = self.storage_of(src)
                gst, acc = self._grad_target(src)
                self.bwd.append(Launch("l2norm_bwd", (self._ref(s_st.buf, s_st.ch_off), s_st.ld,
                                                      self._ref("g:" + st.buf, off), st.ld, self.nb, src.c,
                                                      self._ref(f"l2stat:{idx}", 2 * p),
                                                      self._ref(gst.buf, gst.ch_off), gst.ld, acc), tag="l2norm-bwd"))
            off += src.c

    # ---- loss terms ----
    def _grad_ref(self, t):
        if t is None or not self._needs_grad(t):
            return None, 0, 0
        gst, acc = self._grad_target(t)
        return self._ref(gst.buf, gst.ch_off), gst.ld, acc

    def _emit_term(self, ti, term):
        nb = self.nb
        a_st = self.storage_of(term.a)
        a_ref = self._ref(a_st.buf, a_st.ch_off)
        da, ldda, acc_a = self._grad_ref(term.a)
        acc_loss = 1 if getattr(self, "_loss_written", False) else 0
        if term.kind == "nce" or not LOSS_SLOTS:
            self._loss_written = True
        if term.kind == "nce":
            self._flush_loss_terms()  # the batched terms before it keep their place in the order of gradient writes
            b_st = self.storage_of(term.b)
            db, lddb, acc_b = self._grad_ref(term.b)
            l = Launch("nce_loss", (a_ref, a_st.ld, self._ref(b_st.buf, b_st.ch_off), b_st.ld, nb, int(term.parts),
                                    int(term.embed), float(term.tau), float(term.weight), self._ref("loss"), acc_loss,
                                    da, ldda, acc_a, db, lddb, acc_b, None), tag="loss-nce")
            self._scratch(l, 17, "scratch_nce", nb + 1024)
            self.fwd.append(l)
            return
        mode = {"mean_sq": 0, "mean_abs": 1, "mean": 2}[term.kind]
        b_ref, ldb, db, lddb, acc_b = None, 0, None, 0, 0
        if term.b is not None:
            b_st = self.storage_of(term.b)
            b_ref, ldb = self._ref(b_st.buf, b_st.ch_off), b_st.ld
            db, lddb, acc_b = self._grad_ref(term.b)
        if LOSS_SLOTS:
            # the term's weighted partial sums go to a slot of their own; ONE finaliser adds every slot of the op up, and
            # terms that write different gradient buffers share ONE launch (_flush_loss_terms)
            coef = float(term.weight) / (nb * term.a.c)
            self._batch_loss_term(self.fwd, dict(mode=mode, a=a_ref, lda=a_st.ld, b=b_ref, ldb=ldb, rows=nb, c=term.a.c,
                                                 target=float(term.target), gcoef=coef, pscale=coef, da=da, ldda=ldda,
                                                 acc_da=acc_a, db=db, lddb=lddb, acc_db=acc_b))
            return
        l = Launch("gan_loss", (mode, a_ref, a_st.ld, b_ref, ldb, nb, term.a.c, float(term.target), float(term.weight),
                                self._ref("loss"), acc_loss, da, ldda, acc_a, db, lddb, acc_b, None),
                   tag="loss-" + term.kind)
        self._scratch(l, 17, "scratch_red")
        self.fwd.append(l)

    def _batch_loss_term(self, lst, t):
        """Collect a deferred loss term; terms of a batch run concurrently in one launch, so a term that writes a gradient
        buffer an earlier term of the batch writes (or reads) closes that batch first."""
        batch = self.__dict__.setdefault("_term_batch", [])
        if batch and self._term_list is not lst:
            self._flush_loss_terms()
            batch = self._term_batch
        def spans(u, keys):
            out = []
            for k, ldk in keys:
                r = u[k]
                if r is not None:
                    n = u["rows"] if u["mode"] == 3 else (u["rows"] - 1) * u[ldk] + u["c"]
                    out.append((id(r.t), r.off, r.off + n))
            return out

        def hit(xs, ys):
            return any(a[0] == b[0] and a[1] < b[2] and b[1] < a[2] for a in xs for b in ys)

        outs, ins = spans(t, (("da", "ldda"), ("db", "lddb"))), spans(t, (("a", "lda"), ("b", "ldb")))
        for u in batch:
            u_outs, u_ins = spans(u, (("da", "ldda"), ("db", "lddb"))), spans(u, (("a", "lda"), ("b", "ldb")))
            if hit(outs, u_outs + u_ins) or hit(ins, u_outs):
                self._flush_loss_terms()
                break
        self._term_list = lst
        self._term_batch.append(t)

    def _flush_loss_terms(self):
        from .backend import LOSS_NONE, LOSS_TERM_DTYPE
        batch = self.__dict__.get("_term_batch") or []
        self._term_batch = []
        if not batch:
            return
        base = Ref(self.sess.params)
        base_ptr = base.ptr()

        def rel(ref):
            if ref is None:
                return LOSS_NONE
            d = ref.ptr() - base_ptr
            assert d % 4 == 0
            return d // 4

        first = getattr(self, "_n_loss_slots", 0)
        self._n_loss_slots = first + len(batch)
        arr = np.array([(rel(t["a"]), rel(t["b"]), rel(t["da"]), rel(t["db"]), t["lda"], t["ldb"], t["ldda"], t["lddb"],
                         t["rows"], t["mode"], t["c"], t["acc_da"], t["acc_db"], t["target"], t["gcoef"], t["pscale"],
                         first + k) for k, t in enumerate(batch)], LOSS_TERM_DTYPE)
        e_t = self.be.upload(arr)
        self.tables.append(e_t)
        l = Launch("loss_terms_slots", (base, Ref(e_t), len(batch), None), tag=f"loss-terms/{len(batch)}")
        self._loss_slot(l, 3)
        self._term_list.append(l)

    def _loss_slot(self, launch, pos):
        """Give a deferred loss term the next 1024-float slot of the op's slot buffer (allocated in _finish_loss_slots)."""
        self._slot_launches = getattr(self, "_slot_launches", [])
        self._slot_launches.append((launch, pos))

    def _finish_loss_slots(self):
        self._flush_loss_terms()
        pend = getattr(self, "_slot_launches", [])
        n_slots = getattr(self, "_n_loss_slots", 0)
        if not pend or not n_slots:
            return
        self._alloc("loss_slots", 1024 * n_slots)
        for launch, pos in pend:
            args = list(launch.args)
            args[pos] = self._ref("loss_slots")
            launch.args = tuple(args)
        self.bwd.append(Launch("loss_finalize_slots", (self._ref("loss_slots"), n_slots, self._ref("loss"),
                                                       1 if getattr(self, "_loss_written", False) else 0),
                               tag="loss-finalize"))
        self._slot_launches = []

    def _emit_regularisers(self):
        """tfgan.gan_loss adds the trained scope's regularisation losses (shadow_data_models.py:96,129).  Variables
        that are neighbours in the flat buffer and share the scale go out as one launch (sum of squares is additive)."""
        seen, items = set(), []
        for node in self.needed:
            for v in self._node_vars(node):
                if v.l2_scale and v.group in self.train_groups and v.name not in seen and v.name in self.param_written:
                    seen.add(v.name)
                    items.append((v.offset, v.size, float(v.l2_scale)))
        items.sort()
        runs = []
        for off, size, scale in items:
            if runs and runs[-1][0] + runs[-1][1] == off and runs[-1][2] == scale:
                runs[-1][1] += size
            else:
                runs.append([off, size, scale])
        for off, size, scale in runs:
            if LOSS_SLOTS:
                self._batch_loss_term(self.bwd, dict(mode=3, a=Ref(self.sess.params, off), lda=0, b=None, ldb=0, rows=size,
                                                     c=1, target=0.0, gcoef=scale, pscale=0.5 * scale,
                                                     da=Ref(self.sess.grads, off), ldda=0, acc_da=1, db=None, lddb=0,
                                                     acc_db=0))
                continue
            else:
                l = Launch("l2_reg", (Ref(self.sess.params, off), size, scale, self._ref("loss"), 1,
                                      Ref(self.sess.grads, off), None), tag="l2-reg")
                self._scratch(l, 6, "scratch_red")
            self.bwd.append(l)
