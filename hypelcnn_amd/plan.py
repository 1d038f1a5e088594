"""Lower a recorded `Tower` (hypelcnn_amd.graph) to a flat list of HIP kernel launches.

Layout decisions:
  * every patch tensor lives PIXEL-MAJOR in HBM: buffer[p][n][c], p = y*W + x.  A SAME
    convolution is then, per output pixel, a sum over its VALID taps of dense
    [N x Cin] x [Cin x Cout] products on contiguous row blocks: exact-tap work (2 620 instead of
    4 116 (pixel, tap) pairs for the four kernels of a 7x7 level), no im2col buffer, no masks.
  * flatten + fully_connected is the same sum over pixel blocks; the FC weight rows
    [p*C, (p+1)*C) are the "tap" of pixel p, so the reference's (h, w, c) flatten order is kept.
  * channel concat / split / crop are views: producers write at channel offsets, consumers read
    with (pixel map, channel offset).
  * the backward pass is derived here per node (no autograd engine): residual-map gradient,
    two-pass batch-norm/activation backward, data gradient (same GEMM, transposed weights),
    filter gradient (same GEMM, transposed activations, split over batch rows with a
    deterministic second-stage reduction).
"""
import os

import numpy as np

from . import graph as G
from .backend import GEMM_BM, MTILE_DTYPE, REDUCE_ENTRY_DTYPE, SEG_DTYPE, Ref
from .gemm_tables import SEG_PAIR_FLAG, GemmTables, Launch  # noqa: F401 (re-exported: tests and tools build tables through plan)

STAT_CHUNK_ROWS = 256  # upper bound; see stat_chunk_rows()
STAT_BLOCKS = 256            # blocks per 64-column stripe aimed at


def stat_chunk_rows(rows):
    """Rows per partial-reduction block: ~256 blocks per 64-column stripe, between 16 and 256 rows each
    (a thread walks chunk/4 rows serially, so short matrices get short chunks)."""
    c = (rows + STAT_BLOCKS - 1) // STAT_BLOCKS
    c = max(16, min(STAT_CHUNK_ROWS, c))
    return (c + 3) // 4 * 4
WGRAD_ROW_CHUNK = 512
TARGET_BLOCKS = 1536  # blocks a filter-gradient product is split towards
WGRAD_MAX_SPLITS = 64
WGRAD_MIN_SPLITS = 0  # 0 = one split per 64 batch rows once a launch fills the device unsplit
SPLIT_BIASED = True  # tap / channel-part splitting also for biased convs
DGRAD_MAX_SEGS = 18  # segments per data-gradient tile (0 = never split)
MAX_TAPS_PER_TILE = 9  # taps per forward tile of a multi-kernel level
L2_CHUNK_BYTES = int(3.5 * (1 << 20))  # X working set an XCD's L2 keeps
FWD_HINT_R2 = True  # round-2 forward tile-width rule (incl. 128x96 tiles)
SPLITK_BELOW = 400    # FC-shaped products with fewer 128x64 blocks are cut along K
SPLITK_TARGET = 768  # ... into slices that give about this many blocks
TAP_SPLIT_MIN_BATCH = 64  # below this a pixel block has too few rows for splitting to pay


# layers without batch norm: the bias-gradient reduction also writes dY (no separate activation-backward launch)
ACT_BIAS_BWD = True
# bias + leaky-ReLU of a normaliser-less fully-connected layer in the product's epilogue (HYPEL_GEMM_ACT_*): no post-op launch
ACT_IN_GEMM = True
# GAN loss terms and regularisers leave weighted partials in slots; one finaliser launch per train op sums them
LOSS_SLOTS = True
GEN_KEEP = True  # generator backward starts from the forward pass's kept activations
TILE_HINTS = True
SMALL_BN = True
SMALL_BN_ROWS = 1024  # hypel_bn_act_small_*: rows kept in registers (32 row lanes x 32 rows)
FOLD_RESIDUAL_GRAD = True
# Filter gradients have no consumer before the optimiser: instead of one launch (+ one reduce) per layer they are
# collected and go out as ONE hypel_seg_gemm_multi_f32 per tile width (+ ONE hypel_reduce_splits_multi_f32) at the end
# of the backward pass (and at every data-parallel sync point).  A step's twenty ~25 us launch ramps/drains become three.
MERGE_WGRAD = True
# Batch-norm statistics of a 1x1 convolution's output in the GEMM epilogue (hypel_seg_gemm_stats_f32) instead of a
# separate pass over Y (hypel_col_stats_partial)
STATS_EPILOGUE = True


class Storage:
    """Where a SymTensor lives for a given batch size."""

    def __init__(self, buf, nb, ld, pixmap, ch_off, c, npix):
        self.buf = buf
        self.nb = nb
        self.ld = ld
        self.pixmap = pixmap
        self.ch_off = ch_off
        self.c = c
        self.npix = npix

    def pix_off(self, p):
        q = p if self.pixmap is None else self.pixmap[p]
        return q * self.nb * self.ld + self.ch_off

    @property
    def contiguous(self):
        return self.pixmap is None

    @property
    def rows(self):
        return self.npix * self.nb


def valid_taps(h, w, k, oy, ox):
    """Taps (i, j) of a kxk SAME kernel that read a real pixel for output (oy, ox); pad_before=(k-1)//2."""
    pb = (k - 1) // 2
    out = []
    for i in range(k):
        iy = oy + i - pb
        if iy < 0 or iy >= h:
            continue
        for j in range(k):
            ix = ox + j - pb
            if 0 <= ix < w:
                out.append((i, j, iy * w + ix))
    return out


LOSS_TAIL = True  # xent / MSE sums, non-finite flag, step counter: one finaliser
MSE_PARTIALS = 1024  # include/hypel.h HYPEL_MSE_PARTIALS
DP_SYNC_WORK = 0.5  # share of the filter-gradient work before the sync point
# Gradient buckets of the data-parallel exchange: two by default (one sync point: >= 60 % of the bytes leave under the
# second half of the backward pass).  More buckets are available for large models -- one sync point per DP_BUCKET_BYTES
# once the weight gradients exceed DP_TWO_BUCKET_BYTES -- but every extra sync point flushes the merged filter-gradient
# launch early: measured on the 1-rank RCCL self-test, DUALCNN (1.03 GB of gradients, 357 ms/step) pays +0.7 % for two
# buckets, +3.8 % for five, +4.6 % for eight, while the tail a second bucket leaves exposed is ~410 MB = 2-5 ms (~1 %)
# of ring all-reduce.  Hence off by default (HYPEL_DP_TWO_BUCKET_MB=256 switches the byte rule on).
DP_TWO_BUCKET_BYTES = int(float(os.environ.get("HYPEL_DP_TWO_BUCKET_MB", "1048576")) * (1 << 20))
DP_BUCKET_BYTES = int(float(os.environ.get("HYPEL_DP_BUCKET_MB", "256")) * (1 << 20))
DP_MAX_BUCKETS = 8
HINT_OVERRIDE = {k: int(v) for k, v in (kv.split("=") for kv in os.environ.get("HYPEL_HINT_OVERRIDE", "").split(",") if kv)}
RESIDENT_BLOCKS_64 = 6 * 256  # 128x64 (and multi-segment 128x32) blocks the device holds at once
GEMM_SINGLE_SEG = 0x800  # include/hypel.h HYPEL_GEMM_SINGLE_SEG
SINGLE_SEG_HINT = True
GEMM_PAIRED_SEGS = 0x400    # ... HYPEL_GEMM_PAIRED_SEGS (bit of `accumulate`)
GEMM_BK = 32  # reduction columns per k-tile of the kernel
# fp32 products on the bf16 matrix cores with three-way split operands and six partial products (include/hypel.h
# HYPEL_GEMM_SPLIT6): "0" = never, "6" = wherever the launch is eligible and large enough (widths by column count),
# "6:W" = the same with the tile width forced to hint W (1 = 128x32, 2 = 128x64, 3 = 128x128; per-launch A/B).
# HYPEL_SPLIT_OVERRIDE="fwd:conv_dec_0=3,dgrad:conv_dec_0=0": per launch tag, 0 = fp32 MFMA kernel.
GEMM_SPLIT6 = 0x8000        # include/hypel.h HYPEL_GEMM_SPLIT6 (bit of `accumulate`)
GEMM_MULTI_SPLIT6 = 0x100   # ... HYPEL_GEMM_MULTI_SPLIT6 (bit of hypel_seg_gemm_multi_f32's tile_width)
_gs = os.environ.get("HYPEL_GEMM_SPLIT", "6").split(":")
GEMM_SPLIT = int(_gs[0] or 0)
GEMM_SPLIT_WIDTH = int(_gs[1]) if len(_gs) > 1 else 0
GEMM_SPLIT_MIN_FLOPS = float(os.environ.get("HYPEL_GEMM_SPLIT_MIN_GFLOP", "2")) * 1e9  # ... of the launch at SPLIT_NOMINAL_BATCH samples
SPLIT_NOMINAL_BATCH = 1024  # batch the size rule prices a classifier launch at (the kernel family is chosen per layer, not per batch)
SPLIT_NOMINAL_BATCH_GAN = 4096  # ... and a GAN train op's (PhasePlan: the pair batch of BASELINE's CUT configuration)
# narrow products stage a whole 128-row A tile per 32 output columns: the split costs more than the matrix rate returns
# (H13 level 1, 30 filters per branch: 382 -> 397 us forward, 103 -> 95 TFLOP/s filter gradient; per-launch A/B, round 5)
GEMM_SPLIT_MIN_N = 32
SPLIT_OVERRIDE = {k: int(v) for k, v in (kv.split("=") for kv in os.environ.get("HYPEL_SPLIT_OVERRIDE", "").split(",") if kv)}
PAIR_SEGS = True  # short data-gradient segments (k <= 16) share k-tiles
# K-slice records for the split-operand kernels (include/hypel.h HYPEL_TILE_PLAIN; round-5 verdict item 1).  The 128-wide
# split blocks have 512 resident slots (two per CU), a third of what the fp32 kernels had: a data gradient of 392 x 4 = 1 568
# equal blocks runs 3.06 rounds and pays for 4, and the 392 blocks of a multi-kernel level's data gradient are all resident
# at once and the launch lasts as long as its heaviest tile (the centre pixel sums 1.57x the mean).  The planner SIMULATES the
# launch -- list scheduling of the blocks in table order on 64 slots per XCD -- for a few slicings (every tile above a K
# threshold cut into equal slices; the tiles of each XCD's last, partly filled round cut into s slices) and takes the best
# one if it beats the unsliced launch by KSLICE_MIN_GAIN.  Slices >= 1 write plain partials to scratch, one
# hypel_reduce_splits_multi_f32 adds them in slice order.
KSLICE = True
KSLICE_MIN_GAIN = 0.10
KSLICE_OVERHEAD_K = 24   # fixed cost of a block in reduction columns (prologue, pipeline fill, epilogue)
KSLICE_MIN_K = 48        # shortest slice (reduction columns)
KSLICE_MAX_SLICES = 16
KSLICE_FRAC_MIN = 0.7    # smallest threshold (fraction of the launch's heaviest tile) above which tiles are sliced.  Same box
#                          (profiles/r6_exp_kslice.txt), the 60-filter level's data gradient, 392 tiles: unsliced 319 us; threshold
#                          0.7 (496 records) 279 us; 0.5 (688: a second round) 299; 0.25 (960) 267 -- but with three times the
#                          partial traffic; the simulation, which knows nothing of two blocks sharing a CU, prefers 0.25
KSLICE_REDUCE_COST_K = 55  # the reduce launch behind a sliced launch (~6 us), in reduction columns of a 128 x 128 block
KSLICE_SLOTS = {1: 768, 2: 512, 3: 512}  # resident blocks of the split kernels by tile-width hint (3 / 2 / 2 per CU)
GEMM_MFMA16X4 = 0x2000  # include/hypel.h HYPEL_GEMM_MFMA16X4: 128x64 blocks on the 16x16x4 MFMA (merged level, <= 16 filters)
GEMM_VAR_N = 0x4000     # ... HYPEL_GEMM_VAR_N: tile records carry their group's column count
# Merged multi-kernel levels (include/hypel.h): the nested branches of a level share one packed weight image
# W_pack[offset][Cin][C]; per output pixel and ring of input offsets ONE product on the column range of the branches
# that contain the ring.  HYPEL_MERGE_LEVELS: comma list of the passes that use it -- "fwd", "dgrad" -- or "0".
# Measured on MI355X (round 4, NOTES 4.A; per-launch and step-level A/B on one box): the merged FORWARD pays only for
# levels with <= 16 filters per branch (128x64 blocks on the 16x16x4 MFMA instead of 128x16: 146 -> 128 us); with 30 / 60
# filters a branch already fills a 32- / 64-column tile, the A stagings per FLOP do not change and the mixed-width launch
# loses 50 - 60 % (380 -> 583 us, 430 -> 687 us).  The merged DATA GRADIENT (49 instead of 84 segments per pixel) gains
# 2 - 7 % per launch.  The merged FILTER GRADIENT moved work between the three tile-width launches without shortening
# their sum (1584 -> 1593 us): removed in round 5 (NOTES 4.A keeps the numbers).  Step: 6.51 -> 6.49 ms, i.e. neutral.
MERGE_LEVELS = set(x for x in os.environ.get("HYPEL_MERGE_LEVELS", "fwd,dgrad,wgrad").split(",") if x and x != "0")
MERGE_LEVELS_MAX_COUT = 64
# per pass: widest branch (filters) the pass is merged for, taps per merged forward tile, forward tile-width hint
MERGE_PASS_MAX_COUT = {"fwd": 16, "dgrad": 1 << 20, "wgrad": 1 << 20}
# The merged FILTER GRADIENT (round 6, back from f4fd7b4^ for the split kernels): per input offset d ONE product
# dW_pack[d] = sum_p X[p + d]^T dY[p][:, col0:] of C - col0[ring] columns into a dense packed image, scattered into the HWIO
# gradient slots by hypel_copy_blocks_f32.  Columns per product grow from cout to up to 4 cout: the 30-filter level leaves the
# fp32 pipe for every ring but the outermost.  Same box (profiles/r6_exp_merged_wgrad.txt), sum of the four filter-gradient
# launches of the H13 step: unmerged 1 324 us; levels of 60 + 30 filters merged 1 227; all three levels 1 187 (+ ~8 us for the
# scatter); step 5.62 -> 5.53 ms with the first form.  Narrowest / widest branch (filters) it is used for:
MERGE_WGRAD_MIN_COUT = 1
MERGE_WGRAD_MAX_COUT = 64
# ... and for the split-operand kernels, whose cost is dominated by staging A: sharing one staged A tile between the
# branches of a ring pays for wider branches too
# Round 6, same box (profiles/r6_exp_merged_levels_split.txt; one staged + split A tile serves up to four branches):
# the 30-filter level of H13 -- fp32 kernel, unmerged: 411 us + 55 us of partial-copy reduces; merged on the 128x128 split
# blocks with the round-5 chunking (9 taps, two channel parts): 329 + 65; 16 taps per tile and NO channel parts: 328 + 38
# (= -100 us).  It needs the rows-first wave deal of the split kernels (seg_gemm.hip HYPEL_SPLIT_WAVE_ROWS_FIRST: 357 -> 329 us;
# a ring-3 group fills one column tile of four).  The 60-filter level loses merged (371 + 49 vs 362 + 39), the 15-filter level
# loses on the split kernels (128x64 blocks: 145 us vs 135 on the 16x16x4 MFMA): both keep their round-5 forms.
MERGE_FWD_MAX_COUT_SPLIT = 32
MERGE_MAX_TAPS = 0  # 0 = MAX_TAPS_PER_TILE
MERGE_SPLIT_MAX_TAPS = 16  # ... of a merged forward that runs on the split kernels
MERGE_SPLIT_KPARTS = False  # channel parts (L2_CHUNK_BYTES) for a merged forward on the split kernels
MERGE_FWD_SPLIT_NARROW = 0  # 1: the merged forward of a <= 16-filter level on the split kernels (128x64 blocks) instead of 16x16x4
MERGE_FWD_HINT = 2


# Same-box A/B of the planner's constants without editing the file: HYPEL_PLAN_SET="GEMM_SPLIT_MIN_N=64,TARGET_BLOCKS=768"
# (integers / floats; names must exist in this module).  Not read by any test or benchmark default.
for _kv in filter(None, os.environ.get("HYPEL_PLAN_SET", "").split(",")):
    _k, _, _v = _kv.partition("=")
    if not isinstance(globals().get(_k), (bool, int, float)) or _k.startswith("_"):
        raise ValueError(f"HYPEL_PLAN_SET: {_k!r} is not a numeric constant of hypelcnn_amd.plan")
    globals()[_k] = type(globals()[_k])(float(_v)) if not isinstance(globals()[_k], bool) else bool(int(_v))


class TowerPlan:
    """Buffers + launch lists of one tower at one batch size."""

    def __init__(self, tower, nb, session, loss=None, labels_c=None, external_masks=False, seed=1234, global_nb=None,
                 sync_bn=False):
        self.tower = tower
        self.nb = int(nb)
        # data parallel: samples in the GLOBAL batch this rank's nb samples are a shard of (None = nb x world)
        self.global_nb = global_nb
        # synchronised batch norm (optional, training towers of a data-parallel session): batch statistics and the
        # two backward sums run over the global batch -- one small all-gather / all-reduce per BN layer and direction
        dist_ = getattr(session, "dist", None)
        self.sync_bn = bool(sync_bn) and dist_ is not None and tower.is_training
        self.world = dist_[0] if dist_ is not None else 1
        self.sess = session
        self.be = session.backend
        self.training = tower.is_training
        self.loss = loss
        self.external_masks = external_masks
        self.seed = seed
        self.buffers = {}  # name -> flat tensor
        self.tables = []  # keep uploaded tables alive
        self.storage = {}  # id(owner SymTensor) -> Storage (for owners)
        self.grad_written = {}  # id(owner) -> bool
        self.fwd = []
        self.bwd = []
        self.node_aux = {}
        self.mask_bufs = {}
        self.scratch_sizes = {"scratch_partial": 1, "scratch_wgrad": 1, "scratch_red": 2048, "sums": 2}
        self._pending_scratch = []
        self.param_written = set()  # variables whose gradient was already written in this plan (shared weights)
        self.sync_points = []  # (index into bwd, lo, hi): grads[lo:hi] are final there (data-parallel overlap)
        self._build()

    # ---- which variables does this plan train / which tensors need a gradient (overridden by PhasePlan) ----
    def _trains(self, variables):
        return True

    def _needs_grad(self, t):
        return t.owner.needs_grad

    def _param_acc(self, var):
        acc = 1 if var.name in self.param_written else 0
        self.param_written.add(var.name)
        return acc

    # ------------------------------------------------------------------ buffers
    def _alloc(self, name, n, dtype=None):
        import torch
        t = self.be.zeros(max(int(n), 1), dtype or torch.float32)
        self.buffers[name] = t
        return t

    def _ref(self, name, off=0):
        return Ref(self.buffers[name], off)

    def storage_of(self, t):
        own = t.owner
        st = self.storage[id(own)]
        if t.root is None:
            return st
        # (st.ch_off: the owner may itself be a row-block view of a batched application group, PhasePlan)
        return Storage(st.buf, st.nb, st.ld, t.pixmap, st.ch_off + t.ch_off, t.c, t.npix)

    def grad_storage_of(self, t):
        st = self.storage_of(t)
        return Storage("g:" + st.buf, st.nb, st.ld, st.pixmap, st.ch_off, st.c, st.npix)

    def _new_value(self, t, name):
        buf = self._alloc(name, t.npix * self.nb * t.c)
        st = Storage(name, self.nb, t.c, None, 0, t.c, t.npix)
        self.storage[id(t)] = st
        return st

    def _ensure_grad(self, owner):
        st = self.storage[id(owner)]
        gname = "g:" + st.buf
        if gname not in self.buffers:
            self._alloc(gname, owner.npix * self.nb * owner.c)
        self.grad_written.setdefault(id(owner), False)  # (row-block views of a batched group share one gradient buffer)
        return gname

    def _grad_target(self, t):
        """Returns (Storage of the gradient view, accumulate flag) and marks it written; emits a
        zero fill first when the first contribution only covers a view of the buffer."""
        own = t.owner
        self._ensure_grad(own)
        gst = self.grad_storage_of(t)
        written = self.grad_written[id(own)]
        if not written and t.root is not None:
            full = self.storage[id(own)]
            self.bwd.append(Launch("fill_f32", (self._ref(gst.buf, full.ch_off), full.rows * full.ld, 0.0),
                                   tag="zero-grad-view"))
            written = True
        self.grad_written[id(own)] = True
        return gst, 1 if written else 0

    # ------------------------------------------------------------------ parameters
    def _p(self, var):
        return Ref(self.sess.params, var.offset)

    def _g(self, var):
        return Ref(self.sess.grads, var.offset)

    def _s(self, var):
        return Ref(self.sess.state, var.offset)

    @staticmethod
    def _assert_contiguous(vars_):
        for a, b in zip(vars_[:-1], vars_[1:]):
            if b.offset != a.offset + a.size:
                raise RuntimeError(f"variables {a.name} / {b.name} of a merged level are not contiguous")

    # ------------------------------------------------------------------ gemm emission
    def _split_k(self, tables, n, lda, ta, ldb, tb, ldc):
        """FC-shaped products have too few 128x128 output tiles to fill 256 CUs (fc_0 forward: 64 blocks;
        image_gen_net_4 data gradient: 32).  When the output is one dense range, cut every group's K into S
        slices that write partial slabs; hypel_reduce_splits_f32 sums them in a fixed order."""
        groups = tables.groups
        if ldc != n or not groups:
            return None
        n_nt = (n + 63) // 64 if n > 32 else 1
        blocks = sum((rows + GEMM_BM - 1) // GEMM_BM for _, _, rows in groups) * n_nt
        kmax = max(sum(k for _, _, k in gs) for _, gs, _ in groups)
        if blocks >= SPLITK_BELOW or kmax < 512:
            return None
        S = min((SPLITK_TARGET + blocks - 1) // blocks, kmax // 128, 32)
        if S < 2:
            return None
        order = sorted(groups, key=lambda g: g[0])
        c_min = order[0][0]
        pos = c_min
        for c_off, _, rows in order:  # dense, gap-free cover
            if c_off != pos:
                return None
            pos += rows * n
        count = pos - c_min
        if c_min % n:
            return None
        a_ks = lda if ta else 1
        b_ks = 1 if tb else ldb
        out = GemmTables()
        for c_off, gs, rows in groups:
            ktot = sum(k for _, _, k in gs)
            cuts = [((ktot * s // S) + 31) // 32 * 32 for s in range(S)] + [ktot]
            cuts = [min(c, ktot) for c in cuts]
            parts = [[] for _ in range(S)]
            base = 0
            for (a_off, b_off, k) in gs:
                for s in range(S):
                    lo, hi = max(cuts[s], base), min(cuts[s + 1], base + k)
                    if hi > lo:
                        parts[s].append((a_off + (lo - base) * a_ks, b_off + (lo - base) * b_ks, hi - lo))
                base += k
            for s in range(S):
                out.add_group(s * count + (c_off - c_min), parts[s], rows)
        return out, S, c_min, count

    @staticmethod
    def _tile_hint(tables, n, ta, tb, folded):
        """Tile-width hint for hypel_seg_gemm_f32 (bits 8-9 of `accumulate`), from the per-launch A/B measurements in
        profiles/r1_gemm_tile_choice*.txt (HYPELCNN at N=1024 and DUALCNN at N=512): 1 = 128x32, 2 = 128x64.
        Wide tiles win whenever a launch has plenty of blocks; narrow ones balance small launches, and the data
        gradients that also gather the shortcut gradient in their epilogue (more, shorter epilogues overlap better)."""
        if n <= 32:
            return 0
        n_tiles = sum((rows + GEMM_BM - 1) // GEMM_BM for _, _, rows in tables.groups)
        blocks64 = n_tiles * ((n + 63) // 64)
        if tb and not ta:
            ks = [k for _, segs, _ in tables.groups for _, _, k in segs]
            if ks and sum(ks) / len(ks) < 48:
                # short segments: per-k-tile overhead dominates, keep the wide tile -- unless that leaves the launch
                # with under ~1000 blocks (the n = 120 data gradient of the 15-filter level: 142 us on 128x32 tiles,
                # 154 on 128x64, round-2 A/B with paired segments)
                if blocks64 < 1000:
                    return 1
                # ... and unless the 64-wide tiling overflows the resident capacity (6 blocks per CU) by a few blocks: the
                # level-1 data gradient of H13 (392 x 4 = 1568 on 1536) then ends in a round of 32 lone blocks, and the
                # 32-wide tiling (3136 blocks, two even rounds) is 46 us faster per step
                over = blocks64 % RESIDENT_BLOCKS_64
                return 1 if blocks64 < 2 * RESIDENT_BLOCKS_64 and 0 < over <= RESIDENT_BLOCKS_64 // 10 else 2
            if folded and blocks64 < 4000:
                return 1
            return 1 if blocks64 < 900 else 2
        if ta:
            return 1 if blocks64 < 900 else 2
        # forward.  Round-2 per-launch A/B of the three tile widths (tools/gemm_microbench.py, HYPEL_GEMM_FORCE_WIDTH):
        # single-segment products (1x1 convolutions, n <= 512) run fastest on 128x32 tiles -- except n = 480, which
        # tiles 5 x 96 without padding and whose 392 x 5 blocks are just under two resident rounds of the 128x96
        # variant (conv_dec_0 218 vs 227 (32) vs 244 (64) us; conv_enc_2 129 vs 134 vs 138) -- multi-segment levels
        # and the very wide dense layers on 128x64.
        if n <= 512 and FWD_HINT_R2 and all(len(segs) == 1 for _, segs, _ in tables.groups):
            if n >= 480 and n % 96 == 0:
                return 3
            return 1
        return 1 if blocks64 < 768 else 2

    def nominal_batch(self):
        return SPLIT_NOMINAL_BATCH

    @staticmethod
    def _split6_width(n):
        """Tile-width hint of a split-operand launch by its column count: 128x128 blocks unless the last column tile
        would be more than half empty."""
        if GEMM_SPLIT_WIDTH:
            return GEMM_SPLIT_WIDTH
        if n <= 32:
            return 1
        rem = n % 128
        return 2 if n <= 64 or 0 < rem <= 64 and n < 256 else 3

    def _split6(self, tag, tables, n, ta, tb, flags, paired, in_multi=False):
        """0 = fp32 MFMA kernel, else the tile-width hint of the split-operand kernel for this launch.  in_multi: a product
        of the merged filter-gradient launch -- it shares its launch with the other products of its width class, so its
        own size does not matter (left on the fp32 kernels, the few small ones formed a 102 us launch of 4.4 GFLOP).
        The choice is a function of the LAYER, not of the batch: the size rule prices the launch at SPLIT_NOMINAL_BATCH
        samples (every product of the path is linear in the batch), so that a sample meets the same arithmetic at batch
        64, 512 and 1024 and on 1 or 8 ranks (round-5 verdict: the FLOP count of the launch itself made it batch-dependent)."""
        eligible = n > 16 and not (ta and tb) and not (flags & ~GEMM_VAR_N) and not paired
        if tag in SPLIT_OVERRIDE:
            w = SPLIT_OVERRIDE[tag]
            if w and not eligible:
                raise ValueError(f"HYPEL_SPLIT_OVERRIDE: launch {tag!r} (n = {n}, trans = {int(bool(ta))}{int(bool(tb))}, "
                                 f"flags = {flags:#x}, paired segments = {bool(paired)}) cannot run on the split-operand "
                                 "kernels (they need n > 16, a plain NN / NT / TN product, no 16x16x4 / activation epilogue)")
            if w not in (0, 1, 2, 3):
                raise ValueError(f"HYPEL_SPLIT_OVERRIDE: width hint {w} for {tag!r} (0 = fp32 kernel, 1 / 2 / 3 = 128x32 / x64 / x128)")
            return w
        if GEMM_SPLIT != 6 or n <= GEMM_SPLIT_MIN_N or not eligible:
            return 0
        if in_multi:
            return self._split6_width(n)
        macs = sum(rows * sum(k for _, _, k in gs) * tables.n_of(gi, n) for gi, (_, gs, rows) in enumerate(tables.groups))
        if 2 * macs * self.nominal_batch() < GEMM_SPLIT_MIN_FLOPS * self.nb:
            return 0
        return self._split6_width(n)

    @staticmethod
    def _kslice_makespan(shares, extra, slots):
        """Simulated duration of a launch: shares = eight lists of block costs in dispatch order, extra = block costs dealt
        heaviest first to the end of the least loaded share (GemmTables.finalize does the same); every XCD runs its list on
        slots / 8 block slots, next block to the slot that frees first."""
        import heapq
        shares = [list(sh) for sh in shares]
        work = [sum(sh) for sh in shares]
        for c in sorted(extra, reverse=True):
            x = min(range(8), key=lambda i: (work[i], i))
            shares[x].append(c)
            work[x] += c
        per = max(1, slots // 8)
        worst = 0.0
        for sh in shares:
            free = [0.0] * per
            for c in sh:
                t = heapq.heappop(free)
                heapq.heappush(free, t + c)
            worst = max(worst, max(free))
        return worst

    def _kslice(self, tables, n, ta, tb, lda, ldb, width_hint):
        """Choose K-slices for a launch on the split kernels (see KSLICE above).  Returns None (leave the launch alone) or
        {group index: number of slices}."""
        groups = tables.groups
        bn = {1: 32, 2: 64, 3: 128}[width_hint]
        slots = KSLICE_SLOTS[width_hint]
        ct = (n + bn - 1) // bn
        ov = KSLICE_OVERHEAD_K
        # tile records in dispatch order (GemmTables.finalize: key-major, heavy first), one entry per record
        order = []
        for gi, (c_off, gs, rows) in enumerate(groups):
            ksum = sum(k for _, _, k in gs)
            for m0 in range(0, rows, GEMM_BM):
                key = (tables.keys[gi] if tables.keys[gi] is not None else m0 // GEMM_BM, tables.subkeys[gi])
                order.append((key, -ksum * min(GEMM_BM, rows - m0), gi, ksum))
        order.sort(key=lambda t: (t[0], t[1]))

        def shares_of(recs):
            q, r = divmod(len(recs), 8)
            out, pos = [], 0
            for x in range(8):
                cnt = q + (1 if x < r else 0)
                out.append(recs[pos:pos + cnt])
                pos += cnt
            return out

        def evaluate(slices):  # slices: {gi: s}
            main = [t for t in order if slices.get(t[2], 1) == 1]
            extra, pieces = [], 0
            for t in order:
                s_ = slices.get(t[2], 1)
                if s_ > 1:
                    extra += [t[3] / s_ + ov] * (s_ * ct)
                    pieces += (s_ - 1) * ct
            sh = [[t[3] + ov for t in share for _ in range(ct)] for share in shares_of(main)]
            # + what the partials cost: a 128 x bn partial is written, read back and added into the output (~3 passes of
            # 512 bn bytes at ~4 TB/s; one reduction column of a 128 x 128 block is ~0.11 us), + the reduce launch (~6 us)
            traffic = pieces * (3 * 512 * bn / 4e12) / (0.11e-6 * bn / 128)
            return self._kslice_makespan(sh, extra, slots) + (traffic + KSLICE_REDUCE_COST_K if pieces else 0)

        base = evaluate({})
        kmax = max(t[3] for t in order)
        cands = []
        for f in (0.75, 0.6, 0.5, 0.4, 0.33, 0.25):  # every tile above a threshold, in equal slices
            if f < KSLICE_FRAC_MIN:
                continue
            T = max(kmax * f, KSLICE_MIN_K)
            sl = {}
            for gi, (c_off, gs, rows) in enumerate(groups):
                ksum = sum(k for _, _, k in gs)
                s_ = min(KSLICE_MAX_SLICES, -(-ksum // int(T)), max(1, ksum // KSLICE_MIN_K))
                if s_ > 1:
                    sl[gi] = s_
            if sl:
                cands.append(sl)
        per = max(1, slots // 8)
        single_tile_groups = all(rows <= GEMM_BM for _, _, rows in groups)
        for s_ in (2, 3, 4, 6, 8, 12, 16):  # the tiles of each XCD's last, partly filled round
            sl = {}
            for share in shares_of(order):
                tail_blocks = (len(share) * ct) % per
                if tail_blocks == 0 or tail_blocks * s_ > per + per // 4:
                    continue
                for t in share[len(share) - (-(-tail_blocks // ct)):]:
                    if t[3] // s_ >= KSLICE_MIN_K:
                        sl[t[2]] = s_
            if sl and single_tile_groups:  # (a group = one tile record there: slicing a group slices exactly that tile)
                cands.append(sl)
        best, best_t = None, base
        for sl in cands:
            t = evaluate(sl)
            if t < best_t:
                best, best_t = sl, t
        if best is None or best_t > base * (1.0 - KSLICE_MIN_GAIN):
            return None
        return best

    @staticmethod
    def _group_per_tile(tables, lda, ldc):
        """The same launch with one group per 128-row tile record (A not transposed: rows of A = rows of C)."""
        out = GemmTables()
        for gi, (c_off, gs, rows) in enumerate(tables.groups):
            for m0 in range(0, rows, GEMM_BM):
                key = tables.keys[gi] if tables.keys[gi] is not None else m0 // GEMM_BM
                out.add_group(c_off + m0 * int(ldc), [(a + m0 * int(lda), b, k) for (a, b, k) in gs], min(GEMM_BM, rows - m0),
                              key=key, subkey=tables.subkeys[gi])
        return out

    @staticmethod
    def _cut_segments(gs, s_, a_ks, b_ks):
        """Cut a segment list into s_ slices of (nearly) equal reduction length, at multiples of 16 columns inside a
        segment; a_ks / b_ks = element distance of one reduction column in A / B."""
        ktot = sum(k for _, _, k in gs)
        cuts = [min(ktot, (ktot * i // s_ + 15) // 16 * 16) for i in range(s_)] + [ktot]
        parts = [[] for _ in range(s_)]
        base = 0
        for (a_off, b_off, k) in gs:
            for i in range(s_):
                lo, hi = max(cuts[i], base), min(cuts[i + 1], base + k)
                if hi > lo:
                    parts[i].append((a_off + (lo - base) * a_ks, b_off + (lo - base) * b_ks, hi - lo))
            base += k
        return parts

    def _emit_gemm(self, lst, tables, n, a_ref, lda, ta, b_ref, ldb, tb, c_ref, ldc, bias_ref, accumulate, tag,
                   allow_split=True, res=None, stats=None, pair=False, hint=None, flags=0, kslice=False):
        """res = (ref, ld, start_ref or None): fold a shortcut gradient into the epilogue (hypel_seg_gemm_res_f32).
        stats = floats of the per-tile statistics scratch: hypel_seg_gemm_stats_f32 (single group, no accumulate).
        kslice: the launch may be balanced with K-slice records (dense output, ldc = n)."""
        split = self._split_k(tables, n, lda, ta, ldb, tb, ldc) if allow_split and res is None else None
        if split is not None:
            stab, S, c_min, count = split
            pos = len(lst)
            self._emit_gemm(lst, stab, n, a_ref, lda, ta, b_ref, ldb, tb, c_ref, ldc, None, 0, tag + "/splitk",
                            allow_split=False)
            self._scratch(lst[pos], 6, "scratch_wgrad", S * count)
            l2 = Launch("reduce_splits_f32", (None, count, S, c_ref + c_min, count, int(accumulate), bias_ref, int(n), 0),
                        nbytes=4 * count * (S + 1), tag="splitk-reduce")
            self._scratch(l2, 0, "scratch_wgrad", S * count)
            lst.append(l2)
            return
        pair = bool(pair and PAIR_SEGS and not ta and tb and n > 16 and stats is None)
        if hint is None:
            hint = self._tile_hint(tables, n, ta, tb, res is not None) if TILE_HINTS else 0
        if HINT_OVERRIDE and tag in HINT_OVERRIDE:  # per-launch A/B: HYPEL_HINT_OVERRIDE="fwd:conv_enc_2=1,dgrad:fc_0=2"
            hint = HINT_OVERRIDE[tag]
        single_seg = bool(SINGLE_SEG_HINT and not ta and
                          all(len(segs) == 1 for _, segs, _ in tables.groups))
        sp6 = self._split6(tag, tables, n, ta, tb, flags,
                           pair and any(k <= 16 for _, gs, _ in tables.groups for _, _, k in gs))
        if sp6:
            pair, hint, single_seg = False, sp6, False
            accumulate = int(accumulate) | GEMM_SPLIT6
        reduce_after = None
        if (sp6 and kslice and KSLICE and stats is None and not flags and int(ldc) == int(n) and not (ta and tb)
                and tables.groups and not any(tables.ns)):
            # (one group per 128-row tile record first: a 1x1 convolution's data gradient is ONE group of 50 176 rows)
            per_tile = tables if ta or all(rows <= GEMM_BM for _, _, rows in tables.groups) else self._group_per_tile(tables, lda, ldc)
            slices = self._kslice(per_tile, n, ta, tb, lda, ldb, sp6)
            if slices:
                tables, reduce_after = self._apply_kslices(per_tile, slices, n, lda, ta, ldb, tb, c_ref, ldc, tag)
        garr, sarr, tarr, macs = tables.finalize(n, pair=pair)
        if len(tarr) == 0:
            return
        if pair and tables.paired:
            accumulate = int(accumulate) | GEMM_PAIRED_SEGS
        g_t, s_t, t_t = self.be.upload(garr), self.be.upload(sarr), self.be.upload(tarr)
        self.tables += [g_t, s_t, t_t]
        if single_seg and not flags:
            accumulate = int(accumulate) | GEMM_SINGLE_SEG
        accumulate = int(accumulate) | int(flags)
        args = (a_ref, int(lda), int(ta), b_ref, int(ldb), int(tb), c_ref, int(ldc), int(n), Ref(g_t), Ref(s_t),
                Ref(t_t), int(len(tarr)), bias_ref, int(accumulate) | (hint << 8))
        name = "seg_gemm_f32"
        if res is not None:
            name = "seg_gemm_res_f32"
            args = args + (res[0], int(res[1]), res[2])
        elif stats is not None:
            assert len(tables.groups) == 1 and not (int(accumulate) & 1) and int(ldc) == int(n)
            name = "seg_gemm_stats_f32"
            args = args + (None,)
        l = Launch(name, args, flops=2 * macs, nbytes=tables.compulsory_bytes(n, lda, ta, ldb, tb), tag=tag)
        if stats is not None:
            self._scratch(l, len(args) - 1, "scratch_partial", stats)
        lst.append(l)
        if reduce_after is not None:
            l.meta["kslices"] = reduce_after.meta["kslices"]
            lst.append(reduce_after)

    def _apply_kslices(self, tables, slices, n, lda, ta, ldb, tb, c_ref, ldc, tag):
        """The tables with the chosen groups cut into K-slices + the launch that adds the partials to the output."""
        from .backend import TILE_PLAIN
        a_ks = int(lda) if ta else 1
        b_ks = 1 if tb else int(ldb)
        kid = self.__dict__.setdefault("_kslice_bufs", 0)
        self._kslice_bufs = kid + 1
        need = sum((s_ - 1) * tables.groups[gi][2] * int(ldc) for gi, s_ in slices.items())
        sname = f"kslice:{kid}"
        self._alloc(sname, need)
        base = Ref(self.sess.params)

        def rel(ref, to):
            d = ref.ptr() - to.ptr()
            assert d % 4 == 0
            return d // 4

        scratch_from_c = rel(self._ref(sname), c_ref)
        out = GemmTables()
        entries, spos = [], 0
        for gi, (c_off, gs, rows) in enumerate(tables.groups):
            s_ = slices.get(gi, 1)
            if s_ == 1:
                out.add_group(c_off, gs, rows, key=tables.keys[gi], subkey=tables.subkeys[gi])
                continue
            parts = self._cut_segments(gs, s_, a_ks, b_ks)
            region = rows * int(ldc)
            out.add_group(c_off, parts[0], rows, key=tables.keys[gi], subkey=tables.subkeys[gi], tail=True)
            for i in range(1, s_):
                out.add_group(scratch_from_c + spos + (i - 1) * region, parts[i], rows, flags=TILE_PLAIN, tail=True)
            entries.append((rel(self._ref(sname, spos), base), rel(c_ref, base) + c_off, region, region, s_ - 1, 1))
            spos += (s_ - 1) * region
        earr = np.array(entries, REDUCE_ENTRY_DTYPE)
        e_t = self.be.upload(earr)
        self.tables.append(e_t)
        red = Launch("reduce_splits_multi_sized_f32", (base, Ref(e_t), int(len(earr)), int(max(e[3] for e in entries))),
                     nbytes=4 * sum(cnt * (S + 2) for (_, _, _, cnt, S, _) in entries), tag="kslice-reduce")
        red.meta = {"kslices": {"tiles": len(entries), "slices": int(sum(slices.values())), "of": tag}}
        return out, red

    def _dp_sync_node(self):
        """Data-parallel overlap: [(node index, lo, hi)] -- walking backward, the nodes after which the weight gradients
        at flat offsets [lo, hi) are final.  Small models (H13: 32.6 MB) get ONE point: the node after which >= 60 % of the
        weight-gradient bytes are final; large ones one point per DP_BUCKET_BYTES.  Weights are laid out in creation order
        (Session.finalize_variables), i.e. in node order, so that range is one contiguous tail.
        The merged filter-gradient launch is flushed at this point, so the point also has to cut the filter-gradient
        WORK into two useful halves: at the first node that satisfies the byte rule (H13: fc_0, 75 % of the bytes but 8 %
        of the multiply-adds) the early launch held a handful of badly shaped products and cost 0.13 ms per step; the
        walk therefore continues until >= DP_SYNC_WORK of the filter-gradient multiply-adds are behind it, as long as
        >= 15 % of them -- backward time to hide the all-reduce under -- remain (H13: the first level, 66 % / 34 %)."""
        if os.environ.get("HYPEL_DP_NO_SYNC") == "1":  # diagnostic: one graph, one all-reduce after the backward pass
            return None
        sized = []
        for idx, node in enumerate(self.tower.nodes):
            if isinstance(node, G.LinearNode):
                ws = [b.w for b in node.branches]
                if ws and all(w.trainable and w.offset is not None for w in ws):
                    macs = sum(w.size for w in ws) * max(1, node.out.npix)  # x batch: the same factor for every node
                    sized.append((idx, min(w.offset for w in ws), max(w.offset + w.size for w in ws), macs))
        if len(sized) < 2:
            return None
        total = sum(hi - lo for _, lo, hi, _ in sized)
        total_macs = sum(m for _, _, _, m in sized)
        hi_all = max(hi for _, _, hi, _ in sized)
        if 4 * total > DP_TWO_BUCKET_BYTES:
            # large model: one sync point per DP_BUCKET_BYTES of finished weight gradients (walking backward), the last
            # one early enough that >= 10 % of the filter-gradient work is still ahead to hide it
            n_buckets = max(3, min(DP_MAX_BUCKETS, -(-4 * total // DP_BUCKET_BYTES)))
            points, acc, acc_macs = [], 0, 0
            for k in range(len(sized) - 1, 0, -1):
                idx, lo, hi, macs = sized[k]
                acc += hi - lo
                acc_macs += macs
                tail_lo = min(l for _, l, _, _ in sized[k:])
                contiguous = sum(h - l for _, l, h, _ in sized[k:]) == hi_all - tail_lo
                if not contiguous or acc_macs > 0.9 * total_macs:
                    continue
                if acc >= total * (len(points) + 1) / n_buckets:
                    points.append((idx, tail_lo, hi_all))
                    if len(points) == n_buckets - 1:
                        break
            return points or None
        acc = acc_macs = 0
        best = None
        for k in range(len(sized) - 1, 0, -1):  # never the first layer: nothing would be left to overlap with
            idx, lo, hi, macs = sized[k]
            acc += hi - lo
            acc_macs += macs
            tail_lo = min(l for _, l, _, _ in sized[k:])
            contiguous = sum(h - l for _, l, h, _ in sized[k:]) == hi_all - tail_lo
            if acc >= 0.6 * total and contiguous:
                if best is None:
                    best = (idx, tail_lo, hi_all)
                if acc_macs > 0.85 * total_macs:
                    break  # too little backward left behind this point
                best = (idx, tail_lo, hi_all)
                if acc_macs >= DP_SYNC_WORK * total_macs:
                    break
        return [best] if best is not None else None

    # ------------------------------------------------------------------ consumers / epilogue fusion
    @staticmethod
    def _inputs_of(node):
        if isinstance(node, G.LinearNode):
            return list(node.sources) + [r for r, _ in node.residuals]
        if isinstance(node, G.PostNode):
            return [node.src] + [r for r, _ in node.residuals]
        if hasattr(node, "srcs"):
            return list(node.srcs)
        return [node.src]

    def _build(self):
        tw = self.tower
        nb = self.nb
        # inputs: the loader hands over NHWC batches; keep a persistent staging tensor and convert
        for name, t in tw.inputs.items():
            if t.hw is None:
                self._alloc("in:" + name, nb * t.c)
                self.storage[id(t)] = Storage("in:" + name, nb, t.c, None, 0, t.c, 1)
            else:
                self._alloc("in:" + name, nb * t.npix * t.c)  # NHWC as delivered
                st = self._new_value(t, "pnc:" + name)
                self.fwd.append(Launch("nhwc_to_pnc", (self._ref("in:" + name), self._ref(st.buf), nb, t.npix, t.c,
                                                       st.ld), nbytes=8 * nb * t.npix * t.c, tag="layout"))
        if self.training:
            import torch
            self._alloc("step_ctr", 1, torch.int64)
        pack_pos = len(self.fwd)
        for idx, node in enumerate(tw.nodes):
            if isinstance(node, G.LinearNode):
                self._fwd_linear(idx, node)
            elif isinstance(node, G.PostNode):
                self._fwd_post(idx, node)
            elif isinstance(node, G.LRNNode):
                self._fwd_lrn(idx, node)
            else:
                raise TypeError(node)
        self._emit_level_packs(pack_pos)
        if self.loss is not None:
            self._emit_loss()
        if self.training and self.loss is not None:
            sync_pts = self._dp_sync_node() if getattr(self.sess, "dist", None) is not None else None
            sync_map = {p[0]: p for p in (sync_pts or [])}
            for idx in range(len(tw.nodes) - 1, -1, -1):
                node = tw.nodes[idx]
                if isinstance(node, G.LinearNode):
                    self._bwd_linear(idx, node)
                elif isinstance(node, G.PostNode):
                    self._bwd_post(idx, node)
                elif isinstance(node, G.LRNNode):
                    self._bwd_lrn(idx, node)
                if idx in sync_map:
                    # every weight gradient at flat offsets >= lo is final once the side stream is joined: the session
                    # starts the all-reduce of what no earlier point covered here, under the rest of the backward pass
                    sync_at = sync_map[idx]
                    self._flush_wgrads()
                    self.sync_points.append((len(self.bwd), sync_at[1], sync_at[2]))
            self._flush_wgrads()
            if tw.n_dropout and not self.external_masks and not getattr(self, "_step_in_loss", False):
                self.bwd.append(Launch("step_inc", (self._ref("step_ctr"),), tag="rng"))
        # shared scratch (stream order makes reuse safe)
        self._finish_scratch()

    def _finish_scratch(self):
        """Shared scratch (stream order makes reuse safe): allocate every region at its largest request, resolve the
        launches' scratch arguments."""
        for name, size in self.scratch_sizes.items():
            self._alloc(name, size)
        for launch, pos, name in self._pending_scratch:
            args = list(launch.args)
            args[pos] = self._ref(name)
            launch.args = tuple(args)

    def _scratch(self, launch, pos, name, size=0):
        self.scratch_sizes[name] = max(self.scratch_sizes.get(name, 1), int(size))
        self._pending_scratch.append((launch, pos, name))

    # ------------------------------------------------------------------ LinearNode forward
    def _vec_params(self, node):
        """Concatenated per-channel vectors of a (possibly merged) node."""
        brs = node.branches
        aux = {}
        if node.has_bn:
            for i, key in enumerate(("beta", "mm", "mv")):
                vs = [b.bn[i] for b in brs]
                self._assert_contiguous(vs)
                aux[key] = vs[0]
        if node.has_bias:
            vs = [b.bias for b in brs]
            self._assert_contiguous(vs)
            aux["bias"] = vs[0]
        ws = [b.w for b in brs]
        self._assert_contiguous(ws)
        aux["w0"] = ws[0]
        aux["wsize"] = sum(w.size for w in ws)
        return aux

    # ------------------------------------------------------------------ merged multi-kernel levels
    def _level_layout(self, idx, node):
        """Packed layout of a multi-kernel level whose branches are nested odd kernels of equal width on one contiguous
        source (HYPELCNNModel.py:167-183), or None.  An input offset (dy, dx) at ring r = max(|dy|, |dx|) belongs to the
        branches with (k - 1) / 2 >= r -- a suffix of the concat order -- i.e. to the output columns [col0[r], C).
        Offsets are numbered ring-major; W_pack[d] is [Cin x C] (columns below col0 are never touched)."""
        cache = self.__dict__.setdefault("_level_layouts", {})
        if idx in cache:
            return cache[idx]
        lay = None
        brs = node.branches
        # (a biased level -- DUALCNN -- keeps its unmerged forward: the bias rides in the GEMM epilogue / the tap-split
        # reduce; its data and filter gradients do not see the bias at all)
        if (MERGE_LEVELS and node.kind == "conv" and len(brs) >= 2
                and self.nb >= TAP_SPLIT_MIN_BATCH and len(node.sources) == 1):
            ks = [b.k for b in brs]
            co = brs[0].cout
            src = node.sources[0]
            ok = (all(k % 2 == 1 for k in ks) and all(a < b for a, b in zip(ks, ks[1:])) and
                  all(b.cout == co for b in brs) and co <= MERGE_LEVELS_MAX_COUT and src.hw is not None and
                  self.storage_of(src).contiguous)
            if ok:
                rmax = (ks[-1] - 1) // 2
                first = [next(i for i, k in enumerate(ks) if (k - 1) // 2 >= r) for r in range(rmax + 1)]
                offs = []
                for r in range(rmax + 1):
                    offs += [(dy, dx, r) for dy in range(-r, r + 1) for dx in range(-r, r + 1) if max(abs(dy), abs(dx)) == r]
                C = co * len(brs)
                lay = dict(offs=offs, index={(dy, dx): d for d, (dy, dx, _) in enumerate(offs)}, rmax=rmax, first=first,
                           col0=[f * co for f in first], co=co, C=C, cin=src.c, buf=f"wpack:{idx}", dense_off=[],
                           has_bias=node.has_bias)
                pos = 0
                for (_, _, r) in offs:  # dense image of the packed filter gradient: [d] -> [Cin x (C - col0[r])]
                    lay["dense_off"].append(pos)
                    pos += src.c * (C - lay["col0"][r])
                lay["dense_size"] = pos
                assert pos == sum(b.w.size for b in brs)
                if any(w_ in MERGE_LEVELS and co <= max(MERGE_PASS_MAX_COUT[w_], MERGE_FWD_MAX_COUT_SPLIT if GEMM_SPLIT == 6 else 0)
                       and not (w_ == "fwd" and node.has_bias) for w_ in ("fwd", "dgrad")):
                    # these passes read the packed image (the filter gradient does not)
                    lay["packed"] = True
                    self._alloc(lay["buf"], len(offs) * src.c * C)
        cache[idx] = lay
        return lay

    def _level_pass(self, idx, node, what):
        """The level's packed layout if pass `what` ("fwd" / "dgrad" / "wgrad") uses the merged form, else None."""
        lay = self._level_layout(idx, node)
        cap = MERGE_PASS_MAX_COUT[what]
        if what == "fwd" and GEMM_SPLIT == 6:
            cap = max(cap, MERGE_FWD_MAX_COUT_SPLIT)
        if lay is None or what not in MERGE_LEVELS or lay["co"] > cap or (what == "fwd" and lay["has_bias"]):
            return None
        if what == "wgrad" and not (GEMM_SPLIT == 6 and MERGE_WGRAD and MERGE_WGRAD_MIN_COUT <= lay["co"] <= MERGE_WGRAD_MAX_COUT):
            return None  # (on the fp32 kernels it moved work between the width classes without shortening their sum, NOTES 4.A)
        return lay

    def _emit_level_packs(self, pos):
        """ONE hypel_copy_blocks_f32 in front of the forward pass (inserted at launch position `pos`): every
        (branch, tap) slice [Cin x cout] of every merged level goes to its offset / column range of the level's packed
        image (the variables keep their TF layouts)."""
        from .backend import COPY_BLOCK_DTYPE
        base = Ref(self.sess.params)
        ents = []
        for idx, node in enumerate(self.tower.nodes):
            lay = self.__dict__.get("_level_layouts", {}).get(idx)
            if lay is None or not lay.get("packed"):
                continue
            dst0 = (self._ref(lay["buf"]).ptr() - base.ptr()) // 4
            for bi, b in enumerate(node.branches):
                pb = (b.k - 1) // 2
                for i in range(b.k):
                    for j in range(b.k):
                        d = lay["index"][(i - pb, j - pb)]
                        ents.append((b.w.offset + (i * b.k + j) * lay["cin"] * lay["co"],
                                     dst0 + d * lay["cin"] * lay["C"] + bi * lay["co"], lay["cin"], lay["co"], lay["co"],
                                     lay["C"], 0, 0))
        if not ents:
            return
        t = self.be.upload(np.array(ents, COPY_BLOCK_DTYPE))
        self.tables.append(t)
        self.fwd.insert(pos, Launch("copy_blocks_f32", (base, Ref(t), len(ents), max(e[2] * e[3] for e in ents)),
                                    nbytes=8 * sum(e[2] * e[3] for e in ents), tag="level-pack"))

    def _fwd_level_merged(self, idx, node, lay, s_st, ybuf, c, h, w):
        """Forward pass of a merged level: per output pixel and ring r the product
        Y[p][:, col0[r]:] (+)= sum_{d in ring r, valid} X[p + d] . W_pack[d][:, col0[r]:], rings with more than
        MAX_TAPS_PER_TILE offsets cut into chunks; every (ring chunk, channel part) writes its own copy of Y (copy 0 is Y
        itself), summed per branch by hypel_reduce_splits_f32 as for the tap splits of the unmerged form.
        Returns the number of copies."""
        nb = self.nb
        src = node.sources[0]
        rows_all = node.out.npix * nb
        C, cin = lay["C"], lay["cin"]
        narrow16 = lay["co"] <= 16 and C <= 64 and not (MERGE_FWD_SPLIT_NARROW and GEMM_SPLIT == 6)
        on_split = GEMM_SPLIT == 6 and not narrow16 and C > GEMM_SPLIT_MIN_N  # (the size rule of _split6 may still say fp32)
        ring_sizes = [sum(1 for o in lay["offs"] if o[2] == r) for r in range(lay["rmax"] + 1)]
        max_taps = MERGE_MAX_TAPS or (MERGE_SPLIT_MAX_TAPS if on_split else MAX_TAPS_PER_TILE)
        S_r = [max(1, -(-sz // max_taps)) for sz in ring_sizes]
        ws = h * w * GEMM_BM * src.c * 4
        kp_n = max(1, min(4, -(-ws // L2_CHUNK_BYTES), src.c // 16)) if not on_split or MERGE_SPLIT_KPARTS else 1
        kcuts = [min(src.c, (src.c * q // kp_n + 15) // 16 * 16) for q in range(kp_n)] + [src.c]
        chunk0 = [sum(S_r[:r]) for r in range(lay["rmax"] + 2)]  # first chunk index of ring r
        n_copies = chunk0[-1] * kp_n
        if n_copies > 1:
            self._alloc(ybuf, rows_all * c * n_copies)
        tb = GemmTables()
        for p in range(h * w):
            py, px = p // w, p % w
            for r in range(lay["rmax"] + 1):
                col0 = lay["col0"][r]
                valid = [(d, (py + dy) * w + (px + dx)) for d, (dy, dx, rr) in enumerate(lay["offs"])
                         if rr == r and 0 <= py + dy < h and 0 <= px + dx < w]
                for kp in range(kp_n):
                    k0, k1 = kcuts[kp], kcuts[kp + 1]
                    segs = [(s_st.pix_off(pin) + k0, (d * cin + k0) * C + col0, k1 - k0) for d, pin in valid]
                    for si in range(S_r[r]):
                        chunk = segs[len(segs) * si // S_r[r]:len(segs) * (si + 1) // S_r[r]]
                        copy = (chunk0[r] + si) * kp_n + kp
                        tb.add_group(copy * rows_all * c + p * nb * c + col0, chunk, nb, subkey=kp, n=C - col0)
        flags = GEMM_VAR_N | (GEMM_MFMA16X4 if narrow16 else 0)
        pos = len(self.fwd)
        self._emit_gemm(self.fwd, tb, C, self._ref(s_st.buf), s_st.ld, 0, self._ref(lay["buf"]), C, 0, self._ref(ybuf), c,
                        None, 0, f"fwd:{node.branches[0].scope}/merged", allow_split=False, hint=MERGE_FWD_HINT,
                        flags=flags)  # (a split-operand launch takes its own width, _split6_width)
        if len(self.fwd) > pos:
            self.fwd[pos].kparts = kp_n
        choff = 0
        for bi, b in enumerate(node.branches):
            copies_b = chunk0[(b.k - 1) // 2 + 1] * kp_n  # the copies that hold columns of this branch: 0 .. copies_b - 1
            if copies_b > 1:
                self.fwd.append(Launch("reduce_splits_f32", (self._ref(ybuf, rows_all * c + choff), rows_all * c,
                                                             copies_b - 1, self._ref(ybuf, choff), rows_all * b.cout, 1,
                                                             None, b.cout, c),
                                       nbytes=4 * rows_all * b.cout * (copies_b + 1), tag="tap-split-reduce"))
            choff += b.cout
        return n_copies

    def _fwd_linear(self, idx, node):
        nb = self.nb
        out = node.out
        c = node.cout
        aux = self._vec_params(node)
        self.node_aux[idx] = aux
        ybuf = f"y:{idx}"
        self._alloc(ybuf, out.npix * nb * c)
        y_st = Storage(ybuf, nb, c, None, 0, c, out.npix)
        aux["y"] = y_st
        bias_ref = self._p(aux["bias"]) if node.has_bias else None
        w_base = aux["w0"].offset

        if node.kind == "conv":
            src = node.sources[0]
            s_st = self.storage_of(src)
            h, w = src.hw
            rows_all = out.npix * nb
            lay = self._level_pass(idx, node, "fwd")
            if lay is not None:
                aux["stats_in_gemm"] = False
                self._fwd_level_merged(idx, node, lay, s_st, ybuf, c, h, w)
            # Tap splitting: a block that walks all 49 taps of a 7x7 branch runs ~4x longer than the average tile
            # and finishes alone at ~40 % MFMA utilisation.  Branches with more than MAX_TAPS_PER_TILE taps have
            # their tap list cut into S chunks; chunk s writes a partial copy Y_s of the output (same layout,
            # stored behind Y in the same buffer) and a strided reduce adds Y_1.. into Y.
            splits = {}
            kparts = 1
            if lay is None and (bias_ref is None or SPLIT_BIASED) and nb >= TAP_SPLIT_MIN_BATCH:
                for b in node.branches:
                    taps = min(b.k, h) * min(b.k, w)
                    if taps > MAX_TAPS_PER_TILE:
                        splits[id(b)] = (taps + MAX_TAPS_PER_TILE - 1) // MAX_TAPS_PER_TILE
                # L2 locality: the tiles of one 128-row chunk re-read the chunk's pixel blocks of X once per tap.  When
                # those blocks (pixels x 128 rows x Cin) exceed what an XCD's 4 MB L2 keeps, every tap goes back to
                # HBM (measured: 1.56 GB fetched for 117 MB of operands at Cin = 240).  The reduction dimension is then
                # cut into channel parts that are processed one after the other inside a chunk (tile order key), each
                # writing its own partial copy like the tap chunks do.
                if splits:
                    ws = h * w * GEMM_BM * src.c * 4
                    kparts = max(1, min(4, -(-ws // L2_CHUNK_BYTES), src.c // 16))
                    if kparts > 1:
                        for b in node.branches:
                            if b.k > 1:
                                splits[id(b)] = splits.get(id(b), 1) * kparts
            s_max = max(splits.values()) if splits else 1
            if s_max > 1:
                self._alloc(ybuf, rows_all * c * s_max)
            choff = 0
            by_cout = {}
            for b in (node.branches if lay is None else ()):
                # with a bias, split branches go into a launch of their own (no bias in the GEMM: the reduce adds it)
                key = (b.cout, bias_ref is not None and splits.get(id(b), 1) > 1)
                by_cout.setdefault(key, []).append((b, choff))
                choff += b.cout
            # batch-norm statistics in the epilogue: a lone 1x1 branch on contiguous input writes ONE [rows x c] matrix
            fuse_stats = (STATS_EPILOGUE and node.has_bn and node.training and len(node.branches) == 1
                          and node.branches[0].k == 1 and s_st.contiguous and s_max == 1 and c > 16
                          and not self._small_bn(node, rows_all) and lay is None)
            aux["stats_in_gemm"] = fuse_stats
            for (cout, split_launch), items in by_cout.items():
                biased_launch = not split_launch
                tb = GemmTables()
                kp_used = 1
                for b, off in items:
                    if b.k == 1 and s_st.contiguous:
                        tb.add_group(off, [(s_st.pix_off(0), b.w.offset, src.c)], out.npix * nb)
                        continue
                    S = splits.get(id(b), 1)
                    kp_n = kparts if S >= kparts and S % kparts == 0 and kparts > 1 else 1
                    kp_used = max(kp_used, kp_n)
                    S_tap = S // kp_n
                    kcuts = [min(src.c, (src.c * q // kp_n + 15) // 16 * 16) for q in range(kp_n)] + [src.c]
                    for p in range(h * w):
                        taps = valid_taps(h, w, b.k, p // w, p % w)
                        for kp in range(kp_n):
                            k0, k1 = kcuts[kp], kcuts[kp + 1]
                            segs = [(s_st.pix_off(pin) + k0, b.w.offset + ((i * b.k + j) * src.c + k0) * cout, k1 - k0)
                                    for (i, j, pin) in taps]
                            for si in range(S_tap):
                                chunk = segs[len(segs) * si // S_tap:len(segs) * (si + 1) // S_tap]
                                tb.add_group((kp * S_tap + si) * rows_all * c + p * nb * c + off, chunk, nb, subkey=kp)
                # the kernel indexes bias by (c_off % ldc) + column, so merged branches share one launch
                pos = len(self.fwd)
                self._emit_gemm(self.fwd, tb, cout, self._ref(s_st.buf), s_st.ld, 0, Ref(self.sess.params), cout, 0,
                                self._ref(ybuf), c, bias_ref if biased_launch else None, 0,
                                f"fwd:{items[0][0].scope}" + ("/split" if not biased_launch and bias_ref is not None else ""),
                                allow_split=False,
                                stats=((rows_all + GEMM_BM - 1) // GEMM_BM) * 2 * c if fuse_stats else None)
                if len(self.fwd) > pos:
                    self.fwd[pos].kparts = kp_used
                for b, off in items:
                    S = splits.get(id(b), 1)
                    if S > 1:
                        # a split branch of a biased convolution gets its bias here (every partial copy would add it)
                        self.fwd.append(Launch("reduce_splits_f32", (self._ref(ybuf, rows_all * c + off), rows_all * c,
                                                                     S - 1, self._ref(ybuf, off), rows_all * cout, 1,
                                                                     None if bias_ref is None else bias_ref + off, cout, c),
                                               nbytes=4 * rows_all * cout * (S + 1), tag="tap-split-reduce"))
        elif node.kind == "blockdense":
            # P independent small dense maps (one per band slice) as ONE grouped GEMM: group p reads the source columns
            # of its slice, multiplies by its own weights and writes columns [p*cout, (p+1)*cout) of the output
            src = node.sources[0]
            s_st = self.storage_of(src)
            cout = node.branches[0].cout
            tb = GemmTables()
            for pi, (off, width) in enumerate(node.in_slices):
                tb.add_group(pi * cout, [(s_st.pix_off(0) + off, node.branches[pi].w.offset, width)], nb)
            act_flag = self._act_in_gemm(node, cout)
            aux["act_in_gemm"] = bool(act_flag)
            self._emit_gemm(self.fwd, tb, cout, self._ref(s_st.buf), s_st.ld, 0, Ref(self.sess.params), cout, 0,
                            self._ref(ybuf), c, bias_ref, 0, f"fwd:{node.branches[0].scope}+{len(node.branches) - 1}",
                            allow_split=False, flags=act_flag)
        else:  # dense
            b = node.branches[0]
            rowbase = 0
            for si, src in enumerate(node.sources):
                s_st = self.storage_of(src)
                tb = GemmTables()
                segs = [(s_st.pix_off(p), b.w.offset + (rowbase + p * src.c) * c, src.c) for p in range(src.npix)]
                tb.add_group(0, segs, nb)
                act_flag = 0
                if len(node.sources) == 1 and self._split_k(tb, c, s_st.ld, 0, c, 0, c) is None:
                    act_flag = self._act_in_gemm(node, c)
                aux["act_in_gemm"] = bool(act_flag)
                self._emit_gemm(self.fwd, tb, c, self._ref(s_st.buf), s_st.ld, 0, Ref(self.sess.params), c, 0,
                                self._ref(ybuf), c, bias_ref if si == 0 else None, 1 if si > 0 else 0,
                                f"fwd:{b.scope}",
                                flags=act_flag)
                rowbase += src.npix * src.c

        rows = out.npix * nb
        # ---- batch-norm statistics ----
        if node.has_bn:
            self._alloc(f"mean:{idx}", c)
            self._alloc(f"rstd:{idx}", c)
            if node.training and self._small_bn(node, rows):
                # short matrix (rows = batch): statistics + finaliser + activation in ONE launch (_emit_post_fwd)
                aux["small_bn"] = True
                aux["mean"] = self._ref(f"mean:{idx}")
            elif node.training and aux.get("stats_in_gemm"):
                # the GEMM left one (mean, M2) pair per 128-row tile: only the finaliser remains
                n_chunks = (rows + GEMM_BM - 1) // GEMM_BM
                self._emit_bn_finalize(idx, node, aux, n_chunks, GEMM_BM, rows, c)
                aux["mean"] = self._ref(f"mean:{idx}")
            elif node.training:
                chunk = stat_chunk_rows(rows)
                n_chunks = (rows + chunk - 1) // chunk
                l1 = Launch("col_stats_partial", (self._ref(ybuf), c, rows, c, chunk, None),
                            nbytes=4 * rows * c, tag="bn-stats")
                self._scratch(l1, 5, "scratch_partial", n_chunks * 2 * c)
                self.fwd.append(l1)
                self._emit_bn_finalize(idx, node, aux, n_chunks, chunk, rows, c)
                aux["mean"] = self._ref(f"mean:{idx}")
            else:
                self.fwd.append(Launch("rstd_from_var", (self._s(aux["mv"]), c, float(node.bn_eps),
                                                         self._ref(f"rstd:{idx}")), tag="bn-infer"))
                aux["mean"] = self._s(aux["mm"])
            aux["rstd"] = self._ref(f"rstd:{idx}")
            aux["beta_ref"] = self._p(aux["beta"])
        # ---- fused post-op ----
        if node.has_post and aux.get("act_in_gemm"):
            # the product's epilogue already applied bias + leaky-ReLU: the "pre-activation" buffer holds the layer output
            # (the backward kernels take act' from its sign, which a positive slope preserves)
            self.storage[id(out)] = y_st
        elif node.has_post:
            zbuf = f"z:{idx}"
            self._alloc(zbuf, rows * c)
            self.storage[id(out)] = Storage(zbuf, nb, c, None, 0, c, out.npix)
            self._emit_post_fwd(idx, node, self._ref(ybuf), c, rows, c, aux, self._ref(zbuf))
        else:
            self.storage[id(out)] = y_st

    def _act_in_gemm(self, node, n):
        """HYPEL_GEMM_ACT_* code when this layer's post-op is nothing but (bias +) leaky-ReLU of a slope the library names
        (include/hypel.h) and its product runs on the 32-wide matrix-core tiles: a tf_slim.fully_connected without a
        normaliser then is ONE launch.  0: keep the separate post-op launch."""
        act = node.act
        if not (ACT_IN_GEMM and node.has_post and not node.has_bn and not node.residuals and node.dropout_keep is None
                and act is not None and act.code == 1 and n > 16 and not self.sync_bn):
            return 0
        for code, slope in ((1, 0.1), (2, 0.18), (3, 0.2), (4, 0.01)):
            if np.float32(act.alpha) == np.float32(slope):
                return code << 16
        return 0

    def _emit_bn_finalize(self, idx, node, aux, n_chunks, chunk, rows, c):
        """Chunk partials (in scratch_partial) -> mean / rstd / moving averages.  Synchronised batch norm: the rank's
        partials are merged into one (mean, M2, rows) record, the records of all ranks are all-gathered (a host-side
        collective between two graph segments) and merged in rank order."""
        mean_ref, rstd_ref = self._ref(f"mean:{idx}"), self._ref(f"rstd:{idx}")
        tail = (float(node.bn_eps), mean_ref, rstd_ref, self._s(aux["mm"]), self._s(aux["mv"]), float(node.bn_decay))
        if not self.sync_bn:
            l2 = Launch("bn_finalize", (None, n_chunks, chunk, rows, c) + tail, tag="bn-finalize")
            self._scratch(l2, 0, "scratch_partial", n_chunks * 2 * c)
            self.fwd.append(l2)
            return
        rec = 2 * c + 1
        l2 = Launch("bn_merge_partials", (None, n_chunks, chunk, rows, c, None), tag="bn-sync-merge")
        self._scratch(l2, 0, "scratch_partial", n_chunks * 2 * c)
        self._scratch(l2, 5, "sbn_local", rec)
        l3 = Launch("_allgather", (None, rec, None), tag="bn-sync-gather")
        self._scratch(l3, 0, "sbn_local", rec)
        self._scratch(l3, 2, "sbn_all", self.world * rec)
        l4 = Launch("bn_finalize_ranks", (None, self.world, c) + tail, tag="bn-finalize")
        self._scratch(l4, 0, "sbn_all", self.world * rec)
        self.fwd += [l2, l3, l4]

    def _mask_ref(self, idx, node, rows, c):
        if node.dropout_keep is None:
            return None
        name = f"mask:{idx}"
        self._alloc(name, rows * c)
        self.mask_bufs[node.dropout_index] = name
        if not self.external_masks:
            self.fwd.append(Launch("dropout_mask", (self._ref(name), rows * c, float(node.dropout_keep),
                                                    int(self.seed * 1000003 + idx), self._ref("step_ctr")), tag="rng"))
        else:
            self.buffers[name].fill_(1.0)
        return self._ref(name)

    def _res_args(self, node):
        out = []
        for (src, ridx) in node.residuals:
            st = self.storage_of(src)
            if not st.contiguous:
                raise NotImplementedError("residual source must cover the same pixels")
            iref = None
            if ridx is not None:
                t = self.be.upload(np.asarray(ridx, np.int32))
                self.tables.append(t)
                iref = Ref(t)
            out.append((self._ref(st.buf, st.ch_off), st.ld, iref))
        while len(out) < 2:
            out.append((None, 0, None))
        return out

    def _small_bn(self, node, rows):
        """Batch norm over a short matrix (the fully-connected tail: rows = batch) with no shortcut to add: one block
        per channel stripe covers every row, so the whole BN + activation is one launch per direction."""
        return (SMALL_BN and rows <= SMALL_BN_ROWS and not node.residuals and node.has_post
                and not (self.sync_bn and node.has_bn))

    def _emit_post_fwd(self, idx, node, y_ref, ldy, rows, c, aux, z_ref):
        has_bn = isinstance(node, G.LinearNode) and node.has_bn
        mask = self._mask_ref(idx, node, rows, c)
        aux["mask"] = mask
        (r1, ld1, i1), (r2, ld2, i2) = self._res_args(node)
        act = node.act
        if aux.get("small_bn"):
            self.fwd.append(Launch("bn_act_small_fwd", (
                y_ref, ldy, rows, c, float(node.bn_eps), aux["beta_ref"], act.code if act else 0,
                act.alpha if act else 0.0, mask, c, aux["mean"], aux["rstd"], self._s(aux["mm"]), self._s(aux["mv"]),
                float(node.bn_decay), z_ref, c), nbytes=12 * rows * c, tag="post-fwd-small"))
            return
        self.fwd.append(Launch("bn_act_fwd", (
            y_ref, ldy, rows, c, aux.get("mean") if has_bn else None, aux.get("rstd") if has_bn else None,
            aux.get("beta_ref") if has_bn else None, act.code if act else 0, act.alpha if act else 0.0, mask, c,
            r1, ld1, i1, r2, ld2, i2, z_ref, c), nbytes=4 * rows * c * (2 + len(node.residuals)), tag="post-fwd"))

    def _fwd_post(self, idx, node):
        src, out = node.src, node.out
        s_st = self.storage_of(src)
        if not s_st.contiguous:
            raise NotImplementedError
        rows, c = out.npix * self.nb, out.c
        st = self._new_value(out, f"z:{idx}")
        aux = {}
        self.node_aux[idx] = aux
        self._emit_post_fwd(idx, node, self._ref(s_st.buf, s_st.ch_off), s_st.ld, rows, c, aux, self._ref(st.buf))

    def _fwd_lrn(self, idx, node):
        src, out = node.src, node.out
        s_st = self.storage_of(src)
        rows, c = out.npix * self.nb, out.c
        st = self._new_value(out, f"z:{idx}")
        self.fwd.append(Launch("lrn_fwd", (self._ref(s_st.buf, s_st.ch_off), s_st.ld, rows, c, node.radius,
                                           float(node.bias), float(node.alpha), float(node.beta), self._ref(st.buf), c),
                               nbytes=8 * rows * c, tag="lrn"))

    # ------------------------------------------------------------------ loss
    def _emit_loss(self):
        loss = self.loss
        ps = loss.per_sample
        nb = self.nb
        logits = ps.logits
        l_st = self.storage_of(logits)
        lab = ps.labels
        lab_st = self.storage_of(lab)
        self._alloc("loss_ps", nb)
        self._alloc("loss_ce", 1)
        self._alloc("loss_mse", 1)
        dl = None
        lddl = 0
        if self.training:
            gst, acc = self._grad_target(logits)
            assert acc == 0
            dl, lddl = self._ref(gst.buf, gst.ch_off), gst.ld
        # data parallel: the gradient of the GLOBAL mean loss = sum over ranks of (nb_rank / nb_global) x the local
        # mean-loss gradients (= 1/world for equal shards; the ragged last batch of an epoch-limited run gives ranks
        # unequal shards).  The factor goes in here, at the source, so the exchange is a plain SUM all-reduce with no
        # rescaling pass over the 32 MB buffer afterwards (the reported loss values stay local means)
        dist = getattr(self.sess, "dist", None)
        gworld = 1.0
        if dist is not None:
            gworld = nb / float(self.global_nb) if self.global_nb else 1.0 / dist[0]
        self.fwd.append(Launch("softmax_xent", (self._ref(l_st.buf, l_st.ch_off), l_st.ld, nb, logits.c,
                                                self._ref(lab_st.buf), lab_st.ld, self._ref("loss_ps"), dl, lddl,
                                                gworld / nb), tag="loss"))
        guard = self.sess.guard_ref() if self.training and hasattr(self.sess, "guard_ref") else None
        # the dropout step counter advances once per training step, after the last mask of the step was drawn (masks
        # are drawn in the forward pass): it rides in the loss finaliser
        step = None
        if LOSS_TAIL and self.training and self.tower.n_dropout and not self.external_masks:
            step = self._ref("step_ctr")
            self._step_in_loss = True
        mse_args = None
        if ps.extra_mse is not None:
            m = ps.extra_mse
            a_st = self.storage_of(m.a)
            src = m.b.sources[0]  # flatten(image_original): the NHWC staging tensor already has (h, w, c) order
            assert src.node is None and src.root is None, "reconstruction target must be the network input"
            feat = src.npix * src.c
            assert feat == m.a.c
            da, ldda = None, 0
            if self.training:
                gst, acc = self._grad_target(m.a)
                assert acc == 0
                da, ldda = self._ref(gst.buf), gst.ld
            mse_args = (self._ref(a_st.buf), a_st.ld, self._ref("in:" + src.name), feat, nb, feat)
        if LOSS_TAIL:
            # xent rows + MSE block partials -> ONE finaliser (means, non-finite flag, step counter)
            ws = None
            if mse_args is not None:
                self._alloc("mse_ws", MSE_PARTIALS)
                ws = self._ref("mse_ws")
                self.fwd.append(Launch("mse_partial_f32", mse_args + (da, ldda, gworld, ws), nbytes=12 * nb * feat,
                                       tag="loss"))
            self.fwd.append(Launch("loss_finalize_f32", (
                self._ref("loss_ps"), nb, ws, 1.0 / (nb * feat) if mse_args is not None else 0.0, self._ref("loss_ce"),
                self._ref("loss_mse") if mse_args is not None else None, guard, step), tag="loss"))
            return
        l2 = Launch("sum_f32", (self._ref("loss_ps"), nb, 1.0 / nb, self._ref("loss_ce"), None), tag="loss")
        self._scratch(l2, 4, "scratch_red")
        self.fwd.append(l2)
        if mse_args is not None:
            l3 = Launch("mse", mse_args + (self._ref("loss_mse"), da, ldda, gworld, None), nbytes=12 * nb * feat,
                        tag="loss")
            self._scratch(l3, 10, "scratch_red")
            self.fwd.append(l3)
        if guard is not None:
            # NanTensorHook / check_numerics on the device: flag behind the gradient buffer, read by the optimiser
            self.fwd.append(Launch("loss_guard_f32", (self._ref("loss_ce"),
                                                      self._ref("loss_mse") if ps.extra_mse is not None else None,
                                                      guard), tag="loss-guard"))

    # ------------------------------------------------------------------ backward
    @staticmethod
    def _chanmap_start(ridx, cin):
        """Transpose of a monotone channel map: input channel ci receives output channels [start[ci], start[ci+1])."""
        ridx = np.asarray(ridx)
        assert (np.diff(ridx) >= 0).all(), "channel maps are monotone"
        return np.searchsorted(ridx, np.arange(cin + 1), side="left").astype(np.int32)

    def _foldable_residual(self, node):
        """Index of a shortcut of `node` whose gradient can ride in the epilogue of the node's own data-gradient
        GEMM: the shortcut source IS the convolution input (net = f(conv(net)) + map(net)), same pixel grid, plain
        (un-cropped) storage on both sides."""
        if not FOLD_RESIDUAL_GRAD or node.kind != "conv" or not node.has_post:
            return None
        src = node.sources[0]
        if not self._needs_grad(src) or src.npix != node.out.npix:
            return None
        z_st = self.storage[id(node.out)]
        if z_st.pixmap is not None or z_st.ch_off != 0 or z_st.ld != node.cout:
            return None
        if not self.storage_of(src).contiguous:
            return None
        for k, (rsrc, _) in enumerate(node.residuals):
            if rsrc is src:
                return k
        return None

    def _bwd_residuals(self, node, dz_ref, lddz, rows, c, skip=None):
        for k, (src, ridx) in enumerate(node.residuals):
            if k == skip:
                continue
            if not self._needs_grad(src):
                continue
            gst, acc = self._grad_target(src)
            sref = None
            if ridx is not None:
                t = self.be.upload(self._chanmap_start(ridx, src.c))
                self.tables.append(t)
                sref = Ref(t)
            self.bwd.append(Launch("chanmap_bwd", (dz_ref, lddz, rows, c, self._ref(gst.buf, gst.ch_off), gst.ld,
                                                   src.c, sref, acc), nbytes=4 * rows * (c + 2 * src.c),
                                   tag="res-bwd"))

    def _bwd_linear(self, idx, node):
        nb = self.nb
        out = node.out
        c = node.cout
        rows = out.npix * nb
        aux = self.node_aux[idx]
        own = out.owner
        if not self.grad_written.get(id(own), False):
            raise RuntimeError(f"node {idx} ({node.branches[0].scope}) receives no gradient")
        z_st = self.storage[id(out)]
        dz = self._ref("g:" + z_st.buf)
        y_ref = self._ref(aux["y"].buf)
        dy = dz  # backward post-op runs in place
        trains = self._trains([b.w for b in node.branches])
        fold = self._foldable_residual(node)
        fold_res = None
        if fold is not None:
            # the shortcut gradient is added by the data-gradient epilogue, which therefore needs dZ intact: the
            # post-op backward writes dY into its own buffer instead of over dZ
            ridx = node.residuals[fold][1]
            sref = None
            if ridx is not None:
                t = self.be.upload(self._chanmap_start(ridx, node.sources[0].c))
                self.tables.append(t)
                sref = Ref(t)
            fold_res = (dz, c, sref)
            if node.has_bn or (node.act and node.act.code != 0) or aux.get("mask") is not None:
                self._alloc("dy:" + z_st.buf, rows * c)
                dy = self._ref("dy:" + z_st.buf)
        if node.has_post:
            self._bwd_residuals(node, dz, c, rows, c, skip=fold)
            self._emit_post_bwd(node, aux, dz, y_ref, rows, c, dy, want_param=trains)
        elif node.has_bias and trains:
            self._emit_post_bwd(node, aux, dz, y_ref, rows, c, None, want_param=True)

        # ---- data gradient ----
        if node.kind == "conv":
            src = node.sources[0]
            s_st = self.storage_of(src)
            h, w = src.hw
            if self._needs_grad(src):
                gst, acc = self._grad_target(src)
                choff = 0
                by_cout = {}
                for b in node.branches:
                    by_cout.setdefault(b.cout, []).append((b, choff))
                    choff += b.cout
                lay = self._level_pass(idx, node, "dgrad")
                if lay is not None:
                    by_cout = {lay["co"]: [(b, None) for b in node.branches]}  # one launch over the packed image
                w_ref, w_ld, mtag = Ref(self.sess.params), None, ""
                for cout, items in by_cout.items():
                    tb = GemmTables()
                    if lay is not None:
                        # merged level: an input pixel sums, per input offset d (output pixel pin - d), ONE segment over the
                        # columns [col0[ring], C) of dY against the same columns of W_pack[d] -- 49 instead of 84 segments
                        # of 15 .. 60 columns at the centre of a 7x7 patch
                        w_ref, w_ld, mtag = self._ref(lay["buf"]), lay["C"], "/merged"
                        per_pixel = []
                        for pin in range(h * w):
                            iy, ix = pin // w, pin % w
                            segs = []
                            for d, (ofy, ofx, r) in enumerate(lay["offs"]):
                                oy, ox = iy - ofy, ix - ofx
                                if 0 <= oy < h and 0 <= ox < w:
                                    col0 = lay["col0"][r]
                                    segs.append(((oy * w + ox) * nb * c + col0, d * lay["cin"] * lay["C"] + col0,
                                                 lay["C"] - col0))
                            per_pixel.append(segs)
                    elif len(items) == 1 and items[0][0].k == 1 and gst.contiguous:
                        b, off = items[0]
                        tb.add_group(gst.pix_off(0), [(off, b.w.offset, cout)], src.npix * nb)
                        per_pixel = None
                    else:
                        per_pixel = []
                        for pin in range(h * w):
                            iy, ix = pin // w, pin % w
                            segs = []
                            for b, off in items:
                                pb = (b.k - 1) // 2
                                for i in range(b.k):
                                    oy = iy - (i - pb)
                                    if oy < 0 or oy >= h:
                                        continue
                                    for j in range(b.k):
                                        ox = ix - (j - pb)
                                        if 0 <= ox < w:
                                            segs.append(((oy * w + ox) * nb * c + off,
                                                         b.w.offset + (i * b.k + j) * src.c * cout, cout))
                            per_pixel.append(segs)
                    # Segment splitting, the data-gradient twin of the forward tap splitting: an input pixel of a
                    # multi-kernel level sums up to sum(k^2) segments (165 for the five kernels of DUALCNN) and its
                    # block runs that much longer than a corner pixel's.  The segment list is cut into S chunks
                    # that write partial copies of dX (same layout, in scratch), processed chunk after chunk
                    # inside a row chunk, and a reduce adds them.  Not with a folded shortcut gradient (one epilogue).
                    max_segs = max(len(sg) for sg in per_pixel) if per_pixel is not None else 0
                    S = 1
                    if (DGRAD_MAX_SEGS > 0 and fold_res is None and nb >= TAP_SPLIT_MIN_BATCH and gst.contiguous
                            and gst.ch_off == 0 and gst.ld == src.c and max_segs > DGRAD_MAX_SEGS):
                        S = -(-max_segs // DGRAD_MAX_SEGS)
                    if S > 1:
                        copy = h * w * nb * gst.ld
                        for pin, segs in enumerate(per_pixel):
                            for si in range(S):
                                chunk = segs[len(segs) * si // S:len(segs) * (si + 1) // S]
                                tb.add_group(si * copy + gst.pix_off(pin), chunk, nb, subkey=si)
                        pos = len(self.bwd)
                        self._emit_gemm(self.bwd, tb, src.c, dy, c, 0, w_ref, w_ld or cout, 1,
                                        self._ref(gst.buf), gst.ld, None, 0,
                                        f"dgrad:{items[0][0].scope}/split{mtag}")
                        self._scratch(self.bwd[pos], 6, "scratch_dgrad", S * copy)
                        l2 = Launch("reduce_splits_f32", (None, copy, S, self._ref(gst.buf), h * w * nb * src.c,
                                                          acc, None, src.c, gst.ld),
                                    nbytes=4 * h * w * nb * src.c * (S + 1), tag="dgrad-split-reduce")
                        self._scratch(l2, 0, "scratch_dgrad", S * copy)
                        self.bwd.append(l2)
                        acc = 1
                        continue
                    for pin, segs in enumerate(per_pixel or ()):
                        tb.add_group(gst.pix_off(pin), segs, nb)
                    self._emit_gemm(self.bwd, tb, src.c, dy, c, 0, w_ref, w_ld or cout, 1,
                                    self._ref(gst.buf), gst.ld, None, acc, f"dgrad:{items[0][0].scope}{mtag}",
                                    res=fold_res, pair=cout <= 16, kslice=True)
                    fold_res = None
                    acc = 1
            # ---- filter gradient ----
            if trains:
                self._wgrad_conv(idx, node, aux, s_st, src, dy, c, h, w)
        elif node.kind == "blockdense":
            src = node.sources[0]
            s_st = self.storage_of(src)
            cout = node.branches[0].cout
            if self._needs_grad(src):
                gst, acc = self._grad_target(src)
                by_width = {}
                for pi, (off, width) in enumerate(node.in_slices):
                    by_width.setdefault(width, []).append((pi, off))
                for width, items in by_width.items():  # the last slice may be ragged: one launch per input width
                    tb = GemmTables()
                    for pi, off in items:
                        tb.add_group(gst.pix_off(0) + off, [(pi * cout, node.branches[pi].w.offset, cout)], nb)
                    self._emit_gemm(self.bwd, tb, width, dy, c, 0, Ref(self.sess.params), cout, 1, self._ref(gst.buf),
                                    gst.ld, None, acc, f"dgrad:{node.branches[0].scope}+", allow_split=False)
            if trains:
                self._wgrad_blockdense(idx, node, s_st, dy, c, cout)
        else:
            b = node.branches[0]
            rowbase = 0
            for src in node.sources:
                s_st = self.storage_of(src)
                if self._needs_grad(src):
                    gst, acc = self._grad_target(src)
                    tb = GemmTables()
                    for p in range(src.npix):
                        tb.add_group(gst.pix_off(p), [(0, b.w.offset + (rowbase + p * src.c) * c, c)], nb)
                    self._emit_gemm(self.bwd, tb, src.c, dy, c, 0, Ref(self.sess.params), c, 1, self._ref(gst.buf),
                                    gst.ld, None, acc, f"dgrad:{b.scope}")
                rowbase += src.npix * src.c
            if trains:
                self._wgrad_dense(idx, node, aux, dy, c)

    def _emit_post_bwd(self, node, aux, dz, y_ref, rows, c, dy, want_param):
        has_bn = isinstance(node, G.LinearNode) and node.has_bn
        act = node.act
        code, alpha = (act.code, act.alpha) if act else (0, 0.0)
        mean = aux.get("mean") if has_bn else None
        rstd = aux.get("rstd") if has_bn else None
        beta = aux.get("beta_ref") if has_bn else None
        mask = aux.get("mask")
        dparam = None
        if want_param and isinstance(node, G.LinearNode):
            if has_bn:
                dparam = self._g(aux["beta"])
            elif node.has_bias:
                dparam = self._g(aux["bias"])
        if aux.get("small_bn") and has_bn and dy is not None:
            pacc = self._param_acc(aux["beta"]) if dparam is not None else 0
            self.bwd.append(Launch("bn_act_small_bwd", (dz, c, y_ref, c, rows, c, mean, rstd, beta, code, alpha, mask, c,
                                                        dy, c, dparam, pacc), nbytes=16 * rows * c,
                                   tag="post-bwd-small"))
            return
        sums = None
        if has_bn or dparam is not None:
            chunk = stat_chunk_rows(rows)
            n_chunks = (rows + chunk - 1) // chunk
            pacc = 0
            if dparam is not None:
                pacc = self._param_acc(aux["beta"] if has_bn else aux["bias"])
            if (ACT_BIAS_BWD and not has_bn and dy is not None and (code != 0 or mask is not None)
                  and not self.sync_bn):
                # no batch norm: dY = dZ * act'(y) needs no column sum -- the bias-gradient reduction writes it too
                l1 = Launch("act_bias_bwd_reduce", (dz, c, y_ref, c, rows, c, code, alpha, mask, c, chunk, None, dy, c),
                            nbytes=12 * rows * c, tag="post-bwd-reduce+apply")
                if dparam is None:
                    # the phase does not train this layer: only dY is wanted, the chunk sums have no reader
                    self._scratch(l1, 11, "scratch_partial", n_chunks * 2 * c)
                    self.bwd.append(l1)
                    return
                if getattr(self, "_defer_bias_sums", False):
                    # GAN train op: the chunk sums of every such layer stay in a buffer of their own and ONE launch at the end
                    # of the backward pass turns them all into bias gradients (PhasePlan._flush_slab_reduces)
                    k = self.__dict__.setdefault("_bias_sum_bufs", 0)
                    self._bias_sum_bufs = k + 1
                    name = f"bias_sums:{k}"
                    self._alloc(name, n_chunks * 2 * c)
                    args = list(l1.args)
                    args[11] = self._ref(name)
                    l1.args = tuple(args)
                    self.bwd.append(l1)
                    self.__dict__.setdefault("_bias_sum_entries", []).append((self._ref(name), dparam, 2 * c, c, n_chunks,
                                                                             pacc))
                    return
                self._scratch(l1, 11, "scratch_partial", n_chunks * 2 * c)
                l2 = Launch("bwd_reduce_finalize", (None, n_chunks, c, None, dparam, pacc), tag="post-bwd-finalize")
                self._scratch(l2, 0, "scratch_partial", n_chunks * 2 * c)
                self._scratch(l2, 3, "sums", 2 * c)
                self.bwd += [l1, l2]
                return
            else:
                l1 = Launch("bn_act_bwd_reduce", (dz, c, y_ref, c, rows, c, mean, rstd, beta, code, alpha, mask, c,
                                                  chunk, None), nbytes=8 * rows * c, tag="post-bwd-reduce")
                self._scratch(l1, 14, "scratch_partial", n_chunks * 2 * c)
                l2 = Launch("bwd_reduce_finalize", (None, n_chunks, c, None, dparam, pacc), tag="post-bwd-finalize")
                self._scratch(l2, 0, "scratch_partial", n_chunks * 2 * c)
                self._scratch(l2, 3, "sums", 2 * c)
                self.bwd += [l1, l2]
            sums = "pending"
        sync = self.sync_bn and has_bn and node.training and sums is not None
        if sync:
            # sum(dyh), sum(dyh * xhat) over the GLOBAL batch (the parameter gradient above stays the local sum: it
            # travels in the flat gradient all-reduce)
            lc = Launch("_allreduce", (None, 2 * c), tag="bn-sync-reduce")
            self._scratch(lc, 0, "sums", 2 * c)
            self.bwd.append(lc)
        if dy is not None and (has_bn or code != 0 or mask is not None):
            if sync:
                stat_rows = rows // self.nb * (self.global_nb if self.global_nb is not None else self.nb * self.world)
                l3 = Launch("bn_act_bwd_apply_global", (dz, c, y_ref, c, rows, c, mean, rstd, beta, code, alpha, mask, c,
                                                        None, int(stat_rows), dy, c), nbytes=12 * rows * c,
                            tag="post-bwd-apply")
            else:
                l3 = Launch("bn_act_bwd_apply", (dz, c, y_ref, c, rows, c, mean, rstd, beta, code, alpha, mask, c, None,
                                                 dy, c), nbytes=12 * rows * c, tag="post-bwd-apply")
            if sums is not None:
                self._scratch(l3, 13, "sums", 2 * c)
            self.bwd.append(l3)

    def _wgrad_splits(self, base_blocks, n_pairs):
        """Filter-gradient reduction = S_pix x S_row splits: the pixel-pair list is cut into S_pix contiguous chunks
        and the batch rows into S_row ranges (the SAME ranges for every tap).  Tiles are ordered split-major, so each
        XCD streams its own rows of X and dY through its L2 once while all taps consume them (the per-tap
        formulation re-fetched them 10-20x).  Multi-tap levels use row ranges only; single-tap layers also cut the
        pixel list, otherwise the launch would have too few blocks."""
        want = max(1, min(WGRAD_MAX_SPLITS, (TARGET_BLOCKS + base_blocks - 1) // max(base_blocks, 1)))
        max_row = max(1, self.nb // 64)
        # 64-row ranges also when the unsplit launch already fills the device: with split-major tile order an XCD then streams one
        # row range of X and dY for all taps instead of whole pixel blocks per tap (DUALCNN levels, 165 taps x 10 x 8
        # blocks unsplit: 82 -> 117 TFLOP/s with 8 row ranges)
        if WGRAD_MIN_SPLITS > 0:
            want = max(want, min(WGRAD_MAX_SPLITS, WGRAD_MIN_SPLITS))
        elif base_blocks >= TARGET_BLOCKS:
            want = max(want, min(WGRAD_MAX_SPLITS, max_row))
        s_row = min(want, max_row)
        if s_row >= 5:  # a multiple of 8 row ranges maps whole ranges onto the 8 XCDs
            s_row = min(max(8, max_row // 8 * 8), (s_row + 7) // 8 * 8)
        s_pix = max(1, min(n_pairs, want // s_row))
        return s_pix, s_row

    def _split_ranges(self, s_pix, s_row, n_pairs):
        """[(pair_lo, pair_hi, r0, r1)] in split order (row range major: the XCD-locality key)."""
        nb = self.nb
        rcuts = [min(nb, (nb * s // s_row + 31) // 32 * 32) for s in range(s_row)] + [nb]
        out = []
        for sr in range(s_row):
            for sp in range(s_pix):
                out.append((n_pairs * sp // s_pix, n_pairs * (sp + 1) // s_pix, rcuts[sr], rcuts[sr + 1]))
        return out

    def _emit_wgrad(self, tables_by_split_builder, n_groups_blocks, max_segs, slab, w0_offset, n, a_ref, lda, b_ref, ldb,
                    tag, acc=0, unpack=None):
        """tables_by_split_builder(S) -> GemmTables whose groups write to c_off = split*slab + local.
        unpack (merged levels): dict(buf, entries) -- the product's output is a packed image in buffer `buf`, scattered into
        the gradient slots by block copies after the reduction."""
        s_pix, s_row = self._wgrad_splits(n_groups_blocks, max_segs)
        S = s_pix * s_row
        tb = tables_by_split_builder((s_pix, s_row))
        if MERGE_WGRAD:
            pend = self.__dict__.setdefault("_pending_wgrads", [])
            if (acc or (unpack and any(e[6] for e in unpack["entries"]))) and pend:
                # a second application of shared weights adds to what an earlier pending product writes: keep the order
                self._flush_wgrads()
                pend = self._pending_wgrads
            pend.append(dict(tb=tb, S=S, slab=int(slab), w0=int(w0_offset), n=int(n), a_ref=a_ref, lda=int(lda),
                             b_ref=b_ref, ldb=int(ldb), tag=tag, acc=int(acc), unpack=unpack))
            return
        assert unpack is None, "merged-level filter gradients need the merged launch (MERGE_WGRAD)"
        if S == 1:
            self._emit_gemm(self.bwd, tb, n, a_ref, lda, 1, b_ref, ldb, 0, Ref(self.sess.grads, w0_offset), n, None,
                            acc, tag, allow_split=False)
            return
        launch_pos = len(self.bwd)
        self._emit_gemm(self.bwd, tb, n, a_ref, lda, 1, b_ref, ldb, 0, Ref(self.sess.grads), n, None, 0, tag,
                        allow_split=False)
        l = self.bwd[launch_pos]
        self._scratch(l, 6, "scratch_wgrad", S * slab)
        l2 = Launch("reduce_splits_f32", (None, slab, S, Ref(self.sess.grads, w0_offset), slab, acc, None, 0, 0),
                    nbytes=4 * slab * (S + 1), tag="wgrad-reduce")
        self._scratch(l2, 0, "scratch_wgrad", S * slab)
        self.bwd.append(l2)

    def _flush_wgrads(self):
        """Emit the collected filter gradients: one hypel_seg_gemm_multi_f32 per tile width over the blocks of every
        pending product, then one hypel_reduce_splits_multi_f32 that sums the split slabs into the gradient buffer.

        Block order: the blocks of one (product, split) pair -- all taps and column tiles that read the same batch-row
        range of X and dY -- form a locality group; groups are dealt to the 8 XCDs heaviest first onto the least
        loaded XCD (each XCD's L2 then streams a row range once for all its taps, and the XCDs finish together), and
        the record array is laid out so that the kernel's XCD remap (block b runs on XCD b % 8 and takes record
        (b % 8) * L + b / 8) hands XCD x exactly its list; short lists are padded with empty records."""
        pend = self.__dict__.get("_pending_wgrads") or []
        self._pending_wgrads = []
        if not pend:
            return
        base = Ref(self.sess.params)
        base_ptr = base.ptr()

        def rel(ref):
            d = ref.ptr() - base_ptr
            assert d % 4 == 0
            return d // 4

        # scratch for the split slabs of this flush
        need = sum(e["S"] * e["slab"] for e in pend if e["S"] > 1)
        fid = self.__dict__.setdefault("_wgrad_flushes", 0)
        self._wgrad_flushes = fid + 1
        sname = f"wgrad_partials:{fid}"
        self._alloc(sname, max(need, 1))
        spos = 0
        grads0 = rel(Ref(self.sess.grads))
        entries = []
        unpacks = []  # block copies packed gradient image -> TF-layout gradient slots (merged levels)

        def col_tiles(n, split6=0):
            """[(n0, tile width)] of a product with n output columns: one width per product.  split6 = tile-width hint of
            the split-operand kernel (widths | GEMM_MULTI_SPLIT6: their own launches)."""
            if split6:
                wdt = {1: 32, 2: 64, 3: 128}[split6]
                return [(n0, wdt | GEMM_MULTI_SPLIT6) for n0 in range(0, n, wdt)]
            wdt = 16 if n <= 16 else (32 if n <= 32 else 64)
            return [(n0, wdt) for n0 in range(0, n, wdt)]

        per_width = {}  # width -> dict(segs, lists, macs, nbytes, tags)
        for e in pend:
            n = e["n"]
            up = e.get("unpack")
            if up is not None:
                # the reduction target is the level's packed gradient image (dense per-offset blocks), scattered into the
                # variables' gradient slots afterwards
                self._alloc(up["buf"], e["slab"])
                out_base = rel(self._ref(up["buf"]))
                unpacks += [(out_base + so, grads0 + do, rows, cols, sld, dld, acc, 0)
                            for (so, do, rows, cols, sld, dld, acc) in up["entries"]]
            else:
                out_base = grads0 + e["w0"]
            if e["S"] > 1:
                c_base = rel(self._ref(sname, spos))
                entries.append((c_base, out_base, e["slab"], e["slab"], e["S"], 0 if up is not None else e["acc"]))
                spos += e["S"] * e["slab"]
                flags = 0
            else:
                c_base = out_base
                flags = 0 if up is not None else e["acc"]
            a0, b0 = rel(e["a_ref"]), rel(e["b_ref"])
            tb = e["tb"]
            sp6 = self._split6(e["tag"], tb, n, 1, 0, 0, False, in_multi=True)
            first_width = None
            loc_groups = {}  # (width, locality key) -> [work, [(work, record)]]
            for gi, (c_off, gs, rows) in enumerate(tb.groups):
                gn = tb.n_of(gi, n)
                # a merged level's per-offset products differ in their column count: each takes the width class of ITS n
                sp_g = sp6 if gn == n else (self._split6_width(gn) if sp6 and gn > GEMM_SPLIT_MIN_N else 0)
                tiles = col_tiles(gn, sp_g)
                ksum = sum(k for _, _, k in gs)
                key = tb.keys[gi] if tb.keys[gi] is not None else 0
                seg_begin = {}
                for wdt in sorted({wd for _, wd in tiles}, reverse=True):
                    pw = per_width.setdefault(wdt, dict(segs=[], lists=[], macs=0, nbytes=0, tags=[]))
                    if first_width is None:
                        first_width = wdt
                    if e["tag"] not in pw["tags"]:
                        pw["tags"].append(e["tag"])
                    # (a split with no reduction rows still owns its slab: the reduce reads it, so it must be zero)
                    seg_begin[wdt] = len(pw["segs"])
                    pw["segs"] += [(a0 + int(a_off), b0 + int(b_off), int(k), 0) for (a_off, b_off, k) in gs]
                first = (a0 + int(gs[0][0]), b0 + int(gs[0][1]), int(gs[0][2])) if gs else (0, 0, 0)
                for m0 in range(0, rows, GEMM_BM):
                    work = ksum * min(GEMM_BM, rows - m0)
                    for n0, wdt in tiles:
                        per_width[wdt]["macs"] += min(GEMM_BM, rows - m0) * ksum * min(wdt & 0xff, gn - n0)
                        recs = loc_groups.setdefault((wdt, key), [0, []])
                        recs[1].append((work, (c_base + int(c_off), first[0], first[1], m0, rows, n0, gn,
                                               seg_begin[wdt], len(gs), first[2], flags, e["lda"], e["ldb"], gn, 0)))
                        recs[0] += work
            if first_width is not None:
                per_width[first_width]["nbytes"] += tb.compulsory_bytes(n, e["lda"], 1, e["ldb"], 0)
            for (wdt, _), (wk, r) in loc_groups.items():  # inside a locality group: heavy blocks first
                r.sort(key=lambda t: -t[0])
                per_width[wdt]["lists"].append((wk, [rec for _, rec in r]))
        for width in sorted(per_width, reverse=True):
            pw = per_width[width]
            segs, lists = pw["segs"], pw["lists"]
            lists.sort(key=lambda t: -t[0])
            xcd_work, xcd_recs = [0] * 8, [[] for _ in range(8)]
            for w, r in lists:
                x = min(range(8), key=lambda i: (xcd_work[i], i))
                xcd_work[x] += w
                xcd_recs[x] += r
            L = max(len(r) for r in xcd_recs)
            arr = np.zeros(8 * L, MTILE_DTYPE)  # zero record = empty block (rows 0, no segments)
            for x in range(8):
                if xcd_recs[x]:
                    arr[x * L:x * L + len(xcd_recs[x])] = np.array(xcd_recs[x], MTILE_DTYPE)
            sarr = np.array(segs, SEG_DTYPE) if segs else np.zeros(1, SEG_DTYPE)
            s_t, r_t = self.be.upload(sarr), self.be.upload(arr)
            self.tables += [s_t, r_t]
            l = Launch("seg_gemm_multi_f32", (base, 1, 0, width, Ref(s_t), Ref(r_t), int(len(arr))),
                       flops=2 * pw["macs"], nbytes=pw["nbytes"],
                       tag=f"wgrad-merged/{'s' if width & GEMM_MULTI_SPLIT6 else ''}{width & 0xff}")
            l.meta = {"products": pw["tags"], "blocks": int(sum(len(r) for r in xcd_recs)),
                      "xcd_work": [int(w) for w in xcd_work]}
            self.bwd.append(l)
        if entries:
            earr = np.array([(p, o, st, cnt, S, acc) for (p, o, st, cnt, S, acc) in entries], REDUCE_ENTRY_DTYPE)
            e_t = self.be.upload(earr)
            self.tables.append(e_t)
            l = Launch("reduce_splits_multi_f32", (base, Ref(e_t), int(len(earr))),
                       nbytes=4 * sum(cnt * (S + 1) for (_, _, _, cnt, S, _) in entries), tag="wgrad-reduce")
            l.meta = {"splits": [int(S) for (_, _, _, _, S, _) in entries]}
            self.bwd.append(l)
        if unpacks:
            from .backend import COPY_BLOCK_DTYPE
            u_t = self.be.upload(np.array(unpacks, COPY_BLOCK_DTYPE))
            self.tables.append(u_t)
            self.bwd.append(Launch("copy_blocks_f32", (base, Ref(u_t), len(unpacks), max(u[2] * u[3] for u in unpacks)),
                                   nbytes=8 * sum(u[2] * u[3] for u in unpacks), tag="level-unpack"))
    @staticmethod
    def _split_even(segs, S):
        """Partition a list into S contiguous chunks (some possibly empty)."""
        n = len(segs)
        return [segs[(n * s) // S:(n * (s + 1)) // S] for s in range(S)]

    def _wgrad_level_merged(self, idx, node, lay, s_st, src, dy, c, h, w):
        """Filter gradient of a merged level: per input offset d ONE product dW_pack[d] = sum_p X[p + d]^T dY[p][:, col0:]
        with n = C - col0[ring] columns (30 .. 120 for the 30-filter HYPELCNN level instead of 30 per (branch, tap)), written
        into a dense packed image; hypel_copy_blocks_f32 scatters the [Cin x cout] slices into the HWIO gradient slots."""
        nb = self.nb
        C, cin, co = lay["C"], lay["cin"], lay["co"]
        group_list = []  # (offset of the block in the dense image, [(a_off, b_off)] pixel pairs, columns)
        for d, (ofy, ofx, r) in enumerate(lay["offs"]):
            col0 = lay["col0"][r]
            pairs = [(s_st.pix_off((oy + ofy) * w + ox + ofx), (oy * w + ox) * nb * c + col0)
                     for oy in range(h) for ox in range(w) if 0 <= oy + ofy < h and 0 <= ox + ofx < w]
            group_list.append((lay["dense_off"][d], pairs, C - col0))
        slab = lay["dense_size"]
        blocks = sum(((cin + GEMM_BM - 1) // GEMM_BM) * ((n_d + 63) // 64) for _, _, n_d in group_list)
        max_segs = max(len(prs) for _, prs, _ in group_list)

        def build(S, group_list=group_list, slab=slab, rows=cin, lda=s_st.ld, ldb=c, npairs=max_segs):
            tb = GemmTables()
            for si, (p0, p1, r0, r1) in enumerate(self._split_ranges(S[0], S[1], npairs)):
                for (loc, pairs, n_d) in group_list:
                    q0, q1 = len(pairs) * p0 // npairs, len(pairs) * p1 // npairs
                    segs = [(a + r0 * lda, b_ + r0 * ldb, r1 - r0) for (a, b_) in pairs[q0:q1]] if r1 > r0 else []
                    tb.add_group(si * slab + loc, segs, rows, key=si, n=n_d)
            return tb

        ents = []
        for bi, b in enumerate(node.branches):
            acc = self._param_acc(b.w)
            pb = (b.k - 1) // 2
            for i in range(b.k):
                for j in range(b.k):
                    d = lay["index"][(i - pb, j - pb)]
                    r = lay["offs"][d][2]
                    n_d = C - lay["col0"][r]
                    dst = b.w.offset + (i * b.k + j) * cin * co
                    ents.append((lay["dense_off"][d] + (bi - lay["first"][r]) * co, dst, cin, co, n_d, co, acc))
        self._emit_wgrad(build, blocks, max_segs, slab, 0, C, self._ref(s_st.buf), s_st.ld, dy, c,
                         f"wgrad:{node.branches[0].scope}/merged", acc=0,
                         unpack=dict(buf=f"dwpack:{idx}", entries=ents))

    def _wgrad_conv(self, idx, node, aux, s_st, src, dy, c, h, w):
        nb = self.nb
        lay = self._level_pass(idx, node, "wgrad")
        # (every offset of the largest kernel must meet at least one pixel pair, or its block of the packed image would
        # never be written: (k - 1) / 2 <= min(h, w) - 1)
        if lay is not None and max(b.k for b in node.branches) <= 2 * min(h, w) - 1:
            self._wgrad_level_merged(idx, node, lay, s_st, src, dy, c, h, w)
            return
        choff = 0
        w_base = aux["w0"].offset
        by_cout = {}
        for b in node.branches:
            by_cout.setdefault(b.cout, []).append((b, choff))
            choff += b.cout
        for cout, items in by_cout.items():
            # the slab of one launch = the contiguous weights of its branches
            lo = min(b.w.offset for b, _ in items)
            hi = max(b.w.offset + b.w.size for b, _ in items)
            slab = hi - lo
            if slab != sum(b.w.size for b, _ in items):
                # the reduce overwrites the whole slab: sibling branches of another width in between would be clobbered
                raise RuntimeError(f"filter-gradient slab of {items[0][0].scope}: branches with {cout} filters are not "
                                   f"contiguous in the parameter buffer")
            group_list = []  # (local c_off, [(a_off, b_off)] pixel pairs)
            for b, off in items:
                pb = (b.k - 1) // 2
                for i in range(b.k):
                    for j in range(b.k):
                        pairs = []
                        for oy in range(h):
                            iy = oy + i - pb
                            if iy < 0 or iy >= h:
                                continue
                            for ox in range(w):
                                ix = ox + j - pb
                                if ix < 0 or ix >= w:
                                    continue
                                pairs.append((s_st.pix_off(iy * w + ix), (oy * w + ox) * nb * c + off))
                        group_list.append((b.w.offset - lo + (i * b.k + j) * src.c * cout, pairs))
            blocks = len(group_list) * ((src.c + GEMM_BM - 1) // GEMM_BM) * ((cout + 63) // 64)
            max_segs = max(len(s) for _, s in group_list)

            def build(S, group_list=group_list, slab=slab, rows=src.c, lda=s_st.ld, ldb=c, npairs=max_segs):
                tb = GemmTables()
                for si, (p0, p1, r0, r1) in enumerate(self._split_ranges(S[0], S[1], npairs)):
                    for (loc, pairs) in group_list:
                        # a tap with fewer valid pixel pairs than the widest one gets proportionally cut chunks
                        q0, q1 = len(pairs) * p0 // npairs, len(pairs) * p1 // npairs
                        segs = [(a + r0 * lda, b_ + r0 * ldb, r1 - r0) for (a, b_) in pairs[q0:q1]] if r1 > r0 else []
                        tb.add_group(si * slab + loc, segs, rows, key=si)
                return tb

            acc = max(self._param_acc(b.w) for b, _ in items)
            self._emit_wgrad(build, blocks, max_segs, slab, lo, cout, self._ref(s_st.buf), s_st.ld, dy, c,
                             f"wgrad:{items[0][0].scope}", acc=acc)

    def _wgrad_dense(self, idx, node, aux, dy, c):
        nb = self.nb
        b = node.branches[0]
        rowbase = 0
        for src in node.sources:
            s_st = self.storage_of(src)
            slab = src.npix * src.c * c
            lo = b.w.offset + rowbase * c
            group_list = [(p * src.c * c, [(s_st.pix_off(p), 0)]) for p in range(src.npix)]
            blocks = len(group_list) * ((src.c + GEMM_BM - 1) // GEMM_BM) * ((c + 63) // 64)
            max_segs = 1

            def build(S, group_list=group_list, slab=slab, rows=src.c, lda=s_st.ld, ldb=c):
                tb = GemmTables()
                for si, (p0, p1, r0, r1) in enumerate(self._split_ranges(1, S[0] * S[1], 1)):
                    for (loc, pairs) in group_list:
                        segs = [(a + r0 * lda, b_ + r0 * ldb, r1 - r0) for (a, b_) in pairs] if r1 > r0 else []
                        tb.add_group(si * slab + loc, segs, rows, key=si)
                return tb

            acc = self._param_acc(b.w) if rowbase == 0 else (1 if b.w.name + f"#{rowbase}" in self.param_written else 0)
            self.param_written.add(b.w.name + f"#{rowbase}")
            self._emit_wgrad(build, blocks, max_segs, slab, lo, c, self._ref(s_st.buf), s_st.ld, dy, c,
                             f"wgrad:{b.scope}", acc=acc)
            rowbase += src.npix * src.c

    def _wgrad_blockdense(self, idx, node, s_st, dy, c, cout):
        """dW_p = X[:, slice_p]^T dY[:, p*cout:(p+1)*cout] for every slice p in one launch (the weights of a merged
        layer are contiguous: one slab per batch-row split, one reduction)."""
        ws = [b.w for b in node.branches]
        lo = ws[0].offset
        slab = sum(w.size for w in ws)
        group_list = [(w.offset - lo, off, pi * cout, width) for pi, (w, (off, width)) in
                      enumerate(zip(ws, node.in_slices))]
        blocks = sum((width + GEMM_BM - 1) // GEMM_BM for _, _, _, width in group_list) * ((cout + 63) // 64)

        def build(S, lda=s_st.ld, ldb=c, base=s_st.pix_off(0)):
            tb = GemmTables()
            for si, (p0, p1, r0, r1) in enumerate(self._split_ranges(1, S[0] * S[1], 1)):
                for (loc, a_off, b_off, width) in group_list:
                    segs = [(base + a_off + r0 * lda, b_off + r0 * ldb, r1 - r0)] if r1 > r0 else []
                    tb.add_group(si * slab + loc, segs, width, key=si)
            return tb

        acc = self._param_acc(ws[0])
        for w in ws[1:]:
            self.param_written.add(w.name)
        self._emit_wgrad(build, blocks, 1, slab, lo, cout, self._ref(s_st.buf), s_st.ld, dy, c,
                         f"wgrad:{node.branches[0].scope}+", acc=acc)

    def _bwd_post(self, idx, node):
        out, src = node.out, node.src
        rows, c = out.npix * self.nb, out.c
        aux = self.node_aux[idx]
        z_st = self.storage[id(out)]
        if not self.grad_written.get(id(out.owner), False):
            raise RuntimeError(f"post node {idx} receives no gradient")
        dz = self._ref("g:" + z_st.buf)
        self._bwd_residuals(node, dz, c, rows, c)
        if not self._needs_grad(src):
            return
        s_st = self.storage_of(src)
        act = node.act
        code, alpha = (act.code, act.alpha) if act else (0, 0.0)
        mask = aux.get("mask")
        if code != 0 or mask is not None:
            self.bwd.append(Launch("bn_act_bwd_apply", (dz, c, self._ref(s_st.buf, s_st.ch_off), s_st.ld, rows, c, None,
                                                        None, None, code, alpha, mask, c, None, dz, c),
                                   nbytes=12 * rows * c, tag="post-bwd-apply"))
        gst, acc = self._grad_target(src)
        self.bwd.append(Launch("chanmap_bwd", (dz, c, rows, c, self._ref(gst.buf, gst.ch_off), gst.ld, c, None, acc),
                               nbytes=12 * rows * c, tag="post-bwd-pass"))

    def _bwd_lrn(self, idx, node):
        out, src = node.out, node.src
        rows, c = out.npix * self.nb, out.c
        z_st = self.storage[id(out)]
        if not self.grad_written.get(id(out.owner), False):
            raise RuntimeError(f"lrn node {idx} receives no gradient")
        if not self._needs_grad(src):
            return
        s_st = self.storage_of(src)
        gst, acc = self._grad_target(src)
        self.bwd.append(Launch("lrn_bwd", (self._ref(s_st.buf, s_st.ch_off), s_st.ld, self._ref("g:" + z_st.buf), c,
                                           rows, c, node.radius, float(node.bias), float(node.alpha), float(node.beta),
                                           self._ref(gst.buf, gst.ch_off), gst.ld, acc), nbytes=16 * rows * c,
                               tag="lrn-bwd"))


# =========================================================================================== GAN phases
