"""Host-side tables of the grouped multi-segment GEMM launches (include/hypel.h: hypel_seg_gemm_f32 and its variants):
the `Launch` record the planner emits and `GemmTables`, the builder of a launch's (groups, segments, 128-row tile
records) arrays incl. the XCD-aware tile order, segment pairing and the algorithmic-byte count of a launch."""
import numpy as np

from .backend import GEMM_BM, GROUP_DTYPE, SEG_DTYPE, TILE_DTYPE

SEG_PAIR_FLAG = 0x40000000  # include/hypel.h: HYPEL_SEG_PAIR_FLAG


class Launch:
    __slots__ = ("name", "args", "flops", "bytes", "tag", "kparts", "meta")

    def __init__(self, name, args, flops=0, nbytes=0, tag=""):
        self.name = name
        self.args = args
        self.flops = flops
        self.bytes = nbytes
        self.tag = tag
        self.kparts = 1  # channel parts the reduction dimension of a level forward was cut into (diagnostic)
        self.meta = {}  # diagnostics (e.g. the products a merged filter-gradient launch contains)


class GemmTables:
    """Host-side builder of the (groups, segments, tiles) tables of one hypel_seg_gemm_f32 launch."""

    def __init__(self):
        self.groups = []  # (c_off, [segs], rows)
        self.keys = []  # optional locality key per group (tiles are ordered key-major)
        self.subkeys = []  # secondary locality key (phase inside a row chunk)
        self.ns = []  # per-group column count (0 = the launch's n): the ring groups of a merged multi-kernel level
        self.flags = []  # per-group hypel_tile_t.flags (TILE_PLAIN: a K-slice partial)
        self.tail = []   # per-group: dealt to the END of the XCDs' work lists (the K-slices of a launch, see finalize)

    def add_group(self, c_off, segs, rows, key=None, subkey=0, n=0, flags=0, tail=False):
        self.groups.append((int(c_off), segs, int(rows)))
        self.keys.append(key)
        self.subkeys.append(subkey)
        self.ns.append(int(n))
        self.flags.append(int(flags))
        self.tail.append(bool(tail))

    def n_of(self, gi, n):
        return self.ns[gi] or n

    def finalize(self, n, pair=False):
        """pair: consecutive segments of a group with k <= 16 each are marked to share one k-tile (SEG_PAIR_FLAG on
        the first; the kernel's data-gradient variant for short segments).  self.paired = number of pairs made."""
        self.paired = 0
        segs = []
        garr = np.zeros(len(self.groups), GROUP_DTYPE)
        tiles = []
        macs = 0
        for gi, (c_off, gs, rows) in enumerate(self.groups):
            garr[gi] = (c_off, len(segs), len(gs), rows, 0)
            ksum = 0
            first = len(segs)
            for (a_off, b_off, k) in gs:
                segs.append((int(a_off), int(b_off), int(k), 0))
                ksum += k
            if pair:
                i = first
                while i + 1 < len(segs):
                    (a0, b0, k0, _), (a1, b1, k1, _) = segs[i], segs[i + 1]
                    if k0 <= 16 and k1 <= 16 and abs(a0 - a1) * 4 < (1 << 31) and abs(b0 - b1) * 4 < (1 << 31):
                        segs[i] = (a0, b0, k0 | SEG_PAIR_FLAG, 0)
                        self.paired += 1
                        i += 2
                    else:
                        i += 1
            macs += rows * ksum * self.n_of(gi, n)
            for m0 in range(0, rows, GEMM_BM):
                key = (self.keys[gi] if self.keys[gi] is not None else m0 // GEMM_BM, self.subkeys[gi])
                tiles.append((ksum * min(GEMM_BM, rows - m0), gi, m0, key))
        # Row-chunk major, heavy tiles first inside a chunk.  With the kernel's XCD remap each XCD works through a
        # contiguous range of this list, i.e. through whole row chunks: the ~49 pixel blocks of activations that
        # all the (pixel, branch) tiles of one chunk keep re-reading stay resident in that XCD's 4 MB L2.
        # Filter-gradient launches pass key = split index (= batch-row range), for the same reason.
        tiles.sort(key=lambda t: (t[3], -t[0]))

        def record(g, m0):
            c_off, gs, rows = self.groups[g]
            sb, sc = int(garr[g]["seg_begin"]), len(gs)
            a0, b0, k0 = segs[sb][:3] if sc else (0, 0, 0)  # incl. the pair flag
            return (g, m0, rows, sb, sc, k0, c_off, a0, b0, self.flags[g], self.ns[g])

        if any(self.tail):
            # K-slices (plan.py::_kslice): the kernel's XCD remap hands XCD x the x-th EIGHTH of the record array, in order.
            # The ordinary tiles keep their order and are cut into eight contiguous shares; the slices -- the small pieces
            # that are to fill the end of the launch -- are dealt heaviest first onto the XCD with the least work and go
            # to the END of its share; shares are padded with empty records (rows = 0: the block exits) to one length.
            main = [t for t in tiles if not self.tail[t[1]]]
            extra = sorted((t for t in tiles if self.tail[t[1]]), key=lambda t: -t[0])
            q, r = divmod(len(main), 8)
            shares, pos = [], 0
            for x in range(8):
                cnt = q + (1 if x < r else 0)
                shares.append(main[pos:pos + cnt])
                pos += cnt
            work = [sum(t[0] for t in sh) for sh in shares]
            for t in extra:
                x = min(range(8), key=lambda i: (work[i], i))
                shares[x].append(t)
                work[x] += t[0]
            L = max(len(sh) for sh in shares)
            empty = (0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0)
            recs = []
            for sh in shares:
                recs += [record(g, m0) for (_, g, m0, _) in sh] + [empty] * (L - len(sh))
        else:
            recs = [record(g, m0) for (_, g, m0, _) in tiles]
        sarr = np.array(segs, SEG_DTYPE) if segs else np.zeros(1, SEG_DTYPE)
        tarr = np.array(recs, TILE_DTYPE) if recs else np.zeros(0, TILE_DTYPE)
        return garr, sarr, tarr, macs

    def compulsory_bytes(self, n, lda, ta, ldb, tb):
        """Algorithmic HBM bytes of the launch: every DISTINCT operand element read once, every output element
        written once (overlapping tap windows of A and weights shared by groups count once)."""
        def union(spans):
            tot, end = 0, None
            for lo, hi in sorted(spans):
                if end is None or lo > end:
                    tot += hi - lo
                    end = hi
                elif hi > end:
                    tot += hi - end
                    end = hi
            return tot

        a_sp, b_sp, c_el = {}, {}, set()
        n_launch = n
        for gi, (c_off, gs, rows) in enumerate(self.groups):
            n = self.n_of(gi, n_launch)
            c_el.add((c_off, rows, n))
            for a_off, b_off, k in gs:
                if ta:  # A stored [k, rows]
                    a_sp.setdefault((a_off % lda, rows), []).append((a_off // lda, a_off // lda + k))
                else:  # A stored [rows, k]
                    a_sp.setdefault((a_off % lda, k), []).append((a_off // lda, a_off // lda + rows))
                if tb:  # B stored [n, k]
                    b_sp.setdefault((b_off % ldb, k), []).append((b_off // ldb, b_off // ldb + n))
                else:  # B stored [k, n]
                    b_sp.setdefault((b_off % ldb, n), []).append((b_off // ldb, b_off // ldb + k))
        elems = sum(w * union(sp) for (_, w), sp in a_sp.items()) + sum(w * union(sp) for (_, w), sp in b_sp.items())
        elems += sum(rows * gn for _, rows, gn in c_el)
        return 4 * elems
