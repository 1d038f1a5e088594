"""TEST INFRASTRUCTURE: a numpy emulation of the libhypel_hip.so entry points (include/hypel.h).

There is no GPU in the build container, so the host-side logic of the product (graph recording,
planning, table building, the hand-derived backward pass, optimiser wiring) is exercised here
against this executable specification of every kernel's contract; the `-m gpu` tests then check the
real HIP kernels against the oracle.  Never imported by the hypelcnn_amd package.
"""
import numpy as np
import torch

import ctypes

from hypelcnn_amd.backend import (COLLECTIVES, GROUP_DTYPE, LOSS_NONE, LOSS_TERM_DTYPE, MTILE_DTYPE, REDUCE_ENTRY_DTYPE, SEG_DTYPE, TILE_DTYPE, Ref,
                                  bind_collective)


def _arr(ref, dtype=np.float32):
    """numpy view of the tensor behind a Ref starting at its offset."""
    if ref is None:
        return None
    if isinstance(ref, _Addr):  # nominal length: callers slice what they use
        return np.ctypeslib.as_array((ctypes.c_float * (1 << 26)).from_address(ref.addr)).view(dtype)
    a = ref.t.numpy()
    if a.dtype != dtype:
        a = a.view(dtype) if a.dtype.itemsize == np.dtype(dtype).itemsize else a
    return a[ref.off:]


def _at(base_ref, off, count):
    """float32 view of `count` elements at element offset `off` (any sign) from a base Ref: the multi-product kernels
    address every operand relative to ONE base pointer, across allocations."""
    addr = base_ref.ptr() + int(off) * 4
    return np.ctypeslib.as_array((ctypes.c_float * int(count)).from_address(addr))


class _Addr:
    """A Ref-like for `base + elems` when the result may lie in another allocation than `base` (the *_apps entry points
    reach application g's variables and gradient slabs by a stride from application 0's)."""
    __slots__ = ("addr",)

    def __init__(self, ref, elems):
        self.addr = ref.ptr() + int(elems) * 4

    def ptr(self):
        return self.addr


def _shift(ref, elems):
    return None if ref is None else (ref if int(elems) == 0 else _Addr(ref, elems))


def _mat(ref, ld, rows, cols, dtype=np.float32):
    a = _arr(ref, dtype)
    return np.lib.stride_tricks.as_strided(a, shape=(rows, cols), strides=(ld * a.itemsize, a.itemsize))


def _act(v, code, alpha):
    if code == 1:
        return np.where(v > 0, v, v * alpha)
    if code == 2:
        return np.where(v > 0, v, 0)
    if code == 3:
        return 1.0 / (1.0 + np.exp(-v))
    if code == 4:
        return np.tanh(v)
    return v


def _act_grad(v, code, alpha):
    if code == 1:
        return np.where(v > 0, 1.0, alpha)
    if code == 2:
        return np.where(v > 0, 1.0, 0.0)
    if code == 3:
        s = 1.0 / (1.0 + np.exp(-v))
        return s * (1 - s)
    if code == 4:
        t = np.tanh(v)
        return 1 - t * t
    return np.ones_like(v)


class EmuBackend:
    name = "emu"

    def __init__(self):
        self.device = torch.device("cpu")
        self.launch_log = []

    def empty(self, n, dtype=torch.float32):
        return torch.zeros(int(n), dtype=dtype)

    def zeros(self, n, dtype=torch.float32):
        return torch.zeros(int(n), dtype=dtype)

    def upload(self, array):
        a = np.ascontiguousarray(array)
        if a.dtype.fields is not None:
            return torch.from_numpy(a.view(np.uint8).reshape(-1).copy())
        return torch.from_numpy(a.reshape(-1).copy())

    def synchronize(self):
        pass

    def bind(self, name, args, stream=None):
        if name in COLLECTIVES:
            return bind_collective(name, args)
        fn = getattr(self, "k_" + name)
        return lambda: fn(*args)

    def call(self, name, *args):
        getattr(self, "k_" + name)(*args)

    def capture(self, launches, settled=False):
        def replay():
            for f in launches:
                f()
        return replay

    # ------------------------------------------------------------------ kernels
    def k_nhwc_to_pnc(self, x, out, n, p, c, ld):
        xv = _arr(x)[: n * p * c].reshape(n, p, c)
        o = _arr(out)[: p * n * ld].reshape(p, n, ld)
        o[:, :, :c] = xv.transpose(1, 0, 2)
        o[:, :, c:] = 0

    def k_pnc_to_nhwc(self, inp, ld, x, n, p, c):
        i = _arr(inp)[: p * n * ld].reshape(p, n, ld)
        _arr(x)[: n * p * c].reshape(n, p, c)[...] = i[:, :, :c].transpose(1, 0, 2)

    # ---- data side (specification written with numpy's own rot90 / flip, not with index arithmetic)
    def k_gather_patches_f32(self, casi, lidar, hp, wp, cc, cl, points, n, p, out):
        cs = _arr(casi)[: hp * wp * cc].reshape(hp, wp, cc)
        ls = None if lidar is None else _arr(lidar)[: hp * wp * cl].reshape(hp, wp, cl)
        pts = _arr(points, np.int32)[: 2 * n].reshape(n, 2)
        o = _arr(out)[: n * p * p * (cc + cl)].reshape(n, p, p, cc + cl)
        for i, (x0, y0) in enumerate(pts):
            o[i, :, :, :cc] = cs[y0:y0 + p, x0:x0 + p]
            if ls is not None:
                o[i, :, :, cc:] = ls[y0:y0 + p, x0:x0 + p]

    def k_gather_patches_2x_f32(self, casi, lidar, casi_wp, lidar_wp, cc, cl, nb, points, n, p, out):
        ca, la = _arr(casi), _arr(lidar)
        pts = _arr(points, np.int32)[: 2 * n].reshape(n, 2)
        o = _arr(out)[: n * p * p * (cc + cl)].reshape(n, p, p, cc + cl)
        for i, (x0, y0) in enumerate(pts):
            sx, sy = int(x0 * 0.5) + nb - int(nb * 0.5), int(y0 * 0.5) + nb - int(nb * 0.5)
            for py in range(p):
                for px in range(p):
                    q = ((sy + int(py * 0.5)) * casi_wp + sx + int(px * 0.5)) * cc
                    o[i, py, px, :cc] = ca[q:q + cc]
                    q = ((y0 + py) * lidar_wp + x0 + px) * cl
                    o[i, py, px, cc:] = la[q:q + cl]

    def k_gather_pairs_f32(self, normal, shadow, idx, n, bands, ratio, u1, u2, rate, out_x, out_y):
        """gan_train_for_shadow.py:147-182: gather the pairs, then the two independent regulariser swaps -- the second
        one built from the already swapped normal spectrum, as in the reference."""
        ix = _arr(idx, np.int64)[:n]
        x = _arr(normal).reshape(-1, bands)[ix].copy()
        y = _arr(shadow).reshape(-1, bands)[ix].copy()
        if ratio is not None:
            rt = _arr(ratio)[:bands]
            a, b = _arr(u1)[:n] < np.float32(rate), _arr(u2)[:n] < np.float32(rate)
            x[a] = y[a] * rt
            y[b] = x[b] / rt
        _arr(out_x)[: n * bands] = x.reshape(-1)
        _arr(out_y)[: n * bands] = y.reshape(-1)

    def k_augment_patches_f32(self, x, idx, n, p, c, rot_k, pick, ratio, alt, flip_lr, flip_ud, delta, out):
        ix = np.arange(n) if idx is None else _arr(idx, np.int64)[:n]
        xa = _arr(x)
        o = _arr(out)[: n * p * p * c].reshape(n, p, p, c)
        rk = None if rot_k is None else _arr(rot_k, np.int32)[:n]
        pk = None if pick is None else _arr(pick, np.uint8)[:n]
        fl = None if flip_lr is None else _arr(flip_lr, np.uint8)[:n]
        fu = None if flip_ud is None else _arr(flip_ud, np.uint8)[:n]
        dl = None if delta is None else _arr(delta)[: n * c].reshape(n, c)
        al = None if alt is None else _arr(alt)[: n * p * p * c].reshape(n, p, p, c)
        rt = None if ratio is None else _arr(ratio)[:c]
        for s in range(n):
            v = xa[ix[s] * p * p * c:(ix[s] + 1) * p * p * c].reshape(p, p, c)
            if pk is not None and pk[s]:
                v = al[s] if al is not None else v / rt
            if rk is not None:
                v = np.rot90(v, int(rk[s]), axes=(0, 1))
            if fl is not None and fl[s]:
                v = v[:, ::-1]
            if fu is not None and fu[s]:
                v = v[::-1]
            if dl is not None:
                v = v + dl[s]
            o[s] = v

    def k_argmax_scatter(self, logits, ld, n, c, points, raster, raster_w):
        z = _mat(logits, ld, n, c)
        pts = _arr(points, np.int32)[: 2 * n].reshape(n, 2)
        r = _arr(raster, np.uint8)
        r[pts[:, 1].astype(np.int64) * raster_w + pts[:, 0]] = np.argmax(z, axis=1).astype(np.uint8)

    def k_fill_f32(self, dst, count, value):
        _arr(dst)[:count] = value

    def k_seg_gemm_res_f32(self, a, lda, ta, b, ldb, tb, c, ldc, n, groups, segs, tiles, n_tiles, bias, accumulate,
                           res, ldr, res_start):
        """Specification of the folded shortcut gradient: the plain product first, then the transposed channel
        map of `res` added row by row (row order of C)."""
        self.k_seg_gemm_f32(a, lda, ta, b, ldb, tb, c, ldc, n, groups, segs, tiles, n_tiles, bias, accumulate)
        C, R = _arr(c), _arr(res)
        g = groups.t.numpy()[groups.off:].view(GROUP_DTYPE)
        t = tiles.t.numpy()[tiles.off:].view(TILE_DTYPE)[:n_tiles]
        st = None if res_start is None else _arr(res_start, np.int32)[: n + 1]
        plain = {int(tt["group"]) for tt in t if int(tt["rows"]) and int(tt["flags"]) & 1}  # HYPEL_TILE_PLAIN: no gather
        for gi in sorted({int(tt["group"]) for tt in t if int(tt["rows"])} - plain):
            rows, co = int(g[gi]["rows"]), int(g[gi]["c_off"])
            cm = np.lib.stride_tricks.as_strided(C[co:], (rows, n), (ldc * 4, 4))
            r0 = co // ldc
            for col in range(n):
                o0, o1 = (col, col + 1) if st is None else (int(st[col]), int(st[col + 1]))
                if o1 > o0:
                    rm = np.lib.stride_tricks.as_strided(R[r0 * ldr + o0:], (rows, o1 - o0), (ldr * 4, 4))
                    cm[:, col] += rm.astype(np.float64).sum(1).astype(np.float32)

    def k_seg_gemm_stats_f32(self, a, lda, ta, b, ldb, tb, c, ldc, n, groups, segs, tiles, n_tiles, bias, accumulate,
                             stats):
        """The product, then per 128-row tile of its single group the (mean, sum of squared deviations) of every
        column -- the chunk format of k_col_stats_partial with chunk_rows = 128."""
        assert (accumulate & 1) == 0 and ldc == n
        self.k_seg_gemm_f32(a, lda, ta, b, ldb, tb, c, ldc, n, groups, segs, tiles, n_tiles, bias, accumulate)
        g = groups.t.numpy()[groups.off:].view(GROUP_DTYPE)
        t = tiles.t.numpy()[tiles.off:].view(TILE_DTYPE)[:n_tiles]
        assert not any(int(tt["flags"]) for tt in t), "HYPEL_TILE_PLAIN records are not valid in a statistics launch"
        gis = {int(tt["group"]) for tt in t}
        assert len(gis) == 1, "statistics epilogue: single-group launches only"
        grp = g[gis.pop()]
        self.k_col_stats_partial(c + int(grp["c_off"]), ldc, int(grp["rows"]), n, 128, stats)

    def k_seg_gemm_f32(self, a, lda, ta, b, ldb, tb, c, ldc, n, groups, segs, tiles, n_tiles, bias, accumulate):
        accumulate_raw = accumulate
        accumulate &= 1  # bits 8-9 carry a tile-width hint, bit 10 the paired-segments marker: no effect on the result
        A, B, C = _arr(a), _arr(b), _arr(c)
        g = groups.t.numpy()[groups.off:].view(GROUP_DTYPE)
        s = segs.t.numpy()[segs.off:].view(SEG_DTYPE)
        t = tiles.t.numpy()[tiles.off:].view(TILE_DTYPE)[:n_tiles]
        # tiles must cover every group's rows exactly once; all-zero records are empty blocks.  The result is specified
        # per group, from the group's own segment list.
        seen = {}
        any_split = False
        group_n = {}  # hypel_tile_t.n: a group's own column count (merged levels); every tile of a group agrees
        group_plain = {}  # hypel_tile_t.flags & HYPEL_TILE_PLAIN: K-slice partial (no bias / accumulate / activation)
        for tt in t:
            if int(tt["rows"]) != 0:
                gn = int(tt["n"])
                assert 0 <= gn <= n and group_n.setdefault(int(tt["group"]), gn) == gn, "per-tile n"
                pl = int(tt["flags"]) & 1
                assert int(tt["flags"]) in (0, 1) and group_plain.setdefault(int(tt["group"]), pl) == pl, "per-tile flags"
        if any(group_plain.values()):
            assert not ((accumulate_raw >> 16) & 7), "HYPEL_TILE_PLAIN records are not valid with HYPEL_GEMM_ACT_*"
        if any(group_n.values()):
            assert accumulate_raw & 0x4000, "tile records with their own n need HYPEL_GEMM_VAR_N"
        if accumulate_raw & 0x2000:
            assert n <= 64 and not ta and not tb, "HYPEL_GEMM_MFMA16X4: forward products with n <= 64"
        if accumulate_raw & 0x8000:  # HYPEL_GEMM_SPLIT6: same result to fp32 rounding, plain products only
            assert n > 16 and not (ta and tb) and not (accumulate_raw & 0x2400) and not ((accumulate_raw >> 16) & 7), \
                "HYPEL_GEMM_SPLIT6: plain NN / NT / TN products with n > 16"
        assert not (ta and tb), "A^T B^T products are not part of the path"
        n_launch = n
        for tt in t:
            if int(tt["rows"]) == 0:
                assert not any(int(tt[f]) for f in ("group", "m0", "seg_count")), "malformed empty record"
                continue
            seen.setdefault(int(tt["group"]), []).append(int(tt["m0"]))
        for gi, m0s in seen.items():
            rows = int(g[gi]["rows"])
            assert sorted(m0s) == list(range(0, rows, 128)), (gi, m0s, rows)
        for tt in t:  # every tile record repeats its group and first segment (include/hypel.h)
            if int(tt["rows"]) == 0:
                continue
            gg = g[int(tt["group"])]
            assert (int(tt["rows"]), int(tt["seg_begin"]), int(tt["seg_count"]), int(tt["c_off"])) == \
                (int(gg["rows"]), int(gg["seg_begin"]), int(gg["seg_count"]), int(gg["c_off"]))
            if int(gg["seg_count"]):
                s0 = s[int(gg["seg_begin"])]
                assert (int(tt["a_off0"]), int(tt["b_off0"]), int(tt["k0"])) == \
                    (int(s0["a_off"]), int(s0["b_off"]), int(s0["k"]))
        bv = _arr(bias)
        for gi in seen:
            grp = g[gi]
            rows = int(grp["rows"])
            n = group_n.get(gi, 0) or n_launch
            acc = np.zeros((rows, n), np.float64)
            s_lo, s_hi = int(grp["seg_begin"]), int(grp["seg_begin"]) + int(grp["seg_count"])
            for si in range(s_lo, s_hi):
                sg = s[si]
                k = int(sg["k"])
                if k & 0x40000000:  # HYPEL_SEG_PAIR_FLAG: shares a k-tile with the next segment (no effect on the result)
                    assert (accumulate_raw & 0x400) and not ta and tb and n > 16, "pair flag without HYPEL_GEMM_PAIRED_SEGS"
                    k &= ~0x40000000
                    assert k <= 16 and si + 1 < s_hi and 0 < int(s[si + 1]["k"]) <= 16, "malformed segment pair"
                    assert abs(int(sg["a_off"]) - int(s[si + 1]["a_off"])) * 4 < 2 ** 31
                    assert abs(int(sg["b_off"]) - int(s[si + 1]["b_off"])) * 4 < 2 ** 31
                if k == 0:
                    continue
                ao, bo = int(sg["a_off"]), int(sg["b_off"])
                if ta:
                    am = np.lib.stride_tricks.as_strided(A[ao:], (k, rows), (lda * 4, 4)).T
                else:
                    am = np.lib.stride_tricks.as_strided(A[ao:], (rows, k), (lda * 4, 4))
                if tb:
                    bm = np.lib.stride_tricks.as_strided(B[bo:], (n, k), (ldb * 4, 4)).T
                else:
                    bm = np.lib.stride_tricks.as_strided(B[bo:], (k, n), (ldb * 4, 4))
                acc += am.astype(np.float64) @ bm.astype(np.float64)
            co = int(grp["c_off"])
            # (c_off may point into another allocation: the K-slice partials live in a scratch buffer)
            cm = np.lib.stride_tricks.as_strided(_at(c, co, (rows - 1) * ldc + n), (rows, n), (ldc * 4, 4))
            plain = bool(group_plain.get(gi))
            if bv is not None and not plain:
                col0 = co % ldc
                acc += bv[col0:col0 + n]
            act_idx = (accumulate_raw >> 16) & 7  # HYPEL_GEMM_ACT_*: leaky-ReLU of (product + bias)
            if act_idx:
                assert not ta and not tb and not accumulate and not any_split and act_idx <= 4 and \
                    not (accumulate_raw & 0x6000), "HYPEL_GEMM_ACT_*: plain forward products only"
                alpha = np.float32([0.0, 0.1, 0.18, 0.2, 0.01][act_idx]).astype(np.float64)
                acc = np.where(acc > 0, acc, alpha * acc)
            if accumulate and not plain:
                cm += acc.astype(np.float32)
            else:
                cm[...] = acc.astype(np.float32)

    def k_copy_blocks_f32(self, base, entries, n_entries, max_block_elems):
        from hypelcnn_amd.backend import COPY_BLOCK_DTYPE
        ents = entries.t.numpy()[entries.off:].view(COPY_BLOCK_DTYPE)[:n_entries]
        assert all(int(e["rows"]) * int(e["cols"]) <= max_block_elems for e in ents), "max_block_elems"
        for e in ents:
            rows, cols, sld, dld = int(e["rows"]), int(e["cols"]), int(e["src_ld"]), int(e["dst_ld"])
            src = np.lib.stride_tricks.as_strided(_at(base, int(e["src_off"]), (rows - 1) * sld + cols), (rows, cols),
                                                  (sld * 4, 4))
            dst = np.lib.stride_tricks.as_strided(_at(base, int(e["dst_off"]), (rows - 1) * dld + cols), (rows, cols),
                                                  (dld * 4, 4))
            if int(e["flags"]) & 1:
                dst += src
            else:
                dst[...] = src

    def k_seg_gemm_multi_f32(self, base, ta, tb, tile_width, segs, blocks, n_blocks):
        """Specification of the merged filter-gradient launch: every block record is one 128 x tile_width output
        block of its own product C = A^T B (+ C when flagged), all offsets relative to `base`."""
        split6 = bool(tile_width & 0x100)  # HYPEL_GEMM_MULTI_SPLIT6: same result to fp32 rounding
        tile_width &= ~0x100
        assert ta == 1 and tb == 0 and tile_width in ((32, 64, 128) if split6 else (16, 32, 64))
        s = segs.t.numpy()[segs.off:].view(SEG_DTYPE)
        recs = blocks.t.numpy()[blocks.off:].view(MTILE_DTYPE)[:n_blocks]
        seen = set()
        for r in recs:
            rows, n, m0, n0 = int(r["rows"]), int(r["n"]), int(r["m0"]), int(r["n0"])
            if rows == 0:  # padding record (the XCD lists of the launch are padded to equal length)
                assert int(r["seg_count"]) == 0
                continue
            assert m0 % 128 == 0 and n0 % tile_width == 0 and m0 < rows and n0 < n
            key = (int(r["c_off"]), m0, n0)
            assert key not in seen, "two blocks write the same output tile"
            seen.add(key)
            mr, nc = min(128, rows - m0), min(tile_width, n - n0)
            lda, ldb, ldc = int(r["lda"]), int(r["ldb"]), int(r["ldc"])
            acc = np.zeros((mr, nc), np.float64)
            sb, sc = int(r["seg_begin"]), int(r["seg_count"])
            if sc:
                assert (int(r["a_off0"]), int(r["b_off0"]), int(r["k0"])) == \
                    (int(s[sb]["a_off"]), int(s[sb]["b_off"]), int(s[sb]["k"]))
            for sg in s[sb:sb + sc]:
                k = int(sg["k"])
                if k == 0:
                    continue
                a = _at(base, int(sg["a_off"]) + m0, (k - 1) * lda + mr)
                am = np.lib.stride_tricks.as_strided(a, (k, mr), (lda * 4, 4)).T
                b = _at(base, int(sg["b_off"]) + n0, (k - 1) * ldb + nc)
                bm = np.lib.stride_tricks.as_strided(b, (k, nc), (ldb * 4, 4))
                acc += am.astype(np.float64) @ bm.astype(np.float64)
            c = _at(base, int(r["c_off"]) + m0 * ldc + n0, (mr - 1) * ldc + nc)
            cm = np.lib.stride_tricks.as_strided(c, (mr, nc), (ldc * 4, 4))
            if int(r["flags"]) & 1:
                cm += acc.astype(np.float32)
            else:
                cm[...] = acc.astype(np.float32)

    def k_reduce_splits_multi_f32(self, base, entries, n_entries):
        ents = entries.t.numpy()[entries.off:].view(REDUCE_ENTRY_DTYPE)[:n_entries]
        for e in ents:
            count, stride, S = int(e["count"]), int(e["stride"]), int(e["n_splits"])
            p = _at(base, int(e["partial_off"]), (S - 1) * stride + count)
            o = _at(base, int(e["out_off"]), count)
            tot = np.zeros(count, np.float64)
            for k in range(S):
                tot += p[k * stride:k * stride + count]
            if int(e["flags"]) & 1:
                o += tot.astype(np.float32)
            else:
                o[...] = tot.astype(np.float32)

    def k_reduce_splits_multi_sized_f32(self, base, entries, n_entries, max_count):
        ents = entries.t.numpy()[entries.off:].view(REDUCE_ENTRY_DTYPE)[:n_entries]
        assert all(int(e["count"]) <= max_count for e in ents), "max_count below an entry's count"
        self.k_reduce_splits_multi_f32(base, entries, n_entries)

    def k_reduce_splits_wave_multi_f32(self, base, entries, n_entries, total_count):
        ents = entries.t.numpy()[entries.off:].view(REDUCE_ENTRY_DTYPE)[:n_entries]
        assert sum(int(e["count"]) for e in ents) == total_count
        outs = [(int(e["out_off"]), int(e["out_off"]) + int(e["count"])) for e in ents]
        assert all(a[1] <= b[0] or b[1] <= a[0] for i, a in enumerate(outs) for b in outs[i + 1:]), "overlapping outputs"
        self.k_reduce_splits_multi_f32(base, entries, n_entries)

    def k_reduce_splits_f32(self, partial, stride, n_splits, out, count, accumulate, bias=None, n=0, ldc=0):
        p = _arr(partial)
        idx = np.arange(count)
        if ldc > 0:
            idx = (idx // n) * ldc + idx % n
        ov = _arr(out)
        tot = np.zeros(count, np.float64)
        if bias is not None:
            tot += np.resize(_arr(bias)[:n], count)
        for k in range(n_splits):
            tot += p[k * stride + idx]
        if accumulate:
            ov[idx] += tot.astype(np.float32)
        else:
            ov[idx] = tot.astype(np.float32)

    def k_reduce_splits_pair_f32(self, p0, stride0, count0, out0, p1, stride1, count1, out1, n_splits, accumulate):
        self.k_reduce_splits_f32(p0, stride0, n_splits, out0, count0, accumulate)
        self.k_reduce_splits_f32(p1, stride1, n_splits, out1, count1, accumulate)

    def k_col_stats_partial(self, x, ld, rows, c, chunk_rows, partial):
        xm = _mat(x, ld, rows, c)
        n_chunks = (rows + chunk_rows - 1) // chunk_rows
        po = _arr(partial)[: n_chunks * 2 * c].reshape(n_chunks, 2, c)
        for k in range(n_chunks):
            blk = xm[k * chunk_rows:(k + 1) * chunk_rows].astype(np.float64)
            m = blk.mean(0)
            po[k, 0] = m
            po[k, 1] = ((blk - m) ** 2).sum(0)

    def k_act_bias_bwd_reduce(self, dz, lddz, y, ldy, rows, c, act, alpha, mask, ldm, chunk_rows, partial, dy, lddy):
        dyh, _ = self._dyh(dz, lddz, y, ldy, rows, c, None, None, None, act, alpha, mask, ldm)
        self.k_bn_act_bwd_reduce(dz, lddz, y, ldy, rows, c, None, None, None, act, alpha, mask, ldm, chunk_rows, partial)
        _mat(dy, lddy, rows, c)[...] = dyh.astype(np.float32)

    def k_bn_act_small_fwd(self, y, ldy, rows, c, eps, beta, act, alpha, mask, ldm, mean, rstd, mm, mv, decay, z, ldz):
        ym = _mat(y, ldy, rows, c).astype(np.float64)
        mu = ym.mean(0)
        var = ym.var(0)
        _arr(mean)[:c] = mu.astype(np.float32)
        _arr(rstd)[:c] = (1.0 / np.sqrt(var + eps)).astype(np.float32)
        if mm is not None:
            unbiased = var * rows / max(rows - 1, 1)
            _arr(mm)[:c] = (_arr(mm)[:c].astype(np.float64) * decay + mu * (1 - decay)).astype(np.float32)
            _arr(mv)[:c] = (_arr(mv)[:c].astype(np.float64) * decay + unbiased * (1 - decay)).astype(np.float32)
        self.k_bn_act_fwd(y, ldy, rows, c, mean, rstd, beta, act, alpha, mask, ldm, None, 0, None, None, 0, None, z, ldz)

    def k_bn_act_small_bwd(self, dz, lddz, y, ldy, rows, c, mean, rstd, beta, act, alpha, mask, ldm, dy, lddy, dparam,
                           accumulate):
        dyh, xhat = self._dyh(dz, lddz, y, ldy, rows, c, mean, rstd, beta, act, alpha, mask, ldm)
        s0, s1 = dyh.sum(0), (dyh * xhat).sum(0)
        if dparam is not None:
            dp = _arr(dparam)
            dp[:c] = (dp[:c] if accumulate else 0) + s0.astype(np.float32)
        g = _arr(rstd)[:c] * (dyh - s0 / rows - xhat * s1 / rows)
        _mat(dy, lddy, rows, c)[...] = g.astype(np.float32)

    def k_bn_finalize(self, partial, n_chunks, chunk_rows, rows, c, eps, mean, rstd, mm, mv, decay):
        po = _arr(partial)[: n_chunks * 2 * c].reshape(n_chunks, 2, c).astype(np.float64)
        n_a, mean_a, m2_a = 0.0, np.zeros(c), np.zeros(c)
        for k in range(n_chunks):
            n_b = min(rows, (k + 1) * chunk_rows) - k * chunk_rows
            d = po[k, 0] - mean_a
            n_ab = n_a + n_b
            mean_a = mean_a + d * n_b / n_ab
            m2_a = m2_a + po[k, 1] + d * d * n_a * n_b / n_ab
            n_a = n_ab
        var = m2_a / n_a
        _arr(mean)[:c] = mean_a
        _arr(rstd)[:c] = 1.0 / np.sqrt(var + eps)
        if mm is not None:
            unb = m2_a / (n_a - 1) if n_a > 1 else var
            mmv, mvv = _arr(mm)[:c], _arr(mv)[:c]
            mmv[...] = mmv * np.float64(decay) + mean_a * (1 - np.float64(decay))
            mvv[...] = mvv * np.float64(decay) + unb * (1 - np.float64(decay))

    def k_bn_merge_partials(self, partial, n_chunks, chunk_rows, rows, c, out):
        po = _arr(partial)[: n_chunks * 2 * c].reshape(n_chunks, 2, c).astype(np.float64)
        n_k = np.array([min(rows, (k + 1) * chunk_rows) - k * chunk_rows for k in range(n_chunks)], np.float64)
        mean_a = (n_k[:, None] * po[:, 0]).sum(0) / rows
        m2_a = (po[:, 1] + n_k[:, None] * (po[:, 0] - mean_a) ** 2).sum(0)
        o = _arr(out)
        o[:c], o[c:2 * c], o[2 * c] = mean_a, m2_a, rows

    def k_bn_finalize_ranks(self, gathered, world, c, eps, mean, rstd, mm, mv, decay):
        g = _arr(gathered)[: world * (2 * c + 1)].reshape(world, 2 * c + 1).astype(np.float64)
        n_k = g[:, 2 * c]
        n_a = n_k.sum()
        mean_a = (n_k[:, None] * g[:, :c]).sum(0) / n_a
        m2_a = (g[:, c:2 * c] + n_k[:, None] * (g[:, :c] - mean_a) ** 2).sum(0)
        var = m2_a / n_a
        _arr(mean)[:c] = mean_a
        _arr(rstd)[:c] = 1.0 / np.sqrt(var + eps)
        if mm is not None:
            unb = m2_a / (n_a - 1) if n_a > 1 else var
            mmv, mvv = _arr(mm)[:c], _arr(mv)[:c]
            mmv[...] = mmv * np.float64(decay) + mean_a * (1 - np.float64(decay))
            mvv[...] = mvv * np.float64(decay) + unb * (1 - np.float64(decay))

    def k_rstd_from_var(self, var, c, eps, rstd):
        _arr(rstd)[:c] = 1.0 / np.sqrt(_arr(var)[:c].astype(np.float64) + eps)

    def _pre(self, y, ldy, rows, c, mean, rstd, beta):
        v = _mat(y, ldy, rows, c).astype(np.float64)
        if mean is not None:
            xhat = (v - _arr(mean)[:c]) * _arr(rstd)[:c]
            return xhat, xhat + _arr(beta)[:c]
        return v, v

    def k_bn_act_fwd(self, y, ldy, rows, c, mean, rstd, beta, act, alpha, mask, ldm, r1, ld1, i1, r2, ld2, i2, z, ldz):
        _, pre = self._pre(y, ldy, rows, c, mean, rstd, beta)
        v = _act(pre, act, alpha)
        if mask is not None:
            v = v * _mat(mask, ldm, rows, c)
        for (r, ld, idx) in ((r1, ld1, i1), (r2, ld2, i2)):
            if r is None:
                continue
            if idx is None:
                v = v + _mat(r, ld, rows, c)
            else:
                ii = _arr(idx, np.int32)[:c]
                src = _mat(r, ld, rows, int(ii.max()) + 1)
                v = v + src[:, ii]
        _mat(z, ldz, rows, c)[...] = v.astype(np.float32)

    def _dyh(self, dz, lddz, y, ldy, rows, c, mean, rstd, beta, act, alpha, mask, ldm):
        xhat, pre = self._pre(y, ldy, rows, c, mean, rstd, beta)
        g = _mat(dz, lddz, rows, c).astype(np.float64)
        if mask is not None:
            g = g * _mat(mask, ldm, rows, c)
        return g * _act_grad(pre, act, alpha), xhat

    def k_bn_act_bwd_reduce(self, dz, lddz, y, ldy, rows, c, mean, rstd, beta, act, alpha, mask, ldm, chunk_rows,
                            partial):
        dyh, xhat = self._dyh(dz, lddz, y, ldy, rows, c, mean, rstd, beta, act, alpha, mask, ldm)
        n_chunks = (rows + chunk_rows - 1) // chunk_rows
        po = _arr(partial)[: n_chunks * 2 * c].reshape(n_chunks, 2, c)
        for k in range(n_chunks):
            sl = slice(k * chunk_rows, (k + 1) * chunk_rows)
            po[k, 0] = dyh[sl].sum(0)
            po[k, 1] = (dyh[sl] * xhat[sl]).sum(0)

    def k_bwd_reduce_finalize(self, partial, n_chunks, c, sums, dparam, accumulate):
        po = _arr(partial)[: n_chunks * 2 * c].reshape(n_chunks, 2, c).astype(np.float64)
        s = po.sum(0)
        _arr(sums)[: 2 * c] = s.reshape(-1)
        if dparam is not None:
            d = _arr(dparam)[:c]
            d[...] = (d if accumulate else 0) + s[0]

    def k_bn_act_bwd_apply(self, dz, lddz, y, ldy, rows, c, mean, rstd, beta, act, alpha, mask, ldm, sums, dy, lddy):
        self.k_bn_act_bwd_apply_global(dz, lddz, y, ldy, rows, c, mean, rstd, beta, act, alpha, mask, ldm, sums, rows,
                                       dy, lddy)

    def k_bn_act_bwd_apply_global(self, dz, lddz, y, ldy, rows, c, mean, rstd, beta, act, alpha, mask, ldm, sums,
                                  stat_rows, dy, lddy):
        assert stat_rows >= rows
        dyh, xhat = self._dyh(dz, lddz, y, ldy, rows, c, mean, rstd, beta, act, alpha, mask, ldm)
        g = dyh
        if mean is not None:
            s = _arr(sums)[: 2 * c].astype(np.float64).reshape(2, c)
            g = _arr(rstd)[:c] * (dyh - s[0] / stat_rows - xhat * s[1] / stat_rows)
        _mat(dy, lddy, rows, c)[...] = g.astype(np.float32)

    def k_chanmap_bwd(self, dz, lddz, rows, c, dr, lddr, cin, start, accumulate):
        g = _mat(dz, lddz, rows, c).astype(np.float64)
        if start is None:
            assert cin == c
            s = g
        else:
            st = _arr(start, np.int32)[: cin + 1]
            cs = np.concatenate([np.zeros((rows, 1)), np.cumsum(g, 1)], 1)
            s = cs[:, st[1:]] - cs[:, st[:-1]]
        d = _mat(dr, lddr, rows, cin)
        if accumulate:
            d += s.astype(np.float32)
        else:
            d[...] = s.astype(np.float32)

    def k_softmax_xent(self, logits, ld, n, c, labels, ldl, loss, dlogits, lddl, gscale):
        z = _mat(logits, ld, n, c).astype(np.float64)
        lab = _mat(labels, ldl, n, c).astype(np.float64)
        zm = z.max(1, keepdims=True)
        e = np.exp(z - zm)
        se = e.sum(1, keepdims=True)
        if loss is not None:
            _arr(loss)[:n] = -(lab * (z - zm - np.log(se))).sum(1)
        if dlogits is not None:
            _mat(dlogits, lddl, n, c)[...] = gscale * (e / se * lab.sum(1, keepdims=True) - lab)

    def k_mse(self, a, lda, b, ldb, rows, c, out, da, ldda, gscale, ws):
        d = _mat(a, lda, rows, c).astype(np.float64) - _mat(b, ldb, rows, c)
        _arr(out)[0] = (d * d).mean()
        if da is not None:
            _mat(da, ldda, rows, c)[...] = gscale * 2.0 * d / (rows * c)

    MSE_PARTIALS = 1024  # include/hypel.h HYPEL_MSE_PARTIALS

    def k_mse_partial_f32(self, a, lda, b, ldb, rows, c, da, ldda, gscale, ws):
        d = _mat(a, lda, rows, c).astype(np.float64) - _mat(b, ldb, rows, c)
        w = _arr(ws)[: self.MSE_PARTIALS]
        w[...] = 0.0
        w[0] = (d * d).sum()   # any split over the partials: only their sum is specified
        if da is not None:
            _mat(da, ldda, rows, c)[...] = gscale * 2.0 * d / (rows * c)

    def k_loss_finalize_f32(self, loss_rows, n_rows, mse_ws, mse_scale, out_ce, out_mse, flag, step):
        ce = np.float32(_arr(loss_rows)[:n_rows].astype(np.float64).mean())
        _arr(out_ce)[0] = ce
        ok = np.isfinite(ce)
        assert (mse_ws is None) == (out_mse is None)
        if mse_ws is not None:
            mse = np.float32(_arr(mse_ws)[: self.MSE_PARTIALS].astype(np.float64).sum() * mse_scale)
            _arr(out_mse)[0] = mse
            ok = ok and np.isfinite(mse)
        if flag is not None:
            _arr(flag)[0] = 0.0 if ok else 1.0
        if step is not None:
            step.t[step.off] += 1

    def k_sum_f32(self, x, count, scale, out, ws):
        _arr(out)[0] = _arr(x)[:count].astype(np.float64).sum() * scale

    def k_adam_tf1(self, p, g, m, v, count, lr_t, b1, b2, eps):
        pv, gv, mv, vv = (_arr(t)[:count] for t in (p, g, m, v))
        mv[...] = np.float32(b1) * mv + np.float32(1 - b1) * gv
        vv[...] = np.float32(b2) * vv + np.float32(1 - b2) * gv * gv
        pv -= np.float32(lr_t) * mv / (np.sqrt(vv) + np.float32(eps))

    def k_loss_guard_f32(self, loss_a, loss_b, flag):
        ok = np.isfinite(_arr(loss_a)[0]) and (loss_b is None or np.isfinite(_arr(loss_b)[0]))
        _arr(flag)[0] = 0.0 if ok else 1.0

    def k_adam_tf1_guarded(self, p, g, m, v, count, lr_t, b1, b2, eps, skip):
        if skip is None or _arr(skip)[0] == 0:
            self.k_adam_tf1(p, g, m, v, count, lr_t, b1, b2, eps)

    def k_momentum_tf1_guarded(self, p, g, a, count, lr, mu, skip):
        if skip is None or _arr(skip)[0] == 0:
            self.k_momentum_tf1(p, g, a, count, lr, mu)

    def k_momentum_tf1(self, p, g, a, count, lr, mu):
        pv, gv, av = (_arr(t)[:count] for t in (p, g, a))
        av[...] = np.float32(mu) * av + gv
        pv -= np.float32(lr) * av

    def k_dropout_mask(self, mask, count, keep, seed, step_dev):
        step = int(step_dev.t.numpy()[step_dev.off])
        rng = np.random.default_rng([int(seed) & 0xFFFFFFFF, step])
        _arr(mask)[:count] = (rng.random(count) < keep) / keep

    def k_step_inc(self, step_dev):
        step_dev.t[step_dev.off] += 1

    def k_argmax_confusion(self, logits, ld, n, c, labels, pred, confusion):
        z = _mat(logits, ld, n, c)
        best = z.argmax(1)
        if pred is not None:
            _arr(pred, np.int32)[:n] = best
        if confusion is not None and labels is not None:
            lab = _arr(labels, np.int32)[:n]
            conf = _arr(confusion, np.int32)[: c * c].reshape(c, c)
            np.add.at(conf, (lab, best), 1)

    def _lrn_s(self, xm, radius, bias, alpha):
        c = xm.shape[1]
        sq = xm * xm
        cs = np.concatenate([np.zeros((xm.shape[0], 1)), np.cumsum(sq, 1)], 1)
        lo = np.maximum(np.arange(c) - radius, 0)
        hi = np.minimum(np.arange(c) + radius + 1, c)
        return bias + alpha * (cs[:, hi] - cs[:, lo]), lo, hi

    def k_lrn_fwd(self, x, ldx, rows, c, radius, bias, alpha, beta, y, ldy):
        xm = _mat(x, ldx, rows, c).astype(np.float64)
        s, _, _ = self._lrn_s(xm, radius, bias, alpha)
        _mat(y, ldy, rows, c)[...] = (xm * s ** (-beta)).astype(np.float32)

    def k_lrn_bwd(self, x, ldx, dy, lddy, rows, c, radius, bias, alpha, beta, dx, lddx, accumulate):
        xm = _mat(x, ldx, rows, c).astype(np.float64)
        g = _mat(dy, lddy, rows, c).astype(np.float64)
        s, lo, hi = self._lrn_s(xm, radius, bias, alpha)
        t = g * xm * s ** (-beta - 1)
        ct = np.concatenate([np.zeros((rows, 1)), np.cumsum(t, 1)], 1)
        res = g * s ** (-beta) - 2 * alpha * beta * xm * (ct[:, hi] - ct[:, lo])
        d = _mat(dx, lddx, rows, c)
        if accumulate:
            d += res.astype(np.float32)
        else:
            d[...] = res.astype(np.float32)


# ----------------------------------------------------------------------------------------------- GAN kernels
def _gen_layout(bands):
    ks = [bands, bands // 2, bands // 4, bands // 8, bands // 4, bands // 2, bands]
    offs = np.concatenate([[0], np.cumsum(ks)]).astype(int)
    return ks, offs


def _conv1d_same(a, w):
    """a [N,B] float64, w [k]; out[p] = sum_j w[j] a[p + j - pl], pl = (k-1)//2."""
    n, bands = a.shape
    k = len(w)
    pl = (k - 1) // 2
    ap = np.pad(a, ((0, 0), (pl, k - 1 - pl)))
    out = np.zeros_like(a)
    for j in range(k):
        out += w[j] * ap[:, j:j + bands]
    return out


def _conv1d_same_T(g, w):
    """adjoint of _conv1d_same w.r.t. its input."""
    n, bands = g.shape
    k = len(w)
    pl = (k - 1) // 2
    gp = np.zeros((n, bands + k - 1))
    for j in range(k):
        gp[:, j:j + bands] += w[j] * g
    return gp[:, pl:pl + bands]


def _gen_forward(x, w, b, bands, only_encoder):
    ks, offs = _gen_layout(bands)
    a = [x]
    slopes = []
    hidden = 4 if only_encoder else 6
    for i in range(1, hidden + 1):
        c = _conv1d_same(a[-1], w[offs[i - 1]:offs[i]]) + b[i - 1]
        s = np.where(c > 0, 1.0, 0.1)
        slopes.append(s)
        v = c * s + a[-1]
        if i >= 2:
            v = v + a[-2]
        a.append(v)
    c7 = None
    if not only_encoder:
        c7 = _conv1d_same(a[6], w[offs[6]:offs[7]]) + b[6]
    return a, slopes, c7


def generator_knife_edge_rows(x, w, b, bands, only_encoder, eps=1e-6):
    """Samples with a hidden pre-activation within eps of 0: there the leaky-ReLU branch is decided by the rounding of an
    fp32 sum of up to 360 terms, and a flipped branch changes that sample's gradients by O(1) -- not a parity question.
    The generator tests re-draw such samples (tests/test_gpu_kernels.py)."""
    ks, offs = _gen_layout(bands)
    a = [np.asarray(x, np.float64)]
    w, b = np.asarray(w, np.float64), np.asarray(b, np.float64)
    near = np.zeros(a[0].shape[0], bool)
    for i in range(1, (4 if only_encoder else 6) + 1):
        c = _conv1d_same(a[-1], w[offs[i - 1]:offs[i]]) + b[i - 1]
        near |= (np.abs(c) < eps).any(axis=1)
        v = c * np.where(c > 0, 1.0, 0.1) + a[-1]
        if i >= 2:
            v = v + a[-2]
        a.append(v)
    return np.nonzero(near)[0]


def _emu_generator_blocks(n):
    return int(max(1, min(512, (n + 3) // 4)))


def _k_gan_generator_fwd(self, x, ldx, n, bands, w, b, only_encoder, out, ldo):
    xs = _mat(x, ldx, n, bands).astype(np.float64)
    ks, offs = _gen_layout(bands)
    wv, bv = _arr(w)[:offs[7]].astype(np.float64), _arr(b)[:7].astype(np.float64)
    a, _, c7 = _gen_forward(xs, wv, bv, bands, only_encoder)
    _mat(out, ldo, n, bands)[...] = (a[4] if only_encoder else np.tanh(c7)).astype(np.float32)


def _k_gan_generator_bwd(self, x, ldx, dout, lddo, n, bands, w, b, only_encoder, dx, lddx, accumulate_dx, pw, pb,
                         d_enc=None, ld_denc=0):
    xs = _mat(x, ldx, n, bands).astype(np.float64)
    g = _mat(dout, lddo, n, bands).astype(np.float64)
    ks, offs = _gen_layout(bands)
    wv, bv = _arr(w)[:offs[7]].astype(np.float64), _arr(b)[:7].astype(np.float64)
    a, slopes, c7 = _gen_forward(xs, wv, bv, bands, only_encoder)
    dw, db = np.zeros(offs[7]), np.zeros(8)
    hidden = 4 if only_encoder else 6
    da = [np.zeros_like(xs) for _ in range(7)]

    def layer_bwd(li, dc):
        k = ks[li]
        pl = (k - 1) // 2
        ap = np.pad(a[li], ((0, 0), (pl, k - 1 - pl)))
        for j in range(k):
            dw[offs[li] + j] += (dc * ap[:, j:j + bands]).sum()
        db[li] += dc.sum()
        da[li] += _conv1d_same_T(dc, wv[offs[li]:offs[li + 1]])

    if only_encoder:
        da[4] = g.copy()
    else:
        t = np.tanh(c7)
        layer_bwd(6, g * (1 - t * t))
    for i in range(hidden, 0, -1):
        if i == 4 and d_enc is not None:  # encoder tap: the gradient of the encoder-only application's output joins dn_4
            assert not only_encoder
            da[4] = da[4] + _mat(d_enc, ld_denc, n, bands).astype(np.float64)
        gi = da[i]
        da[i - 1] = da[i - 1] + gi
        if i >= 2:
            da[i - 2] = da[i - 2] + gi
        layer_bwd(i - 1, gi * slopes[i - 1])
    if dx is not None:
        d = _mat(dx, lddx, n, bands)
        if accumulate_dx:
            d += da[0].astype(np.float32)
        else:
            d[...] = da[0].astype(np.float32)
    blocks = _emu_generator_blocks(n)
    pwv = _arr(pw)[: blocks * offs[7]].reshape(blocks, offs[7])
    pbv = _arr(pb)[: blocks * 8].reshape(blocks, 8)
    pwv[...] = 0
    pbv[...] = 0
    pwv[0] = dw
    pbv[0] = db


def _k_gan_loss(self, mode, a, lda, b, ldb, rows, c, target, weight, loss, accumulate_loss, da, ldda, acc_da, db, lddb,
                acc_db, ws):
    av = _mat(a, lda, rows, c).astype(np.float64)
    cnt = rows * c
    if mode == 0:
        d = av - target
        val, ga, gb = (d * d).sum(), 2 * d, None
    elif mode == 1:
        d = av - _mat(b, ldb, rows, c)
        val, ga = np.abs(d).sum(), np.sign(d)
        gb = -ga
    else:
        val, ga, gb = av.sum(), np.ones_like(av), None
    lv = _arr(loss)
    lv[0] = (lv[0] if accumulate_loss else 0.0) + weight * val / cnt
    for ref, ld, acc, gr in ((da, ldda, acc_da, ga), (db, lddb, acc_db, gb)):
        if ref is not None and gr is not None:
            m = _mat(ref, ld, rows, c)
            upd = (weight / cnt * gr).astype(np.float32)
            if acc:
                m += upd
            else:
                m[...] = upd


def _k_l2_reg(self, w, count, scale, loss, accumulate_loss, dw, ws):
    wv = _arr(w)[:count].astype(np.float64)
    lv = _arr(loss)
    lv[0] = (lv[0] if accumulate_loss else 0.0) + 0.5 * scale * (wv * wv).sum()
    if dw is not None:
        _arr(dw)[:count] += (scale * wv).astype(np.float32)


def _k_loss_terms_slots(self, base, terms, n_terms, slots):
    ents = terms.t.numpy()[terms.off:].view(LOSS_TERM_DTYPE)[:n_terms]
    sl = _arr(slots)
    for e in ents:
        rows, c, mode = int(e["rows"]), int(e["c"]), int(e["mode"])
        slot = sl[int(e["slot"]) * 1024:(int(e["slot"]) + 1) * 1024]
        slot[...] = 0.0

        def mat(off, ld):
            if int(off) == LOSS_NONE:
                return None
            flat = _at(base, int(off), (rows - 1) * int(ld) + c)
            return np.lib.stride_tricks.as_strided(flat, shape=(rows, c), strides=(int(ld) * 4, 4))

        if mode == 3:
            w = _at(base, int(e["a_off"]), rows).astype(np.float64)
            slot[0] = np.float32(float(e["pscale"]) * (w * w).sum())
            if int(e["da_off"]) != LOSS_NONE:
                _at(base, int(e["da_off"]), rows)[...] += (float(e["gcoef"]) * w).astype(np.float32)
            continue
        av = mat(e["a_off"], e["lda"]).astype(np.float64)
        if mode == 0:
            d = av - float(e["target"])
            val, ga, gb = (d * d).sum(), 2 * d, None
        elif mode == 1:
            d = av - mat(e["b_off"], e["ldb"])
            val, ga = np.abs(d).sum(), np.sign(d)
            gb = -ga
        else:
            val, ga, gb = av.sum(), np.ones_like(av), None
        slot[0] = np.float32(float(e["pscale"]) * val)
        for off, ld, acc, gr in ((e["da_off"], e["ldda"], e["acc_da"], ga), (e["db_off"], e["lddb"], e["acc_db"], gb)):
            m = mat(off, ld)
            if m is not None and gr is not None:
                upd = (float(e["gcoef"]) * gr).astype(np.float32)
                if int(acc):
                    m += upd
                else:
                    m[...] = upd


def _k_loss_finalize_slots(self, slots, n_slots, loss, accumulate_loss):
    lv = _arr(loss)
    lv[0] = (lv[0] if accumulate_loss else 0.0) + _arr(slots)[: n_slots * 1024].astype(np.float64).sum()


def _k_l2norm_fwd(self, x, ldx, rows, c, y, ldy, stat):
    xv = _mat(x, ldx, rows, c).astype(np.float64)
    ss = (xv * xv).sum()
    inv = 1.0 / np.sqrt(max(ss, 1e-12))
    st = _arr(stat)
    st[0], st[1] = ss, inv
    _mat(y, ldy, rows, c)[...] = (xv * inv).astype(np.float32)


def _k_l2norm_bwd(self, x, ldx, dy, lddy, rows, c, stat, dx, lddx, accumulate):
    xv = _mat(x, ldx, rows, c).astype(np.float64)
    g = _mat(dy, lddy, rows, c).astype(np.float64)
    st = _arr(stat)
    inv = float(st[1])
    coef = float((g * xv).sum()) * inv ** 3 if st[0] > 1e-12 else 0.0
    res = g * inv - xv * coef
    d = _mat(dx, lddx, rows, c)
    if accumulate:
        d += res.astype(np.float32)
    else:
        d[...] = res.astype(np.float32)


def _k_l2norm_parts_fwd(self, x, ldx, rows, c, parts, y, ldy, stat):
    for p in range(parts):
        _k_l2norm_fwd(self, x + p * c, ldx, rows, c, y + p * c, ldy, stat + 2 * p)


def _k_l2norm_parts_bwd(self, x, ldx, dy, lddy, rows, c, parts, stat, dx, lddx, accumulate):
    for p in range(parts):
        _k_l2norm_bwd(self, x + p * c, ldx, dy + p * c, lddy, rows, c, stat + 2 * p, dx + p * c, lddx, accumulate)


def _k_l2norm_segs_fwd(self, x, ldx, rows, c, parts, segs, y, ldy, stat):
    for g in range(segs):
        _k_l2norm_parts_fwd(self, x + g * rows * ldx, ldx, rows, c, parts, y + g * rows * ldy, ldy, stat + 2 * parts * g)


def _k_l2norm_segs_bwd(self, x, ldx, dy, lddy, rows, c, parts, segs, stat, dx, lddx, accumulate):
    for g in range(segs):
        _k_l2norm_parts_bwd(self, x + g * rows * ldx, ldx, dy + g * rows * lddy, lddy, rows, c, parts,
                            stat + 2 * parts * g, dx + g * rows * lddx, lddx, accumulate)


def _k_nce_loss(self, g, ldg, r, ldr, n, p, e, tau, weight, loss, accumulate_loss, dg, lddg, acc_dg, dr, lddr, acc_dr,
                ws):
    gv = _mat(g, ldg, n, p * e).astype(np.float64).reshape(n, p, e)
    rv = _mat(r, ldr, n, p * e).astype(np.float64).reshape(n, p, e)
    logits = np.einsum("npe,nqe->npq", gv, rv) / tau
    flat = logits.reshape(n, -1)
    mx = flat.max(1, keepdims=True)
    ex = np.exp(flat - mx)
    se = ex.sum(1, keepdims=True)
    lse = (mx + np.log(se))[:, 0]
    per = p * lse - np.trace(logits, axis1=1, axis2=2)
    lv = _arr(loss)
    lv[0] = (lv[0] if accumulate_loss else 0.0) + weight * per.mean()
    dl = (p * ex / se).reshape(n, p, p) - np.eye(p)
    coef = weight / n / tau
    for ref, ld, acc, gr in ((dg, lddg, acc_dg, np.einsum("npq,nqe->npe", dl, rv)),
                             (dr, lddr, acc_dr, np.einsum("npq,npe->nqe", dl, gv))):
        if ref is not None:
            m = _mat(ref, ld, n, p * e)
            upd = (coef * gr).reshape(n, p * e).astype(np.float32)
            if acc:
                m += upd
            else:
                m[...] = upd


EmuBackend.k_l2norm_parts_fwd = _k_l2norm_parts_fwd
EmuBackend.k_l2norm_parts_bwd = _k_l2norm_parts_bwd
EmuBackend.k_l2norm_segs_fwd = _k_l2norm_segs_fwd
EmuBackend.k_l2norm_segs_bwd = _k_l2norm_segs_bwd
def _k_gan_generator_fwd_tap(self, x, ldx, n, bands, w, b, out, ldo, enc_out, ld_enc, keep):
    """Specification of the encoder tap: the full forward, and what the encoder-only forward on the same input writes."""
    assert 16 <= bands <= 384, "matrix-core kernels only"
    _k_gan_generator_fwd(self, x, ldx, n, bands, w, b, 0, out, ldo)
    _k_gan_generator_fwd(self, x, ldx, n, bands, w, b, 1, enc_out, ld_enc)


def _k_gan_generator_bwd_tap(self, x, ldx, dout, lddo, d_enc, ld_denc, n, bands, w, b, dx, lddx, acc, pw, pb, keep):
    assert 16 <= bands <= 384 and d_enc is not None
    _k_gan_generator_bwd(self, x, ldx, dout, lddo, n, bands, w, b, 0, dx, lddx, acc, pw, pb, d_enc, ld_denc)


EmuBackend.k_gan_generator_fwd = _k_gan_generator_fwd
EmuBackend.k_gan_generator_fwd_tap = _k_gan_generator_fwd_tap
EmuBackend.k_gan_generator_bwd_tap = _k_gan_generator_bwd_tap
EmuBackend.gan_generator_tap_supported = lambda self, bands: 16 <= bands <= 384
EmuBackend.k_gan_generator_bwd = _k_gan_generator_bwd
EmuBackend.k_gan_loss = _k_gan_loss
EmuBackend.k_l2_reg = _k_l2_reg
EmuBackend.k_loss_finalize_slots = _k_loss_finalize_slots
EmuBackend.k_loss_terms_slots = _k_loss_terms_slots
EmuBackend.k_l2norm_fwd = _k_l2norm_fwd
EmuBackend.k_l2norm_bwd = _k_l2norm_bwd
EmuBackend.k_nce_loss = _k_nce_loss
def _ds_layers(n_layers, widths, act_mask, w, b):
    ws, bs, wo, bo = [], [], 0, 0
    for l in range(n_layers):
        cin, cout = widths[l], widths[l + 1]
        ws.append(_arr(w)[wo:wo + cin * cout].reshape(cin, cout).astype(np.float64))
        bs.append(_arr(b)[bo:bo + cout].astype(np.float64))
        wo += cin * cout
        bo += cout
    return ws, bs, [bool((act_mask >> l) & 1) for l in range(n_layers)], wo, bo


def _ds_forward(x, ws, bs, lrelu, alpha):
    a = [x]
    for wl, bl, act in zip(ws, bs, lrelu):
        v = a[-1] @ wl + bl
        a.append(np.where(v > 0, v, alpha * v) if act else v)
    return a


def _k_dense_stack_fwd(self, x, ldx, n, n_layers, w0, w1, w2, w3, w4, act_mask, alpha, w, b, out, ldo):
    widths = [w0, w1, w2, w3, w4]
    ws, bs, lrelu, _, _ = _ds_layers(n_layers, widths, act_mask, w, b)
    a = _ds_forward(_mat(x, ldx, n, w0).astype(np.float64), ws, bs, lrelu, alpha)
    _mat(out, ldo, n, widths[n_layers])[...] = a[-1].astype(np.float32)


def _k_dense_stack_bwd(self, x, ldx, dout, lddo, n, n_layers, w0, w1, w2, w3, w4, act_mask, alpha, w, b, dx, lddx, acc,
                       pw, pb):
    widths = [w0, w1, w2, w3, w4]
    ws, bs, lrelu, wtotal, btotal = _ds_layers(n_layers, widths, act_mask, w, b)
    a = _ds_forward(_mat(x, ldx, n, w0).astype(np.float64), ws, bs, lrelu, alpha)
    g = _mat(dout, lddo, n, widths[n_layers]).astype(np.float64)
    dws, dbs = [None] * n_layers, [None] * n_layers
    for l in range(n_layers - 1, -1, -1):
        if lrelu[l]:
            g = np.where(a[l + 1] > 0, g, alpha * g)
        dbs[l] = g.sum(0)
        dws[l] = a[l].T @ g
        g = g @ ws[l].T
    if dx is not None:
        d = _mat(dx, lddx, n, w0)
        d[...] = (d if acc else 0) + g.astype(np.float32)
    blocks = _emu_dense_stack_blocks(n)
    pwv = _arr(pw)[: blocks * wtotal].reshape(blocks, wtotal)
    pbv = _arr(pb)[: blocks * btotal].reshape(blocks, btotal)
    pwv[...] = 0
    pbv[...] = 0
    pwv[0] = np.concatenate([d.reshape(-1) for d in dws])
    pbv[0] = np.concatenate(dbs)


def _emu_dense_stack_blocks(n):
    return int(max(1, min(256, (n + 15) // 16)))


EmuBackend.k_dense_stack_fwd = _k_dense_stack_fwd
EmuBackend.k_dense_stack_bwd = _k_dense_stack_bwd
EmuBackend.dense_stack_blocks = lambda self, n: _emu_dense_stack_blocks(n)
def _emu_dense_stack_supported(widths):
    """mirror of hypel_dense_stack_supported: <= 4 layers, widths <= 128, all images + weights within 160 KB of LDS"""
    if not (2 <= len(widths) <= 5 and all(1 <= v <= 128 for v in widths)):
        return False
    w16 = (max(widths) + 15) // 16 * 16
    p = w16 // 32 * 32 + 18
    p = p if p >= w16 else p + 32
    return (8 * 16 * p + sum((c + 3) // 4 * 4 * p for c in widths[:-1]) + sum(widths[1:])) * 4 <= 160 * 1024


EmuBackend.dense_stack_supported = lambda self, widths: _emu_dense_stack_supported(list(widths))
EmuBackend.gan_generator_blocks = lambda self, n: _emu_generator_blocks(n)
# kept activations: the emulation recomputes (the product's two pairs are bit-identical); a nominal buffer size so that
# the planner takes the same path as on the device
EmuBackend.gan_generator_keep_floats = lambda self, n, bands, only_encoder: 16 if 16 <= bands <= 384 else 0
EmuBackend.k_gan_generator_fwd_keep = lambda self, x, ldx, n, bands, w, b, enc, out, ldo, keep: _k_gan_generator_fwd(
    self, x, ldx, n, bands, w, b, enc, out, ldo)
EmuBackend.k_gan_generator_bwd_kept = lambda self, x, ldx, dout, lddo, n, bands, w, b, enc, dx, lddx, acc, pw, pb, keep: \
    _k_gan_generator_bwd(self, x, ldx, dout, lddo, n, bands, w, b, enc, dx, lddx, acc, pw, pb)


# ---- several same-shaped applications with different variables in one launch (include/hypel.h: *_apps) ----------------
def _emu_generator_blocks_apps(n, n_apps):
    return int(max(1, min(max(1, 512 // n_apps), (n + 15) // 16))) * n_apps


def _emu_dense_stack_blocks_apps(n, n_apps):
    return int(max(1, min(max(1, 256 // n_apps), (n + 15) // 16))) * n_apps


def _apps_slabs(pw, pb, n_apps, bpa, pw_stride, pb_stride, wtotal, btotal):
    """per application: Refs of its first slab, after zeroing its bpa slabs"""
    out = []
    for g in range(n_apps):
        rw = _shift(pw, g * (pw_stride if pw_stride else bpa * wtotal))
        rb = _shift(pb, g * (pb_stride if pw_stride else bpa * btotal))
        _arr(rw)[: bpa * wtotal] = 0
        _arr(rb)[: bpa * btotal] = 0
        out.append((rw, rb))
    return out


def _k_gan_generator_fwd_apps(self, x, ldx, n, n_apps, w_stride, b_stride, bands, w, b, only_encoder, out, ldo, keep):
    for g in range(n_apps):
        _k_gan_generator_fwd(self, x + g * n * ldx, ldx, n, bands, _shift(w, g * w_stride), _shift(b, g * b_stride),
                             only_encoder, out + g * n * ldo, ldo)


def _k_gan_generator_bwd_apps(self, x, ldx, dout, lddo, n, n_apps, w_stride, b_stride, pw_stride, pb_stride, bands, w, b,
                              only_encoder, dx, lddx, acc, pw, pb, keep):
    _, offs = _gen_layout(bands)
    bpa = _emu_generator_blocks_apps(n, n_apps) // n_apps
    one = _emu_generator_blocks(n)
    tw, tb = torch.zeros(one * offs[7]), torch.zeros(one * 8)
    for g, (rw, rb) in enumerate(_apps_slabs(pw, pb, n_apps, bpa, pw_stride, pb_stride, offs[7], 8)):
        _k_gan_generator_bwd(self, x + g * n * ldx, ldx, dout + g * n * lddo, lddo, n, bands, _shift(w, g * w_stride),
                             _shift(b, g * b_stride), only_encoder, None if dx is None else dx + g * n * lddx, lddx, acc,
                             Ref(tw), Ref(tb))
        _arr(rw)[: offs[7]] = tw.numpy()[: offs[7]]
        _arr(rb)[:8] = tb.numpy()[:8]


def _k_dense_stack_fwd_apps(self, x, ldx, n, n_apps, w_stride, b_stride, n_layers, w0, w1, w2, w3, w4, act_mask, alpha, w,
                            b, out, ldo):
    for g in range(n_apps):
        _k_dense_stack_fwd(self, x + g * n * ldx, ldx, n, n_layers, w0, w1, w2, w3, w4, act_mask, alpha,
                           _shift(w, g * w_stride), _shift(b, g * b_stride), out + g * n * ldo, ldo)


def _k_dense_stack_bwd_apps(self, x, ldx, dout, lddo, n, n_apps, w_stride, b_stride, pw_stride, pb_stride, n_layers, w0, w1,
                            w2, w3, w4, act_mask, alpha, w, b, dx, lddx, acc, pw, pb):
    widths = [w0, w1, w2, w3, w4]
    wtotal = sum(widths[l] * widths[l + 1] for l in range(n_layers))
    btotal = sum(widths[1:n_layers + 1])
    bpa = _emu_dense_stack_blocks_apps(n, n_apps) // n_apps
    one = _emu_dense_stack_blocks(n)
    tw, tb = torch.zeros(one * wtotal), torch.zeros(one * btotal)
    for g, (rw, rb) in enumerate(_apps_slabs(pw, pb, n_apps, bpa, pw_stride, pb_stride, wtotal, btotal)):
        _k_dense_stack_bwd(self, x + g * n * ldx, ldx, dout + g * n * lddo, lddo, n, n_layers, w0, w1, w2, w3, w4, act_mask,
                           alpha, _shift(w, g * w_stride), _shift(b, g * b_stride),
                           None if dx is None else dx + g * n * lddx, lddx, acc, Ref(tw), Ref(tb))
        _arr(rw)[:wtotal] = tw.numpy()[:wtotal]
        _arr(rb)[:btotal] = tb.numpy()[:btotal]


EmuBackend.k_gan_generator_fwd_apps = _k_gan_generator_fwd_apps
EmuBackend.k_gan_generator_bwd_apps = _k_gan_generator_bwd_apps
EmuBackend.k_dense_stack_fwd_apps = _k_dense_stack_fwd_apps
EmuBackend.k_dense_stack_bwd_apps = _k_dense_stack_bwd_apps
EmuBackend.gan_generator_blocks_apps = lambda self, n, n_apps: _emu_generator_blocks_apps(n, n_apps)
EmuBackend.dense_stack_blocks_apps = lambda self, n, n_apps: _emu_dense_stack_blocks_apps(n, n_apps)


def _k_copy_pair_f32(self, dst0, src0, n0, dst1, src1, n1):
    _arr(dst0)[:n0] = _arr(src0)[:n0]
    if n1 > 0:
        _arr(dst1)[:n1] = _arr(src1)[:n1]


EmuBackend.k_copy_pair_f32 = _k_copy_pair_f32
