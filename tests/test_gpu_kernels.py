"""-m gpu: every HIP kernel, called through the C-ABI, against the executable spec (tests/emu_backend.py,
itself validated against the oracle by tests/test_host_plan_emu.py) on identical random inputs, at the
ragged shapes the models actually produce (K = 145, n = 15/60/120, rows % 128 != 0, channel offsets)."""
import numpy as np
import pytest
import torch

from hypelcnn_amd.backend import GROUP_DTYPE, SEG_DTYPE, TILE_DTYPE, Ref
from hypelcnn_amd.plan import GemmTables, TowerPlan
from tests import emu_backend
from tests.emu_backend import EmuBackend

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from hypelcnn_amd.backend import HipBackend
    return HipBackend()


class Both:
    """Mirror host arrays onto both backends; run the same launch; compare named outputs."""

    def __init__(self, hip):
        self.hip, self.emu = hip, EmuBackend()
        self.h, self.e = {}, {}

    def arr(self, name, a):
        a = np.ascontiguousarray(a)
        self.e[name] = self.emu.upload(a)
        self.h[name] = self.hip.upload(a)
        return name

    def run(self, kernel, *args):
        def conv(store):
            out = []
            for a in args:
                if isinstance(a, tuple) and len(a) == 2 and isinstance(a[0], str):
                    out.append(Ref(store[a[0]], a[1]))
                elif isinstance(a, str) and a in store:
                    out.append(Ref(store[a]))
                else:
                    out.append(a)
            return out
        self.emu.call(kernel, *conv(self.e))
        self.hip.call(kernel, *conv(self.h))
        self.hip.synchronize()

    def check(self, name, rtol=1e-4, atol=1e-5, dtype=np.float32):
        got = self.h[name].cpu().numpy().view(dtype) if dtype != np.float32 else self.h[name].cpu().numpy()
        ref = self.e[name].numpy().view(dtype) if dtype != np.float32 else self.e[name].numpy()
        scale = max(1.0, float(np.abs(ref).max()))
        np.testing.assert_allclose(got, ref, rtol=rtol, atol=atol * scale, err_msg=name)


def _tables(b, groups):  # (b: unused, kept for the call sites)
    tb = GemmTables()
    for g in groups:
        tb.add_group(*g)
    return tb


@pytest.mark.parametrize("rows,k,n,ta,tb_,acc,bias", [
    (300, 145, 120, 0, 0, 0, False),   # conv_enc_0 shape, ragged rows
    (128, 32, 32, 0, 0, 0, False),
    (1, 1, 1, 0, 0, 0, True),
    (257, 60, 15, 0, 0, 1, True),      # level-2 branch, n=15, accumulate + bias
    (200, 480, 480, 0, 0, 0, False),
    (131, 120, 145, 0, 1, 1, False),   # dgrad: B transposed
    (64, 980, 60, 0, 1, 0, False),
    (145, 333, 120, 1, 0, 0, False),   # wgrad: A transposed, M = Cin = 145
    (60, 1000, 60, 1, 0, 0, False),
    (15, 77, 45, 1, 0, 1, False),
    # n <= 16: the 128x16 blocks on the 16x16x4 MFMA, the three operand layouts of the path, ragged K / rows, k-tile halves
    (300, 120, 15, 0, 0, 0, True),
    (120, 1085, 15, 1, 0, 0, False),   # level-2 filter gradient: M = Cin = 120, n = 15
    (257, 49, 16, 0, 1, 1, False),
    (129, 7, 1, 0, 0, 1, False),
])
def test_seg_gemm_single_segment(hip, rows, k, n, ta, tb_, acc, bias):
    rng = np.random.default_rng(rows * 7 + k)
    b = Both(hip)
    lda = (rows if ta else k) + 3
    ldb = (k if tb_ else n) + 5
    ldc = n + 2
    a = rng.standard_normal(((k if ta else rows), lda)).astype(np.float32)
    bm = rng.standard_normal(((n if tb_ else k), ldb)).astype(np.float32)
    c0 = rng.standard_normal((rows, ldc)).astype(np.float32)
    bv = rng.standard_normal(ldc).astype(np.float32)
    tabs = _tables(b, [(0, [(0, 0, k)], rows)])
    garr, sarr, tarr, _ = tabs.finalize(n)
    for nm, arr in (("a", a), ("b", bm), ("c", c0), ("bias", bv), ("g", garr), ("s", sarr), ("t", tarr)):
        b.arr(nm, arr)
    b.run("seg_gemm_f32", "a", lda, ta, "b", ldb, tb_, "c", ldc, n, "g", "s", "t", len(tarr), "bias" if bias else None,
          acc)
    b.check("c", rtol=2e-4, atol=2e-5)


def test_seg_gemm_multi_group_multi_segment(hip):
    """Conv-like: 9 groups (pixels) x up to 9 segments (taps) with channel offsets, two cout branches merged."""
    rng = np.random.default_rng(3)
    nb, cin, cout, c_total, P = 70, 37, 20, 45, 9
    x = rng.standard_normal(P * nb * cin).astype(np.float32)
    w = rng.standard_normal(9 * cin * cout + 11).astype(np.float32)
    y = np.zeros(P * nb * c_total, np.float32)
    groups = []
    for p in range(P):
        oy, ox = divmod(p, 3)
        segs = []
        for i in range(3):
            for j in range(3):
                iy, ix = oy + i - 1, ox + j - 1
                if 0 <= iy < 3 and 0 <= ix < 3:
                    segs.append(((iy * 3 + ix) * nb * cin, 11 + (i * 3 + j) * cin * cout, cin))
        groups.append((p * nb * c_total + 25, segs, nb))
    b = Both(hip)
    garr, sarr, tarr, macs = _tables(b, groups).finalize(cout)
    bias = rng.standard_normal(c_total).astype(np.float32)
    for nm, arr in (("x", x), ("w", w), ("y", y), ("bias", bias), ("g", garr), ("s", sarr), ("t", tarr)):
        b.arr(nm, arr)
    b.run("seg_gemm_f32", "x", cin, 0, "w", cout, 0, "y", c_total, cout, "g", "s", "t", len(tarr), "bias", 0)
    b.check("y", rtol=2e-4, atol=2e-5)
    got = b.h["y"].cpu().numpy().reshape(P * nb, c_total)
    assert (got[:, :25] == 0).all(), "columns outside the branch must stay untouched"


@pytest.mark.parametrize("cin,cout,mapped,acc", [(120, 240, True, 0), (240, 120, True, 1), (60, 60, False, 0),
                                                  (145, 480, True, 0)])
def test_seg_gemm_res_epilogue(hip, cin, cout, mapped, acc):
    """Data gradient with the folded shortcut gradient: dX[P*nb, cin] = dY @ W^T (+)= ... + map^T(dZ)."""
    rng = np.random.default_rng(cin + cout)
    nb, P = 70, 3
    rows = P * nb
    dy = rng.standard_normal((rows, cout)).astype(np.float32)
    dz = rng.standard_normal((rows, cout)).astype(np.float32)
    w = rng.standard_normal((cin, cout)).astype(np.float32)
    ld = cin + 5
    dx = rng.standard_normal((rows, ld)).astype(np.float32)
    sc = cin / cout
    idx = (np.arange(cout) // (cout // cin) if cout % cin == 0 else
           np.minimum(np.round(np.arange(cout) * sc), cin - 1).astype(int))
    start = np.searchsorted(idx, np.arange(cin + 1), side="left").astype(np.int32)
    b = Both(hip)
    # one group per pixel, writing at channel offset 2 of a wider gradient buffer
    groups = [(p * nb * ld + 2, [(p * nb * cout, 0, cout)], nb) for p in range(P)]
    garr, sarr, tarr, _ = _tables(b, groups).finalize(cin)
    for nm, arr in (("dy", dy), ("dz", dz), ("w", w), ("dx", dx), ("g", garr), ("s", sarr), ("t", tarr),
                    ("start", start)):
        b.arr(nm, arr)
    b.run("seg_gemm_res_f32", "dy", cout, 0, "w", cout, 1, "dx", ld, cin, "g", "s", "t", len(tarr), None, acc, "dz",
          cout, "start" if mapped else None)
    b.check("dx", rtol=2e-4, atol=2e-5)
    got = b.h["dx"].cpu().numpy().reshape(rows, ld)
    want = (dx[:, 2:2 + cin] if acc else 0) + dy @ w.T
    if mapped:
        for ci in range(cin):
            want[:, ci] += dz[:, start[ci]:start[ci + 1]].sum(1)
    else:
        want = want + dz[:, :cin]
    np.testing.assert_allclose(got[:, 2:2 + cin], want, rtol=2e-4, atol=2e-4)
    np.testing.assert_array_equal(got[:, :2], dx[:, :2])


@pytest.mark.parametrize("rows,k,n,hint", [(50176 // 8, 120, 240, 2), (300, 145, 120, 0), (1000, 480, 480, 1),
                                          (129, 33, 60, 2), (128 * 5, 64, 33, 0)])
def test_seg_gemm_stats_epilogue(hip, rows, k, n, hint):
    """1x1 convolution forward with the batch-norm statistics in the epilogue: the product itself, and per 128-row
    tile the (mean, sum of squared deviations) of every column in the chunk format hypel_bn_finalize merges; then the
    finaliser on those partials against the plain statistics of Y (both tile widths, ragged rows / columns)."""
    rng = np.random.default_rng(rows + n)
    b = Both(hip)
    a = (rng.standard_normal((rows, k)) + 0.5).astype(np.float32)
    w = rng.standard_normal((k, n)).astype(np.float32)
    garr, sarr, tarr, _ = _tables(b, [(0, [(0, 0, k)], rows)]).finalize(n)
    n_chunks = (rows + 127) // 128
    for nm, arr in (("a", a), ("w", w), ("y", np.zeros(rows * n, np.float32)), ("g", garr), ("s", sarr), ("t", tarr),
                    ("part", np.zeros(n_chunks * 2 * n, np.float32)), ("mean", np.zeros(n, np.float32)),
                    ("rstd", np.zeros(n, np.float32)), ("mm", np.zeros(n, np.float32)), ("mv", np.ones(n, np.float32))):
        b.arr(nm, arr)
    b.run("seg_gemm_stats_f32", "a", k, 0, "w", n, 0, "y", n, n, "g", "s", "t", len(tarr), None, hint << 8, "part")
    b.check("y", rtol=2e-4, atol=2e-5)
    b.check("part", rtol=1e-3, atol=2e-4)
    b.run("bn_finalize", "part", n_chunks, 128, rows, n, 1e-3, "mean", "rstd", "mm", "mv", 0.95)
    y = a.astype(np.float64) @ w.astype(np.float64)
    got_mean, got_rstd = b.h["mean"].cpu().numpy(), b.h["rstd"].cpu().numpy()
    np.testing.assert_allclose(got_mean, y.mean(0), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(got_rstd, 1.0 / np.sqrt(y.var(0) + 1e-3), rtol=1e-4)
    np.testing.assert_allclose(b.h["mv"].cpu().numpy(), 0.95 + 0.05 * y.var(0, ddof=1), rtol=1e-4)


SPLIT6 = 0x8000  # include/hypel.h HYPEL_GEMM_SPLIT6


@pytest.mark.parametrize("rows,k,n,ta,tb_,hint,kind", [
    (128 * 5 + 37, 480, 480, 0, 0, 3, "stats"), (128 * 3, 145, 120, 0, 0, 1, "bias"), (300, 145, 120, 0, 0, 2, "bias"),
    (128 * 4 + 5, 240, 240, 0, 1, 2, "res"), (128 * 2 + 100, 300, 200, 0, 1, 3, "acc"), (128 * 3, 64, 60, 0, 0, 2, "bias"),
    (131, 120, 145, 0, 1, 1, "acc"), (64, 980, 60, 0, 1, 2, "plain"), (257, 17, 33, 0, 0, 3, "plain"),
    (145, 333, 120, 1, 0, 1, "plain"), (60, 1000, 60, 1, 0, 2, "plain"), (480, 1100, 480, 1, 0, 3, "acc"),
    (240, 77, 45, 1, 0, 2, "acc"), (129, 7, 17, 0, 0, 0, "plain"), (200, 480, 480, 0, 1, 0, "plain"),
    (1, 1, 17, 0, 0, 1, "bias")])
def test_seg_gemm_split6(hip, rows, k, n, ta, tb_, hint, kind):
    """HYPEL_GEMM_SPLIT6: three-way split operands, six bf16 MFMAs per step -- every operand layout (forward, data
    gradient, filter gradient), the three block widths, every epilogue (bias, accumulate, shortcut gradient, statistics),
    ragged rows / columns / K (quads cut by the end of a segment), unaligned leading dimensions and offsets; against
    the emulation at the tolerance of the fp32 kernel's own tests."""
    rng = np.random.default_rng(rows + 3 * n + k)
    b = Both(hip)
    lda = (rows if ta else k) + 3
    ldb = (k if tb_ else n) + 5
    a = (rng.standard_normal(((k if ta else rows), lda)) * 0.5).astype(np.float32)
    w = (rng.standard_normal(((n if tb_ else k), ldb)) * 0.5).astype(np.float32)
    tb = _tables(b, [(0, [(1, 2, k)], rows)])  # operand offsets 1 / 2 elements: nothing is 16-byte aligned
    garr, sarr, tarr, _ = tb.finalize(n)
    flags = SPLIT6 | (hint << 8)
    c0 = rng.standard_normal(rows * n).astype(np.float32)
    for nm, arr in (("a", np.concatenate([np.zeros(1, np.float32), a.ravel()])),
                    ("w", np.concatenate([np.zeros(2, np.float32), w.ravel()])), ("y", c0), ("g", garr), ("s", sarr),
                    ("t", tarr), ("bias", rng.standard_normal(n).astype(np.float32)),
                    ("res", rng.standard_normal((rows, n)).astype(np.float32)),
                    ("part", np.zeros(((rows + 127) // 128) * 2 * n, np.float32))):
        b.arr(nm, arr)
    if kind == "stats":
        b.run("seg_gemm_stats_f32", "a", lda, ta, "w", ldb, tb_, "y", n, n, "g", "s", "t", len(tarr), None, flags, "part")
        b.check("part", rtol=1e-3, atol=2e-4)
    elif kind == "res":
        b.run("seg_gemm_res_f32", "a", lda, ta, "w", ldb, tb_, "y", n, n, "g", "s", "t", len(tarr), None, flags, "res", n, None)
    else:
        b.run("seg_gemm_f32", "a", lda, ta, "w", ldb, tb_, "y", n, n, "g", "s", "t", len(tarr),
              "bias" if kind == "bias" else None, flags | (1 if kind == "acc" else 0))
    b.check("y", rtol=2e-4, atol=2e-5)


def test_seg_gemm_split6_multi_segment_levels(hip):
    """A multi-kernel level as the split kernel sees it: groups = output pixels, segments = valid taps (K = 120, not a
    multiple of the 32-column k-tile), two branches at channel offsets of one concatenated output."""
    rng = np.random.default_rng(11)
    nb, cin, cout, c_total, P = 200, 120, 60, 180, 9
    x = rng.standard_normal(P * nb * cin).astype(np.float32)
    w = rng.standard_normal(9 * cin * cout).astype(np.float32)
    groups = []
    for p in range(P):
        oy, ox = divmod(p, 3)
        segs = [((iy * 3 + ix) * nb * cin, (i * 3 + j) * cin * cout, cin) for i in range(3) for j in range(3)
                for iy, ix in [(oy + i - 1, ox + j - 1)] if 0 <= iy < 3 and 0 <= ix < 3]
        groups.append((p * nb * c_total + 60, segs, nb))
    b = Both(hip)
    garr, sarr, tarr, _ = _tables(b, groups).finalize(cout)
    for nm, arr in (("x", x), ("w", w), ("y", np.zeros(P * nb * c_total, np.float32)), ("g", garr), ("s", sarr), ("t", tarr)):
        b.arr(nm, arr)
    for hint in (1, 2):
        b.run("seg_gemm_f32", "x", cin, 0, "w", cout, 0, "y", c_total, cout, "g", "s", "t", len(tarr), None, SPLIT6 | (hint << 8))
        b.check("y", rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("cin,cout,nbr,hint", [(320, 128, 128, 3), (320, 128, 96, 2), (145, 60, 200, 2), (120, 30, 64, 1),
                                               (480, 480, 256, 3)])
def test_seg_gemm_split6_filter_gradient_pixel_pairs(hip, cin, cout, nbr, hint):
    """A filter gradient as the planner emits it: dW[tap] = sum over (input pixel, output pixel) pairs of X[p_in]^T dY[p_out]
    -- groups whose segments are batch-row blocks of DIFFERENT pixel pairs (k = rows per block, not a multiple of 16 for
    one case), X contiguous, dY at a channel offset of a wider tensor, ragged Cin tiles; two groups, three to five pairs."""
    rng = np.random.default_rng(cin + cout)
    P, c_tot, ch0 = 6, cout * 2 + 8, cout + 8
    x = rng.standard_normal(P * nbr * cin).astype(np.float32)
    dy = rng.standard_normal(P * nbr * c_tot).astype(np.float32)
    groups = []
    for gi, pairs in enumerate(([(0, 1), (1, 2), (2, 3), (3, 4), (4, 5)], [(5, 0), (4, 2), (1, 1)])):
        segs = [(pi * nbr * cin, po * nbr * c_tot + ch0, nbr) for pi, po in pairs]
        groups.append((gi * cin * cout, segs, cin))
    b = Both(hip)
    garr, sarr, tarr, _ = _tables(b, groups).finalize(cout)
    for nm, arr in (("x", x), ("dy", dy), ("w", np.zeros(2 * cin * cout, np.float32)), ("g", garr), ("s", sarr), ("t", tarr)):
        b.arr(nm, arr)
    b.run("seg_gemm_f32", "x", cin, 1, "dy", c_tot, 0, "w", cout, cout, "g", "s", "t", len(tarr), None, SPLIT6 | (hint << 8))
    b.check("w", rtol=2e-4, atol=2e-5)


def _hard_operands(kind, rng, rows, k, n):
    """(A [rows, k], W [k, n]) float64 draws that stress the three-way split: see test_seg_gemm_split6_hard_operands."""
    a, w = rng.standard_normal((rows, k)), rng.standard_normal((k, n)) * 0.3
    if kind == "cancel":      # every product has a partner of opposite sign and almost equal size: sum a b << sum |a b|
        half = k // 2
        a[:, half:2 * half] = -a[:, :half] * (1.0 + rng.uniform(-1e-6, 1e-6, (rows, half)))
        w[half:2 * half] = w[:half]
    elif kind == "range":     # 2^-40 .. 2^40 inside every row and column
        a *= np.exp2(rng.integers(-40, 41, (rows, k)))
        w *= np.exp2(rng.integers(-40, 41, (k, n)))
    elif kind == "tiny":      # 1e-30: the lo parts sit at bf16's last normal binades
        a *= 1e-30
    elif kind == "huge":      # 1e30 x 1e-3: sums near 1e29, far from ordinary training values
        a *= 1e30
        w *= 1e-3
    elif kind == "mixed_sign_bits":  # mantissas of all ones / alternating bits: worst cases of the round-to-nearest split
        bits = rng.choice(np.array([0x3fffffff, 0x3faaaaaa, 0x3f955555, 0x3f800001, 0x3f7fffff], np.uint32), (rows, k))
        a = bits.view(np.float32).astype(np.float64) * rng.choice([-1.0, 1.0], (rows, k))
    return a, w


def _gemm_errors(hip, rows, k, n, ta, tb_, hint, seed, hard=None):
    """max |c - fp64| / sum |a b| (over sampled rows) of the split kernel and of the fp32 MFMA kernel on one product."""
    rng = np.random.default_rng(seed)
    a = (rng.standard_normal((k, rows) if ta else (rows, k)) * rng.uniform(0.1, 2.0)).astype(np.float32)
    w = (rng.standard_normal((n, k) if tb_ else (k, n)) * rng.uniform(0.01, 0.5)).astype(np.float32)
    if hard is not None:
        am_, wm_ = _hard_operands(hard, rng, rows, k, n)
        a = np.ascontiguousarray((am_.T if ta else am_).astype(np.float32))
        w = np.ascontiguousarray((wm_.T if tb_ else wm_).astype(np.float32))
    garr, sarr, tarr, _ = _tables(None, [(0, [(0, 0, k)], rows)]).finalize(n)
    h = {nm: hip.upload(np.ascontiguousarray(v)) for nm, v in (("a", a), ("w", w), ("g", garr), ("s", sarr), ("t", tarr))}
    sel = np.unique(np.concatenate([np.arange(min(rows, 8)), rng.integers(0, rows, 24), [rows - 1]]))
    am = (a.T if ta else a)[sel].astype(np.float64)
    wm = (w.T if tb_ else w).astype(np.float64)
    ref = am @ wm
    mag = np.abs(am) @ np.abs(wm)
    errs = []
    for flags in (SPLIT6 | (hint << 8), 0):
        y = hip.zeros(rows * n)
        hip.call("seg_gemm_f32", Ref(h["a"]), a.shape[1], ta, Ref(h["w"]), w.shape[1], tb_, Ref(y), n, n, Ref(h["g"]),
                 Ref(h["s"]), Ref(h["t"]), len(tarr), None, flags)
        hip.synchronize()
        got = y.cpu().numpy().reshape(rows, n)[sel].astype(np.float64)
        errs.append(float((np.abs(got - ref) / mag).max()))
    return errs


def test_seg_gemm_split6_error_vs_fp32_chain(hip):
    """The accuracy claim of HYPEL_GEMM_SPLIT6, measured: on every (rows, K, n, layout) class of the 42 GEMM launches of the
    benchmark step (GRSS2013 HYPELCNN, batch 1024: per-pixel blocks of 1024 rows, the 50 176-row 1x1 products sampled at
    4 096 rows, filter gradients over 1 024-row slices) and DUALCNN's six heaviest products, the split kernel's error
    against the float64 product, relative to sum |a b|, is at most 1.25 x the fp32 MFMA chain's (+ 1e-8: both are a few
    1e-7; single outliers of rounding luck must not fail the build)."""
    shapes = [  # (rows, K, n, ta, tb, hint)
        (4096, 120, 240, 0, 0, 3), (4096, 240, 480, 0, 0, 3), (4096, 480, 480, 0, 0, 3), (4096, 480, 240, 0, 0, 3),
        (4096, 240, 120, 0, 0, 2), (1024, 120 * 9, 60, 0, 0, 2), (1024, 240 * 9, 30, 0, 0, 1), (4096, 240, 240, 0, 0, 3),
        (4096, 120, 120, 0, 0, 2), (1024, 2940, 980, 0, 0, 3), (1024, 405, 7105, 0, 0, 3),
        (4096, 240, 120, 0, 1, 2), (4096, 480, 240, 0, 1, 3), (4096, 480, 480, 0, 1, 3), (4096, 240, 480, 0, 1, 3),
        (4096, 120, 240, 0, 1, 3), (1024, 60 * 9, 120, 0, 1, 2), (1024, 30 * 9, 240, 0, 1, 3), (1024, 980, 2940, 0, 1, 3),
        (1024, 7105, 405, 0, 1, 3),
        (480, 1024, 480, 1, 0, 3), (240, 1024, 480, 1, 0, 3), (480, 1024, 240, 1, 0, 3), (120, 1024, 240, 1, 0, 3),
        (120, 1024, 60, 1, 0, 2), (240, 1024, 30, 1, 0, 1), (240, 1024, 240, 1, 0, 3), (2940, 1024, 980, 1, 0, 3),
        # DUALCNN (11 x 11 x 48, batch 512 per GPU): level 3 / 4 branches, 1x1 connectors, fc
        (512, 9 * 9 * 256, 256, 0, 0, 3), (512, 7 * 7 * 256, 256, 0, 1, 3), (256, 4096, 256, 1, 0, 3),
        (4096, 1280, 1280, 0, 0, 3), (4096, 1280, 1280, 0, 1, 3), (1280, 2048, 1280, 1, 0, 3)]
    worst = 0.0
    rows_out = []
    for i, (rows, k, n, ta, tb_, hint) in enumerate(shapes):
        e_split, e_f32 = _gemm_errors(hip, rows, k, n, ta, tb_, hint, 100 + i)
        rows_out.append((rows, k, n, ta, tb_, e_split, e_f32))
        worst = max(worst, e_split / (e_f32 + 1e-30))
        assert e_split <= 1.25 * e_f32 + 1e-8, (rows, k, n, ta, tb_, e_split, e_f32)
    print("split6 error / fp32-MFMA error per shape:")
    for r in rows_out:
        print("  rows=%5d K=%5d n=%5d ta=%d tb=%d  split %.2e  fp32 %.2e" % r)


@pytest.mark.parametrize("hard", ["cancel", "range", "tiny", "huge", "mixed_sign_bits"])
def test_seg_gemm_split6_hard_operands(hip, hard):
    """The error bound of HYPEL_GEMM_SPLIT6 on operands the Gaussian draws of the test above never produce (round-5 verdict):
    rows whose products cancel to 1e-6 of their magnitude, 80 binades of dynamic range inside a row, operands at 1e-30 and at
    1e30, mantissa patterns that sit on the rounding boundaries of the split.  Same measure (max |c - fp64| / sum |a b|), the
    three operand layouts.  Bound: 1.25 x the fp32 MFMA chain + 1e-8 as on ordinary data -- except for "range", where ONE
    product dominates a row's sum: its six partial products enter the accumulator as three non-negligible additions, each
    rounded at the magnitude of the sum, where the fp32 chain rounds once (measured 1.5 x: 7.3e-7 vs 4.7e-7 of sum |a b| at
    K = 480, 1.12e-6 vs 7.8e-7 at K = 2 048); the bound there is 3 x the chain's error and 2^-19 absolute -- the limit
    include/hypel.h states."""
    for i, (rows, k, n, ta, tb_, hint) in enumerate([(1024, 480, 480, 0, 0, 3), (1024, 240, 120, 0, 1, 2), (240, 2048, 480, 1, 0, 3),
                                                     (512, 145, 60, 0, 0, 2)]):
        e_split, e_f32 = _gemm_errors(hip, rows, k, n, ta, tb_, hint, 700 + i, hard=hard)
        factor = 3.0 if hard == "range" else 1.25
        assert np.isfinite(e_split) and e_split <= factor * e_f32 + 1e-8 and e_split < 2.0 ** -19, \
            (hard, rows, k, n, ta, tb_, e_split, e_f32)
        print(f"  {hard:16s} rows={rows:5d} K={k:5d} n={n:4d} ta={ta} tb={tb_}  split {e_split:.2e}  fp32 {e_f32:.2e}")


def test_seg_gemm_split6_nonfinite_and_extreme_operands(hip):
    """The written limits of HYPEL_GEMM_SPLIT6 (include/hypel.h): an Inf or a finite value above bf16's maximum in an operand
    makes every output that meets it NON-FINITE (NaN where the fp32 chain says Inf or a huge number) and leaves all other
    outputs exact to the usual bound; operands below 2^-110 lose their low parts gracefully (relative error of the affected
    products <= 2^-15) instead of producing garbage."""
    rng = np.random.default_rng(9)
    rows, k, n = 256, 96, 64
    a = rng.standard_normal((rows, k)).astype(np.float32)
    w = (rng.standard_normal((k, n)) * 0.3).astype(np.float32)
    a[3, 5] = np.inf
    a[7, 9] = np.float32(3.4e38)          # finite in fp32, above bf16's largest finite value
    a[11, :] = (rng.standard_normal(k) * 1e-36).astype(np.float32)   # lo parts are bf16 subnormals
    garr, sarr, tarr, _ = _tables(None, [(0, [(0, 0, k)], rows)]).finalize(n)
    h = {nm: hip.upload(np.ascontiguousarray(v)) for nm, v in (("a", a), ("w", w), ("g", garr), ("s", sarr), ("t", tarr))}
    outs = []
    for flags in (SPLIT6 | (2 << 8), 0):
        y = hip.zeros(rows * n)
        hip.call("seg_gemm_f32", Ref(h["a"]), k, 0, Ref(h["w"]), n, 0, Ref(y), n, n, Ref(h["g"]), Ref(h["s"]), Ref(h["t"]),
                 len(tarr), None, flags)
        hip.synchronize()
        outs.append(y.cpu().numpy().reshape(rows, n))
    split, f32 = outs
    assert not np.isfinite(split[3]).any() and not np.isfinite(f32[3]).any()       # Inf operand: NaN here, Inf there
    assert not np.isfinite(split[7]).any()                                          # above bf16's range: non-finite
    ok = np.ones(rows, bool)
    ok[[3, 7, 11]] = False
    ref = a[ok].astype(np.float64) @ w.astype(np.float64)
    mag = np.abs(a[ok].astype(np.float64)) @ np.abs(w.astype(np.float64))
    assert np.isfinite(split[ok]).all() and (np.abs(split[ok] - ref) / mag).max() < 1e-6
    ref11 = a[11].astype(np.float64) @ w.astype(np.float64)
    mag11 = np.abs(a[11].astype(np.float64)) @ np.abs(w.astype(np.float64))
    assert np.isfinite(split[11]).all() and (np.abs(split[11] - ref11) / mag11).max() < 2.0 ** -15


@pytest.mark.parametrize("cin,cout,hint,with_res,acc", [(120, 15, 2, True, 0), (120, 15, 1, False, 1), (60, 7, 0, False, 0),
                                                          (33, 16, 2, True, 1), (240, 9, 1, False, 0)])
def test_seg_gemm_paired_short_segments(hip, cin, cout, hint, with_res, acc):
    """Data gradient of a level of four branches with `cout` <= 16 filters each: per input pixel many segments of
    K = cout; consecutive ones share a k-tile (HYPEL_SEG_PAIR_FLAG).  Odd segment counts, both tile widths, with and
    without the folded shortcut epilogue, accumulate -- against the specification, and bit-identical to the unpaired
    launch of the same tables (the k columns are summed in the same order)."""
    from hypelcnn_amd.plan import GEMM_PAIRED_SEGS
    rng = np.random.default_rng(cin + cout)
    nb, P, branches = 130, 5, 4
    c_total = branches * cout
    dy = rng.standard_normal(P * nb * c_total).astype(np.float32)
    w = rng.standard_normal(branches * 3 * cin * cout).astype(np.float32)
    dz = rng.standard_normal((P * nb, cin)).astype(np.float32)
    dx0 = rng.standard_normal(P * nb * cin).astype(np.float32)
    groups = []
    for p in range(P):
        segs = []
        for b_ in range(branches):
            for t in range(3):
                q = (p + t + b_) % P
                if (p + b_ + t) % 5 == 4:
                    continue  # ragged: some groups get an odd number of segments
                segs.append((q * nb * c_total + b_ * cout, (b_ * 3 + t) * cin * cout, cout))
        groups.append((p * nb * cin, segs, nb))
    outs = []
    for pair in (True, False):
        b = Both(hip)
        tabs = _tables(b, groups)
        garr, sarr, tarr, _ = tabs.finalize(cin, pair=pair)
        assert (tabs.paired > 0) == pair
        for nm, arr in (("dy", dy), ("w", w), ("dz", dz), ("dx", dx0.copy()), ("g", garr), ("s", sarr), ("t", tarr)):
            b.arr(nm, arr)
        flags = acc | (hint << 8) | (GEMM_PAIRED_SEGS if pair else 0)
        if with_res:
            b.run("seg_gemm_res_f32", "dy", c_total, 0, "w", cout, 1, "dx", cin, cin, "g", "s", "t", len(tarr), None,
                  flags, "dz", cin, None)
        else:
            b.run("seg_gemm_f32", "dy", c_total, 0, "w", cout, 1, "dx", cin, cin, "g", "s", "t", len(tarr), None, flags)
        b.check("dx", rtol=2e-4, atol=2e-5)
        outs.append(b.h["dx"].cpu().numpy().copy())
    np.testing.assert_allclose(outs[0], outs[1], rtol=1e-5, atol=1e-5)


def test_seg_gemm_multi_products(hip):
    """Three filter-gradient-shaped products (C_p = A_p^T B_p) of different n / ld / split structure in ONE launch per
    tile width, addressed relative to one base pointer across separate allocations, then the merged reduce."""
    from hypelcnn_amd.backend import MTILE_DTYPE, REDUCE_ENTRY_DTYPE
    rng = np.random.default_rng(17)
    prods = [dict(m=145, n=120, k=600, S=3, acc=0), dict(m=60, n=60, k=200, S=1, acc=1), dict(m=130, n=70, k=333, S=2, acc=0)]
    for be in (hip, EmuBackend()):
        rng = np.random.default_rng(17)
        base = be.zeros(16)
        bp = Ref(base).ptr()
        rel = lambda t, off=0: (Ref(t, off).ptr() - bp) // 4
        segs, recs, ents, outs, wants = [], [], [], [], []
        for p in prods:
            m, n, k, S = p["m"], p["n"], p["k"], p["S"]
            x = rng.standard_normal((k, m + 3)).astype(np.float32)       # A stored [k, m], lda = m + 3
            dy = rng.standard_normal((k, n + 1)).astype(np.float32)      # B stored [k, n], ldb = n + 1
            out0 = rng.standard_normal(m * n).astype(np.float32)
            xt, dt, ot = be.upload(x), be.upload(dy), be.upload(out0)
            part = be.zeros(S * m * n) if S > 1 else None
            cuts = [k * s // S for s in range(S + 1)]
            for s_ in range(S):
                k0, k1 = cuts[s_], cuts[s_ + 1]
                sb = len(segs)
                segs.append((rel(xt, k0 * (m + 3)), rel(dt, k0 * (n + 1)), k1 - k0, 0))
                c_off = rel(part, s_ * m * n) if S > 1 else rel(ot)
                for m0 in range(0, m, 128):
                    for n0 in range(0, n, 64):
                        recs.append((c_off, segs[sb][0], segs[sb][1], m0, m, n0, n, sb, 1, k1 - k0,
                                     p["acc"] if S == 1 else 0, m + 3, n + 1, n, 0))
            if S > 1:
                ents.append((rel(part), rel(ot), m * n, m * n, S, p["acc"]))
            outs.append((ot, xt, dt, part))
            wants.append((out0.reshape(m, n) if p["acc"] else 0) + x[:, :m].astype(np.float64).T @ dy[:, :n].astype(np.float64))
        pad = [tuple([0] * 15)] * 5  # empty records (rows = 0) anywhere in the array are legal
        rarr = np.array(recs[:4] + pad + recs[4:], MTILE_DTYPE)
        s_t, r_t = be.upload(np.array(segs, SEG_DTYPE)), be.upload(rarr)
        be.call("seg_gemm_multi_f32", Ref(base), 1, 0, 64, Ref(s_t), Ref(r_t), len(rarr))
        e_t = be.upload(np.array(ents, REDUCE_ENTRY_DTYPE))
        be.call("reduce_splits_multi_f32", Ref(base), Ref(e_t), len(ents))
        be.synchronize()
        for (ot, _, _, _), want, p in zip(outs, wants, prods):
            got = ot.cpu().numpy().reshape(p["m"], p["n"])
            np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-4 * max(1.0, np.abs(want).max()), err_msg=be.name)


def test_seg_gemm_empty_group_writes_zero(hip):
    b = Both(hip)
    garr, sarr, tarr, _ = _tables(b, [(0, [], 50)]).finalize(40)
    for nm, arr in (("a", np.ones(10, np.float32)), ("c", np.full(50 * 40, 7.0, np.float32)), ("g", garr), ("s", sarr),
                    ("t", tarr)):
        b.arr(nm, arr)
    b.run("seg_gemm_f32", "a", 1, 1, "a", 1, 0, "c", 40, 40, "g", "s", "t", len(tarr), None, 0)
    assert (b.h["c"].cpu().numpy() == 0).all()


def test_layout_roundtrip(hip):
    rng = np.random.default_rng(0)
    n, p, c = 37, 49, 145
    b = Both(hip)
    b.arr("x", rng.standard_normal(n * p * c).astype(np.float32))
    b.arr("o", np.zeros(p * n * c, np.float32))
    b.arr("x2", np.zeros(n * p * c, np.float32))
    b.run("nhwc_to_pnc", "x", "o", n, p, c, c)
    b.check("o", rtol=0, atol=0)
    b.run("pnc_to_nhwc", "o", c, "x2", n, p, c)
    assert torch.equal(b.h["x2"], b.h["x"])


@pytest.mark.parametrize("rows,c", [(1000, 145), (50176 // 8, 480), (257, 15), (5, 7), (256, 64)])
def test_bn_stats_and_finalize(hip, rows, c):
    rng = np.random.default_rng(rows)
    b = Both(hip)
    x = (rng.standard_normal((rows, c)) * rng.random(c) * 3 + rng.standard_normal(c) * 10).astype(np.float32)
    chunk = 256
    nch = (rows + chunk - 1) // chunk
    b.arr("x", x)
    b.arr("part", np.zeros(nch * 2 * c, np.float32))
    for nm in ("mean", "rstd"):
        b.arr(nm, np.zeros(c, np.float32))
    b.arr("mm", rng.standard_normal(c).astype(np.float32))
    b.arr("mv", (rng.random(c) + 0.5).astype(np.float32))
    b.run("col_stats_partial", "x", c, rows, c, chunk, "part")
    b.check("part", rtol=1e-4, atol=1e-4)
    b.run("bn_finalize", "part", nch, chunk, rows, c, 1e-3, "mean", "rstd", "mm", "mv", 0.95)
    for nm in ("mean", "rstd", "mm", "mv"):
        b.check(nm, rtol=1e-5, atol=1e-6)
    # against the definition
    x64 = x.astype(np.float64)
    np.testing.assert_allclose(b.h["mean"].cpu().numpy(), x64.mean(0), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(b.h["rstd"].cpu().numpy(), 1 / np.sqrt(x64.var(0) + 1e-3), rtol=1e-5)


@pytest.mark.parametrize("rows,c,act,use_mask", [(1024, 980, 1, True), (1024, 15, 0, False), (1024, 7105, 3, False),
                                                  (77, 45, 1, False), (1000, 33, 2, True), (1, 5, 1, False)])
def test_small_rows_bn_kernels(hip, rows, c, act, use_mask):
    """One-launch BN + activation (forward) and BN backward for short matrices vs the numpy specification."""
    rng = np.random.default_rng(rows + c)
    b = Both(hip)
    ld = c + 3
    b.arr("y", (rng.standard_normal((rows, ld)) * 1.5 + rng.standard_normal(ld) * 2).astype(np.float32))
    b.arr("dz", rng.standard_normal((rows, ld)).astype(np.float32))
    b.arr("beta", rng.standard_normal(c).astype(np.float32))
    b.arr("mask", (rng.random((rows, c)) < 0.7).astype(np.float32) / 0.7)
    for nm in ("mean", "rstd"):
        b.arr(nm, np.zeros(c, np.float32))
    b.arr("mm", rng.standard_normal(c).astype(np.float32))
    b.arr("mv", (rng.random(c) + 0.5).astype(np.float32))
    b.arr("z", np.zeros(rows * ld, np.float32))
    b.arr("dy", np.zeros(rows * ld, np.float32))
    b.arr("dbeta", rng.standard_normal(c).astype(np.float32))
    m = "mask" if use_mask else None
    b.run("bn_act_small_fwd", "y", ld, rows, c, 1e-3, "beta", act, 0.18, m, c, "mean", "rstd", "mm", "mv", 0.95, "z", ld)
    for nm in ("mean", "rstd", "mm", "mv"):
        b.check(nm, rtol=2e-5, atol=2e-6)
    b.check("z", rtol=1e-4, atol=1e-5)
    b.run("bn_act_small_bwd", "dz", ld, "y", ld, rows, c, "mean", "rstd", "beta", act, 0.18, m, c, "dy", ld, "dbeta", 1)
    b.check("dy", rtol=2e-4, atol=2e-5)
    b.check("dbeta", rtol=2e-4, atol=2e-4)
    assert (b.h["z"].cpu().numpy().reshape(rows, ld)[:, c:] == 0).all(), "pad columns untouched"


@pytest.mark.parametrize("cin,c,ld_pad", [(60, 120, 0), (240, 120, 0), (60, 120, 2), (240, 120, 4), (30, 120, 0),
                                           (120, 120, 0), (145, 120, 0), (62, 124, 1)])
def test_post_op_forward_channel_maps(hip, cin, c, ld_pad):
    """The shortcut add of bn_act_fwd for every kind of scale_in_to_out map: repeat (Cout = r Cin), every other channel
    (Cout = Cin / 2), identity via indices, irregular gather; with padded source rows."""
    rng = np.random.default_rng(cin * 3 + c + ld_pad)
    rows, ld = 301, cin + ld_pad
    b = Both(hip)
    b.arr("y", rng.standard_normal((rows, c)).astype(np.float32))
    b.arr("r1", rng.standard_normal((rows, ld)).astype(np.float32))
    if c % cin == 0:
        idx = (np.arange(c) // (c // cin)).astype(np.int32)
    else:
        idx = np.minimum(np.round(np.arange(c) * cin / c), cin - 1).astype(np.int32)
    b.arr("idx", idx)
    b.arr("mean", rng.standard_normal(c).astype(np.float32) * 0.1)
    b.arr("rstd", (rng.random(c) + 0.5).astype(np.float32))
    b.arr("beta", rng.standard_normal(c).astype(np.float32) * 0.1)
    b.arr("z", np.zeros(rows * c, np.float32))
    b.run("bn_act_fwd", "y", c, rows, c, "mean", "rstd", "beta", 1, 0.18, None, 0, "r1", ld, "idx", None, 0, None, "z", c)
    b.check("z", rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("act,use_bn,use_mask,nres", [(1, True, False, 2), (3, True, False, 0), (0, True, False, 0),
                                                     (1, False, True, 1), (2, False, False, 1), (4, False, False, 0),
                                                     (1, True, True, 0)])
def test_post_op_forward_backward(hip, act, use_bn, use_mask, nres):
    rng = np.random.default_rng(act * 10 + nres)
    rows, c, cin = 777, 60, 145
    b = Both(hip)
    b.arr("y", rng.standard_normal((rows, c)).astype(np.float32))
    b.arr("dz", rng.standard_normal((rows, c)).astype(np.float32))
    b.arr("mean", rng.standard_normal(c).astype(np.float32) * 0.1)
    b.arr("rstd", (rng.random(c) + 0.5).astype(np.float32))
    b.arr("beta", rng.standard_normal(c).astype(np.float32) * 0.1)
    b.arr("mask", ((rng.random((rows, c)) < 0.3) / 0.3).astype(np.float32))
    b.arr("r1", rng.standard_normal((rows, cin)).astype(np.float32))
    b.arr("r2", rng.standard_normal((rows, c)).astype(np.float32))
    idx = np.minimum(np.round(np.arange(c) * cin / c), cin - 1).astype(np.int32)
    b.arr("idx", idx)
    b.arr("z", np.zeros(rows * c, np.float32))
    bn = ("mean", "rstd", "beta") if use_bn else (None, None, None)
    mask = "mask" if use_mask else None
    r1 = ("r1", cin, "idx") if nres >= 1 else (None, 0, None)
    r2 = ("r2", c, None) if nres >= 2 else (None, 0, None)
    b.run("bn_act_fwd", "y", c, rows, c, *bn, act, 0.18, mask, c, *r1, *r2, "z", c)
    b.check("z", rtol=1e-5, atol=1e-6)
    chunk = 256
    nch = (rows + chunk - 1) // chunk
    b.arr("part", np.zeros(nch * 2 * c, np.float32))
    b.arr("sums", np.zeros(2 * c, np.float32))
    b.arr("dparam", np.zeros(c, np.float32))
    b.arr("dy", np.zeros(rows * c, np.float32))
    b.run("bn_act_bwd_reduce", "dz", c, "y", c, rows, c, *bn, act, 0.18, mask, c, chunk, "part")
    b.check("part", rtol=1e-4, atol=1e-5)
    b.run("bwd_reduce_finalize", "part", nch, c, "sums", "dparam", 0)
    b.check("sums", rtol=1e-5, atol=1e-6)
    b.check("dparam", rtol=1e-5, atol=1e-6)
    b.run("bn_act_bwd_apply", "dz", c, "y", c, rows, c, *bn, act, 0.18, mask, c, "sums", "dy", c)
    b.check("dy", rtol=1e-4, atol=1e-5)
    # channel-map gradient
    start = np.searchsorted(idx, np.arange(cin + 1), side="left").astype(np.int32)
    b.arr("start", start)
    b.arr("dr", rng.standard_normal((rows, cin)).astype(np.float32))
    b.run("chanmap_bwd", "dz", c, rows, c, "dr", cin, cin, "start", 1)
    b.check("dr", rtol=1e-5, atol=1e-6)
    b.run("chanmap_bwd", "dz", c, rows, c, "dy", c, c, None, 0)
    b.check("dy", rtol=0, atol=0)


@pytest.mark.parametrize("rows,c,act,use_mask,in_place", [(2048, 64, 1, False, True), (4096, 360, 1, False, True),
                                                          (1000, 37, 1, True, False), (41472, 240, 2, False, True)])
def test_act_bias_bwd_reduce_writes_dy(hip, rows, c, act, use_mask, in_place):
    """hypel_act_bias_bwd_reduce == hypel_bn_act_bwd_reduce + hypel_bn_act_bwd_apply for a layer without batch norm."""
    rng = np.random.default_rng(rows + c)
    b = Both(hip)
    b.arr("y", rng.standard_normal((rows, c)).astype(np.float32))
    b.arr("dz", rng.standard_normal((rows, c)).astype(np.float32))
    b.arr("mask", (rng.random((rows, c)) < 0.7).astype(np.float32) / 0.7)
    chunk = 256
    nch = (rows + chunk - 1) // chunk
    b.arr("part", np.zeros(nch * 2 * c, np.float32))
    b.arr("sums", np.zeros(2 * c, np.float32))
    b.arr("dparam", rng.standard_normal(c).astype(np.float32))
    b.arr("dy", np.zeros(rows * c, np.float32))
    out = "dz" if in_place else "dy"
    b.run("act_bias_bwd_reduce", "dz", c, "y", c, rows, c, act, 0.18, "mask" if use_mask else None, c, chunk, "part", out, c)
    b.check(out, rtol=1e-6, atol=1e-7)
    b.check("part", rtol=1e-4, atol=1e-5)
    b.run("bwd_reduce_finalize", "part", nch, c, "sums", "dparam", 1)
    b.check("dparam", rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("splits,c0,c1,acc", [(128, 10240, 160, 0), (512, 1305, 7, 1), (3, 1000, 8, 1), (40, 70000, 5, 0)])
def test_reduce_splits_pair_bit_identical(hip, splits, c0, c1, acc):
    """hypel_reduce_splits_pair_f32 == two hypel_reduce_splits_f32, bit for bit (whichever kernel form the shapes pick)."""
    rng = np.random.default_rng(splits + c0)
    p0 = torch.from_numpy(rng.standard_normal(splits * c0).astype(np.float32)).cuda()
    p1 = torch.from_numpy(rng.standard_normal(splits * (c1 + 1)).astype(np.float32)).cuda()
    init0 = torch.from_numpy(rng.standard_normal(c0).astype(np.float32)).cuda()
    init1 = torch.from_numpy(rng.standard_normal(c1).astype(np.float32)).cuda()
    a0, a1, b0, b1 = init0.clone(), init1.clone(), init0.clone(), init1.clone()
    hip.call("reduce_splits_f32", Ref(p0), c0, splits, Ref(a0), c0, acc, None, 0, 0)
    hip.call("reduce_splits_f32", Ref(p1), c1 + 1, splits, Ref(a1), c1, acc, None, 0, 0)
    hip.call("reduce_splits_pair_f32", Ref(p0), c0, c0, Ref(b0), Ref(p1), c1 + 1, c1, Ref(b1), splits, acc)
    hip.synchronize()
    assert torch.equal(a0, b0) and torch.equal(a1, b1)


def test_losses(hip):
    rng = np.random.default_rng(1)
    n, c = 1000, 15
    b = Both(hip)
    b.arr("logits", (rng.standard_normal((n, c)) * 5).astype(np.float32))
    b.arr("lab", np.eye(c, dtype=np.float32)[rng.integers(0, c, n)])
    b.arr("loss", np.zeros(n, np.float32))
    b.arr("dl", np.zeros(n * c, np.float32))
    b.run("softmax_xent", "logits", c, n, c, "lab", c, "loss", "dl", c, 1.0 / n)
    b.check("loss", rtol=1e-5, atol=1e-6)
    b.check("dl", rtol=1e-4, atol=1e-7)
    # K6: zero logits -> log(classes)
    b.arr("z0", np.zeros((4, c), np.float32))
    b.run("softmax_xent", "z0", c, 4, c, "lab", c, "loss", None, 0, 1.0)
    np.testing.assert_allclose(b.h["loss"].cpu().numpy()[:4], np.log(c), rtol=1e-6)
    rows, f = 64, 7105
    b.arr("a", rng.random((rows, f)).astype(np.float32))
    b.arr("bb", rng.random((rows, f)).astype(np.float32))
    b.arr("out", np.zeros(1, np.float32))
    b.arr("da", np.zeros(rows * f, np.float32))
    b.arr("ws", np.zeros(2048, np.float32))
    b.run("mse", "a", f, "bb", f, rows, f, "out", "da", f, 1.0, "ws")
    b.check("out", rtol=1e-5)
    b.check("da", rtol=1e-5, atol=1e-9)
    # strided operands (a window of a wider buffer) and a count that is no multiple of 4: the general kernel
    b.arr("da", np.zeros(rows * f, np.float32))
    b.run("mse", "a", f, "bb", f, rows - 1, f - 3, "out", "da", f, 0.5, "ws")
    b.check("out", rtol=1e-5)
    b.check("da", rtol=1e-5, atol=1e-9)
    b.run("sum_f32", "loss", n, 1.0 / n, "out", "ws")
    b.check("out", rtol=1e-5)
    # the fused loss tail: MSE block partials + one finaliser (means, non-finite flag, step counter)
    b.arr("da2", np.zeros(rows * f, np.float32))
    b.arr("ws2", np.full(1024, 7.0, np.float32))
    b.run("mse_partial_f32", "a", f, "bb", f, rows, f, "da2", f, 1.0, "ws2")
    b.check("da2", rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(b.h["ws2"].cpu().numpy().astype(np.float64).sum(), b.e["ws2"].numpy().astype(np.float64).sum(),
                               rtol=1e-5)
    for nm in ("ce", "mse", "flag"):
        b.arr(nm, np.full(1, -1.0, np.float32))
    b.arr("step", np.array([41], np.int64))
    b.run("loss_finalize_f32", "loss", n, "ws2", 1.0 / (rows * f), "ce", "mse", "flag", "step")
    b.check("ce", rtol=1e-5)
    b.check("mse", rtol=1e-5)
    a64, b64 = b.e["a"].numpy().astype(np.float64), b.e["bb"].numpy().astype(np.float64)
    np.testing.assert_allclose(b.h["mse"].cpu().numpy()[0], ((a64 - b64) ** 2).mean(), rtol=1e-5)
    assert float(b.h["flag"].cpu()[0]) == 0.0 and int(b.h["step"].cpu()[0]) == 42 and int(b.e["step"][0]) == 42
    b.arr("bad", np.array([np.inf] + [0.0] * (n - 1), np.float32))
    b.run("loss_finalize_f32", "bad", n, None, 0.0, "ce", None, "flag", None)
    assert float(b.h["flag"].cpu()[0]) == 1.0 and float(b.e["flag"][0]) == 1.0 and int(b.h["step"].cpu()[0]) == 42


def test_optimizers_and_reduce(hip):
    rng = np.random.default_rng(2)
    n = 100003
    b = Both(hip)
    for nm in ("p", "g", "m"):
        b.arr(nm, rng.standard_normal(n).astype(np.float32))
    b.arr("v", rng.random(n).astype(np.float32))
    b.run("adam_tf1", "p", "g", "m", "v", n, 3e-4, 0.9, 0.999, 1e-8)
    for nm in ("p", "m", "v"):
        b.check(nm, rtol=2e-5, atol=1e-6)
    b.run("momentum_tf1", "p", "g", "m", n, 1e-3, 0.9)
    b.check("p", rtol=2e-5, atol=1e-6)
    b.arr("part", rng.standard_normal(5 * 1000).astype(np.float32))
    b.arr("out", rng.standard_normal(900).astype(np.float32))
    b.run("reduce_splits_f32", "part", 1000, 5, "out", 900, 1, None, 0, 0)
    b.check("out", rtol=1e-5, atol=1e-6)
    b.arr("rb", rng.standard_normal(90).astype(np.float32))
    b.run("reduce_splits_f32", "part", 1000, 5, "out", 900, 0, "rb", 90, 0)
    b.check("out", rtol=1e-5, atol=1e-6)
    # strided window: 20 rows x 7 columns of a matrix with leading dimension 13, 3 slabs of stride 400
    b.arr("spart", rng.standard_normal(3 * 400).astype(np.float32))
    b.arr("sout", rng.standard_normal(400).astype(np.float32))
    b.run("reduce_splits_f32", ("spart", 5), 400, 3, ("sout", 5), 140, 1, None, 7, 13)
    b.check("sout", rtol=1e-5, atol=1e-6)
    # float4 variant: bias with n % 4 == 0, and a strided window (24 rows x 8 columns, leading dimension 16)
    b.arr("rb4", rng.standard_normal(100).astype(np.float32))
    b.run("reduce_splits_f32", "part", 1000, 5, "out", 900, 1, "rb4", 100, 0)
    b.check("out", rtol=1e-5, atol=1e-6)
    b.arr("vpart", rng.standard_normal(6 * 400).astype(np.float32))
    b.arr("vout", rng.standard_normal(400).astype(np.float32))
    b.run("reduce_splits_f32", ("vpart", 4), 400, 6, ("vout", 4), 192, 1, None, 8, 16)
    b.check("vout", rtol=1e-5, atol=1e-6)
    # many slabs, few outputs (wave-per-output variant): plain, with bias + accumulate, and strided
    b.arr("mpart", rng.standard_normal(300 * 500).astype(np.float32))
    b.arr("mout", rng.standard_normal(500).astype(np.float32))
    b.run("reduce_splits_f32", "mpart", 500, 300, "mout", 239, 0, None, 0, 0)
    b.check("mout", rtol=1e-5, atol=1e-5)
    b.run("reduce_splits_f32", "mpart", 500, 77, "mout", 450, 1, "rb", 90, 0)
    b.check("mout", rtol=1e-5, atol=1e-5)
    b.run("reduce_splits_f32", ("mpart", 3), 500, 64, ("mout", 3), 140, 1, None, 7, 13)
    b.check("mout", rtol=1e-5, atol=1e-5)
    # K4: first Adam step from zero slots ~ -lr*sign(g)
    b.arr("p0", np.zeros(3, np.float32)); b.arr("g0", np.array([0.3, -2.0, 1e-3], np.float32))
    b.arr("m0", np.zeros(3, np.float32)); b.arr("v0", np.zeros(3, np.float32))
    lr_t = 3e-4 * np.sqrt(1 - 0.999) / (1 - 0.9)
    b.run("adam_tf1", "p0", "g0", "m0", "v0", 3, float(lr_t), 0.9, 0.999, 1e-8)
    g = np.array([0.3, -2.0, 1e-3])
    np.testing.assert_allclose(b.h["p0"].cpu().numpy(), -3e-4 * g / (np.abs(g) + 1e-8 / np.sqrt(1 - 0.999)), rtol=1e-4)


def test_dropout_mask_statistics_and_replay(hip):
    n = 1 << 20
    mask = hip.zeros(n)
    step = hip.zeros(1, torch.int64)
    hip.call("dropout_mask", Ref(mask), n, 0.3, 1234, Ref(step))
    m1 = mask.clone()
    hip.call("dropout_mask", Ref(mask), n, 0.3, 1234, Ref(step))
    assert torch.equal(m1, mask), "same (seed, step) -> same mask"
    vals = torch.unique(mask).cpu().numpy()
    np.testing.assert_allclose(vals, [0.0, 1 / 0.3], rtol=1e-6)
    frac = float((mask > 0).float().mean())
    assert abs(frac - 0.3) < 5e-3
    hip.call("step_inc", Ref(step))
    hip.call("dropout_mask", Ref(mask), n, 0.3, 1234, Ref(step))
    assert not torch.equal(m1, mask) and int(step[0]) == 1


def test_argmax_confusion_and_lrn(hip):
    rng = np.random.default_rng(4)
    n, c = 5000, 15
    b = Both(hip)
    logits = rng.standard_normal((n, c)).astype(np.float32)
    logits[::7, 3] = logits[::7, 9] = 50.0  # ties: first maximum wins
    b.arr("logits", logits)
    b.arr("lab", rng.integers(0, c, n).astype(np.int32))
    b.arr("pred", np.zeros(n, np.int32))
    b.arr("conf", np.zeros(c * c, np.int32))
    b.run("argmax_confusion", "logits", c, n, c, "lab", "pred", "conf")
    assert torch.equal(b.h["pred"].cpu(), b.e["pred"])
    assert torch.equal(b.h["conf"].cpu(), b.e["conf"])
    rows, cc = 300, 384
    b.arr("x", rng.standard_normal((rows, cc)).astype(np.float32))
    b.arr("dy", rng.standard_normal((rows, cc)).astype(np.float32))
    b.arr("y", np.zeros(rows * cc, np.float32))
    b.arr("dx", np.zeros(rows * cc, np.float32))
    b.run("lrn_fwd", "x", cc, rows, cc, 5, 1.0, 1.0, 0.5, "y", cc)
    b.check("y", rtol=1e-5, atol=1e-6)
    b.run("lrn_bwd", "x", cc, "dy", cc, rows, cc, 5, 1.0, 1.0, 0.5, "dx", cc, 0)
    b.check("dx", rtol=1e-4, atol=1e-5)


# ----------------------------------------------------------------------------------------------- GAN kernels
@pytest.mark.parametrize("bands,n,only_enc", [(64, 37, 0), (64, 2048, 0), (144, 130, 0), (360, 70, 0), (64, 50, 1),
                                              (360, 33, 1), (16, 5, 0), (130, 9, 0), (200, 50, 0), (258, 21, 1), (360, 4096, 0),
                                              (128, 40, 0)])
def test_gan_generator_fwd_bwd(hip, bands, n, only_enc):
    rng = np.random.default_rng(bands + n)
    ks = [bands, bands // 2, bands // 4, bands // 8, bands // 4, bands // 2, bands]
    wtot = sum(ks)
    b = Both(hip)
    x = rng.random((n, bands)).astype(np.float32)
    w = (rng.standard_normal(wtot) * 0.3 / np.sqrt(np.repeat(ks, ks))).astype(np.float32)
    bias = (rng.standard_normal(8) * 0.05).astype(np.float32)
    # samples whose leaky-ReLU branch hangs on the rounding of an fp32 sum are re-drawn ((360, 70): one pre-activation of
    # 2.2e-8 in layer 1 -- whichever way it falls, that sample's dx differs by 0.04 from the other answer)
    for _ in range(20):
        rows = emu_backend.generator_knife_edge_rows(x, w, bias, bands, only_enc)
        if len(rows) == 0:
            break
        x[rows] = rng.random((len(rows), bands)).astype(np.float32)
    assert len(rows) == 0
    b.arr("x", x)
    b.arr("w", w)
    b.arr("bias", bias)
    b.arr("out", np.zeros(n * bands, np.float32))
    b.run("gan_generator_fwd", "x", bands, n, bands, "w", "bias", only_enc, "out", bands)
    b.check("out", rtol=5e-5, atol=5e-6)  # fp32 sums over up to 360 taps x 7 layers vs the float64 spec
    blocks = hip.gan_generator_blocks(n)
    assert blocks == b.emu.gan_generator_blocks(n)
    b.arr("dout", rng.standard_normal((n, bands)).astype(np.float32))
    b.arr("dx", rng.standard_normal((n, bands)).astype(np.float32))
    b.arr("pw", np.zeros(blocks * wtot, np.float32))
    b.arr("pb", np.zeros(blocks * 8, np.float32))
    b.arr("dw", np.zeros(wtot, np.float32))
    b.arr("db", np.zeros(8, np.float32))
    b.run("gan_generator_bwd", "x", bands, "dout", bands, n, bands, "w", "bias", only_enc, "dx", bands, 1, "pw", "pb")
    b.check("dx", rtol=1e-4, atol=1e-5)
    b.run("reduce_splits_f32", "pw", wtot, blocks, "dw", wtot, 0, None, 0, 0)
    b.run("reduce_splits_f32", "pb", 8, blocks, "db", 7, 0, None, 0, 0)
    b.check("dw", rtol=2e-4, atol=2e-5)
    b.check("db", rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("widths,n,act_mask,want_dx", [((64, 64, 64, 32), 2048, 0b011, True), ((64, 64, 64, 32), 37, 0b011, False),
                                                       ((96, 96, 48), 300, 0b01, True), ((60, 60, 15, 30, 2), 1000, 0b1111, True),
                                                       ((17, 5), 20, 0b0, True), ((100, 37, 64, 9), 5000, 0b101, True)])
def test_dense_stack_fwd_bwd(hip, widths, n, act_mask, want_dx):
    """The one-launch fully-connected stack (discriminator at narrow band counts) against the float64 definition:
    output, input gradient (accumulating into a pre-filled buffer) and the slab-reduced filter / bias gradients."""
    rng = np.random.default_rng(sum(widths) + n)
    L = len(widths) - 1
    wq = list(widths) + [0] * (5 - len(widths))
    assert hip.dense_stack_supported(widths)
    wtot = sum(a * b_ for a, b_ in zip(widths, widths[1:]))
    btot = sum(widths[1:])
    b = Both(hip)
    b.arr("x", rng.standard_normal((n, widths[0])).astype(np.float32))
    b.arr("w", np.concatenate([(rng.standard_normal(a * c) * np.sqrt(2.0 / a)) for a, c in zip(widths, widths[1:])]).astype(np.float32))
    b.arr("bias", (rng.standard_normal(btot) * 0.1).astype(np.float32))
    b.arr("out", np.zeros(n * widths[-1], np.float32))
    b.run("dense_stack_fwd", "x", widths[0], n, L, *wq, act_mask, 0.1, "w", "bias", "out", widths[-1])
    b.check("out", rtol=2e-5, atol=2e-6)
    blocks = hip.dense_stack_blocks(n)
    assert blocks == b.emu.dense_stack_blocks(n)
    b.arr("dout", rng.standard_normal((n, widths[-1])).astype(np.float32))
    b.arr("dx", rng.standard_normal((n, widths[0])).astype(np.float32))
    b.arr("pw", np.full(blocks * wtot, 7.0, np.float32))
    b.arr("pb", np.full(blocks * btot, 7.0, np.float32))
    b.arr("dw", np.zeros(wtot, np.float32))
    b.arr("db", np.zeros(btot, np.float32))
    b.run("dense_stack_bwd", "x", widths[0], "dout", widths[-1], n, L, *wq, act_mask, 0.1, "w", "bias",
          "dx" if want_dx else None, widths[0], 1, "pw", "pb")
    if want_dx:
        b.check("dx", rtol=5e-5, atol=5e-6)
    b.run("reduce_splits_f32", "pw", wtot, blocks, "dw", wtot, 0, None, 0, 0)
    b.run("reduce_splits_f32", "pb", btot, blocks, "db", btot, 0, None, 0, 0)
    b.check("dw", rtol=1e-4, atol=1e-5)
    b.check("db", rtol=1e-4, atol=1e-5)
    assert not hip.dense_stack_supported((360, 360, 360, 180)) and not hip.dense_stack_supported((128, 128, 128, 64))
    assert b.emu.dense_stack_supported(widths) and not b.emu.dense_stack_supported((128, 128, 128, 64))


@pytest.mark.parametrize("bands,n,only_enc", [(360, 4096, 0), (360, 70, 1), (64, 2048, 0), (144, 37, 0), (16, 5, 1),
                                              (200, 1000, 1), (32, 9000, 0), (48, 8300, 1)])
def test_gan_generator_kept_activations_bit_identical(hip, bands, n, only_enc):
    """hypel_gan_generator_fwd_keep + _bwd_kept (the backward pass starts from the activations the forward pass left
    behind; round 4: the register-ring kernel gan_generator_bwd2_mfma_kernel) == hypel_gan_generator_fwd + _bwd (the LDS-ring
    kernel, which recomputes them), bit for bit: out, dx and every partial slab.  The last two cases have more row tiles
    than blocks (a block walks two tiles: its filter / bias gradient sums carry over)."""
    rng = np.random.default_rng(bands + n)
    ks = [bands, bands // 2, bands // 4, bands // 8, bands // 4, bands // 2, bands]
    wtot = sum(ks)
    x = torch.from_numpy(rng.random((n, bands)).astype(np.float32)).cuda()
    w = torch.from_numpy((rng.standard_normal(wtot) * 0.3 / np.sqrt(np.repeat(ks, ks))).astype(np.float32)).cuda()
    bias = torch.from_numpy((rng.standard_normal(8) * 0.05).astype(np.float32)).cuda()
    dout = torch.from_numpy(rng.standard_normal((n, bands)).astype(np.float32)).cuda()
    blocks = hip.gan_generator_blocks(n)
    keep_n = hip.gan_generator_keep_floats(n, bands, only_enc)
    assert keep_n > 0
    res = []
    for kept in (False, True):
        out = torch.zeros(n, bands, device="cuda")
        dx = torch.ones(n, bands, device="cuda")
        pw = torch.zeros(blocks * wtot, device="cuda")
        pb = torch.zeros(blocks * 8, device="cuda")
        if kept:
            keep = torch.full((keep_n,), float("nan"), device="cuda")
            hip.call("gan_generator_fwd_keep", Ref(x), bands, n, bands, Ref(w), Ref(bias), only_enc, Ref(out), bands, Ref(keep))
            hip.call("gan_generator_bwd_kept", Ref(x), bands, Ref(dout), bands, n, bands, Ref(w), Ref(bias), only_enc, Ref(dx),
                     bands, 1, Ref(pw), Ref(pb), Ref(keep))
        else:
            hip.call("gan_generator_fwd", Ref(x), bands, n, bands, Ref(w), Ref(bias), only_enc, Ref(out), bands)
            hip.call("gan_generator_bwd", Ref(x), bands, Ref(dout), bands, n, bands, Ref(w), Ref(bias), only_enc, Ref(dx),
                     bands, 1, Ref(pw), Ref(pb))
        hip.synchronize()
        res.append([t.cpu().numpy() for t in (out, dx, pw, pb)])
    for a, b_, name in zip(res[0], res[1], ("out", "dx", "pw", "pb")):
        assert np.isfinite(b_).all(), name
        np.testing.assert_array_equal(a, b_, err_msg=name)
    # band counts of gan.hip's VALU kernels keep nothing
    assert hip.gan_generator_keep_floats(n, 8, only_enc) == 0


@pytest.mark.parametrize("bands,n,with_ratio", [(64, 2048, True), (360, 100, True), (144, 33, False)])
def test_gather_pairs_bit_exact(hip, bands, n, with_ratio):
    """The GAN trainer's input stage as one library launch: gather + the two regulariser swaps, bit for bit."""
    rng = np.random.default_rng(bands + n)
    pool = 3000
    b = Both(hip)
    b.arr("normal", rng.random((pool, bands)).astype(np.float32))
    b.arr("shadow", rng.random((pool, bands)).astype(np.float32))
    b.arr("idx", rng.integers(0, pool, n).astype(np.int64))
    b.arr("ratio", (1.0 + rng.random(bands)).astype(np.float32))
    b.arr("u", (rng.random(2 * n) * 0.98 + 0.01).astype(np.float32))
    b.arr("ox", np.zeros(n * bands, np.float32))
    b.arr("oy", np.zeros(n * bands, np.float32))
    if with_ratio:
        b.run("gather_pairs_f32", "normal", "shadow", "idx", n, bands, "ratio", "u", ("u", n), 0.4, "ox", "oy")
    else:
        b.run("gather_pairs_f32", "normal", "shadow", "idx", n, bands, None, None, None, 0.0, "ox", "oy")
    for nm in ("ox", "oy"):
        np.testing.assert_array_equal(b.h[nm].cpu().numpy(), b.e[nm].numpy(), err_msg=nm)


def test_gan_generator_valu_kernels_in_a_subprocess():
    """The generator runs on the matrix cores (gan_mfma.hip) from 16 bands on; gan.hip's wave-per-sample and register-tiled
    kernels remain the path for B < 16 / B > 384 and under HYPEL_GAN_MFMA=0 (the library reads the switch once per
    process): the same parity cases once more on them."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, HYPEL_GAN_MFMA="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k",
                        "test_gan_generator_fwd_bwd", "-p", "no:cacheprovider"], env=env, capture_output=True, text=True,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "12 passed" in r.stdout, r.stdout[-500:]


@pytest.mark.parametrize("rows,e,parts", [(4096, 2, 6), (100, 3, 7), (8192, 2, 2), (20000, 5, 3)])
def test_l2norm_parts(hip, rows, e, parts):
    """register-resident form for parts of <= 16 K elements, the looping form beyond"""
    rng = np.random.default_rng(5)
    b = Both(hip)
    b.arr("x", rng.standard_normal((rows, parts * e)).astype(np.float32))
    b.arr("dy", rng.standard_normal((rows, parts * e)).astype(np.float32))
    b.arr("y", np.zeros(rows * parts * e, np.float32))
    b.arr("dx", rng.standard_normal(rows * parts * e).astype(np.float32))
    b.arr("stat", np.zeros(2 * parts, np.float32))
    b.run("l2norm_parts_fwd", "x", parts * e, rows, e, parts, "y", parts * e, "stat")
    b.check("y", rtol=1e-5, atol=1e-7)
    b.check("stat", rtol=1e-5, atol=1e-7)
    b.run("l2norm_parts_bwd", "x", parts * e, "dy", parts * e, rows, e, parts, "stat", "dx", parts * e, 1)
    b.check("dx", rtol=1e-4, atol=1e-6)


def test_gan_losses_l2norm_nce(hip):
    rng = np.random.default_rng(11)
    rows, c = 2048, 32
    b = Both(hip)
    b.arr("a", rng.standard_normal((rows, c)).astype(np.float32))
    b.arr("bb", rng.standard_normal((rows, c)).astype(np.float32))
    b.arr("loss", np.zeros(1, np.float32))
    b.arr("ws", np.zeros(rows + 4096, np.float32))
    for mode in (0, 1, 2):
        b.arr("da", rng.standard_normal((rows, c)).astype(np.float32))
        b.arr("db", np.zeros(rows * c, np.float32))
        b.run("gan_loss", mode, "a", c, "bb" if mode == 1 else None, c, rows, c, 1.0, 0.5, "loss", 0 if mode == 0 else 1,
              "da", c, 1, "db" if mode == 1 else None, c, 0, "ws")
        b.check("da", rtol=1e-5, atol=1e-7)
        if mode == 1:
            b.check("db", rtol=1e-5, atol=1e-8)
    b.check("loss", rtol=1e-5)
    b.arr("w", rng.standard_normal(10007).astype(np.float32))
    b.arr("dw", rng.standard_normal(10007).astype(np.float32))
    b.run("l2_reg", "w", 10007, 1e-4, "loss", 1, "dw", "ws")
    b.check("loss", rtol=1e-5)
    b.check("dw", rtol=1e-6, atol=1e-7)
    # whole-tensor l2 normalise on a strided [N, E] slice of a [N, P*E] stack
    n, p, e = 777, 7, 2
    b.arr("x", rng.standard_normal((n, e)).astype(np.float32))
    b.arr("stack", np.zeros(n * p * e, np.float32))
    b.arr("stat", np.zeros(2, np.float32))
    b.run("l2norm_fwd", "x", e, n, e, ("stack", 3 * e), p * e, "stat")
    b.check("stack", rtol=1e-5, atol=1e-8)
    b.check("stat", rtol=1e-5)
    b.arr("dstack", rng.standard_normal((n, p * e)).astype(np.float32))
    b.arr("dxn", np.zeros(n * e, np.float32))
    b.run("l2norm_bwd", "x", e, ("dstack", 3 * e), p * e, n, e, "stat", "dxn", e, 0)
    b.check("dxn", rtol=1e-4, atol=1e-7)
    # patch-NCE
    b.arr("g", rng.standard_normal((n, p * e)).astype(np.float32) * 0.3)
    b.arr("r", rng.standard_normal((n, p * e)).astype(np.float32) * 0.3)
    b.arr("dg", np.zeros(n * p * e, np.float32))
    b.arr("dr", rng.standard_normal((n, p * e)).astype(np.float32))
    b.run("nce_loss", "g", p * e, "r", p * e, n, p, e, 0.07, 10.0, "loss", 0, "dg", p * e, 0, "dr", p * e, 1, "ws")
    b.check("loss", rtol=1e-4)
    b.check("dg", rtol=2e-3, atol=1e-5)
    b.check("dr", rtol=2e-3, atol=1e-5)
    # K6: all-equal logits -> P * log(P^2) per sample
    b.arr("g0", np.zeros((n, p * e), np.float32))
    b.run("nce_loss", "g0", p * e, "r", p * e, n, p, e, 0.07, 1.0, "loss", 0, None, 0, 0, None, 0, 0, "ws")
    np.testing.assert_allclose(b.h["loss"].cpu().numpy()[0], p * np.log(p * p), rtol=1e-5)


@pytest.mark.parametrize("seed", range(12))
def test_seg_gemm_random_tables_all_variants(hip, seed):
    """Randomised tables through every epilogue / staging variant added in round 2 (statistics epilogue, 128x96 tiles,
    paired short segments, multi-segment groups with ragged rows) against the specification."""
    from hypelcnn_amd.plan import GEMM_PAIRED_SEGS
    rng = np.random.default_rng(1000 + seed)
    mode = ["fwd", "dgrad", "dgrad_pair", "stats"][seed % 4]
    n = int(rng.choice([17, 30, 60, 96, 120, 145, 192, 240]))
    hint = int(rng.integers(0, 4))
    if mode == "stats":
        rows, k = int(rng.integers(1, 700)), int(rng.integers(1, 300))
        b = Both(hip)
        a = rng.standard_normal((rows, k)).astype(np.float32)
        w = rng.standard_normal((k, n)).astype(np.float32)
        garr, sarr, tarr, _ = _tables(b, [(0, [(0, 0, k)], rows)]).finalize(n)
        n_chunks = (rows + 127) // 128
        for nm, arr in (("a", a), ("w", w), ("y", np.zeros(rows * n, np.float32)), ("g", garr), ("s", sarr), ("t", tarr),
                        ("part", np.zeros(n_chunks * 2 * n, np.float32))):
            b.arr(nm, arr)
        b.run("seg_gemm_stats_f32", "a", k, 0, "w", n, 0, "y", n, n, "g", "s", "t", len(tarr), None, hint << 8, "part")
        b.check("y", rtol=2e-4, atol=2e-5)
        b.check("part", rtol=2e-3, atol=5e-4)
        return
    tb_ = 0 if mode == "fwd" else 1
    kmax = 16 if mode == "dgrad_pair" else 200
    n_groups = int(rng.integers(1, 6))
    lda = 300
    ldb = (kmax + 3) if tb_ else (n + 5)
    ldc = n + int(rng.integers(0, 4))
    a = rng.standard_normal(4000 * lda).astype(np.float32)
    bm = rng.standard_normal(900 * ldb).astype(np.float32)
    groups, c_pos = [], 0
    for _ in range(n_groups):
        rows = int(rng.integers(1, 400))
        segs = []
        for _ in range(int(rng.integers(0, 7))):
            k = int(rng.integers(1, kmax + 1))
            a_off = int(rng.integers(0, 4000 - rows)) * lda + int(rng.integers(0, lda - k))
            b_off = int(rng.integers(0, 900 - (n if tb_ else k))) * ldb + (int(rng.integers(0, ldb - k)) if tb_ else 0)
            segs.append((a_off, b_off, k))
        groups.append((c_pos * ldc, segs, rows))
        c_pos += rows
    c0 = rng.standard_normal(c_pos * ldc).astype(np.float32)
    bias = rng.standard_normal(ldc).astype(np.float32)
    acc = int(rng.integers(0, 2))
    b = Both(hip)
    tabs = _tables(b, groups)
    pair = mode == "dgrad_pair"
    garr, sarr, tarr, _ = tabs.finalize(n, pair=pair)
    for nm, arr in (("a", a), ("b", bm), ("c", c0), ("bias", bias), ("g", garr), ("s", sarr), ("t", tarr)):
        b.arr(nm, arr)
    flags = acc | (hint << 8) | (GEMM_PAIRED_SEGS if (pair and tabs.paired) else 0)
    b.run("seg_gemm_f32", "a", lda, 0, "b", ldb, tb_, "c", ldc, n, "g", "s", "t", len(tarr),
          "bias" if seed % 3 == 0 else None, flags)
    b.check("c", rtol=3e-4, atol=3e-5)


@pytest.mark.parametrize("rows,c,world", [(1000, 120, 1), (777, 45, 3), (128, 480, 8)])
def test_sync_bn_kernels_match_the_global_statistics(hip, rows, c, world):
    """hypel_bn_merge_partials (a rank's chunk partials -> one (mean, M2, rows) record) and hypel_bn_finalize_ranks
    (records of all ranks -> mean / rstd / moving averages) against the emulation and the definition on the
    concatenated rows; hypel_bn_act_bwd_apply_global against the emulation."""
    rng = np.random.default_rng(rows + world)
    chunk = 128
    nch = (rows + chunk - 1) // chunk
    xs = [(rng.standard_normal((rows - 7 * r, c)) * (rng.random(c) * 3 + 0.1) + rng.standard_normal(c) * 5 + r).astype(np.float32)
          for r in range(world)]
    b = Both(hip)
    rec = 2 * c + 1
    b.arr("all", np.zeros(world * rec, np.float32))
    for r, x in enumerate(xs):
        rr = x.shape[0]
        nch_r = (rr + chunk - 1) // chunk
        b.arr(f"x{r}", x)
        b.arr(f"part{r}", np.zeros(nch * 2 * c, np.float32))
        b.run("col_stats_partial", f"x{r}", c, rr, c, chunk, f"part{r}")
        b.run("bn_merge_partials", f"part{r}", nch_r, chunk, rr, c, ("all", r * rec))
    b.check("all", rtol=1e-4, atol=1e-4)
    for nm in ("mean", "rstd"):
        b.arr(nm, np.zeros(c, np.float32))
    b.arr("mm", rng.standard_normal(c).astype(np.float32))
    b.arr("mv", (rng.random(c) + 0.5).astype(np.float32))
    b.run("bn_finalize_ranks", "all", world, c, 1e-3, "mean", "rstd", "mm", "mv", 0.95)
    for nm in ("mean", "rstd", "mm", "mv"):
        b.check(nm, rtol=1e-5, atol=1e-6)
    x64 = np.concatenate(xs).astype(np.float64)
    np.testing.assert_allclose(b.h["mean"].cpu().numpy(), x64.mean(0), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(b.h["rstd"].cpu().numpy(), 1 / np.sqrt(x64.var(0) + 1e-3), rtol=1e-5)
    # backward apply with the statistic row count of the global batch
    r0 = xs[0].shape[0]
    b.arr("dz", rng.standard_normal((r0, c)).astype(np.float32))
    b.arr("beta", rng.standard_normal(c).astype(np.float32) * 0.1)
    b.arr("sums", rng.standard_normal(2 * c).astype(np.float32) * 10)
    b.arr("dy", np.zeros(r0 * c, np.float32))
    b.run("bn_act_bwd_apply_global", "dz", c, "x0", c, r0, c, "mean", "rstd", "beta", 1, 0.18, None, c, "sums",
          int(x64.shape[0]), "dy", c)
    b.check("dy", rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("rows,k,n,tb_,kind", [
    (1000, 145, 120, 0, "plain"), (777, 120, 240, 0, "stats"), (300, 240, 120, 1, "plain"), (515, 480, 240, 1, "res"),
    (129, 33, 17, 0, "plain"), (64, 60, 30, 1, "res"),
])
def test_seg_gemm_single_segment_hint(hip, rows, k, n, tb_, kind):
    """HYPEL_GEMM_SINGLE_SEG: single-segment launches (1x1 convolutions and their data gradients) run on the build of
    the 128x32 kernel that is capped at 96 scalar registers; same results as the specification and as the uncapped
    kernel, for the plain, statistics and shortcut-gradient entry points."""
    from hypelcnn_amd.plan import GEMM_SINGLE_SEG
    rng = np.random.default_rng(rows + n)
    a = rng.standard_normal((rows, k)).astype(np.float32)
    w = rng.standard_normal((n, k) if tb_ else (k, n)).astype(np.float32)
    outs = []
    for flag in (GEMM_SINGLE_SEG, 0):
        b = Both(hip)
        # three groups of ragged rows, one segment each
        cuts = [0, rows // 3, rows // 3 + 1, rows]
        groups = [(cuts[i] * n, [(cuts[i] * k, 0, k)], cuts[i + 1] - cuts[i]) for i in range(3)]
        if kind == "stats":
            groups = [(0, [(0, 0, k)], rows)]
        garr, sarr, tarr, _ = _tables(b, groups).finalize(n)
        for nm, arr in (("a", a), ("w", w), ("y", np.zeros(rows * n, np.float32)), ("g", garr), ("s", sarr), ("t", tarr)):
            b.arr(nm, arr)
        acc = (1 << 8) | flag   # 128x32 tiles
        if kind == "stats":
            b.arr("part", np.zeros(((rows + 127) // 128) * 2 * n, np.float32))
            b.run("seg_gemm_stats_f32", "a", k, 0, "w", n, 0, "y", n, n, "g", "s", "t", len(tarr), None, acc, "part")
            b.check("part", rtol=1e-3, atol=2e-4)
        elif kind == "res":
            b.arr("dz", rng.standard_normal((rows, n)).astype(np.float32))
            b.run("seg_gemm_res_f32", "a", k, 0, "w", k, 1, "y", n, n, "g", "s", "t", len(tarr), None, acc, "dz", n, None)
        else:
            b.run("seg_gemm_f32", "a", k, 0, "w", k if tb_ else n, tb_, "y", n, n, "g", "s", "t", len(tarr), None, acc)
        b.check("y", rtol=2e-4, atol=2e-5)
        outs.append(b.h["y"].cpu().numpy().copy())
        rng = np.random.default_rng(rows + n + 1)  # same dz for both passes
    if kind != "res":
        np.testing.assert_array_equal(outs[0], outs[1])


@pytest.mark.parametrize("co,branches,rows,cin,acc,flags_kind", [
    (15, 4, 300, 120, 0, "mfma16x4"),   # narrowest HYPELCNN level: groups of 60 / 45 / 30 / 15 columns, 128x64 blocks on 16x16x4
    (15, 4, 128, 33, 1, "mfma16x4"),
    (7, 3, 70, 40, 0, "mfma16x4"),
    (16, 4, 257, 64, 0, "mfma16x4"),
    (30, 4, 300, 240, 0, "var_n"),      # level 1: groups of 120 / 90 / 60 / 30 columns on 128x64 blocks (one- / two-tile phases)
    (30, 4, 129, 50, 1, "var_n"),
    (60, 4, 200, 120, 0, "var_n"),      # level 0: 240 / 180 / 120 / 60
    (20, 5, 90, 37, 0, "var_n"),        # DUALCNN-like: five rings
    (30, 4, 300, 240, 0, "none"),       # the flags are hints: the plain kernels give the same result
    (15, 4, 300, 120, 0, "none"),
    # the same launches on the split-operand kernels: 128-wide blocks = the 4 x 2 / 32x64 variant with one- and two-tile
    # MFMA phases per wave (round 6), 64-wide blocks = the ordinary split kernel
    (30, 4, 300, 240, 0, "split128"), (30, 4, 129, 50, 1, "split128"), (60, 4, 200, 120, 0, "split128"),
    (20, 5, 90, 37, 1, "split128"), (30, 4, 1024, 240, 0, "split128"), (15, 4, 300, 120, 0, "split64"),
])
def test_seg_gemm_per_tile_column_counts(hip, co, branches, rows, cin, acc, flags_kind):
    """Merged multi-kernel levels (hypel_tile_t.n): the groups of ONE launch write column ranges [r * co, C) of different
    widths, each from its own segments over a packed image with ldb = C; columns left of a group's range must stay
    untouched, the grid is sized for the widest group."""
    from hypelcnn_amd.plan import GEMM_MFMA16X4, GEMM_VAR_N
    rng = np.random.default_rng(co * 1000 + rows)
    C = co * branches
    n_off = 6
    x = rng.standard_normal((n_off + 2) * rows * cin).astype(np.float32)
    wp = rng.standard_normal(n_off * cin * C).astype(np.float32)
    ncopy = branches
    y0 = rng.standard_normal(ncopy * rows * C).astype(np.float32)
    tb = GemmTables()
    for r in range(branches):
        col0 = r * co
        segs = [((d + (r % 2)) * rows * cin, d * cin * C + col0, cin) for d in range(n_off) if (d + r) % 3 != 0]
        if r == branches - 1:
            segs = segs[:1] + [(segs[0][0] + 3, segs[0][1] + 3 * C, cin - 3)]  # a ragged second segment
        tb.add_group(r * rows * C + col0, segs, rows, n=C - col0)
    garr, sarr, tarr, _ = tb.finalize(C)
    assert set(int(v) for v in tarr["n"]) == {C - r * co for r in range(branches)}
    b = Both(hip)
    for nm, arr in (("x", x), ("w", wp), ("y", y0), ("g", garr), ("s", sarr), ("t", tarr)):
        b.arr(nm, arr)
    flags = {"mfma16x4": GEMM_VAR_N | GEMM_MFMA16X4, "var_n": GEMM_VAR_N | (2 << 8), "none": GEMM_VAR_N,
             "split128": GEMM_VAR_N | SPLIT6 | (3 << 8), "split64": GEMM_VAR_N | SPLIT6 | (2 << 8)}[flags_kind]
    b.run("seg_gemm_f32", "x", cin, 0, "w", C, 0, "y", C, C, "g", "s", "t", len(tarr), None, acc | flags)
    b.check("y", rtol=3e-4, atol=3e-5)
    got = b.h["y"].cpu().numpy().reshape(ncopy, rows, C)
    for r in range(branches):
        assert np.array_equal(got[r][:, :r * co], y0.reshape(ncopy, rows, C)[r][:, :r * co]), "columns left of the group"


def test_copy_blocks(hip):
    rng = np.random.default_rng(5)
    buf = rng.standard_normal(200000).astype(np.float32)
    from hypelcnn_amd.backend import COPY_BLOCK_DTYPE
    ents = [(1000, 100000, 120, 60, 60, 240, 0, 0), (9000, 100060, 120, 60, 60, 240, 0, 0),
            (20000, 150000, 7, 3, 5, 11, 1, 0), (30000, 160000, 1, 1, 1, 1, 1, 0), (40000, 170000, 33, 15, 45, 15, 0, 0),
            (50000, 180000, 50, 64, 64, 64, 1, 0), (60000, 190000, 20, 8, 12, 16, 0, 0)]  # the last two: float4 path
    b = Both(hip)
    b.arr("buf", buf)
    b.arr("e", np.array(ents, COPY_BLOCK_DTYPE))
    b.run("copy_blocks_f32", "buf", "e", len(ents), 120 * 60)
    got, ref = b.h["buf"].cpu().numpy(), b.e["buf"].numpy()
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("rows,e,parts,segs", [(4096, 2, 6, 4), (100, 3, 7, 2), (20000, 5, 3, 3)])
def test_l2norm_segments(hip, rows, e, parts, segs):
    """The applications of a row-concatenated batch keep their own whole-tensor norms: one set per (row segment, part)."""
    rng = np.random.default_rng(6)
    b = Both(hip)
    c = parts * e
    b.arr("x", (rng.standard_normal((segs * rows, c)) * (1 + np.arange(segs * rows)[:, None] // rows)).astype(np.float32))
    b.arr("dy", rng.standard_normal((segs * rows, c)).astype(np.float32))
    b.arr("y", np.zeros(segs * rows * c, np.float32))
    b.arr("dx", rng.standard_normal(segs * rows * c).astype(np.float32))
    b.arr("stat", np.zeros(2 * parts * segs, np.float32))
    b.run("l2norm_segs_fwd", "x", c, rows, e, parts, segs, "y", c, "stat")
    b.check("y", rtol=1e-5, atol=1e-7)
    b.check("stat", rtol=1e-5, atol=1e-7)
    st = b.h["stat"].cpu().numpy().reshape(segs, parts, 2)
    assert st[1, 0, 0] > 3 * st[0, 0, 0], "each segment has its own norm"
    b.run("l2norm_segs_bwd", "x", c, "dy", c, rows, e, parts, segs, "stat", "dx", c, 1)
    b.check("dx", rtol=1e-4, atol=1e-6)


def test_reduce_splits_wave_multi(hip):
    """Every slab reduction of a GAN train op in one launch: one wave per output, entries of different slab counts."""
    from hypelcnn_amd.backend import REDUCE_ENTRY_DTYPE
    rng = np.random.default_rng(8)
    buf = rng.standard_normal(600000).astype(np.float32)
    ents = [(1000, 500000, 1305, 1305, 256, 0), (400000, 510000, 8, 7, 256, 1), (450000, 520000, 232, 232, 37, 0),
            (470000, 530000, 8, 8, 1, 1)]
    b = Both(hip)
    b.arr("buf", buf)
    b.arr("e", np.array(ents, REDUCE_ENTRY_DTYPE))
    b.run("reduce_splits_wave_multi_f32", "buf", "e", len(ents), sum(e[3] for e in ents))
    b.check("buf", rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("bands,n,kept", [(360, 200, True), (64, 2048, True), (144, 37, False), (16, 5, False)])
def test_gan_generator_encoder_tap(hip, bands, n, kept):
    """The full generator also leaves n_4: bit for bit what the encoder-only application on the same input writes; its
    backward pass takes the gradient of that value and returns the input / filter / bias gradients of BOTH applications."""
    from tests.emu_backend import generator_knife_edge_rows
    rng = np.random.default_rng(bands + n)
    ks = [bands >> s for s in (0, 1, 2, 3, 2, 1, 0)]
    wt = sum(ks)
    w = (rng.standard_normal(wt) * 0.4 / np.sqrt(np.repeat(ks, ks))).astype(np.float32)
    bias = (rng.standard_normal(8) * 0.1).astype(np.float32)
    x = rng.random((n, bands)).astype(np.float32)
    for _ in range(20):  # re-draw samples whose leaky-ReLU branch hangs on fp32 rounding (see the generator tests)
        bad = generator_knife_edge_rows(x.astype(np.float64), w.astype(np.float64), bias.astype(np.float64), bands, False)
        if not len(bad):
            break
        x[bad] = rng.random((len(bad), bands)).astype(np.float32)
    blocks = hip.gan_generator_blocks(n)
    b = Both(hip)
    for nm, arr in (("x", x), ("w", w), ("b", bias), ("dout", rng.standard_normal((n, bands)).astype(np.float32)),
                    ("denc", rng.standard_normal((n, bands)).astype(np.float32)), ("out", np.zeros(n * bands, np.float32)),
                    ("enc", np.zeros(n * bands, np.float32)), ("enc_ref", np.zeros(n * bands, np.float32)),
                    ("dx", np.zeros(n * bands, np.float32)), ("pw", np.zeros(blocks * wt, np.float32)),
                    ("pb", np.zeros(blocks * 8, np.float32)), ("dw", np.zeros(wt, np.float32)), ("db", np.zeros(8, np.float32))):
        b.arr(nm, arr)
    keep = None
    if kept:
        keep = "keep"
        b.arr("keep", np.zeros(max(16, hip.gan_generator_keep_floats(n, bands, 0)), np.float32))
    b.run("gan_generator_fwd_tap", "x", bands, n, bands, "w", "b", "out", bands, "enc", bands, keep)
    b.run("gan_generator_fwd", "x", bands, n, bands, "w", "b", 1, "enc_ref", bands)
    assert torch.equal(b.h["enc"], b.h["enc_ref"]), "the tap IS the encoder-only application"
    b.check("out", rtol=2e-4, atol=2e-5)
    b.check("enc", rtol=2e-4, atol=2e-5)
    b.run("gan_generator_bwd_tap", "x", bands, "dout", bands, "denc", bands, n, bands, "w", "b", "dx", bands, 0, "pw", "pb", keep)
    b.run("reduce_splits_pair_f32", "pw", wt, wt, "dw", "pb", 8, 7, "db", blocks, 0)
    b.check("dx", rtol=5e-4, atol=5e-5)
    b.check("dw", rtol=5e-4, atol=5e-5)
    b.check("db", rtol=5e-4, atol=5e-5)


@pytest.mark.parametrize("bands,n,enc,kept,own_regions", [(64, 2048, 0, True, True), (64, 300, 0, False, False),
                                                          (360, 100, 1, True, True), (16, 37, 0, True, False),
                                                          (144, 16 * 300, 0, True, True)])
def test_gan_generator_two_variable_sets_in_one_launch(hip, bands, n, enc, kept, own_regions):
    """hypel_gan_generator_{fwd,bwd}_apps: application g on rows [g*n, (g+1)*n) with the variables `w_stride` / `b_stride`
    behind application 0's.  Per application bit-identical to the single-application entry points (outputs, input
    gradients); its slabs -- consecutive, or in a region of their own reached by pw_stride / pb_stride, as the planner lays
    them out -- sum to the same filter / bias gradients."""
    from tests.emu_backend import generator_knife_edge_rows
    rng = np.random.default_rng(bands + n)
    ks = [bands >> s for s in (0, 1, 2, 3, 2, 1, 0)]
    wt = sum(ks)
    A, gap_w, gap_b = 2, 40, 24  # variables of the second model: a fixed distance behind the first's
    w = np.zeros(A * (wt + gap_w), np.float32)
    bias = np.zeros(A * (8 + gap_b), np.float32)
    x = rng.random((A * n, bands)).astype(np.float32)
    for g in range(A):
        wg = (rng.standard_normal(wt) * 0.4 / np.sqrt(np.repeat(ks, ks))).astype(np.float32)
        bg = (rng.standard_normal(8) * 0.1).astype(np.float32)
        w[g * (wt + gap_w): g * (wt + gap_w) + wt] = wg
        bias[g * (8 + gap_b): g * (8 + gap_b) + 8] = bg
        xs = x[g * n:(g + 1) * n]
        for _ in range(20):
            bad = generator_knife_edge_rows(xs.astype(np.float64), wg.astype(np.float64), bg.astype(np.float64), bands, bool(enc))
            if not len(bad):
                break
            xs[bad] = rng.random((len(bad), bands)).astype(np.float32)
    blocks = hip.gan_generator_blocks_apps(n, A)
    bpa, one = blocks // A, hip.gan_generator_blocks(n)
    assert blocks % A == 0 and blocks <= 512
    pad = 3 * wt + 5  # (own_regions) other applications' slabs sit between the two sets' regions
    pws, pbs = (bpa * wt + pad, bpa * 8 + 16) if own_regions else (0, 0)
    b = Both(hip)
    for nm, arr in (("x", x), ("w", w), ("b", bias), ("dout", rng.standard_normal((A * n, bands)).astype(np.float32)),
                    ("out", np.zeros(A * n * bands, np.float32)), ("out1", np.zeros(A * n * bands, np.float32)),
                    ("dx", np.ones(A * n * bands, np.float32)), ("dx1", np.ones(A * n * bands, np.float32)),
                    ("pw", np.full(A * (bpa * wt + pad), 7.0, np.float32)), ("pb", np.full(A * (bpa * 8 + 16), 7.0, np.float32)),
                    ("pw1", np.zeros(one * wt, np.float32)), ("pb1", np.zeros(one * 8, np.float32)),
                    ("dw", np.zeros(A * wt, np.float32)), ("db", np.zeros(A * 8, np.float32)),
                    ("dw1", np.zeros(A * wt, np.float32)), ("db1", np.zeros(A * 8, np.float32))):
        b.arr(nm, arr)
    keep = keep1 = None
    if kept:
        kf = max(16, hip.gan_generator_keep_floats(n, bands, enc))
        keep, keep1 = b.arr("keep", np.zeros(A * kf, np.float32)), b.arr("keep1", np.zeros(kf, np.float32))
    b.run("gan_generator_fwd_apps", "x", bands, n, A, wt + gap_w, 8 + gap_b, bands, "w", "b", enc, "out", bands, keep)
    b.run("gan_generator_bwd_apps", "x", bands, "dout", bands, n, A, wt + gap_w, 8 + gap_b, pws, pbs, bands, "w", "b", enc,
          "dx", bands, 1, "pw", "pb", keep)
    for g in range(A):
        wg, bg, rows = ("w", g * (wt + gap_w)), ("b", g * (8 + gap_b)), g * n * bands
        b.run("gan_generator_fwd_keep", ("x", rows), bands, n, bands, wg, bg, enc, ("out1", rows), bands, keep1)
        b.run("gan_generator_bwd_kept" if kept else "gan_generator_bwd", ("x", rows), bands, ("dout", rows), bands, n, bands,
              wg, bg, enc, ("dx1", rows), bands, 1, "pw1", "pb1", *([keep1] if kept else []))
        b.run("reduce_splits_pair_f32", "pw1", wt, wt, ("dw1", g * wt), "pb1", 8, 7, ("db1", g * 8), one, 0)
        b.run("reduce_splits_pair_f32", ("pw", g * (pws if pws else bpa * wt)), wt, wt, ("dw", g * wt),
              ("pb", g * (pbs if pbs else bpa * 8)), 8, 7, ("db", g * 8), bpa, 0)
    assert torch.equal(b.h["out"], b.h["out1"]) and torch.equal(b.h["dx"], b.h["dx1"]), "per application: the same kernel work"
    b.check("out", rtol=2e-4, atol=2e-5)
    b.check("dx", rtol=5e-4, atol=5e-5)
    b.check("dw", rtol=5e-4, atol=5e-5)
    b.check("db", rtol=5e-4, atol=5e-5)
    for nm in ("dw", "db"):
        got, ref = b.h[nm].cpu().numpy(), b.h[nm + "1"].cpu().numpy()
        np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-6 * max(1.0, float(np.abs(ref).max())), err_msg=nm)
    if own_regions:  # nothing outside the two regions' slabs was written
        pw = b.h["pw"].cpu().numpy()
        assert (pw[bpa * wt:bpa * wt + pad] == 7.0).all() and (pw[pws + bpa * wt:] == 7.0).all()


@pytest.mark.parametrize("widths,n,own_regions", [([64, 64, 64, 64, 32], 2048, True), ([64, 64, 64, 64, 32], 4096, True),
                                                  ([64, 64, 64, 64, 32], 133, False), ([24, 24, 12], 50, True),
                                                  ([100, 50, 25, 10], 16 * 200, False)])
def test_dense_stack_two_variable_sets_in_one_launch(hip, widths, n, own_regions):
    """hypel_dense_stack_{fwd,bwd}_apps: two critics of one shape (cycle_gan_wrapper.py: D_x, D_y), each on its half of the
    rows and its share of the blocks -- per application bit-identical to hypel_dense_stack_fwd / _bwd, slabs summing to
    the same gradients."""
    rng = np.random.default_rng(n + sum(widths))
    L = len(widths) - 1
    wt, bt = sum(a * c for a, c in zip(widths[:-1], widths[1:])), sum(widths[1:])
    A, gap_w, gap_b = 2, 52, 12
    w = np.zeros(A * (wt + gap_w), np.float32)
    bias = np.zeros(A * (bt + gap_b), np.float32)
    for g in range(A):
        w[g * (wt + gap_w): g * (wt + gap_w) + wt] = np.concatenate(
            [(rng.standard_normal(a * c) / np.sqrt(a)).astype(np.float32) for a, c in zip(widths[:-1], widths[1:])])
        bias[g * (bt + gap_b): g * (bt + gap_b) + bt] = (rng.standard_normal(bt) * 0.1).astype(np.float32)
    shape = (L, *(widths + [0] * (5 - len(widths))), (1 << min(2, L - 1)) - 1, 0.1)
    c0, cL = widths[0], widths[-1]
    blocks = hip.dense_stack_blocks_apps(n, A)
    bpa, one = blocks // A, hip.dense_stack_blocks(n)
    assert blocks % A == 0 and blocks <= 256
    pad = wt + 9
    pws, pbs = (bpa * wt + pad, bpa * bt + 8) if own_regions else (0, 0)
    b = Both(hip)
    for nm, arr in (("x", rng.standard_normal((A * n, c0)).astype(np.float32)), ("w", w), ("b", bias),
                    ("dout", rng.standard_normal((A * n, cL)).astype(np.float32)),
                    ("out", np.zeros(A * n * cL, np.float32)), ("out1", np.zeros(A * n * cL, np.float32)),
                    ("dx", np.ones(A * n * c0, np.float32)), ("dx1", np.ones(A * n * c0, np.float32)),
                    ("pw", np.full(A * (bpa * wt + pad), 7.0, np.float32)), ("pb", np.full(A * (bpa * bt + 8), 7.0, np.float32)),
                    ("pw1", np.zeros(one * wt, np.float32)), ("pb1", np.zeros(one * bt, np.float32)),
                    ("dw", np.zeros(A * wt, np.float32)), ("db", np.zeros(A * bt, np.float32)),
                    ("dw1", np.zeros(A * wt, np.float32)), ("db1", np.zeros(A * bt, np.float32))):
        b.arr(nm, arr)
    b.run("dense_stack_fwd_apps", "x", c0, n, A, wt + gap_w, bt + gap_b, *shape, "w", "b", "out", cL)
    b.run("dense_stack_bwd_apps", "x", c0, "dout", cL, n, A, wt + gap_w, bt + gap_b, pws, pbs, *shape, "w", "b", "dx", c0, 1,
          "pw", "pb")
    for g in range(A):
        wg, bg = ("w", g * (wt + gap_w)), ("b", g * (bt + gap_b))
        b.run("dense_stack_fwd", ("x", g * n * c0), c0, n, *shape, wg, bg, ("out1", g * n * cL), cL)
        b.run("dense_stack_bwd", ("x", g * n * c0), c0, ("dout", g * n * cL), cL, n, *shape, wg, bg, ("dx1", g * n * c0), c0, 1,
              "pw1", "pb1")
        b.run("reduce_splits_pair_f32", "pw1", wt, wt, ("dw1", g * wt), "pb1", bt, bt, ("db1", g * bt), one, 0)
        b.run("reduce_splits_pair_f32", ("pw", g * (pws if pws else bpa * wt)), wt, wt, ("dw", g * wt),
              ("pb", g * (pbs if pbs else bpa * bt)), bt, bt, ("db", g * bt), bpa, 0)
    assert torch.equal(b.h["out"], b.h["out1"]) and torch.equal(b.h["dx"], b.h["dx1"]), "per application: the same kernel work"
    b.check("out", rtol=2e-4, atol=2e-5)
    b.check("dx", rtol=5e-4, atol=5e-5)
    b.check("dw", rtol=5e-4, atol=5e-5)
    b.check("db", rtol=5e-4, atol=5e-5)
    for nm in ("dw", "db"):
        got, ref = b.h[nm].cpu().numpy(), b.h[nm + "1"].cpu().numpy()
        np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-6 * max(1.0, float(np.abs(ref).max())), err_msg=nm)
    if own_regions:
        pw = b.h["pw"].cpu().numpy()
        assert (pw[bpa * wt:bpa * wt + pad] == 7.0).all() and (pw[pws + bpa * wt:] == 7.0).all()


@pytest.mark.parametrize("n0,n1,off", [(2048 * 64, 2048 * 64, 0), (1000, 37, 0), (4096, 0, 0), (513, 2052, 1)])
def test_copy_pair(hip, n0, n1, off):
    """hypel_copy_pair_f32: two flat copies in one launch (vector path when aligned, scalar otherwise), nothing else touched."""
    rng = np.random.default_rng(n0 + n1)
    b = Both(hip)
    b.arr("s0", rng.standard_normal(n0 + off).astype(np.float32))
    b.arr("s1", rng.standard_normal(max(n1, 1) + off).astype(np.float32))
    b.arr("d0", np.full(n0 + 8, 3.0, np.float32))
    b.arr("d1", np.full(max(n1, 1) + 8, 3.0, np.float32))
    b.run("copy_pair_f32", "d0", ("s0", off), n0, "d1", ("s1", off), n1)
    for d, s_, n in (("d0", "s0", n0), ("d1", "s1", n1)):
        got = b.h[d].cpu().numpy()
        assert np.array_equal(got[:n], b.h[s_].cpu().numpy()[off:off + n]) and (got[n:] == 3.0).all()
        assert np.array_equal(got, b.e[d].numpy())


@pytest.mark.parametrize("rows,k,n,groups,bias,code,hint", [
    (16384, 60, 60, 6, True, 1, 0),    # CUT feature discriminator, first layer: six band slices, slope 0.1
    (8192, 360, 360, 1, True, 1, 0),   # the 360-band critic's first layer
    (300, 77, 33, 2, False, 2, 1),     # ragged, no bias, slope 0.18, 128x32 blocks
    (129, 19, 70, 1, True, 4, 2),      # slope 0.01, 128x64 blocks
    (4096, 180, 90, 1, True, 3, 0),
])
def test_seg_gemm_leaky_relu_epilogue(hip, rows, k, n, groups, bias, code, hint):
    """HYPEL_GEMM_ACT_*: C = leaky_relu(product + bias) straight from the accumulators -- a normaliser-less
    tf_slim.fully_connected (gan/shadow_data_models.py:95-149) in one launch.  Equal, bit for bit, to the plain product
    followed by the element-wise activation kernel; columns outside the groups stay untouched."""
    rng = np.random.default_rng(rows + k + n)
    ldc = groups * n + 3
    a = rng.standard_normal((rows, groups * k)).astype(np.float32)
    w = (rng.standard_normal(groups * k * n) / np.sqrt(k)).astype(np.float32)
    bv = rng.standard_normal(ldc).astype(np.float32)
    slope = [0.0, 0.1, 0.18, 0.2, 0.01][code]
    b = Both(hip)
    garr, sarr, tarr, _ = _tables(b, [(g * n, [(g * k, g * k * n, k)], rows) for g in range(groups)]).finalize(n)
    for nm, arr in (("a", a), ("w", w), ("c", np.full((rows, ldc), 5.0, np.float32)), ("c2", np.full((rows, ldc), 5.0, np.float32)),
                    ("z2", np.zeros((rows, ldc), np.float32)), ("bias", bv), ("g", garr), ("s", sarr), ("t", tarr)):
        b.arr(nm, arr)
    b.run("seg_gemm_f32", "a", groups * k, 0, "w", n, 0, "c", ldc, n, "g", "s", "t", len(tarr), "bias" if bias else None,
          (code << 16) | (hint << 8))
    b.check("c", rtol=2e-4, atol=2e-5)
    # the two-launch form on the device
    b.run("seg_gemm_f32", "a", groups * k, 0, "w", n, 0, "c2", ldc, n, "g", "s", "t", len(tarr), "bias" if bias else None,
          hint << 8)
    b.run("bn_act_fwd", "c2", ldc, rows, groups * n, None, None, None, 1, slope, None, 0, None, 0, None, None, 0, None, "z2", ldc)
    got, two = b.h["c"].cpu().numpy().reshape(rows, ldc), b.h["z2"].cpu().numpy().reshape(rows, ldc)
    assert np.array_equal(got[:, :groups * n], two[:, :groups * n]), "same arithmetic as product -> activation kernel"
    assert (got[:, groups * n:] == 5.0).all()


def test_loss_terms_slots(hip):
    """hypel_loss_terms_slots + hypel_loss_finalize_slots: four terms (least squares, L1 with two gradients, Wasserstein
    mean, l2 regulariser) in one launch, base-relative operands, accumulate flags on known gradient contents; three rounds."""
    from hypelcnn_amd.backend import LOSS_NONE, LOSS_TERM_DTYPE
    rng = np.random.default_rng(5)
    rows, c, nw = 300, 37, 5000
    # one arena: a0 | a1 | b1 | a2 | w | da0 | da1 | db1 | dw
    sizes = [rows * c] * 4 + [nw] + [rows * c] * 3 + [nw]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    arena = rng.standard_normal(int(offs[-1])).astype(np.float32)
    arena[offs[5]:] = 1.5
    N = LOSS_NONE
    terms = np.array([
        (offs[0], N, offs[5], N, c, 0, c, 0, rows, 0, c, 0, 0, 1.0, 0.5 / (rows * c), 0.5 / (rows * c), 0),
        (offs[1], offs[2], offs[6], offs[7], c, c, c, c, rows, 1, c, 1, 0, 0.0, 10.0 / (rows * c), 10.0 / (rows * c), 1),
        (offs[3], N, N, N, c, 0, 0, 0, rows, 2, c, 0, 0, 0.0, 1.0 / (rows * c), 1.0 / (rows * c), 2),
        (offs[4], N, offs[8], N, 0, 0, 0, 0, nw, 3, 1, 1, 0, 0.0, 1e-3, 0.5e-3, 3)], LOSS_TERM_DTYPE)
    b = Both(hip)
    for nm, arr in (("arena", arena), ("terms", terms), ("slots", np.zeros(4 * 1024, np.float32)),
                    ("loss", np.full(1, 3.0, np.float32))):
        b.arr(nm, arr)
    for acc in (0, 1, 1):
        b.run("loss_terms_slots", "arena", "terms", 4, "slots")
        b.run("loss_finalize_slots", "slots", 4, "loss", acc)
    b.check("arena", rtol=1e-5, atol=1e-6)
    b.check("loss", rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("split,hint,with_res,acc", [(True, 3, True, 1), (True, 2, False, 0), (True, 1, True, 0),
                                                     (False, 2, True, 1), (False, 1, False, 1)])
def test_seg_gemm_kslice_plain_records(hip, split, hint, with_res, acc):
    """K-slice records (HYPEL_TILE_PLAIN): a data gradient over 5 pixels whose heavy tiles are cut into slices -- slice 0 keeps
    the launch's accumulate bit / bias-less shortcut gather, the other slices write plain partials into a scratch region
    addressed relative to C, tail records dealt to the end of the eight XCD shares with empty padding records; then
    hypel_reduce_splits_multi_f32 adds the partials.  Against the emulation, and the sum against the unsliced launch."""
    rng = np.random.default_rng(5 + hint)
    nb, P, cin, cout = 200, 5, 120, 60
    rows = P * nb
    dy = (rng.standard_normal((rows, cout)) * 0.5).astype(np.float32)
    dz = rng.standard_normal((rows, cin)).astype(np.float32)
    w = (rng.standard_normal((9, cin, cout)) * 0.5).astype(np.float32)
    dx0 = rng.standard_normal((rows, cin)).astype(np.float32)
    nseg = [9, 4, 6, 9, 2]
    base_groups = [(p * nb * cin, [(((p + s) % P) * nb * cout, s * cin * cout, cout) for s in range(nseg[p])], nb)
                   for p in range(P)]
    region = nb * cin
    scratch0 = rows * cin  # the partials live behind the output in the same test buffer
    tb = GemmTables()
    entries, spos = [], 0
    for p, (c_off, gs, r) in enumerate(base_groups):
        s_ = 3 if len(gs) == 9 else (2 if len(gs) == 6 else 1)
        if s_ == 1:
            tb.add_group(c_off, gs, r)
            continue
        parts = TowerPlan._cut_segments(gs, s_, 1, 1)
        tb.add_group(c_off, parts[0], r, tail=True)
        for i in range(1, s_):
            tb.add_group(scratch0 + spos + (i - 1) * region, parts[i], r, flags=1, tail=True)
        entries.append((scratch0 + spos, c_off, region, region, s_ - 1, 1))
        spos += (s_ - 1) * region
    garr, sarr, tarr, _ = tb.finalize(cin)
    assert len(tarr) % 8 == 0 and (tarr["flags"] & 1).sum() == 2 * 2 * 2 + 1 * 2  # two row tiles per pixel
    g0, s0, t0, _ = _tables(None, base_groups).finalize(cin)
    flags = (SPLIT6 if split else 0) | (hint << 8) | acc
    from hypelcnn_amd.backend import REDUCE_ENTRY_DTYPE
    b = Both(hip)
    buf = np.concatenate([dx0.ravel(), np.full(spos, 7.0, np.float32)])  # (the partials must not depend on what is there)
    for nm, arr in (("dy", dy), ("dz", dz), ("w", w), ("y", buf), ("yref", dx0.ravel().copy()), ("g", garr), ("s", sarr),
                    ("t", tarr), ("g0", g0), ("s0", s0), ("t0", t0), ("e", np.array(entries, REDUCE_ENTRY_DTYPE))):
        b.arr(nm, arr)
    if with_res:
        b.run("seg_gemm_res_f32", "dy", cout, 0, "w", cout, 1, "y", cin, cin, "g", "s", "t", len(tarr), None, flags, "dz", cin, None)
        b.run("seg_gemm_res_f32", "dy", cout, 0, "w", cout, 1, "yref", cin, cin, "g0", "s0", "t0", len(t0), None, flags, "dz", cin, None)
    else:
        b.run("seg_gemm_f32", "dy", cout, 0, "w", cout, 1, "y", cin, cin, "g", "s", "t", len(tarr), None, flags)
        b.run("seg_gemm_f32", "dy", cout, 0, "w", cout, 1, "yref", cin, cin, "g0", "s0", "t0", len(t0), None, flags)
    b.run("reduce_splits_multi_sized_f32", "y", "e", len(entries), region)
    b.check("y", rtol=2e-4, atol=2e-5)
    got = b.h["y"].cpu().numpy()[:rows * cin]
    want = b.h["yref"].cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-4 * float(np.abs(want).max()))
