// Grouped multi-segment fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32).
//
//   C_g[rows_g x n] (+)= sum_s op(A_s)[rows_g x k_s] * op(B_s)[k_s x n]  (+ bias)
//
// One kernel template serves the exact-tap convolution forward (segments = valid taps on
// pixel-major activations), its data gradient (trans_b) and its filter gradient (trans_a,
// split over batch rows), and every fully-connected layer.  Design notes:
//  * block = 256 threads = 4 wave64; each wave owns TM x TN tiles of 32x32 accumulators.
//  * operands are staged global -> registers -> LDS; the next k-tile's global loads are in
//    flight while the current one feeds the MFMAs.  fp32 MFMA issues once per 64 cycles per
//    SIMD, so dword loads (128 B per half-wave, fully coalesced) are nowhere near the issue
//    limit and remove every alignment requirement (K = 145, channel offsets 15/30/45 ...).
//  * LDS tiles keep the global row order; a tile that is read DOWN its strided dimension uses an
//    odd pitch (33 floats) so the 32 lanes of a ds_read_b32 group hit 32 distinct banks.
//  * MFMA work is skipped for 32x32 tiles that lie wholly outside rows_g / n and for k-steps
//    beyond the segment's K, so ragged shapes (n = 60, 120, K = 145) cost only LDS zero fill.
//  * 1-D grid with the bijective XCD remap: the blocks that share an A row-tile (different
//    column tiles) run on the same XCD and hit its L2.
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// SPLIT variants (HYPEL_GEMM_SPLIT6): an fp32 value is EXACTLY hi + mid + lo with three round-to-nearest bf16 parts
// (8 + 8 + 8 significand bits, the exponent range of fp32); every part x part product is exact in the fp32 accumulator
// of v_mfma_f32_32x32x16_bf16, and the three smallest of the nine partial products (mid lo, lo mid, lo lo) are below
// 2^-24 |a b|, i.e. below the rounding unit of the fp32 accumulate that follows.  Six bf16 MFMAs per k-step therefore
// give an fp32-grade product (tests/test_gpu_kernels.py measures it against the fp32 MFMA chain per launch shape) at
// 16 / 6 = 2.67x the rate of v_mfma_f32_32x32x2_f32.
// (a, b) -> bf16(a) | bf16(b) << 16, round to nearest even.  Inline assembly: as a C++ cast pair hipcc re-converts element 0
// alone wherever `word << 16` is needed (5 instead of 3 conversions per pair).
__device__ __forceinline__ unsigned hypel_cvt_pk_bf16(float a, float b) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// two floats -> their (hi, mid, lo) bf16 parts, element 0 in the low half of each word: 11 VALU operations per pair.
// Measured and dropped (round-5 notes): the residual as one v_dot2c_f32_bf16 per element with the constants (-1, 0) /
// (0, -1) -- 7 operations per pair, but the instruction runs below the full VALU rate on gfx950 (171 vs 160 us on the
// M = 50 176, K = n = 480 product) and does not return the exact residual; as one v_pk_add_f32 per pair -- hipcc pays
// for the register pairing with more moves than it saves.
__device__ __forceinline__ void hypel_residual2(float& x0, float& x1, unsigned part) {
    x0 -= __builtin_bit_cast(float, part << 16);
    x1 -= __builtin_bit_cast(float, part & 0xffff0000u);
}
__device__ __forceinline__ void hypel_split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    h = hypel_cvt_pk_bf16(x0, x1);
    hypel_residual2(x0, x1, h);
    m = hypel_cvt_pk_bf16(x0, x1);
    hypel_residual2(x0, x1, m);
    l = hypel_cvt_pk_bf16(x0, x1);
}

namespace {

// f(integral_constant<int, LO>), ..., f(integral_constant<int, LO + N - 1>)
template <int LO, class F, int... Is>
__device__ __forceinline__ void hypel_for_range(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, LO + Is>{}), ...);
}

#ifndef HYPEL_GEMM_BK
#define HYPEL_GEMM_BK 32  // reduction columns per LDS tile (a multiple of 16)
#endif
constexpr int BK_WIDE = HYPEL_GEMM_BK;
#ifndef HYPEL_GEMM_BK_NARROW
#define HYPEL_GEMM_BK_NARROW 32  // reduction columns per LDS tile of the 128x16 variant (experiments: 64)
#endif
constexpr int BK_NARROW = HYPEL_GEMM_BK_NARROW;
#ifndef HYPEL_GEMM_CLK
#define HYPEL_GEMM_CLK 0  // tools/gemm_quantisation.py: 1 = --clk (shader clock the kernel ran at), 2 = --timeline
#endif
#ifndef HYPEL_GEMM_CHUNK
#define HYPEL_GEMM_CHUNK 8  // granularity (reduction columns) at which a short k-tile stops issuing MFMAs
#endif
#ifndef HYPEL_GEMM_RD64
#define HYPEL_GEMM_RD64 51  // operand fragments of two consecutive MFMA k-steps from ONE ds_read_b64 (see RD64 below); bits:
                           // 1 = A B products, 2 = A B^T, 4 = A^T B, 8 = A^T B^T; 16 = also the 128x32 variants;
                           // 32 = k-contiguous operands are transposed into the pair-interleaved image as well
#endif
#ifndef HYPEL_GEMM_BATCHED_EPILOGUE
#define HYPEL_GEMM_BATCHED_EPILOGUE 2  // 1: read-modify-write epilogues of full tiles issue all loads before the stores;
                                       // 2: every full tile's epilogue on raw buffer accesses (one offset + scalar row step)
#endif
#ifndef HYPEL_GEMM_HOIST_EPI
#define HYPEL_GEMM_HOIST_EPI 1  // bias / shortcut column ranges requested before the k loop
#endif
constexpr int CHUNK = HYPEL_GEMM_CHUNK;
static_assert(CHUNK % 4 == 0 && BK_WIDE % CHUNK == 0 && BK_NARROW % CHUNK == 0, "chunk of k2 / k4 steps");
#ifndef HYPEL_SPLIT_WAVE_ROWS_FIRST
#define HYPEL_SPLIT_WAVE_ROWS_FIRST 1  // split variants: wave -> (wave % WM, wave / WM) instead of (wave / WN, wave % WN)
#endif
#ifndef HYPEL_OCC_BN32
#define HYPEL_OCC_BN32 3  // waves per SIMD the 128x32 variant is compiled for
#endif
#ifndef HYPEL_OCC_BN64_TA
#define HYPEL_OCC_BN64_TA 3  // waves per SIMD the 128x64 filter-gradient (A transposed) variants are compiled for
#endif
#ifndef HYPEL_OCC_BN32_TA
#define HYPEL_OCC_BN32_TA 3  // ... the 128x32 filter-gradient variants
#endif
#ifndef HYPEL_OCC_BN96
#define HYPEL_OCC_BN96 5  // ... and the 128x96 data-gradient variant (5 blocks per CU = 1280 resident: 392 x 3 row x column
                          // tiles fit); the forward variant needs ~104 registers: 4 waves per SIMD, 1024 resident
#endif

// NARROW: 128x16 blocks on v_mfma_f32_16x16x4_f32 for n <= 16 (the Cout = 15 level of HYPELCNN, fc_final): each wave
// owns 32 rows x 16 columns as two 16x16 accumulators, so a 15-column output wastes 1/16 of the MFMA work instead of
// the 17/32 it wastes on a 32-wide tile.
// MULTI: one launch covers the tiles of SEVERAL products (the filter gradients of many layers): `tiles` is then an
// array of hypel_mtile_t, one record per BLOCK, that carries the block's column tile, its product's n / lda / ldb /
// ldc and accumulate flag; operand offsets (records and segments) are relative to the one base pointer passed as
// A = B = C.
// PAIR (data gradients, !TA && TB): two consecutive segments of at most 16 reduction columns each share ONE k-tile
// (columns 0-15 from the first, 16-31 from the second).  A data gradient through a convolution with 15 filters (the
// narrowest HYPELCNN level, DUALCNN's last levels) otherwise stages, synchronises and walks a whole 32-column k-tile
// for 15 useful columns.  A segment whose k has HYPEL_SEG_PAIR_FLAG set is paired with the next one of its group.
// VARN (forward 128x64 blocks): the groups of the launch differ in their column count (hypel_tile_t.n, merged levels);
// the MFMA phase then has one variant per number of active accumulator tiles.  A separate instantiation: as a shared
// code path the second variant cost the 128x64 forward kernel 20 registers (82 -> 102: 4 instead of 5 waves per SIMD).
// ACT (plain forward products, a separate instantiation like VARN): C = act(product + bias) with act = leaky-ReLU of the
// slope HYPEL_GEMM_ACT_* names (bits 16-18 of `accumulate`): the bias + activation launch behind a BN-less
// tf_slim.fully_connected (the GAN critics' and feature-discriminator layers, gan/shadow_data_models.py:95-149) is gone.
// SPLIT: operands split three ways on their way into LDS (three bf16 planes per operand, reduction dimension
// contiguous), six v_mfma_f32_32x32x16_bf16 per 32x32x16 step; prologue, segment walk and epilogues are shared.
template <int WM, int WN, int TM, int TN, bool TA, bool TB, bool NARROW = false, bool MULTI = false, bool PAIR = false,
          bool VARN = false, bool ACT = false, bool SPLIT = false>
__device__ __forceinline__ void seg_gemm_body(const float* __restrict__ A, int64_t lda,
                                                        const float* __restrict__ B, int64_t ldb,
                                                        float* __restrict__ C, int64_t ldc, int n,
                                                        const hypel_group_t* __restrict__ groups,
                                                        const hypel_seg_t* __restrict__ segs,
                                                        const void* __restrict__ tiles_v, int n_tiles,
                                                        int n_ntiles, const float* __restrict__ bias,
                                                        int accumulate, const float* __restrict__ res, int64_t ldr,
                                                        const int32_t* __restrict__ res_start,
                                                        float* __restrict__ stats) {
    constexpr int BM = WM * TM * 32;
    // NARROW with TN = 4 (forward only): 128x64 blocks whose waves own 32 rows x FOUR 16-column tiles -- the merged
    // form of a multi-kernel level with <= 16 filters per branch (groups of 15 / 30 / 45 / 60 output columns, see
    // hypel_tile_t.n): the A tile is staged once for up to four branches and the MFMA work stays exact to 16 columns
    constexpr int NT16 = NARROW ? TN : 1;  // 16-column accumulator tiles per wave
    constexpr int BN = NARROW ? 16 * NT16 : WN * TN * 32;
    constexpr int NT = 64 * WM * WN;  // threads per block
    static_assert(WM * WN == 4 || (SPLIT && WM * WN == 8), "4 waves per block (split variants: 4 or 8)");
    static_assert(BM == HYPEL_GEMM_BM, "tile table is built for BM = 128");
    static_assert(!ACT || (!TA && !TB && !NARROW && !MULTI && !PAIR && !VARN), "activation epilogue: plain forward only");
    static_assert(!SPLIT || (!NARROW && !PAIR && !VARN && !ACT && !(TA && TB)), "split variants: plain NN / NT / TN products");
#if HYPEL_GEMM_CLK
    const float* const clk_dbg = bias;  // (the debug buffer travels in `bias`; K-slice records null the bias itself below)
#endif
    [[maybe_unused]] int act_idx = 0;
    if constexpr (ACT) {
        act_idx = accumulate >> 16;
        accumulate &= 1;
    }
    static_assert(!NARROW || (WM == 4 && TM == 1 && (TN == 1 || (TN == 4 && !TA && !TB && !MULTI))),
                  "narrow variant: 4 x 1 waves of 32 x 16 (forward: 32 x 64)");
    // LDS images (rows x pitch), global row order preserved
    constexpr int BK = NARROW ? BK_NARROW : BK_WIDE;
    constexpr int A_ROWS = TA ? BK : BM;
    constexpr int A_COLS = TA ? BM : BK;
    // pitches: a fragment read must hit 32 distinct banks per half-wave.  32x32x2 fragments read 32 rows at one k
    // (odd pitch 33 for the k-contiguous image); 16x16x4 fragments read 16 rows x 2 k per half-wave (pitch 34 resp.
    // 16 extra floats per k row of the transposed image)
    // RD64 (32x32x2 variants): the k-step PAIR t of a k-tile multiplies the reduction columns {4t, 4t+2} (first MFMA:
    // lane half h supplies column 4t + 2h) and {4t+1, 4t+3} (second MFMA), so that a lane needs the two ADJACENT
    // columns 4t + 2h, 4t + 2h + 1: one ds_read_b64 per operand tile per step pair instead of two ds_read_b32 (the b64
    // form moves 256 B per LDS clock, the b32 form 128; MI355X_MICROARCH.md LDS table).  k-contiguous images ([m][k],
    // [n][k]) keep their natural order with an even pitch of BK + 2 = 34 floats (32 rows x 8 bytes then cover the 64
    // banks exactly once; the 2-way conflict of the staging stores is hidden under their register transfer); the
    // k-strided images ([k][m], [k][n]) interleave the rows of a column pair: (k, x) -> ((k >> 1) * W + x) * 2 + (k & 1).
    // The order of the products inside a k-tile changes (0,2,1,3,4,6,...), the result stays one fixed fmaf chain.
    constexpr bool RD64 = !NARROW && ((HYPEL_GEMM_RD64 >> ((TA ? 2 : 0) + (TB ? 1 : 0))) & 1) &&
                          (TM * TN > 1 || (HYPEL_GEMM_RD64 & 16));
    // LIN: a k-contiguous operand is transposed on its way into LDS, into the same pair-interleaved image the
    // k-strided operands use -- (x, k) -> (k >> 1) * PP + 2 x + (k & 1), pair pitch PP = 2 W + 2 (the 32 lanes that
    // store one row's 32 columns then hit 32 banks) -- so that its fragment reads are 32 consecutive 8-byte words too
    constexpr bool LIN = RD64 && (HYPEL_GEMM_RD64 & 32);
    constexpr int PPA = 2 * BM + 2, PPB = 2 * BN + 2;
    constexpr int A_PITCH = TA ? (NARROW ? BM + 16 : BM) : (NARROW || RD64 ? BK + 2 : BK + 1);
    constexpr int B_ROWS = TB ? BN : BK;
    constexpr int B_COLS = TB ? BK : BN;
    constexpr int B_PITCH = TB ? (NARROW || RD64 ? BK + 2 : BK + 1) : (NARROW && NT16 > 1 ? BN + 16 : BN);
    constexpr int A_PER_THREAD = A_ROWS * A_COLS / 256;
    // a 96-column B row does not divide the 256 threads: then only the first (256 / B_COLS) * B_COLS = 192 threads
    // stage B (two rows of 96 per pass), the fourth wave sits that part out
    constexpr int B_THREADS = (256 / B_COLS) * B_COLS;
    constexpr int B_PER_THREAD = B_ROWS * B_COLS / B_THREADS;
    constexpr int A_RSTEP = 256 / A_COLS;
    constexpr int B_RSTEP = 256 / B_COLS;
    static_assert(!RD64 || (A_RSTEP % 2 == 0 && B_RSTEP % 2 == 0 && (A_ROWS * A_PITCH) % 2 == 0), "RD64 staging layout");
    // SPLIT: k-tiles of SP_BK = 16 reduction columns (one 32x32x16 step); three bf16 planes per operand, each [rows][16]
    // with the reduction dimension contiguous and rows of 48 bytes (16 + 8 halves): the 16 lanes that a ds_read_b128
    // services per LDS cycle (rows 0-3, 12-15, 20-27 of a 32-row fragment) then start in 16 different 4-bank groups
    // (12 r mod 64) -- conflict-free fragment reads; so are the 16-byte staging stores of eight consecutive rows.
    constexpr int SP_BK = 16;
    constexpr int SP_PITCH = (SP_BK + 8) / 2;  // floats (48 bytes) per plane row
    constexpr int SP_PLANE_A = BM * SP_PITCH, SP_PLANE_B = BN * SP_PITCH;  // floats per plane
    constexpr int SP_BUF = 3 * (SP_PLANE_A + SP_PLANE_B);                  // floats per buffer; the split variants keep TWO
    __shared__ __attribute__((aligned(16))) float lds[SPLIT ? 2 * SP_BUF : A_ROWS * A_PITCH + B_ROWS * B_PITCH];
    float* As = lds;
    float* Bs = lds + (SPLIT ? 3 * SP_PLANE_A : A_ROWS * A_PITCH);

    // ---- block -> (row tile, column tile), XCD-aware (bijective remap, guide §5 T1) ----
    const int nblk = gridDim.x;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, loc = bid >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    struct { int64_t c_off; int seg_begin, seg_count, rows; } grp;
    struct { int64_t a_off0, b_off0; int k0; } tile;
    int m0, n0;
    if constexpr (MULTI) {
        if (lid >= n_tiles) return;
        const hypel_mtile_t rec = reinterpret_cast<const hypel_mtile_t*>(tiles_v)[lid];
        grp = {rec.c_off, rec.seg_begin, rec.seg_count, rec.rows};
        tile = {rec.a_off0, rec.b_off0, rec.k0};
        m0 = rec.m0;
        n0 = rec.n0;
        n = rec.n;
        lda = rec.lda;
        ldb = rec.ldb;
        ldc = rec.ldc;
        accumulate = rec.flags & 1;
    } else {
        const int tile_id = lid / n_ntiles;
        n0 = (lid - tile_id * n_ntiles) * BN;
        if (tile_id >= n_tiles) return;
        // one record per tile: the group's fields and its first segment travel with it (no tiles -> groups -> segs
        // chain in front of the first operand loads)
        const hypel_tile_t t = reinterpret_cast<const hypel_tile_t*>(tiles_v)[tile_id];
        grp = {t.c_off, t.seg_begin, t.seg_count, t.rows};
        tile = {t.a_off0, t.b_off0, t.k0};
        m0 = t.m0;
        if (t.n > 0) n = t.n;  // this tile's group has its own column count (merged multi-kernel levels)
        if (t.flags & HYPEL_TILE_PLAIN) {  // K-slice partial: the plain sum into its own region (block-uniform)
            bias = nullptr;
            res = nullptr;
            accumulate &= ~1;
        }
    }
    const int rows_left = grp.rows - m0;  // valid rows in this tile (may exceed BM)
    if (rows_left <= 0) return;           // empty record (padding of an XCD's share of the table)
    const int cols_left = n - n0;
    if (cols_left <= 0) return;  // the grid is sized for the widest group of the launch

    const int tid = threadIdx.x;
    const int lane = tid & 63;
#if HYPEL_GEMM_CLK  // diagnostic build: shader-clock and 100 MHz wall counters over this block's k loop, returned through `bias`
    const long long clk0 = clock64(), wall0 = wall_clock64();
#endif
    // wave index as a SCALAR: every predicate derived from it is wave-uniform and compiles to s_cbranch,
    // not to exec-masked regions around the MFMAs
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // SPLIT: consecutive waves (= the four SIMDs of a CU) take different ROW slabs of the same column tiles first, so that a
    // group that fills only the first column tile(s) of the block (the outer rings of a merged multi-kernel level:
    // hypel_tile_t.n = 30 of 120 columns) still has MFMA work on every SIMD; the fp32 kernels keep the column-major deal
    const int wm = SPLIT && HYPEL_SPLIT_WAVE_ROWS_FIRST ? wave % WM : wave / WN;
    const int wn = SPLIT && HYPEL_SPLIT_WAVE_ROWS_FIRST ? wave / WM : wave % WN;
    const int l31 = lane & 31, lhi = lane >> 5;

    // per-thread staging coordinates
    const int a_col = tid % A_COLS, a_row0 = tid / A_COLS;
    const bool b_stager = B_THREADS == 256 || tid < B_THREADS;
    const int b_col = b_stager ? tid % B_COLS : 0x3fffffff / 4, b_row0 = b_stager ? tid / B_COLS : 0;

    f32x16 acc[NARROW ? 1 : TM][NARROW ? 1 : TN];
    f32x4 acc16[2 * NT16];  // NARROW: [t * NT16 + j] = rows [16 t, 16 t + 16) x columns [16 j, 16 j + 16) of the wave's tile
    if constexpr (NARROW) {
#pragma unroll
        for (int t = 0; t < 2 * NT16; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc16[t][e] = 0.0f;
    } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    }

    // wave-uniform activity of each 32x32 accumulator tile
    bool row_act[TM], col_act[TN];
    bool any_row = false, any_col = false;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        row_act[i] = (wm * TM + i) * 32 < rows_left;
        any_row = any_row || row_act[i];
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        col_act[j] = (wn * TN + j) * (NARROW ? 16 : 32) < cols_left;
        any_col = any_col || col_act[j];
    }
    // active accumulator column tiles are a prefix: their count, as a scalar (merged levels: a 64-wide block may
    // serve a group of 15 - 60 columns; the MFMA phase below runs the variant for exactly that many)
    int tn_act = 0;
#pragma unroll
    for (int j = 0; j < TN; ++j) tn_act += col_act[j] ? 1 : 0;
    tn_act = __builtin_amdgcn_readfirstlane(tn_act);
    const bool any_act = any_row && any_col;
    // per-lane LDS read bases (k advances by immediate offsets in the unrolled loop)
    const int l15 = lane & 15, lq = lane >> 4;  // 16x16x4 fragments: row / column = lane & 15, k = lane >> 4
    const int a_rd = NARROW ? (TA ? (lq * A_PITCH + wm * 32 + l15) : ((wm * 32 + l15) * A_PITCH + lq))
                     : RD64 ? (TA ? (lhi * BM + wm * TM * 32 + l31) * 2
                                  : LIN ? lhi * PPA + 2 * (wm * TM * 32 + l31) : ((wm * TM * 32 + l31) * A_PITCH + 2 * lhi))
                            : (TA ? (lhi * A_PITCH + wm * TM * 32 + l31) : ((wm * TM * 32 + l31) * A_PITCH + lhi));
    const int b_rd = NARROW ? (TB ? (l15 * B_PITCH + lq) : (lq * B_PITCH + l15))
                     : RD64 ? (!TB ? (lhi * BN + wn * TN * 32 + l31) * 2
                                   : LIN ? lhi * PPB + 2 * (wn * TN * 32 + l31) : ((wn * TN * 32 + l31) * B_PITCH + 2 * lhi))
                            : (TB ? ((wn * TN * 32 + l31) * B_PITCH + lhi) : (lhi * B_PITCH + wn * TN * 32 + l31));
    // RD64: LDS distance of one k-step pair (4 reduction columns) / between the wave's 32-row (column) tiles
    constexpr int A_PSTEP = TA ? 4 * BM : (LIN ? 2 * PPA : 4);
    constexpr int A_PTILE = TA || LIN ? 64 : 32 * A_PITCH;
    constexpr int B_PSTEP = TB ? (LIN ? 2 * PPB : 4) : 4 * BN;
    constexpr int B_PTILE = TB && !LIN ? 32 * B_PITCH : 64;
    constexpr int A_K4STEP = TA ? 4 * A_PITCH : 4;     // NARROW: LDS distance of one k4 step / of the second 16 rows
    constexpr int A_TILE16 = TA ? 16 : 16 * A_PITCH;
    constexpr int B_K4STEP = TB ? 4 : 4 * B_PITCH;
    constexpr int A_KSTEP = TA ? 2 * A_PITCH : 2;      // LDS distance of one k2 step
    constexpr int A_TILE = TA ? 32 : 32 * A_PITCH;     // LDS distance between the wave's 32-row tiles
    constexpr int B_KSTEP = TB ? 2 : 2 * B_PITCH;
    constexpr int B_TILE = TB ? 32 * B_PITCH : 32;

    float ra[SPLIT ? 1 : A_PER_THREAD], rb[SPLIT ? 1 : B_PER_THREAD];
    // SPLIT staging.  An operand stored with its reduction dimension contiguous ([x][k]: A of a forward / data gradient,
    // B = W^T of a data gradient) is fetched as 16-byte quads: thread -> quad tid % SP_QPR (= tid & 3) of rows tid / SP_QPR + SP_RP i; one stored
    // with k strided ([k][x]: both operands of a filter gradient, the weights of a forward product) by dword loads along
    // x, a thread keeping KR consecutive k of ONE x -- the transpose happens in the registers.  Either way a thread ends
    // up with PAIRS of k-adjacent elements: SP_NPA + SP_NPB pairs per k-tile.
    constexpr int SP_QPR = SP_BK / 4;                     // quads per row of a k-contiguous operand tile
    constexpr int SP_RP = NT / SP_QPR;                    // rows one pass of quad loads covers
    constexpr int SA_N = BM / SP_RP, SB_N = BN >= SP_RP ? BN / SP_RP : 1;  // k-contiguous operand: quads per thread (BN <
                                                                          // rows per pass: the threads beyond BN rows idle)
    // k-strided operand of X columns: KR consecutive k per thread (all of the thread's share of the tile, at most 8),
    // (X / 64) * (SP_BK / KR) wave-wide units dealt to the waves; X = 32: 32 x per 32 lanes, 2 k each
    constexpr int SA_KR = SP_BK * BM / NT >= 8 ? 8 : SP_BK * BM / NT, SA_U = (BM / 64) * (SP_BK / SA_KR) / (NT / 64);
    constexpr int SB_KR = SP_BK * BN / NT >= 8 ? 8 : SP_BK * BN / NT;
    constexpr int SB_U = BN >= 64 ? (BN / 64) * (SP_BK / SB_KR) / (NT / 64) : 1;
    static_assert(!SPLIT || (SA_N >= 1 && SB_N >= 1 && SA_U >= 1 && SB_U >= 1 && SA_KR >= 2 && SB_KR >= 2 &&
                             (BN >= 64 || (NT == 256 && SB_KR == 2))), "split staging geometry");
    int ls = grp.seg_begin;
    const int s_end = grp.seg_begin + grp.seg_count;
    int lk = 0;
    hypel_seg_t seg;
    seg.a_off = 0; seg.b_off = 0; seg.k = 0; seg.reserved = 0;
    bool have = ls < s_end;
    if (have) {
        seg.a_off = tile.a_off0;
        seg.b_off = tile.b_off0;
        seg.k = tile.k0;
    }
    hypel_seg_t seg2;  // PAIR: partner of `seg` while (seg.k & HYPEL_SEG_PAIR_FLAG)
    seg2.a_off = 0; seg2.b_off = 0; seg2.k = 0; seg2.reserved = 0;
    if constexpr (PAIR) {
        static_assert(!TA && TB && !NARROW && !MULTI, "segment pairing: data-gradient operand layout only");
        if (have && (seg.k & HYPEL_SEG_PAIR_FLAG)) seg2 = segs[ls + 1];
    }

    // Staging loads are raw buffer loads: the tile base lives in a scalar descriptor, each thread keeps ONE
    // 32-bit offset per operand and the per-load row step is a scalar soffset.  An invalid element is fetched
    // at an out-of-range voffset, which the hardware returns as 0 (no masks, no 64-bit address registers).
    auto stage = [&](const float* base, int64_t ld, int row0, int col, int rows_valid, int cols_valid,
                     auto& regs, auto rstep_c, auto count_c) {
        constexpr int RSTEP = decltype(rstep_c)::value;
        constexpr int COUNT = decltype(count_c)::value;
        // wave-uniform by construction; say so, or hipcc wraps every buffer_load in a waterfall loop
        const int ld4 = __builtin_amdgcn_readfirstlane((int)ld * 4);
        // bytes up to the end of the last valid row
        const int span = __builtin_amdgcn_readfirstlane(
            rows_valid > 0 && cols_valid > 0 ? ((rows_valid - 1) * (int)ld + cols_valid) * 4 : 0);
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, span, 0x00020000);
        const int kOOB = 0x7fffffff;
        const int voff = col < cols_valid ? (row0 * (int)ld + col) * 4 : kOOB;
        // the row step as ONE running scalar (an opaque s_add between the loads) instead of COUNT precomputed soffsets
        // that stay live across the whole staging burst: up to 20 scalar registers less, the scalar spills of the
        // 128x32 data-gradient variant drop from 11 to 2 (step 6.87 -> 6.81 ms, same-box A/B)
        const int step = RSTEP * ld4;
        int so = 0;
        if (rows_valid >= COUNT * RSTEP) {
#pragma unroll
            for (int i = 0; i < COUNT; ++i) {
                regs[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, so, 0));
                asm volatile("s_add_u32 %0, %0, %1" : "+s"(so) : "s"(step) : "scc");
            }
        } else {
#pragma unroll
            for (int i = 0; i < COUNT; ++i) {
                const int v = (row0 + i * RSTEP) < rows_valid ? voff : kOOB;
                regs[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, v, so, 0));
                asm volatile("s_add_u32 %0, %0, %1" : "+s"(so) : "s"(step) : "scc");
            }
        }
    };

    // PAIR: the k columns [0, 16) of the tile come from (offX, kx), [16, 32) from (offY, ky): ONE descriptor based at
    // the lower of the two addresses, the other one's distance goes into the per-lane offset (both operands of a data
    // gradient live in one allocation each, < 2 GB apart); rows = the non-reduction dimension.
    auto stage_pair = [&](const float* basep, int64_t offx, int64_t offy, int kx, int ky, int64_t ld, int row0, int col,
                          int rows_valid, auto& regs, auto rstep_c, auto count_c) {
        constexpr int RSTEP = decltype(rstep_c)::value;
        constexpr int COUNT = decltype(count_c)::value;
        const int ld4 = __builtin_amdgcn_readfirstlane((int)ld * 4);
        const int64_t lo = offx < offy ? offx : offy;
        const int dx = __builtin_amdgcn_readfirstlane((int)((offx - lo) * 4));
        const int dy = __builtin_amdgcn_readfirstlane((int)((offy - lo) * 4));
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(basep + lo), 0, 0x7ffffff0, 0x00020000);
        const int kOOB = 0x7fffffff;
        const int c2 = col - 16;
        const int voff = col < 16 ? (col < kx ? dx + (row0 * (int)ld + col) * 4 : kOOB)
                                  : (c2 < ky ? dy + (row0 * (int)ld + c2) * 4 : kOOB);
        const int step = RSTEP * ld4;
        int so = 0;
#pragma unroll
        for (int i = 0; i < COUNT; ++i) {
            const int v = (row0 + i * RSTEP) < rows_valid ? voff : kOOB;
            regs[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, v, so, 0));
            asm volatile("s_add_u32 %0, %0, %1" : "+s"(so) : "s"(step) : "scc");
        }
    };

    // SPLIT staging loads are branch-free: every fetch issues the SAME number of buffer loads on every path, so that hipcc's
    // s_waitcnt insertion can count exactly how many younger loads may stay in flight (any branch that changes the count
    // makes it fall back to vmcnt(0), which empties the whole prefetch ring).  Validity rides in the descriptor: rows at or
    // beyond rows_valid (and everything, when the stream has ended: rows_valid = 0) lie beyond `span` and come back as 0;
    // columns at or beyond the valid count are fetched at an out-of-range per-lane offset.
    // Operand with its reduction dimension contiguous: [rows][16 k] as 16-byte quads (dword-aligned addresses suffice for
    // buffer_load_dwordx4; each dword is range-checked on its own).  A k-tile that ends inside a quad (segment k not a
    // multiple of 4) reads the row's next columns with it: split_quad() masks them.
    [[maybe_unused]] auto stage_q = [&](const float* base, int64_t ld, int span, int k_valid, auto& regs, auto count_c) {
        constexpr int COUNT = decltype(count_c)::value;
        const int ld4 = __builtin_amdgcn_readfirstlane((int)ld * 4);
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, span, 0x00020000);
        const int q4 = (tid % SP_QPR) * 4;
        const int voff = q4 < k_valid ? ((tid / SP_QPR) * (int)ld + q4) * 4 : 0x7fffffff;
        const int step = SP_RP * ld4;
        int so = 0;
#pragma unroll
        for (int i = 0; i < COUNT; ++i) {
            regs[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, so, 0));
            asm volatile("s_add_u32 %0, %0, %1" : "+s"(so) : "s"(step) : "scc");
        }
    };
    // Operand with its reduction dimension strided: [16 k][XW x], dword loads along x (fully coalesced), a thread keeps KR
    // consecutive k of one x.  XW >= 64: unit = wave * U + u -> x = (unit % (XW / 64)) * 64 + lane, k = KR * (unit /
    // (XW / 64)) + r; XW = 32 (256 threads): x = tid & 31, k = 2 * (tid >> 5) + r
    [[maybe_unused]] auto stage_s = [&](const float* base, int64_t ld, int span, int x_valid, auto& regs, auto xw_c,
                                        auto kr_c, auto u_c) {
        constexpr int XW = decltype(xw_c)::value, KR = decltype(kr_c)::value, U = decltype(u_c)::value;
        const int ld4 = __builtin_amdgcn_readfirstlane((int)ld * 4);
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, span, 0x00020000);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int x, kr0;
            if constexpr (XW >= 64) {
                const int unit = wave * U + u;
                x = (unit % (XW / 64)) * 64 + lane;
                kr0 = (unit / (XW / 64)) * KR;
            } else {
                x = tid & 31;
                kr0 = (tid >> 5) * KR;
            }
            const int voff = x < x_valid ? (kr0 * (int)ld + x) * 4 : 0x7fffffff;
            int so = 0;
#pragma unroll
            for (int r = 0; r < KR; ++r) {
                regs[u][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, so, 0));
                asm volatile("s_add_u32 %0, %0, %1" : "+s"(so) : "s"(ld4) : "scc");
            }
        }
    };

    auto load_tiles = [&](const hypel_seg_t& sg, int k0) {
        if constexpr (PAIR) {
            if (sg.k & HYPEL_SEG_PAIR_FLAG) {
                const int kx = sg.k & ~HYPEL_SEG_PAIR_FLAG, ky = seg2.k;
                stage_pair(A + (int64_t)m0 * lda, sg.a_off, seg2.a_off, kx, ky, lda, a_row0, a_col, min(BM, rows_left),
                           ra, std::integral_constant<int, A_RSTEP>{}, std::integral_constant<int, A_PER_THREAD>{});
                stage_pair(B + (int64_t)n0 * ldb, sg.b_off, seg2.b_off, kx, ky, ldb, b_row0, b_col, min(BN, cols_left),
                           rb, std::integral_constant<int, B_RSTEP>{}, std::integral_constant<int, B_PER_THREAD>{});
                return;
            }
        }
        const int k_left = min(BK, sg.k - k0);
        const int m_left = min(BM, rows_left);
        const int n_left = min(BN, cols_left);
        if (!TA)  // rows = output rows (m), cols = k
            stage(A + sg.a_off + (int64_t)m0 * lda + k0, lda, a_row0, a_col, m_left, k_left, ra,
                  std::integral_constant<int, A_RSTEP>{}, std::integral_constant<int, A_PER_THREAD>{});
        else  // rows = k (reduction rows), cols = output rows (m)
            stage(A + sg.a_off + (int64_t)k0 * lda + m0, lda, a_row0, a_col, k_left, m_left, ra,
                  std::integral_constant<int, A_RSTEP>{}, std::integral_constant<int, A_PER_THREAD>{});
        if (!TB)  // rows = k, cols = n
            stage(B + sg.b_off + (int64_t)k0 * ldb + n0, ldb, b_row0, b_col, k_left, n_left, rb,
                  std::integral_constant<int, B_RSTEP>{}, std::integral_constant<int, B_PER_THREAD>{});
        else  // rows = n, cols = k
            stage(B + sg.b_off + (int64_t)n0 * ldb + k0, ldb, b_row0, b_col, n_left, k_left, rb,
                  std::integral_constant<int, B_RSTEP>{}, std::integral_constant<int, B_PER_THREAD>{});
    };

    // Epilogue operands that only depend on the block's position -- the shortcut gradient's column ranges and the bias --
    // are requested BEFORE the k loop: at the end of the block they would be one more dependent round trip in front of
    // the gather passes (HYPEL_GEMM_HOIST_EPI=0: load them in the epilogue).
    [[maybe_unused]] int h_o0[TN], h_o1[TN];
    [[maybe_unused]] float h_bv[TN];
    constexpr bool HOIST = HYPEL_GEMM_HOIST_EPI && !NARROW && !TA && TM * TN == 1;  // wider tiles: 6 more live
    if constexpr (HOIST) {                                                                 // registers cost a wave per SIMD
        const int bias_c0 = bias ? (int)(grp.c_off % ldc) + n0 : 0;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = (wn * TN + j) * 32 + l31;
            const bool okc = col < cols_left;
            h_bv[j] = bias && okc ? bias[bias_c0 + col] : 0.0f;
            h_o0[j] = n0 + col;
            h_o1[j] = n0 + col + 1;
            if (res && res_start && okc) {
                h_o0[j] = res_start[n0 + col];
                h_o1[j] = res_start[n0 + col + 1];
            }
        }
    }
    if constexpr (SPLIT) {
        // ---- split-operand pipeline ----
        // The fp32 -> 3 x bf16 split costs ~5 VALU operations per element.  Done between two barriers (load -> split -> LDS
        // -> MFMA, as the fp32 kernel stages) it serialises with an MFMA phase that is 2.67x shorter here: that form ran at
        // the fp32 kernel's own speed (round-5 notes).  And with two or three blocks per CU little but a block's own loads
        // covers the memory latency: at the MFMA rate a CU consumes ~21 bytes per cycle, i.e. ~90 KB must be in flight
        // across ~2 us.  So the k-tiles (16 reduction columns each) run through a software pipeline inside every wave:
        //     LDS[t & 1]             tile t feeds the MFMAs
        //     raw[(t + 1) % 4]       tile t + 1 is split and stored into LDS[(t + 1) & 1] UNDER those MFMAs, chunk by chunk
        //     raw[(t + 2 .. 4) % 4]  the loads of tiles t + 2 .. t + 4 are in flight (tile t + 4 goes out at the phase start)
        // One barrier per k-tile.
        struct Raw {
            f32x4 qa[!TA ? SA_N : 1], qb[TB ? SB_N : 1];
            float sa[TA ? SA_U : 1][SA_KR], sb[!TB ? SB_U : 1][SB_KR];
            int kv;  // valid reduction columns of the tile (scalar)
        };
        Raw raw0, raw1, raw2, raw3;
        auto raw_of = [&](auto i_c) -> Raw& {
            constexpr int I = decltype(i_c)::value & 3;
            if constexpr (I == 0) return raw0;
            else if constexpr (I == 1) return raw1;
            else if constexpr (I == 2) return raw2;
            else return raw3;
        };
        const int m_left = min(BM, rows_left), n_left = min(BN, cols_left);
        // cursor of the load stream: (seg, lk) = the next k-tile, nseg = the record behind seg (always requested one fetch
        // ahead, clamped at the group's last); have = the stream has not ended.  No branches (see stage_q).
        // The operand addresses of the next k-tile are kept as two running 64-bit scalars (a_cur, b_cur): + one tile's
        // distance per fetch, or the next segment's start when the segment ends -- ~30 scalar instructions less per
        // k-tile than rebuilding base + offset + k * ld (the scalar port issues one instruction per SIMD and 4 cycles;
        // at ~95 per wave and k-tile it was as busy as the matrix pipe).
        // (a group without segments -- a filter-gradient split with no reduction rows still owns its slab -- has s_end ==
        // seg_begin: the clamp must not step in front of the table)
        hypel_seg_t nseg = segs[max(grp.seg_begin, min(ls + 1, s_end - 1))];
        int n_fetched = 0;  // real k-tiles requested so far: tile t + 1 exists iff t + 1 < n_fetched
        const uint64_t a_base0 = (uint64_t)(A + (!TA ? (int64_t)m0 * lda : (int64_t)m0));
        const uint64_t b_base0 = (uint64_t)(B + (TB ? (int64_t)n0 * ldb : (int64_t)n0));
        const uint64_t a_step = !TA ? (uint64_t)(4 * SP_BK) : (uint64_t)lda * (4 * SP_BK);
        const uint64_t b_step = TB ? (uint64_t)(4 * SP_BK) : (uint64_t)ldb * (4 * SP_BK);
        uint64_t a_cur = a_base0 + (uint64_t)seg.a_off * 4, b_cur = b_base0 + (uint64_t)seg.b_off * 4;
        int seg_k = seg.k;
        // descriptor spans (bytes up to the end of the last valid row) of a k-tile with kl valid reduction columns
        auto span_q = [&](int rows_valid, int64_t ld, int kl) {
            return __builtin_amdgcn_readfirstlane(rows_valid > 0 && kl > 0 ? ((rows_valid - 1) * (int)ld + kl) * 4 : 0);
        };
        auto span_s = [&](int x_valid, int64_t ld, int kl) {
            return __builtin_amdgcn_readfirstlane(x_valid > 0 && kl > 0 ? ((kl - 1) * (int)ld + x_valid) * 4 : 0);
        };
        const int span_a_full = !TA ? span_q(m_left, lda, SP_BK) : span_s(m_left, lda, SP_BK);
        const int span_b_full = TB ? span_q(n_left, ldb, SP_BK) : span_s(n_left, ldb, SP_BK);
        auto fetch = [&](Raw& r) {
            // what this fetch loads, and the cursor behind it -- ONE uniform branch of scalar work, the loads follow the
            // join (a branch that contains loads makes hipcc's s_waitcnt insertion drain the whole prefetch ring)
            const uint64_t a_ptr = a_cur, b_ptr = b_cur;
            int k_left, span_a, span_b;
            if (have && seg_k - lk > SP_BK) {  // a whole tile that is not its segment's last: constants, the cursor steps
                k_left = SP_BK;
                span_a = span_a_full;
                span_b = span_b_full;
                n_fetched += 1;
                lk += SP_BK;
                a_cur += a_step;
                b_cur += b_step;
            } else {
                k_left = have ? min(SP_BK, seg_k - lk) : 0;  // 0: every load of this fetch is out of range
                span_a = !TA ? span_q(m_left, lda, k_left) : span_s(m_left, lda, k_left);
                span_b = TB ? span_q(n_left, ldb, k_left) : span_s(n_left, ldb, k_left);
                n_fetched += have ? 1 : 0;
                lk += SP_BK;
                const bool adv = have && lk >= seg_k;
                const uint64_t a_nxt = a_base0 + (uint64_t)nseg.a_off * 4, b_nxt = b_base0 + (uint64_t)nseg.b_off * 4;
                a_cur = adv ? a_nxt : a_cur + a_step;
                b_cur = adv ? b_nxt : b_cur + b_step;
                seg_k = adv ? nseg.k : seg_k;
                lk = adv ? 0 : lk;
                ls += adv ? 1 : 0;
                have = ls < s_end;
                nseg = segs[max(grp.seg_begin, min(ls + 1, s_end - 1))];
            }
            r.kv = k_left;
            if constexpr (!TA)
                stage_q(reinterpret_cast<const float*>(a_ptr), lda, span_a, k_left, r.qa, std::integral_constant<int, SA_N>{});
            else
                stage_s(reinterpret_cast<const float*>(a_ptr), lda, span_a, m_left, r.sa, std::integral_constant<int, BM>{},
                        std::integral_constant<int, SA_KR>{}, std::integral_constant<int, SA_U>{});
            if constexpr (TB)
                stage_q(reinterpret_cast<const float*>(b_ptr), ldb, span_b, k_left, r.qb, std::integral_constant<int, SB_N>{});
            else
                stage_s(reinterpret_cast<const float*>(b_ptr), ldb, span_b, n_left, r.sb, std::integral_constant<int, BN>{},
                        std::integral_constant<int, SB_KR>{}, std::integral_constant<int, SB_U>{});
        };
        // chunk c of an operand's raw registers (a quad of a k-contiguous operand, a KR-run of a k-strided one) -> split
        // -> the three planes of LDS buffer `buf`: plane p of element (x, k) at p * PLANE + x * 80 + 2 k bytes
        constexpr int SP_NCA = !TA ? SA_N : SA_U, SP_NCB = TB ? SB_N : SB_U;
        auto chunk_q = [&](f32x4 v, int kv, float* img, int plane, int i, auto rows_c) {
            constexpr int ROWS = decltype(rows_c)::value;
            if (ROWS < SP_RP && tid / SP_QPR >= ROWS) return;  // (an operand with fewer rows than one pass of the block covers)
            if (kv & 3) {  // the tile ends inside a quad: the elements behind its end are the row's next columns, not zeros
                asm volatile("" ::: "memory");  // (keeps this a rare uniform branch: if-converted it is 4 selects per quad)
                const int q4 = (tid % SP_QPR) * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = q4 + e < kv ? v[e] : 0.0f;
            }
            unsigned h0, h1, m0_, m1_, l0, l1;
            hypel_split2(v[0], v[1], h0, m0_, l0);
            hypel_split2(v[2], v[3], h1, m1_, l1);
            float* o = img + (tid / SP_QPR + SP_RP * i) * SP_PITCH + 2 * (tid % SP_QPR);
            *reinterpret_cast<u32x2*>(o) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(o + plane) = u32x2{m0_, m1_};
            *reinterpret_cast<u32x2*>(o + 2 * plane) = u32x2{l0, l1};
        };
        auto chunk_s = [&](const auto& v, float* img, int plane, int u, auto xw_c, auto kr_c, auto u_c) {
            constexpr int XW = decltype(xw_c)::value, KR = decltype(kr_c)::value, U = decltype(u_c)::value;
            int x, kg;
            if constexpr (XW >= 64) {
                const int unit = wave * U + u;
                x = (unit % (XW / 64)) * 64 + lane;
                kg = unit / (XW / 64);
            } else {
                x = tid & 31;
                kg = tid >> 5;
            }
            unsigned h[KR / 2], m[KR / 2], l[KR / 2];
#pragma unroll
            for (int e = 0; e < KR / 2; ++e) hypel_split2(v[2 * e], v[2 * e + 1], h[e], m[e], l[e]);
            if constexpr (KR == 8) {
                float* o = img + x * SP_PITCH + 4 * kg;
                *reinterpret_cast<u32x4*>(o) = u32x4{h[0], h[1], h[2], h[3]};
                *reinterpret_cast<u32x4*>(o + plane) = u32x4{m[0], m[1], m[2], m[3]};
                *reinterpret_cast<u32x4*>(o + 2 * plane) = u32x4{l[0], l[1], l[2], l[3]};
            } else if constexpr (KR == 2) {
                unsigned* o = reinterpret_cast<unsigned*>(img + x * SP_PITCH + kg);
                o[0] = h[0];
                o[plane] = m[0];
                o[2 * plane] = l[0];
            } else {
                float* o = img + x * SP_PITCH + 2 * kg;
                *reinterpret_cast<u32x2*>(o) = u32x2{h[0], h[1]};
                *reinterpret_cast<u32x2*>(o + plane) = u32x2{m[0], m[1]};
                *reinterpret_cast<u32x2*>(o + 2 * plane) = u32x2{l[0], l[1]};
            }
        };
        auto chunk_a = [&](const Raw& r, int c, int buf) {
            if constexpr (!TA) chunk_q(r.qa[c], r.kv, As + buf * SP_BUF, SP_PLANE_A, c, std::integral_constant<int, BM>{});
            else chunk_s(r.sa[c], As + buf * SP_BUF, SP_PLANE_A, c, std::integral_constant<int, BM>{},
                         std::integral_constant<int, SA_KR>{}, std::integral_constant<int, SA_U>{});
        };
        auto chunk_b = [&](const Raw& r, int c, int buf) {
            if constexpr (TB) chunk_q(r.qb[c], r.kv, Bs + buf * SP_BUF, SP_PLANE_B, c, std::integral_constant<int, BN>{});
            else chunk_s(r.sb[c], Bs + buf * SP_BUF, SP_PLANE_B, c, std::integral_constant<int, BN>{},
                         std::integral_constant<int, SB_KR>{}, std::integral_constant<int, SB_U>{});
        };
        // One 32x32x16 step per k-tile: lane half h supplies k = 8 h .. 8 h + 7 of its row (column) from each plane with one
        // ds_read_b128; six products per accumulator tile, smallest first.  WITH: the chunks of the next tile (raw set
        // P + 1) are split and stored in between, spread over the six groups of MFMAs
        const int sa_rd = (wm * TM * 32 + l31) * SP_PITCH + 4 * lhi;
        const int sb_rd = (wn * TN * 32 + l31) * SP_PITCH + 4 * lhi;
        auto phase = [&](auto p_c, auto with_c, auto mfma_c) {
            constexpr int P = decltype(p_c)::value, CUR = P & 1;
            constexpr bool WITH = decltype(with_c)::value, MF = decltype(mfma_c)::value;
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
            constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
            constexpr int NC = SP_NCA + SP_NCB;
            const Raw& rs = raw_of(std::integral_constant<int, P + 1>{});
            bf16x8 a[TM][3], b[TN][3];
            if constexpr (MF) {
#pragma unroll
                for (int p = 0; p < 3; ++p) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        a[i][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(
                            &As[CUR * SP_BUF + p * SP_PLANE_A + sa_rd + i * 32 * SP_PITCH]));
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        b[j][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(
                            &Bs[CUR * SP_BUF + p * SP_PLANE_B + sb_rd + j * 32 * SP_PITCH]));
                }
            }
            hypel_for_range<0>([&](auto g_c) {
                constexpr int Gi = decltype(g_c)::value;
                if constexpr (MF) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA[Gi]], b[j][PB[Gi]], acc[i][j], 0, 0, 0);
                }
                if constexpr (WITH) {
#pragma unroll
                    for (int c = Gi * NC / 6; c < (Gi + 1) * NC / 6; ++c) {
                        if (c < SP_NCA) chunk_a(rs, c, CUR ^ 1);
                        else chunk_b(rs, c - SP_NCA, CUR ^ 1);
                    }
                }
            }, std::make_integer_sequence<int, 6>{});
        };
        // pipeline stage P (= t mod 4) of a tile that HAS a successor: fetch tile t + 4 into the set tile t left, run the
        // phase, one barrier (tile t is done with, tile t + 1 is complete in the other buffer)
        auto stage_of = [&](auto p_c) {
            constexpr int P = decltype(p_c)::value;
            fetch(raw_of(std::integral_constant<int, P>{}));
            if (any_act) {
                __builtin_amdgcn_s_setprio(1);
                phase(p_c, std::true_type{}, std::true_type{});
                __builtin_amdgcn_s_setprio(0);
            } else {
                phase(p_c, std::true_type{}, std::false_type{});
            }
            __syncthreads();
        };
        if (have) {
            fetch(raw0);
            fetch(raw1);
            fetch(raw2);
            fetch(raw3);
#pragma unroll
            for (int c = 0; c < SP_NCA; ++c) chunk_a(raw0, c, 0);
#pragma unroll
            for (int c = 0; c < SP_NCB; ++c) chunk_b(raw0, c, 0);
            __syncthreads();
            int t = 0;
            while (true) {
                if (!(t + 1 < n_fetched)) break;
                stage_of(std::integral_constant<int, 0>{});
                if (!(t + 2 < n_fetched)) { t += 1; break; }
                stage_of(std::integral_constant<int, 1>{});
                if (!(t + 3 < n_fetched)) { t += 2; break; }
                stage_of(std::integral_constant<int, 2>{});
                if (!(t + 4 < n_fetched)) { t += 3; break; }
                stage_of(std::integral_constant<int, 3>{});
                t += 4;
            }
            if (any_act) {  // the last tile: nothing left to split
                __builtin_amdgcn_s_setprio(1);
                if (t & 1) phase(std::integral_constant<int, 1>{}, std::false_type{}, std::true_type{});
                else phase(std::integral_constant<int, 0>{}, std::false_type{}, std::true_type{});
                __builtin_amdgcn_s_setprio(0);
            }
        }
    } else {
        if (have) load_tiles(seg, lk);

        while (have) {
            const bool paired = PAIR && (seg.k & HYPEL_SEG_PAIR_FLAG);
            const int kvalid = paired ? 16 + seg2.k : min(BK, seg.k - lk);
            // advance the (segment, k) cursor NOW and request the next segment record before the LDS hand-over below: behind
            // the second barrier the scalar load sat directly in front of the address arithmetic of the next tile's loads
            // -- one exposed round trip per segment, i.e. per k-tile in the data gradients of the multi-kernel levels
            // (segments of 15-60 reduction columns)
            lk += BK;
            if (paired || lk >= seg.k) {
                ls += paired ? 2 : 1;
                lk = 0;
                if (ls < s_end) {
                    seg = segs[ls];
                    if constexpr (PAIR)
                        if (seg.k & HYPEL_SEG_PAIR_FLAG) seg2 = segs[ls + 1];
                }
            }
            have = ls < s_end;
            __syncthreads();  // previous tile's MFMAs are done reading LDS
            if constexpr (LIN && !TA) {  // [m][k] tile transposed into the pair-interleaved image
#pragma unroll
                for (int i = 0; i < A_PER_THREAD; ++i)
                    As[(a_col >> 1) * PPA + 2 * (a_row0 + i * A_RSTEP) + (a_col & 1)] = ra[i];
            } else if constexpr (RD64 && TA) {  // [k][m] image, column pairs interleaved; the row step is even
#pragma unroll
                for (int i = 0; i < A_PER_THREAD; ++i)
                    As[(((a_row0 >> 1) + i * (A_RSTEP / 2)) * BM + a_col) * 2 + (a_row0 & 1)] = ra[i];
            } else {
#pragma unroll
                for (int i = 0; i < A_PER_THREAD; ++i) As[(a_row0 + i * A_RSTEP) * A_PITCH + a_col] = ra[i];
            }
            if constexpr (LIN && TB) {
#pragma unroll
                for (int i = 0; i < B_PER_THREAD; ++i)
                    Bs[(b_col >> 1) * PPB + 2 * (b_row0 + i * B_RSTEP) + (b_col & 1)] = rb[i];
            } else if constexpr (RD64 && !TB) {  // [k][n] image, column pairs interleaved
#pragma unroll
                for (int i = 0; i < B_PER_THREAD; ++i)
                    if (B_THREADS == 256 || b_stager)
                        Bs[(((b_row0 >> 1) + i * (B_RSTEP / 2)) * BN + b_col) * 2 + (b_row0 & 1)] = rb[i];
            } else {
#pragma unroll
                for (int i = 0; i < B_PER_THREAD; ++i)
                    if (B_THREADS == 256 || b_stager) Bs[(b_row0 + i * B_RSTEP) * B_PITCH + b_col] = rb[i];
            }
            __syncthreads();

            // put the next tile's loads in flight
            if (have) load_tiles(seg, lk);

            // One straight-line path (no per-tile branches, so the accumulators stay put and the compiler
            // pipelines the ds_reads under the MFMAs).  Ragged shapes rely on the zero-filled LDS image; the
            // only skips are wave-uniform: a wave with no active tile, and the second half of a short k-tile.
            if (any_act) {
                __builtin_amdgcn_s_setprio(1);  // favour the wave that is in its MFMA phase over the ones staging tiles
                if constexpr (NARROW && NT16 > 1) {
                    // one straight-line variant per number of active 16-column tiles (a wave-uniform switch: the
                    // accumulators stay where they are, no MFMA is exec-masked)
                    auto phase = [&](auto nact_c) {
                        constexpr int NACT = decltype(nact_c)::value;
#pragma unroll
                        for (int q = 0; q < BK / CHUNK; ++q) {
                            if (q > 0 && kvalid <= q * CHUNK) break;
#pragma unroll
                            for (int k4 = q * (CHUNK / 4); k4 < (q + 1) * (CHUNK / 4); ++k4) {
                                const float a0 = As[a_rd + k4 * A_K4STEP];
                                const float a1 = As[a_rd + A_TILE16 + k4 * A_K4STEP];
                                float b[NACT];
#pragma unroll
                                for (int j = 0; j < NACT; ++j) b[j] = Bs[b_rd + 16 * j + k4 * B_K4STEP];
#pragma unroll
                                for (int j = 0; j < NACT; ++j) {
                                    acc16[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b[j], acc16[j], 0, 0, 0);
                                    acc16[NT16 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b[j], acc16[NT16 + j], 0, 0, 0);
                                }
                            }
                        }
                    };
                    if (tn_act >= 4) phase(std::integral_constant<int, 4>{});
                    else if (tn_act == 3) phase(std::integral_constant<int, 3>{});
                    else if (tn_act == 2) phase(std::integral_constant<int, 2>{});
                    else phase(std::integral_constant<int, 1>{});
                } else if constexpr (NARROW) {
                    // CHUNK reduction columns at a time, as below
#pragma unroll
                    for (int q = 0; q < BK / CHUNK; ++q) {
                        if (q > 0 && kvalid <= q * CHUNK) break;
#pragma unroll
                        for (int k4 = q * (CHUNK / 4); k4 < (q + 1) * (CHUNK / 4); ++k4) {
                            const float b = Bs[b_rd + k4 * B_K4STEP];
                            const float a0 = As[a_rd + k4 * A_K4STEP];
                            const float a1 = As[a_rd + A_TILE16 + k4 * A_K4STEP];
                            acc16[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b, acc16[0], 0, 0, 0);
                            acc16[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, acc16[1], 0, 0, 0);
                        }
                    }
                } else if constexpr (RD64) {
                    // CHUNK reduction columns (CHUNK / 4 step pairs) at a time, skipped beyond kvalid as below.  NACT = active
                    // 32-column accumulator tiles (a prefix): forward 128x64 blocks of a merged level serve groups of 30 - 120
                    // columns and run the one-tile variant where the second tile lies outside the group (wave-uniform switch)
                    auto phase = [&](auto nact_c) {
                        constexpr int NACT = decltype(nact_c)::value;
#pragma unroll
                        for (int q = 0; q < BK / CHUNK; ++q) {
                            if (q > 0 && kvalid <= q * CHUNK) break;
#pragma unroll
                            for (int t = q * (CHUNK / 4); t < (q + 1) * (CHUNK / 4); ++t) {
                                float2 a[TM], b[NACT];
#pragma unroll
                                for (int i = 0; i < TM; ++i)
                                    a[i] = *reinterpret_cast<const float2*>(&As[a_rd + i * A_PTILE + t * A_PSTEP]);
#pragma unroll
                                for (int j = 0; j < NACT; ++j)
                                    b[j] = *reinterpret_cast<const float2*>(&Bs[b_rd + j * B_PTILE + t * B_PSTEP]);
#pragma unroll
                                for (int i = 0; i < TM; ++i)
#pragma unroll
                                    for (int j = 0; j < NACT; ++j)
                                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
                                for (int i = 0; i < TM; ++i)
#pragma unroll
                                    for (int j = 0; j < NACT; ++j)
                                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                            }
                        }
                    };
                    constexpr bool PREFIX = VARN && TN == 2 && TM == 1 && !TA && !TB && !MULTI && !PAIR;
                    if constexpr (PREFIX) {
                        if (tn_act >= 2) phase(std::integral_constant<int, 2>{});
                        else phase(std::integral_constant<int, 1>{});
                    } else {
                        phase(std::integral_constant<int, TN>{});
                    }
                } else {
                // 16 reduction columns (8 MFMA k-steps) at a time; the chunks of a short k-tile beyond kvalid are skipped
                    // by a wave-uniform branch
        #pragma unroll
                    for (int q = 0; q < BK / CHUNK; ++q) {
                        if (q > 0 && kvalid <= q * CHUNK) break;
        #pragma unroll
                        for (int k2 = q * (CHUNK / 2); k2 < (q + 1) * (CHUNK / 2); ++k2) {
                            float a[TM], b[TN];
        #pragma unroll
                            for (int i = 0; i < TM; ++i) a[i] = As[a_rd + i * A_TILE + k2 * A_KSTEP];
        #pragma unroll
                            for (int j = 0; j < TN; ++j) b[j] = Bs[b_rd + j * B_TILE + k2 * B_KSTEP];
        #pragma unroll
                            for (int i = 0; i < TM; ++i)
        #pragma unroll
                                for (int j = 0; j < TN; ++j)
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
                        }
                    }
                }
                __builtin_amdgcn_s_setprio(0);
            }
        }
    }

#if HYPEL_GEMM_CLK
    if (tid == 0 && clk_dbg) {
        long long* dbg = (long long*)clk_dbg + 2 * (size_t)blockIdx.x;
#if HYPEL_GEMM_CLK == 2  // absolute 100 MHz times of the block's start and end: occupancy timeline of a launch
        dbg[0] = wall0;
        dbg[1] = wall_clock64();
#else
        dbg[0] = clock64() - clk0;
        dbg[1] = wall_clock64() - wall0;
#endif
    }
    bias = nullptr;
#endif
    if constexpr (ACT) {
        // act(product + bias): the bias joins the accumulators here (the stores below then add none); no read-modify-write,
        // no shortcut gather, no statistics with this epilogue (the dispatcher checks)
        const float alpha = act_idx == 1 ? 0.1f : act_idx == 2 ? 0.18f : act_idx == 3 ? 0.2f : act_idx == 4 ? 0.01f : 0.0f;
        const int bc0 = bias ? (int)(grp.c_off % ldc) + n0 : 0;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = (wn * TN + j) * 32 + l31;
            float bv = 0.0f;
            if constexpr (HOIST) {
                bv = h_bv[j];
                h_bv[j] = 0.0f;
            } else {
                bv = bias && col < cols_left ? bias[bc0 + col] : 0.0f;
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float v = acc[i][j][e] + bv;
                    acc[i][j][e] = v > 0.0f ? v : alpha * v;
                }
        }
        bias = nullptr;
    }
    // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5) ----
    float* cbase = C + grp.c_off + (int64_t)m0 * ldc + n0;
    // bias is indexed by the absolute output column: groups of a merged level start at channel offsets
    const int bias_col0 = bias ? (int)(grp.c_off % ldc) + n0 : 0;
    // residual-gradient addend (data gradient of a layer whose input is also its shortcut source): output element
    // (row, c) additionally receives sum_{o in [res_start[c], res_start[c+1])} res[row][o] -- the transpose of the
    // monotone channel map of scale_in_to_out -- from the row-aligned matrix `res` (same pixel-major row order as C)
    const float* rbase = res ? res + (grp.c_off / ldc + m0) * ldr : nullptr;
    // one output element: bias, optional read-modify-write, optional shortcut-gradient gather
    auto put = [&](int row, int col, float accv, float bv, int o0, int o1) -> float {
        float v = 0.0f;
        if (row < rows_left) {
            float* p = cbase + (int64_t)row * ldc + col;
            v = accv + bv;
            if (accumulate) v += *p;
            if (res) {
                const float* rr = rbase + (int64_t)row * ldr;
                for (int o = o0; o < o1; ++o) v += rr[o];
            }
            *p = v;
        }
        return v;
    };
    // Batch-norm statistics in the epilogue (tf_slim.batch_norm's batch moments, HYPELCNNModel.py:37,43-44): the tile's
    // per-column (mean, sum of squared deviations) over its valid rows, straight from the accumulators -- the
    // statistics pass that re-read the whole GEMM output (hypel_col_stats_partial) is gone.  Two passes over the
    // registers per 32-row slab, Chan's merge of the block's slabs through LDS in slab order; one (mean, M2) pair per
    // 128-row tile = exactly the chunk format hypel_bn_finalize merges (chunk_rows = 128).  Single-group launches
    // only (chunk = m0 / 128), no accumulate.
    if constexpr (!NARROW && !MULTI) {
        if (stats) {
            constexpr int SLABS = WM * TM;
            float* red = lds;  // [SLABS][BN][2], after the last MFMA phase
            __syncthreads();
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int slab = wm * TM + i;
                    const int nw = min(32, max(0, rows_left - slab * 32));
                    float s1 = 0.0f;
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (slab * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi < rows_left) s1 += acc[i][j][e];
                    s1 += __shfl_xor(s1, 32, 64);
                    const float mean = nw > 0 ? s1 / (float)nw : 0.0f;
                    float m2 = 0.0f;
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (slab * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi < rows_left) {
                            const float d = acc[i][j][e] - mean;
                            m2 += d * d;
                        }
                    m2 += __shfl_xor(m2, 32, 64);
                    if (lhi == 0) {
                        const int cl = (wn * TN + j) * 32 + l31;
                        red[(slab * BN + cl) * 2 + 0] = mean;
                        red[(slab * BN + cl) * 2 + 1] = m2;
                    }
                }
            __syncthreads();
            if (tid < BN && tid < cols_left) {
                float na = 0.0f, mean_a = 0.0f, m2_a = 0.0f;
#pragma unroll
                for (int sl = 0; sl < SLABS; ++sl) {
                    const float nb_ = (float)min(32, max(0, rows_left - sl * 32));
                    if (nb_ > 0.0f) {
                        const float mb = red[(sl * BN + tid) * 2 + 0], m2b = red[(sl * BN + tid) * 2 + 1];
                        const float d = mb - mean_a, nab = na + nb_;
                        mean_a += d * nb_ / nab;
                        m2_a += m2b + d * d * na * nb_ / nab;
                        na = nab;
                    }
                }
                if (bias) mean_a += bias[bias_col0 + tid];
                const int64_t chunk = m0 / BM;
                stats[(chunk * 2 + 0) * n + n0 + tid] = mean_a;
                stats[(chunk * 2 + 1) * n + n0 + tid] = m2_a;
            }
        }
    }
    if constexpr (NARROW) {
        // C/D layout of 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + e
#pragma unroll
        for (int jt = 0; jt < NT16; ++jt) {
        const int col = 16 * jt + l15;
        if (row_act[0] && col < cols_left) {
            const float bv = bias ? bias[bias_col0 + col] : 0.0f;
            int o0 = n0 + col, o1 = n0 + col + 1;
            if (res && res_start) {
                o0 = res_start[n0 + col];
                o1 = res_start[n0 + col + 1];
            }
            if (HYPEL_GEMM_BATCHED_EPILOGUE > 1 && !res && rows_left >= wm * 32 + 32) {
                // full 32-row slab: raw buffer accesses, one per-lane offset, rows stepped by a scalar (see below)
                const int ldc4 = __builtin_amdgcn_readfirstlane((int)ldc * 4);
                const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc((void*)cbase, 0, 0x7ffffff0, 0x00020000);
                const int cvo = ((wm * 32 + 4 * lq) * (int)ldc + col) * 4;
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = acc16[(q >> 2) * NT16 + jt][q & 3] + bv;
                if (accumulate) {
                    float old[8];
                    int so = 0;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        old[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(crs, cvo, so, 0));
                        const int step = (q & 3) == 3 ? 13 * ldc4 : ldc4;
                        asm volatile("s_add_u32 %0, %0, %1" : "+s"(so) : "s"(step) : "scc");
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] += old[q];
                }
                int so = 0;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[q]), crs, cvo, so, 0);
                    const int step = (q & 3) == 3 ? 13 * ldc4 : ldc4;
                    asm volatile("s_add_u32 %0, %0, %1" : "+s"(so) : "s"(step) : "scc");
                }
            } else {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        put(wm * 32 + t * 16 + 4 * lq + e, col, acc16[t * NT16 + jt][e], bv, o0, o1);
            }
        }
        }
    } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (!(row_act[i] && col_act[j])) continue;
                const int col = (wn * TN + j) * 32 + l31;
                if (col >= cols_left) continue;
                float bv;
                int o0, o1;
                if constexpr (HOIST) {
                    bv = h_bv[j];
                    o0 = h_o0[j];
                    o1 = h_o1[j];
                } else {
                    bv = bias ? bias[bias_col0 + col] : 0.0f;
                    o0 = n0 + col;
                    o1 = n0 + col + 1;
                    if (res && res_start) {
                        o0 = res_start[n0 + col];
                        o1 = res_start[n0 + col + 1];
                    }
                }
                const bool full_tile = rows_left >= (wm * TM + i) * 32 + 32;
                if (HYPEL_GEMM_BATCHED_EPILOGUE && (HYPEL_GEMM_BATCHED_EPILOGUE > 1 || accumulate || res) && full_tile) {
                    // Read-modify-write epilogue of a FULL 32-row accumulator tile with every addend IN FLIGHT before the
                    // first store.  As `put` writes it, hipcc must keep each load behind the previous element's store
                    // (they may alias): 16 x (1 + gathered addends) dependent round trips per lane, 15-40 us of a
                    // data-gradient block's ~60 us life.  Here: one pass of 16 raw buffer loads for the old C values, one
                    // pass per gathered addend (a lane without a g-th addend fetches out of range = 0; the pass count is
                    // wave-uniform), the sums in the same order as `put` (bit-identical), then 16 stores.  One per-lane
                    // offset per pass, the row step as a running scalar: no address registers beyond the 16 values.
                    const int row0 = (wm * TM + i) * 32 + 4 * lhi;
                    const int kOOBe = 0x7fffffff;
                    const int ldc4 = __builtin_amdgcn_readfirstlane((int)ldc * 4);
                    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc((void*)cbase, 0, 0x7ffffff0, 0x00020000);
                    const int cvo = (row0 * (int)ldc + col) * 4;
                    float v[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = acc[i][j][e] + bv;
                    if (accumulate) {
                        float old[16];
                        int so = 0;
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            old[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(crs, cvo, so, 0));
                            const int step = (e & 3) == 3 ? 5 * ldc4 : ldc4;
                            asm volatile("s_add_u32 %0, %0, %1" : "+s"(so) : "s"(step) : "scc");
                        }
#pragma unroll
                        for (int e = 0; e < 16; ++e) v[e] += old[e];
                    }
                    if (res) {
                        const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void*)rbase, 0, 0x7ffffff0, 0x00020000);
                        const int ldr4 = __builtin_amdgcn_readfirstlane((int)ldr * 4);
                        for (int g = 0; __any(o0 + g < o1); ++g) {
                            const int rvo = o0 + g < o1 ? (row0 * (int)ldr + o0 + g) * 4 : kOOBe;
                            float add[16];
                            int so = 0;
#pragma unroll
                            for (int e = 0; e < 16; ++e) {
                                add[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, rvo, so, 0));
                                const int step = (e & 3) == 3 ? 5 * ldr4 : ldr4;
                                asm volatile("s_add_u32 %0, %0, %1" : "+s"(so) : "s"(step) : "scc");
                            }
#pragma unroll
                            for (int e = 0; e < 16; ++e) v[e] += add[e];
                        }
                    }
                    int so = 0;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[e]), crs, cvo, so, 0);
                        const int step = (e & 3) == 3 ? 5 * ldc4 : ldc4;
                        asm volatile("s_add_u32 %0, %0, %1" : "+s"(so) : "s"(step) : "scc");
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        put((wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi, col, acc[i][j][e], bv, o0, o1);
                }
            }
    }
}

#define HYPEL_GEMM_PARAMS                                                                                          \
    const float *__restrict__ A, int64_t lda, const float *__restrict__ B, int64_t ldb, float *__restrict__ C,       \
        int64_t ldc, int n, const hypel_group_t *__restrict__ groups, const hypel_seg_t *__restrict__ segs,           \
        const void *__restrict__ tiles_v, int n_tiles, int n_ntiles, const float *__restrict__ bias, int accumulate, \
        const float *__restrict__ res, int64_t ldr, const int32_t *__restrict__ res_start, float *__restrict__ stats
#define HYPEL_GEMM_ARGS \
    A, lda, B, ldb, C, ldc, n, groups, segs, tiles_v, n_tiles, n_ntiles, bias, accumulate, res, ldr, res_start, stats
#define HYPEL_GEMM_BOUNDS \
    __launch_bounds__(256, (TM * TN == 1 ? (TA ? HYPEL_OCC_BN32_TA : HYPEL_OCC_BN32)       \
                                         : (TM * TN == 3 ? (TB ? HYPEL_OCC_BN96 : 4) : (TA && TM * TN == 2 ? HYPEL_OCC_BN64_TA : 3))))

template <int WM, int WN, int TM, int TN, bool TA, bool TB, bool NARROW = false, bool MULTI = false, bool PAIR = false,
          bool VARN = false, bool ACT = false>
__global__ HYPEL_GEMM_BOUNDS void seg_gemm_kernel(HYPEL_GEMM_PARAMS) {
    seg_gemm_body<WM, WN, TM, TN, TA, TB, NARROW, MULTI, PAIR, VARN, ACT>(HYPEL_GEMM_ARGS);
}

// The same code under a cap of 96 scalar registers: 7 instead of 6 resident 128x32 blocks per CU (hipcc uses all 106
// otherwise; LDS and vector registers allow 7).  For launches whose groups have ONE segment (1x1 convolutions and
// their data gradients): 392 row tiles x 4 column tiles = 1568 blocks of an n = 120 layer -- 3136 for n = 240 -- are
// 2 % over the 1536 blocks that 6 per CU hold, so every such launch ends in a round of 32 - 64 blocks; with 7 per CU
// they are resident at once.  The multi-segment launches lose to the extra scalar spills in their segment loop
// (round-2 A/B), and the attribute cannot depend on a template parameter: hence a second kernel symbol.
template <int WM, int WN, int TM, int TN, bool TA, bool TB>
__global__ __attribute__((amdgpu_num_sgpr(96))) HYPEL_GEMM_BOUNDS void seg_gemm_kernel_s96(HYPEL_GEMM_PARAMS) {
    seg_gemm_body<WM, WN, TM, TN, TA, TB>(HYPEL_GEMM_ARGS);
}

// Split-operand variants (HYPEL_GEMM_SPLIT6): 128x128 blocks of 512 threads (2 x 4 waves of 64x32; 72 KB of LDS: two
// blocks = 16 waves per CU, <= 128 registers), 128x64 blocks of 512 threads (4 x 2 waves of 32x32; 54 KB: two blocks) and
// 128x32 blocks of 256 threads (4 x 1 waves of 32x32; 45 KB: three).  Same-box A/B against 256-thread blocks with
// 64x64 / 32x64 wave tiles (round 5): 155 vs 164 us on M = 50176, K = n = 480 -- four waves per SIMD hide the barrier of
// every 16-column k-tile and the split's VALU work better than two.
template <int WM, int WN, int TM, int TN, bool TA, bool TB, bool MULTI = false>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN == 8 ? 4 : 3)) void seg_gemm_split_kernel(HYPEL_GEMM_PARAMS) {
    seg_gemm_body<WM, WN, TM, TN, TA, TB, false, MULTI, false, false, false, true>(HYPEL_GEMM_ARGS);
}

template <int WM, int WN, int TM, int TN>
int launch_cfg_pair(const float* a, int64_t lda, const float* b, int64_t ldb, float* c, int64_t ldc, int n,
                    const hypel_group_t* groups, const hypel_seg_t* segs, const void* tiles, int n_tiles,
                    const float* bias, int accumulate, const float* res, int64_t ldr, const int32_t* res_start,
                    hipStream_t st) {
    constexpr int BN = WN * TN * 32;
    const int n_nt = (n + BN - 1) / BN;
    hipLaunchKernelGGL((seg_gemm_kernel<WM, WN, TM, TN, false, true, false, false, true>), dim3(n_tiles * n_nt),
                       dim3(256), 0, st, a, lda, b, ldb, c, ldc, n, groups, segs, tiles, n_tiles, n_nt, bias, accumulate,
                       res, ldr, res_start, (float*)nullptr);
    return 0;
}

template <int WM, int WN, int TM, int TN, bool NARROW = false, bool MULTI = false>
int launch_cfg(const float* a, int64_t lda, int ta, const float* b, int64_t ldb, int tb, float* c, int64_t ldc,
               int n, const hypel_group_t* groups, const hypel_seg_t* segs, const void* tiles, int n_tiles,
               const float* bias, int accumulate, const float* res, int64_t ldr, const int32_t* res_start,
               hipStream_t st, float* stats = nullptr, bool cap96 = false) {
    constexpr int BN = NARROW ? 16 * TN : WN * TN * 32;
    const int n_nt = MULTI ? 1 : (n + BN - 1) / BN;
    const int grid = n_tiles * n_nt;
    if constexpr (NARROW && TN > 1) {  // forward products only (dispatch checks trans_a = trans_b = 0)
        hipLaunchKernelGGL((seg_gemm_kernel<WM, WN, TM, TN, false, false, true, false>), dim3(grid), dim3(256), 0, st, a,
                           lda, b, ldb, c, ldc, n, groups, segs, tiles, n_tiles, n_nt, bias, accumulate, res, ldr,
                           res_start, stats);
        return 0;
    } else {
    if constexpr (TM * TN == 1 && !NARROW && !MULTI) {
        if (cap96 && !ta) {  // single-segment launches on the 7-blocks-per-CU build of the 128x32 kernel
            if (tb)
                hipLaunchKernelGGL((seg_gemm_kernel_s96<WM, WN, TM, TN, false, true>), dim3(grid), dim3(256), 0, st, a, lda,
                                   b, ldb, c, ldc, n, groups, segs, tiles, n_tiles, n_nt, bias, accumulate, res, ldr,
                                   res_start, stats);
            else
                hipLaunchKernelGGL((seg_gemm_kernel_s96<WM, WN, TM, TN, false, false>), dim3(grid), dim3(256), 0, st, a, lda,
                                   b, ldb, c, ldc, n, groups, segs, tiles, n_tiles, n_nt, bias, accumulate, res, ldr,
                                   res_start, stats);
            return 0;
        }
    }
#define HYPEL_GO(TA_, TB_)                                                                                         \
    hipLaunchKernelGGL((seg_gemm_kernel<WM, WN, TM, TN, TA_, TB_, NARROW, MULTI>), dim3(grid), dim3(256), 0, st, a,  \
                       lda, b, ldb, c, ldc, n, groups, segs, tiles, n_tiles, n_nt, bias, accumulate, res, ldr,     \
                       res_start, stats)
    if constexpr (MULTI) {  // filter gradients only: A transposed, B as stored
        HYPEL_GO(true, false);
    } else {
        if (!ta && !tb) HYPEL_GO(false, false);
        else if (!ta && tb) HYPEL_GO(false, true);
        else HYPEL_GO(true, false);
    }
#undef HYPEL_GO
    return 0;
    }
}

template <int WM, int WN, int TM, int TN, bool MULTI = false>
int launch_split(const float* a, int64_t lda, int ta, const float* b, int64_t ldb, int tb, float* c, int64_t ldc, int n,
                 const hypel_group_t* groups, const hypel_seg_t* segs, const void* tiles, int n_tiles, const float* bias,
                 int accumulate, const float* res, int64_t ldr, const int32_t* res_start, hipStream_t st, float* stats) {
    constexpr int BN = WN * TN * 32;
    const int n_nt = MULTI ? 1 : (n + BN - 1) / BN;
    const int grid = n_tiles * n_nt;
#define HYPEL_GO(TA_, TB_)                                                                                           \
    hipLaunchKernelGGL((seg_gemm_split_kernel<WM, WN, TM, TN, TA_, TB_, MULTI>), dim3(grid), dim3(64 * WM * WN), 0, st, a, lda, b, \
                       ldb, c, ldc, n, groups, segs, tiles, n_tiles, n_nt, bias, accumulate, res, ldr, res_start, stats)
    if constexpr (MULTI) {
        HYPEL_GO(true, false);
    } else {
        if (!ta && !tb) HYPEL_GO(false, false);
        else if (!ta && tb) HYPEL_GO(false, true);
        else HYPEL_GO(true, false);
    }
#undef HYPEL_GO
    return 0;
}

}  // namespace

static int seg_gemm_dispatch(const float* a, int64_t lda, int32_t trans_a, const float* b, int64_t ldb,
                             int32_t trans_b, float* c, int64_t ldc, int32_t n, const hypel_group_t* groups,
                             const hypel_seg_t* segs, const hypel_tile_t* tiles, int32_t n_tiles, const float* bias,
                             int32_t accumulate, const float* res, int64_t ldr, const int32_t* res_start,
                             hypel_stream_t stream, float* stats = nullptr) {
    HYPEL_REQUIRE(a && b && c && groups && segs && tiles, "hypel_seg_gemm_f32");
    HYPEL_REQUIRE(n > 0 && n_tiles >= 0, "hypel_seg_gemm_f32");
    HYPEL_REQUIRE(!(trans_a && trans_b), "hypel_seg_gemm_f32: A^T B^T products are not part of the path");
    if (n_tiles == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    // Tile choice (measured, profiles/r1_*): 128x64 blocks (4 waves/SIMD -> 1024 resident blocks) beat 128x128
    // (3 waves/SIMD -> 768) on every layer of the model by 1.1-1.7x: the grids here are only 1-4 "waves" of blocks,
    // so the finer grain wastes less of the last wave.
    // bits 8-9 of `accumulate`: tile-width hint of the caller (1 = 128x32, 2 = 128x64, 3 = 128x96), measured per launch
    // class (profiles/r1_gemm_tile_choice.txt): data gradients with >= 48 reduction columns per segment and launches with
    // few blocks run faster on the narrow tile, wide filter gradients on the wide one; without a hint: launches with
    // fewer than ~one round of 128x64 blocks (level data gradients: 784) balance better as 128x32
    int hint = (accumulate >> 8) & 3;
    const bool pairs = (accumulate & HYPEL_GEMM_PAIRED_SEGS) != 0;  // segments carry HYPEL_SEG_PAIR_FLAG
    const bool cap96 = (accumulate & HYPEL_GEMM_SINGLE_SEG) != 0;   // every group has one segment
    const bool mfma16x4 = (accumulate & HYPEL_GEMM_MFMA16X4) != 0;  // merged level with <= 16 filters per branch
    const bool var_n = (accumulate & HYPEL_GEMM_VAR_N) != 0;        // groups differ in their column count
    const bool split6 = (accumulate & HYPEL_GEMM_SPLIT6) != 0;      // three-way split operands on the bf16 matrix cores
    const int act_idx = (accumulate >> 16) & 7;                     // HYPEL_GEMM_ACT_*: leaky-ReLU of (product + bias)
    accumulate &= 1;
    if (split6) {
        HYPEL_REQUIRE(!pairs && !mfma16x4 && !act_idx && n > 16,
                      "hypel_seg_gemm_f32: HYPEL_GEMM_SPLIT6 needs a plain product with n > 16");
        // hint: 1 = 128x32, 2 = 128x64, 3 = 128x128 blocks; 0 = by n
        const int w = hint == 1 || n <= 32 ? 32 : (hint == 2 || n <= 64 ? 64 : (hint == 3 || n > 96 ? 128 : 64));
        // tile records with their own column count are honoured by every variant (blocks beyond them exit).  A variant of
        // its own for the forward VAR_N launches -- eight waves of 32 x 64 in a 4 x 2 deal, one active wave on EVERY SIMD for
        // a one-tile group -- was built and measured SLOWER than the rows-first 2 x 4 deal of 64 x 32 wave tiles (round 6:
        // 349 vs 328 us on the 30-filter level, profiles/r6_exp_varn_kernel.txt); removed.
        (void)var_n;
        if (w == 32)
            launch_split<4, 1, 1, 1>(a, lda, trans_a, b, ldb, trans_b, c, ldc, n, groups, segs, tiles, n_tiles, bias,
                                     accumulate, res, ldr, res_start, st, stats);
        else if (w == 64)
            launch_split<4, 2, 1, 1>(a, lda, trans_a, b, ldb, trans_b, c, ldc, n, groups, segs, tiles, n_tiles, bias,
                                     accumulate, res, ldr, res_start, st, stats);
        else
            launch_split<2, 4, 2, 1>(a, lda, trans_a, b, ldb, trans_b, c, ldc, n, groups, segs, tiles, n_tiles, bias,
                                     accumulate, res, ldr, res_start, st, stats);
        HYPEL_CHECK_LAUNCH("hypel_seg_gemm_f32");
        return 0;
    }
    const bool plain_fwd = !trans_a && !trans_b && !pairs && !stats && !res;
    if (act_idx) {
        HYPEL_REQUIRE(plain_fwd && !accumulate && !mfma16x4 && !var_n && act_idx <= 4,
                      "hypel_seg_gemm_f32: HYPEL_GEMM_ACT_* needs a plain forward product (no accumulate / shortcut / statistics)");
        const bool narrow_a = hint == 1 || n <= 32 || (hint == 0 && (int64_t)n_tiles * ((n + 63) / 64) < 1000);
        const int bn = narrow_a ? 32 : 64;
        const int n_nt = (n + bn - 1) / bn;
        if (narrow_a)
            hipLaunchKernelGGL((seg_gemm_kernel<4, 1, 1, 1, false, false, false, false, false, false, true>),
                               dim3(n_tiles * n_nt), dim3(256), 0, st, a, lda, b, ldb, c, ldc, n, groups, segs, tiles, n_tiles,
                               n_nt, bias, act_idx << 16, res, ldr, res_start, (float*)nullptr);
        else
            hipLaunchKernelGGL((seg_gemm_kernel<4, 1, 1, 2, false, false, false, false, false, false, true>),
                               dim3(n_tiles * n_nt), dim3(256), 0, st, a, lda, b, ldb, c, ldc, n, groups, segs, tiles, n_tiles,
                               n_nt, bias, act_idx << 16, res, ldr, res_start, (float*)nullptr);
        HYPEL_CHECK_LAUNCH("hypel_seg_gemm_f32");
        return 0;
    }
    if (mfma16x4 && n <= 64 && plain_fwd) {
        launch_cfg<4, 1, 1, 4, true>(a, lda, trans_a, b, ldb, trans_b, c, ldc, n, groups, segs, tiles, n_tiles, bias,
                                     accumulate, res, ldr, res_start, st);
        HYPEL_CHECK_LAUNCH("hypel_seg_gemm_f32");
        return 0;
    }
    if (var_n && n > 32 && plain_fwd && hint != 1) {  // 128x64 blocks, one- and two-tile MFMA phases
        const int n_nt = (n + 63) / 64;
        hipLaunchKernelGGL((seg_gemm_kernel<4, 1, 1, 2, false, false, false, false, false, true>),
                           dim3(n_tiles * n_nt), dim3(256), 0, st, a, lda, b, ldb, c, ldc, n, groups, segs, tiles, n_tiles,
                           n_nt, bias, accumulate, res, ldr, res_start, (float*)nullptr);
        HYPEL_CHECK_LAUNCH("hypel_seg_gemm_f32");
        return 0;
    }
    // hint 3 = 128x96 blocks (three 32x32 accumulators per wave, 5 resident blocks per CU): N = 240 / 480 tile without
    // padding (5 x 96, 96 + 96 + 48) and a layer's 392 row tiles x 3 or 5 column tiles fit the resident capacity where
    // 392 x 4 / x 8 of the 64-wide tiling overflow it by 2 %.
    if (hint == 3 && n > 64) {
        launch_cfg<4, 1, 1, 3>(a, lda, trans_a, b, ldb, trans_b, c, ldc, n, groups, segs, tiles, n_tiles, bias,
                               accumulate, res, ldr, res_start, st, stats);
        HYPEL_CHECK_LAUNCH("hypel_seg_gemm_f32");
        return 0;
    }
    if (hint == 3) hint = 2;
    const bool narrow = hint == 1 || (hint == 0 && (int64_t)n_tiles * ((n + 63) / 64) < 1000);
    if (pairs) {  // data gradients only: A as stored, B transposed; n > 16
        HYPEL_REQUIRE(!trans_a && trans_b && n > 16 && !stats, "hypel_seg_gemm_f32: paired segments");
        if (n <= 32 || narrow)
            launch_cfg_pair<4, 1, 1, 1>(a, lda, b, ldb, c, ldc, n, groups, segs, tiles, n_tiles, bias, accumulate, res,
                                        ldr, res_start, st);
        else
            launch_cfg_pair<4, 1, 1, 2>(a, lda, b, ldb, c, ldc, n, groups, segs, tiles, n_tiles, bias, accumulate, res,
                                        ldr, res_start, st);
        HYPEL_CHECK_LAUNCH("hypel_seg_gemm_f32");
        return 0;
    }
    // n <= 16 (the Cout = 15 level, fc_final): 128x16 blocks on the 16x16x4 MFMA (no reduction epilogues there)
    if (n <= 16 && !stats)
        launch_cfg<4, 1, 1, 1, true>(a, lda, trans_a, b, ldb, trans_b, c, ldc, n, groups, segs, tiles, n_tiles, bias,
                                     accumulate, res, ldr, res_start, st);
    else if (n <= 32 || narrow)
        launch_cfg<4, 1, 1, 1>(a, lda, trans_a, b, ldb, trans_b, c, ldc, n, groups, segs, tiles, n_tiles, bias,
                               accumulate, res, ldr, res_start, st, stats, cap96);
    else
        launch_cfg<4, 1, 1, 2>(a, lda, trans_a, b, ldb, trans_b, c, ldc, n, groups, segs, tiles, n_tiles, bias,
                               accumulate, res, ldr, res_start, st, stats);
    HYPEL_CHECK_LAUNCH("hypel_seg_gemm_f32");
    return 0;
}

extern "C" int hypel_seg_gemm_f32(const float* a, int64_t lda, int32_t trans_a, const float* b, int64_t ldb,
                                  int32_t trans_b, float* c, int64_t ldc, int32_t n, const hypel_group_t* groups,
                                  const hypel_seg_t* segs, const hypel_tile_t* tiles, int32_t n_tiles,
                                  const float* bias, int32_t accumulate, hypel_stream_t stream) {
    return seg_gemm_dispatch(a, lda, trans_a, b, ldb, trans_b, c, ldc, n, groups, segs, tiles, n_tiles, bias,
                             accumulate, nullptr, 0, nullptr, stream);
}

extern "C" int hypel_seg_gemm_stats_f32(const float* a, int64_t lda, int32_t trans_a, const float* b, int64_t ldb,
                                        int32_t trans_b, float* c, int64_t ldc, int32_t n,
                                        const hypel_group_t* groups, const hypel_seg_t* segs, const hypel_tile_t* tiles,
                                        int32_t n_tiles, const float* bias, int32_t accumulate, float* stats_partial,
                                        hypel_stream_t stream) {
    HYPEL_REQUIRE(stats_partial && (accumulate & 1) == 0, "hypel_seg_gemm_stats_f32");
    return seg_gemm_dispatch(a, lda, trans_a, b, ldb, trans_b, c, ldc, n, groups, segs, tiles, n_tiles, bias,
                             accumulate, nullptr, 0, nullptr, stream, stats_partial);
}

extern "C" int hypel_seg_gemm_res_f32(const float* a, int64_t lda, int32_t trans_a, const float* b, int64_t ldb,
                                      int32_t trans_b, float* c, int64_t ldc, int32_t n, const hypel_group_t* groups,
                                      const hypel_seg_t* segs, const hypel_tile_t* tiles, int32_t n_tiles,
                                      const float* bias, int32_t accumulate, const float* res, int64_t ldr,
                                      const int32_t* res_start, hypel_stream_t stream) {
    HYPEL_REQUIRE(res && ldr > 0, "hypel_seg_gemm_res_f32");
    return seg_gemm_dispatch(a, lda, trans_a, b, ldb, trans_b, c, ldc, n, groups, segs, tiles, n_tiles, bias,
                             accumulate, res, ldr, res_start, stream);
}

extern "C" int hypel_seg_gemm_multi_f32(const float* base, int32_t trans_a, int32_t trans_b, int32_t tile_width,
                                        const hypel_seg_t* segs, const hypel_mtile_t* blocks, int32_t n_blocks,
                                        hypel_stream_t stream) {
    HYPEL_REQUIRE(base && segs && blocks && n_blocks >= 0, "hypel_seg_gemm_multi_f32");
    HYPEL_REQUIRE(trans_a == 1 && trans_b == 0, "hypel_seg_gemm_multi_f32: only A^T B products (filter gradients)");
    const bool split6 = (tile_width & HYPEL_GEMM_MULTI_SPLIT6) != 0;
    tile_width &= ~HYPEL_GEMM_MULTI_SPLIT6;
    HYPEL_REQUIRE(tile_width == 16 || tile_width == 32 || tile_width == 64 || (split6 && tile_width == 128),
                  "hypel_seg_gemm_multi_f32");
    HYPEL_REQUIRE(!split6 || tile_width >= 32, "hypel_seg_gemm_multi_f32: split operands need 32- / 64- / 128-wide blocks");
    if (n_blocks == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    float* c = const_cast<float*>(base);
    if (split6) {
        if (tile_width == 32)
            launch_split<4, 1, 1, 1, true>(base, 0, 1, base, 0, 0, c, 0, 32, nullptr, segs, blocks, n_blocks, nullptr, 0,
                                           nullptr, 0, nullptr, st, nullptr);
        else if (tile_width == 64)
            launch_split<4, 2, 1, 1, true>(base, 0, 1, base, 0, 0, c, 0, 64, nullptr, segs, blocks, n_blocks, nullptr, 0,
                                           nullptr, 0, nullptr, st, nullptr);
        else
            launch_split<2, 4, 2, 1, true>(base, 0, 1, base, 0, 0, c, 0, 128, nullptr, segs, blocks, n_blocks, nullptr, 0,
                                           nullptr, 0, nullptr, st, nullptr);
    } else if (tile_width == 16)
        launch_cfg<4, 1, 1, 1, true, true>(base, 0, 1, base, 0, 0, c, 0, 16, nullptr, segs, blocks, n_blocks, nullptr, 0,
                                           nullptr, 0, nullptr, st);
    else if (tile_width == 32)
        launch_cfg<4, 1, 1, 1, false, true>(base, 0, 1, base, 0, 0, c, 0, 32, nullptr, segs, blocks, n_blocks, nullptr, 0,
                                            nullptr, 0, nullptr, st);
    else
        launch_cfg<4, 1, 1, 2, false, true>(base, 0, 1, base, 0, 0, c, 0, 64, nullptr, segs, blocks, n_blocks, nullptr, 0,
                                            nullptr, 0, nullptr, st);
    HYPEL_CHECK_LAUNCH("hypel_seg_gemm_multi_f32");
    return 0;
}
