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
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BK = 32;

template <int WM, int WN, int TM, int TN, bool TA, bool TB>
__global__ __launch_bounds__(256) void seg_gemm_kernel(const float* __restrict__ A, int64_t lda,
                                                        const float* __restrict__ B, int64_t ldb,
                                                        float* __restrict__ C, int64_t ldc, int n,
                                                        const hypel_group_t* __restrict__ groups,
                                                        const hypel_seg_t* __restrict__ segs,
                                                        const hypel_tile_t* __restrict__ tiles, int n_tiles,
                                                        int n_ntiles, const float* __restrict__ bias,
                                                        int accumulate) {
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    static_assert(WM * WN == 4, "4 waves per block");
    static_assert(BM == HYPEL_GEMM_BM, "tile table is built for BM = 128");
    // LDS images (rows x pitch), global row order preserved
    constexpr int A_ROWS = TA ? BK : BM;
    constexpr int A_COLS = TA ? BM : BK;
    constexpr int A_PITCH = TA ? BM : BK + 1;
    constexpr int B_ROWS = TB ? BN : BK;
    constexpr int B_COLS = TB ? BK : BN;
    constexpr int B_PITCH = TB ? BK + 1 : BN;
    constexpr int A_PER_THREAD = A_ROWS * A_COLS / 256;
    constexpr int B_PER_THREAD = B_ROWS * B_COLS / 256;
    constexpr int A_RSTEP = 256 / A_COLS;
    constexpr int B_RSTEP = 256 / B_COLS;
    __shared__ float lds[A_ROWS * A_PITCH + B_ROWS * B_PITCH];
    float* As = lds;
    float* Bs = lds + A_ROWS * A_PITCH;

    // ---- block -> (row tile, column tile), XCD-aware (bijective remap, guide §5 T1) ----
    const int nblk = gridDim.x;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, loc = bid >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    const int tile_id = lid / n_ntiles;
    const int n0 = (lid - tile_id * n_ntiles) * BN;
    if (tile_id >= n_tiles) return;
    const hypel_tile_t tile = tiles[tile_id];
    const hypel_group_t grp = groups[tile.group];
    const int m0 = tile.m0;
    const int rows_left = grp.rows - m0;  // valid rows in this tile (may exceed BM)
    const int cols_left = n - n0;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lhi = lane >> 5;

    // per-thread staging coordinates
    const int a_col = tid % A_COLS, a_row0 = tid / A_COLS;
    const int b_col = tid % B_COLS, b_row0 = tid / B_COLS;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // wave-uniform activity of each 32x32 accumulator tile
    bool row_act[TM], col_act[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) row_act[i] = (wm * TM + i) * 32 < rows_left;
#pragma unroll
    for (int j = 0; j < TN; ++j) col_act[j] = (wn * TN + j) * 32 < cols_left;

    float ra[A_PER_THREAD], rb[B_PER_THREAD];

    int ls = grp.seg_begin;
    const int s_end = grp.seg_begin + grp.seg_count;
    int lk = 0;
    hypel_seg_t seg;
    seg.a_off = 0; seg.b_off = 0; seg.k = 0; seg.reserved = 0;
    bool have = ls < s_end;
    if (have) seg = segs[ls];

    auto load_tiles = [&](const hypel_seg_t& sg, int k0) {
        const int k_left = sg.k - k0;
        // ---- A ----
        if (!TA) {
            // rows = output rows (m), cols = k
            const float* base = A + sg.a_off + (int64_t)m0 * lda + k0;
            const bool cok = a_col < k_left;
#pragma unroll
            for (int i = 0; i < A_PER_THREAD; ++i) {
                const int row = a_row0 + i * A_RSTEP;
                ra[i] = (cok && row < rows_left) ? base[(int64_t)row * lda + a_col] : 0.0f;
            }
        } else {
            // rows = k (reduction rows), cols = output rows (m)
            const float* base = A + sg.a_off + (int64_t)k0 * lda + m0;
            const bool cok = a_col < rows_left;
#pragma unroll
            for (int i = 0; i < A_PER_THREAD; ++i) {
                const int row = a_row0 + i * A_RSTEP;
                ra[i] = (cok && row < k_left) ? base[(int64_t)row * lda + a_col] : 0.0f;
            }
        }
        // ---- B ----
        if (!TB) {
            // rows = k, cols = n
            const float* base = B + sg.b_off + (int64_t)k0 * ldb + n0;
            const bool cok = b_col < cols_left;
#pragma unroll
            for (int i = 0; i < B_PER_THREAD; ++i) {
                const int row = b_row0 + i * B_RSTEP;
                rb[i] = (cok && row < k_left) ? base[(int64_t)row * ldb + b_col] : 0.0f;
            }
        } else {
            // rows = n, cols = k
            const float* base = B + sg.b_off + (int64_t)n0 * ldb + k0;
            const bool cok = b_col < k_left;
#pragma unroll
            for (int i = 0; i < B_PER_THREAD; ++i) {
                const int row = b_row0 + i * B_RSTEP;
                rb[i] = (cok && row < cols_left) ? base[(int64_t)row * ldb + b_col] : 0.0f;
            }
        }
    };

    if (have) load_tiles(seg, lk);

    while (have) {
        const int kvalid = min(BK, seg.k - lk);
        __syncthreads();  // previous tile's MFMAs are done reading LDS
#pragma unroll
        for (int i = 0; i < A_PER_THREAD; ++i) As[(a_row0 + i * A_RSTEP) * A_PITCH + a_col] = ra[i];
#pragma unroll
        for (int i = 0; i < B_PER_THREAD; ++i) Bs[(b_row0 + i * B_RSTEP) * B_PITCH + b_col] = rb[i];
        __syncthreads();

        // advance the (segment, k) cursor and put the next tile's loads in flight
        lk += BK;
        if (lk >= seg.k) {
            ++ls;
            lk = 0;
            if (ls < s_end) seg = segs[ls];
        }
        have = ls < s_end;
        if (have) load_tiles(seg, lk);

        const int ksteps = (kvalid + 1) >> 1;
        for (int k2 = 0; k2 < ksteps; ++k2) {
            const int kk = 2 * k2 + lhi;
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int mrow = (wm * TM + i) * 32 + l31;
                a[i] = TA ? As[kk * A_PITCH + mrow] : As[mrow * A_PITCH + kk];
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int ncol = (wn * TN + j) * 32 + l31;
                b[j] = TB ? Bs[ncol * B_PITCH + kk] : Bs[kk * B_PITCH + ncol];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    if (row_act[i] && col_act[j])
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5) ----
    float* cbase = C + grp.c_off + (int64_t)m0 * ldc + n0;
    // bias is indexed by the absolute output column: groups of a merged level start at channel offsets
    const int bias_col0 = bias ? (int)(grp.c_off % ldc) + n0 : 0;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (!(row_act[i] && col_act[j])) continue;
            const int col = (wn * TN + j) * 32 + l31;
            if (col >= cols_left) continue;
            const float bv = bias ? bias[bias_col0 + col] : 0.0f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
                if (row < rows_left) {
                    float* p = cbase + (int64_t)row * ldc + col;
                    float v = acc[i][j][e] + bv;
                    if (accumulate) v += *p;
                    *p = v;
                }
            }
        }
}

template <int WM, int WN, int TM, int TN>
int launch_cfg(const float* a, int64_t lda, int ta, const float* b, int64_t ldb, int tb, float* c, int64_t ldc,
               int n, const hypel_group_t* groups, const hypel_seg_t* segs, const hypel_tile_t* tiles, int n_tiles,
               const float* bias, int accumulate, hipStream_t st) {
    constexpr int BN = WN * TN * 32;
    const int n_nt = (n + BN - 1) / BN;
    const int grid = n_tiles * n_nt;
#define HYPEL_GO(TA_, TB_)                                                                                   \
    hipLaunchKernelGGL((seg_gemm_kernel<WM, WN, TM, TN, TA_, TB_>), dim3(grid), dim3(256), 0, st, a, lda, b, \
                       ldb, c, ldc, n, groups, segs, tiles, n_tiles, n_nt, bias, accumulate)
    if (!ta && !tb) HYPEL_GO(false, false);
    else if (!ta && tb) HYPEL_GO(false, true);
    else if (ta && !tb) HYPEL_GO(true, false);
    else HYPEL_GO(true, true);
#undef HYPEL_GO
    return 0;
}

}  // namespace

extern "C" int hypel_seg_gemm_f32(const float* a, int64_t lda, int32_t trans_a, const float* b, int64_t ldb,
                                  int32_t trans_b, float* c, int64_t ldc, int32_t n, const hypel_group_t* groups,
                                  const hypel_seg_t* segs, const hypel_tile_t* tiles, int32_t n_tiles,
                                  const float* bias, int32_t accumulate, hypel_stream_t stream) {
    HYPEL_REQUIRE(a && b && c && groups && segs && tiles, "hypel_seg_gemm_f32");
    HYPEL_REQUIRE(n > 0 && n_tiles >= 0, "hypel_seg_gemm_f32");
    if (n_tiles == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (n <= 32)
        launch_cfg<4, 1, 1, 1>(a, lda, trans_a, b, ldb, trans_b, c, ldc, n, groups, segs, tiles, n_tiles, bias,
                               accumulate, st);
    else if (n <= 64)
        launch_cfg<4, 1, 1, 2>(a, lda, trans_a, b, ldb, trans_b, c, ldc, n, groups, segs, tiles, n_tiles, bias,
                               accumulate, st);
    else
        launch_cfg<2, 2, 2, 2>(a, lda, trans_a, b, ldb, trans_b, c, ldc, n, groups, segs, tiles, n_tiles, bias,
                               accumulate, st);
    HYPEL_CHECK_LAUNCH("hypel_seg_gemm_f32");
    return 0;
}
