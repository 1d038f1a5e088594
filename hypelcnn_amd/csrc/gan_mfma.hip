// Shadow-GAN generator for WIDE spectra (bands > 128; AVON: 360 bands) on the CDNA4 matrix cores.
//
// shadowdata_generator_model (gan/shadow_data_models.py:43-90): seven 1-channel SAME 1-D convolutions over the band axis
// (kernel sizes B, B/2, B/4, B/8, B/4, B/2, B), leaky-ReLU(0.1), skip sums n_l = c_l + n_{l-1} + n_{l-2}, tanh on the
// last layer.  At B = 360 that is 384 k multiply-adds per sample: a 1-in / 1-out channel convolution of N samples is
//     Y[N x B] = X[N x B] . T[B x B],    T[i][j] = w[i - j + pad]   (banded Toeplitz),
// i.e. matrix-core work.  One block = 16 samples (one v_mfma_f32_16x16x4_f32 row tile) x all bands, 8 wavefronts:
//  * the activations of the 16 samples never leave the CU: three rotating [16 x B] LDS images (n_{l-2}, n_{l-1}, n_l);
//  * T is never materialised: the B operand of lane (column jj, k-slot kq) is ONE LDS word of a zero-margined copy of
//    the layer's taps, wz[B' + k - j + pad] -- the lanes of a fragment read 17 consecutive words (broadcasts);
//  * a column tile only walks the 16-row chunks its band touches (exact taps rounded to chunks of 16);
//  * backward: forward recompute with every layer's output kept in REGISTERS (MFMA C layout: 12 floats per lane and
//    layer), then per layer l = L..1: dz_l = dn_l * act'(.), filter gradient, data gradient dn_{l-1} += dz_l . T^T
//    (the same tap table, index mirrored).  The filter gradient dw[t] = sum_{n,j} dz[n][j] x[n][j + t - pad] is the
//    sum of one diagonal of G = X^T dZ; tiles of G with the same tile offset I - J = 16 a are ACCUMULATED over J in one
//    16x16 accumulator (K = 16 samples x 23 column tiles), so a layer needs only (k / 16 + 3) accumulators and one
//    small diagonal reduction per block, in a fixed order (no float atomics).
// fp32 MFMA is a k-ordered fmaf chain: same arithmetic class as the VALU kernels in gan.hip, which stay in charge
// of bands < 16 and bands > 384 (and of everything under HYPEL_GAN_MFMA=0).
#include <stdlib.h>

#include "common.h"

// No fused multiply-adds outside the MFMAs: whether hipcc contracts `gd * f` into a neighbouring sum depends on the code
// around it, and the two backward kernels (LDS ring / register ring) must round alike (they are tested bit for bit).
#pragma clang fp contract(off)

typedef float gm_f32x4 __attribute__((ext_vector_type(4)));
struct GmApps {
    int n_apps;
    int64_t w_stride, b_stride;    // element distance of application g + 1's variables from application g's
    int64_t pw_stride, pb_stride;  // ... of its first gradient slab (0: the slabs of all applications are consecutive)
};

namespace {

constexpr int GM_ROWS = 16;     // samples per row tile
#ifndef GM_WAVES_N
#define GM_WAVES_N 8
#endif
#ifndef GM_MAXT_N
#define GM_MAXT_N 3
#endif
constexpr int GM_WAVES = GM_WAVES_N;  // wavefronts per block (two per SIMD)
constexpr int GM_THREADS = 64 * GM_WAVES;
constexpr int GM_MAXT = GM_MAXT_N;    // column tiles per wave: bands <= 16 * GM_WAVES * GM_MAXT
constexpr int GM_MAX_BANDS = 16 * GM_WAVES * GM_MAXT;
#ifndef GM_PIPE
#define GM_PIPE 0  // 1: fragments of the next 16-column chunk requested by hand ahead of the current chunk's MFMAs
                   // (measured slower than hipcc's own schedule: forward 62 vs 51 us, backward 177 vs 167 us at N = 4096)
#endif
#ifndef GM_TWO_CHAINS
#define GM_TWO_CHAINS 1
#endif
#ifndef GM_DIAG
#define GM_DIAG 0
#endif
#ifndef GM_FWD_ROLLED
#define GM_FWD_ROLLED 0  // forward kernel: 1 = rolled layer loop (54 vs 51 us); the backward kernel's recompute is always
                         // rolled (unrolled it spills > 200 registers)
#endif
#if GM_DIAG == 7  // forward phase stamps: cycles of thread 0 of block 0 per phase, reported through out[0..7]
__device__ long long gm_fdbg[8];
__device__ long long gm_fmark;
#define GM_FMARK(i) { if (threadIdx.x == 0 && blockIdx.x == 0) { const long long now_ = clock64(); gm_fdbg[i] += now_ - gm_fmark; gm_fmark = now_; } }
#else
#define GM_FMARK(i)
#endif
constexpr int GM_GP = 17;       // pitch of a 16 x 16 filter-gradient tile in LDS (diagonal reads hit distinct banks)

struct GmGeo {
    int bands, bp, pitch, nt;  // bp = bands rounded up to 16; nt = column tiles; pitch = 18 mod 32 (see a_frag)
};

__host__ __device__ inline GmGeo gm_geo(int bands) {
    GmGeo g;
    g.bands = bands;
    g.bp = (bands + 15) / 16 * 16;
    g.nt = g.bp / 16;
    g.pitch = g.bp + ((18 - g.bp % 32) + 32) % 32;
    return g;
}
__host__ __device__ inline int gm_ksz(int bands, int l) {  // kernel size of layer l = 0..6
    const int sh = l < 4 ? l : 6 - l;
    return bands >> sh;
}
__host__ __device__ inline int gm_woff(int bands, int l) {
    int o = 0;
    for (int i = 0; i < l; ++i) o += gm_ksz(bands, i);
    return o;
}
// dynamic LDS: forward = 3 activation images + 2 tap tables; backward = 5 images + 2 tap tables + G tiles
__host__ __device__ inline int gm_raw(int bands) {  // floats of the block's LDS copy of every layer's taps + 8 biases
    return (gm_woff(bands, 7) + 8 + 3) / 4 * 4;
}
#ifndef GM_FWD_ONE_TABLE
#define GM_FWD_ONE_TABLE 1  // forward kernel: ONE tap table (a second barrier per layer instead of a second table): 80.7 instead
                            // of 85.1 KB at 360 bands, i.e. TWO resident blocks per CU (2 x 85.1 KB > 160 KB by 5 KB)
#endif
__host__ __device__ inline size_t gm_fwd_lds(int bands) {
    const GmGeo g = gm_geo(bands);
    return sizeof(float) * (3 * (size_t)GM_ROWS * g.pitch + (GM_FWD_ONE_TABLE ? 1 : 2) * 3 * (size_t)g.bp + gm_raw(bands));
}
__host__ __device__ inline int gm_atiles(int bands) { return (bands + 30) / 16 + 2; }  // upper bound of the tile offsets a
__host__ __device__ inline size_t gm_bwd_lds(int bands) {
    const GmGeo g = gm_geo(bands);
    return sizeof(float) * (5 * (size_t)GM_ROWS * g.pitch + 6 * (size_t)g.bp + (size_t)gm_atiles(bands) * 16 * GM_GP + 128 +
                            gm_raw(bands));
}

// ---- one 16 x 16 output tile of  src[16 x B] . T  (MIRROR = false)  or  src . T^T  (MIRROR = true) -----------------
// A fragment of v_mfma_f32_16x16x4_f32: lane (r = lane & 15, kq = lane >> 4) supplies src[r][k], k = kc + 4 s + kq: with
// pitch = 18 mod 32 the 32 lanes of a ds_read_b32 group (16 rows x 2 k-slots) hit 32 banks.  B fragment: lane
// (c = lane & 15, kq) supplies T[k][j0 + c] = w[k - (j0 + c) + pad] (mirrored: w[(j0 + c) - k + pad]) from the
// zero-margined tap table wz (taps at [bp, bp + ksz)).
#ifndef GM_CONV_SCHED
#define GM_CONV_SCHED 0  // 1: the fetch / mac / fetch / mac order of the loop below pinned with sched_barrier
#endif
#if GM_CONV_SCHED
#define GM_CONV_PIN __builtin_amdgcn_sched_barrier(0);
#else
#define GM_CONV_PIN
#endif
template <bool MIRROR>
__device__ __forceinline__ gm_f32x4 gm_conv_tile(const float* __restrict__ src, const float* __restrict__ wz, const GmGeo g,
                                                 int j0, int ksz, int pad, int lane) {
    const int r = lane & 15, kq = lane >> 4;
    // reduction range of the tile: forward i in [j0 - pad, j0 + 15 + ksz - 1 - pad]; mirrored j in
    // [i0 - (ksz - 1 - pad), i0 + 15 + pad]; clipped to the real bands
    const int lo = max(0, MIRROR ? j0 - (ksz - 1 - pad) : j0 - pad);
    const int hi = min(g.bands - 1, MIRROR ? j0 + 15 + pad : j0 + 15 + ksz - 1 - pad);
    gm_f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    const float* ap = src + r * g.pitch + kq;
    const float* tp = MIRROR ? wz + g.bp + (j0 + r) + pad - kq : wz + g.bp - (j0 + r) + pad + kq;
    // j0 / ksz / pad are wave-uniform (the callers take the wave index through readfirstlane), so this is a scalar
    // loop.  Two chunks per trip, the fragments of one requested while the other's four MFMAs run: left to itself
    // hipcc loaded two fragments, waited, issued two MFMAs, and again -- the LDS round trip exposed twice per chunk
    // (61 cycles per MFMA and SIMD instead of 32).
    const int k0 = lo & ~15;
    const int nch = hi >= k0 ? ((hi - k0) >> 4) + 1 : 0;
#if GM_DIAG == 4  // timing diagnostics: everything but the products
    acc[0] = (float)nch;
    return acc;
#elif GM_DIAG
    for (int c = 0; c < nch; ++c)
        for (int s = 0; s < 4; ++s) {
            const int k = k0 + 16 * c + 4 * s;
            const float av = GM_DIAG == 1 ? ap[k] : 1.0f + (float)s;
            const float bv = GM_DIAG == 2 ? (MIRROR ? tp[-k] : tp[k]) : 2.0f + (float)s;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
        }
    return acc;
#else
    if (nch == 0) return acc;
    float a0[4], b0[4], a1[4], b1[4];
    auto fetch = [&](int c, float (&a)[4], float (&b)[4]) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = k0 + 16 * c + 4 * s;
            a[s] = ap[k];
            b[s] = MIRROR ? tp[-k] : tp[k];
        }
    };
    // two accumulator chains (k-steps 0, 2 / 1, 3 of every chunk), added at the end: a 16x16x4 MFMA issues every 32
    // cycles but its result is ready after 40, and a wave has one partner on its SIMD
    gm_f32x4 acc2 = {0.0f, 0.0f, 0.0f, 0.0f};
    auto mac = [&](const float (&a)[4], const float (&b)[4]) {
#if GM_TWO_CHAINS
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc2, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc2, 0, 0, 0);
#else
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], acc, 0, 0, 0);
#endif
    };
    fetch(0, a0, b0);
    int c = 0;
    for (; c + 2 < nch; c += 2) {
        fetch(c + 1, a1, b1);
        GM_CONV_PIN
        mac(a0, b0);
        GM_CONV_PIN
        fetch(c + 2, a0, b0);
        GM_CONV_PIN
        mac(a1, b1);
        GM_CONV_PIN
    }
    if (c + 1 < nch) {  // two chunks left
        fetch(c + 1, a1, b1);
        mac(a0, b0);
        mac(a1, b1);
    } else {
        mac(a0, b0);
    }
#if GM_TWO_CHAINS
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] += acc2[e];
#endif
    return acc;
#endif
}

// Every layer's taps and the 8 biases are copied into LDS ONCE per block (gm_stage_raw); a layer's zero-margined table
// is then built from that copy: a global round trip in front of every layer's barrier (7 per forward, 14 per
// backward) was ~10 % of the kernels.
__device__ __forceinline__ void gm_stage_raw(float* raw, int bands, const float* __restrict__ w,
                                             const float* __restrict__ bias, int tid) {
    const int wtotal = gm_woff(bands, 7);
    for (int i = tid; i < wtotal; i += GM_THREADS) raw[i] = w[i];
    if (tid < 8) raw[wtotal + tid] = tid < 7 ? bias[tid] : 0.0f;
}
__device__ __forceinline__ void gm_fill_taps(float* wz, const GmGeo g, const float* w, int ksz, int tid) {
    // only [bp, bp + bands) can ever hold taps; the margins were zeroed once
    for (int i = tid; i < g.bands; i += GM_THREADS) wz[g.bp + i] = i < ksz ? w[i] : 0.0f;
}

// Forward of the whole stack for one row tile.  bufs: three [16 x pitch] images, the input in bufs[0] (zero beyond
// `bands` and for rows without a sample).  KEEP: every lane keeps the layers' outputs (its 12 elements per layer, MFMA
// C layout: column = lane & 15, row = 4 (lane >> 4) + e) and the leaky-ReLU branch bits for the backward pass.
// Returns the image index that holds the result (n4 or tanh output); with out != nullptr the result also goes to global.
template <bool ENC>
struct GmKeep {
    static constexpr int SLOTS = ENC ? 3 : 7;  // n_1..n_3 | n_1..n_6 + the tanh output
    static constexpr int V4 = SLOTS * GM_MAXT + 2;  // float4 per thread and row tile
};

typedef unsigned gm_u32x4 __attribute__((ext_vector_type(4)));

// STASH: every layer's outputs / branch bits go straight from the epilogue to this thread's place in the kept-activation
// buffer (`stash` = the row tile's base + tid, in float4): nothing stays in registers, the unrolled forward stays cheap.
template <bool ENC, bool KEEP, bool ROLLED = true, bool STASH = false, bool ONE_TABLE = false, bool TAP = false>
__device__ __forceinline__ int gm_forward(float* lds0, float* wz0, float* wz1, const GmGeo g,
                                          const float* w, const float* bias,  // the block's LDS copy (gm_stage_raw)
                                          float* __restrict__ out, int64_t ldo, int rows_valid, int tid,
                                          float (&keep)[6][GM_MAXT][4], unsigned (&mask)[7],
                                          gm_f32x4* __restrict__ stash = nullptr, float* __restrict__ enc_out = nullptr,
                                          int64_t ld_enc = 0) {
    constexpr int L = ENC ? 4 : 7;
#ifndef GM_FWD_REGSKIP
#define GM_FWD_REGSKIP 1
#endif
    constexpr bool REGSKIP = GM_FWD_REGSKIP && !KEEP;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: tile loops are wave-uniform
    const int col = lane & 15, rg = lane >> 4;
    const int img = GM_ROWS * g.pitch;
    int woff = 0;
    [[maybe_unused]] float nr1[GM_MAXT][4], nr2[GM_MAXT][4];
    if constexpr (KEEP) {
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int m = 0; m < GM_MAXT; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e) keep[q][m][e] = 0.0f;
    }
    // one rolled loop over the layers (kept values go to their slot through wave-uniform selects): the unrolled form
    // spent > 180 vector registers on seven copies of the addressing
#pragma unroll ROLLED ? 1 : 7
    for (int l = 0; l < L; ++l) {
        const int ksz = gm_ksz(g.bands, l), pad = (ksz - 1) / 2;
        float* wz = (l & 1) ? wz1 : wz0;
        if constexpr (ONE_TABLE) {  // wz0 == wz1: every wave must have left layer l - 1's products before its taps go
            if (l > 0) __syncthreads();
        }
        GM_FMARK(4)  // end of the previous layer's epilogue .. (ONE_TABLE barrier)
        gm_fill_taps(wz, g, w + woff, ksz, tid);
        woff += ksz;
        __syncthreads();  // taps of layer l and the outputs of layer l - 1 are in LDS
        GM_FMARK(1)  // tap table + barrier
        const float* src = lds0 + (l % 3) * img;          // n_l (x for l = 0)
        const float* skip2 = lds0 + ((l + 2) % 3) * img;  // n_{l-1}
        float* dst = lds0 + ((l + 1) % 3) * img;
        const float bl = bias[l];
        const bool last_tanh = !ENC && l == 6;
        unsigned mk = 0;
        // REGSKIP (the forward kernels): the skip operands n_{l-1}, n_{l-2} of a lane's 12 elements are the outputs it wrote
        // in the two layers before -- kept in registers (nr1, nr2) instead of read back from LDS, and the columns beyond
        // the bands are zeroed by a select: no divergent branch around 12 dependent LDS round trips per layer
        if constexpr (REGSKIP) {
            if (l == 0) {  // n_{-1} = x: this lane's elements of the input image
#pragma unroll
                for (int m = 0; m < GM_MAXT; ++m)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int jt = wave + GM_WAVES * m;
                        nr1[m][e] = jt < g.nt ? src[(4 * rg + e) * g.pitch + 16 * jt + col] : 0.0f;
                        nr2[m][e] = 0.0f;
                    }
            }
        }
#pragma unroll
        for (int m = 0; m < GM_MAXT; ++m) {
            const int jt = wave + GM_WAVES * m;
            if (jt >= g.nt) break;
            const int j0 = 16 * jt;
            const gm_f32x4 acc = gm_conv_tile<false>(src, wz, g, j0, ksz, pad, lane);
            GM_FMARK(2)  // products
            const int c = j0 + col;
            [[maybe_unused]] gm_f32x4 kept = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = 4 * rg + e;
                const float v = acc[e] + bl;
                float y;
                if (last_tanh) {
                    y = tanhf(v);
                } else {
                    if (v > 0.0f) mk |= 1u << (4 * m + e);
                    y = v > 0.0f ? v : 0.1f * v;
                    if constexpr (REGSKIP) {
                        y += nr1[m][e];              // + n_{l-1}
                        if (l >= 1) y += nr2[m][e];  // + n_{l-2}
                    } else {
                        y += src[row * g.pitch + c];                // + n_{l-1}
                        if (l >= 1) y += skip2[row * g.pitch + c];  // + n_{l-2}
                    }
                }
                if constexpr (REGSKIP) {
                    y = c < g.bands ? y : 0.0f;
                    dst[row * g.pitch + c] = y;  // c < bp: inside the image, the padding columns keep their zeros
                    nr2[m][e] = nr1[m][e];
                    nr1[m][e] = y;
                    if (c < g.bands && row < rows_valid) {
                        if (out != nullptr && l == L - 1) out[(int64_t)row * ldo + c] = y;
                        if constexpr (TAP && !ENC) {
                            if (l == 3) enc_out[(int64_t)row * ld_enc + c] = y;
                        }
                    }
                } else {
                    if (c < g.bands) {
                        dst[row * g.pitch + c] = y;
                        if (out != nullptr && l == L - 1 && row < rows_valid) out[(int64_t)row * ldo + c] = y;
                        // encoder tap: n_4 IS the output of the encoder-only application on the same input
                        if constexpr (TAP && !ENC) {
                            if (l == 3 && row < rows_valid) enc_out[(int64_t)row * ld_enc + c] = y;
                        }
                    } else {
                        y = 0.0f;
                    }
                }
                if constexpr (KEEP) {
#pragma unroll
                    for (int q = 0; q < 6; ++q) keep[q][m][e] = q == l ? y : keep[q][m][e];
                }
                if constexpr (STASH) kept[e] = y;
            }
            if constexpr (STASH) {  // slots n_1..n_3 (encoder) / n_1..n_6 + the tanh output
                if (ENC ? l < 3 : true) stash[(l * GM_MAXT + m) * GM_THREADS] = kept;
            }
            GM_FMARK(3)  // epilogue
        }
        if constexpr (KEEP) {
#pragma unroll
            for (int q = 0; q < 7; ++q) mask[q] = q == l ? mk : mask[q];
        }
        if constexpr (STASH) {  // branch bits of layer l: word l of the two uint4 behind the slots
            if (l < (ENC ? 4 : 6))
                reinterpret_cast<unsigned*>(stash + GmKeep<ENC>::SLOTS * GM_MAXT * GM_THREADS + (l >> 2) * GM_THREADS)[l & 3] = mk;
        }
    }
    return L % 3;
}

__device__ __forceinline__ void gm_zero(float* p, int n, int tid) {
    for (int i = tid; i < n; i += GM_THREADS) p[i] = 0.0f;
}

__device__ __forceinline__ void gm_load_rows(float* img, const GmGeo g, const float* __restrict__ x, int64_t ldx,
                                             int rows_valid, int tid) {
    for (int i = tid; i < GM_ROWS * g.bp; i += GM_THREADS) {
        const int row = i / g.bp, c = i - row * g.bp;
        img[row * g.pitch + c] = (row < rows_valid && c < g.bands) ? x[(int64_t)row * ldx + c] : 0.0f;
    }
}

// The same copy in two halves: every thread REQUESTS its 12 elements (rows wave, wave + 8; columns lane + 64 k) before it
// stores the first -- gm_load_rows' loop is one dependent global round trip (and an integer division) per pass.
constexpr int GM_RCH = (GM_MAX_BANDS + 63) / 64;  // 64-column chunks of a row
__device__ __forceinline__ void gm_rows_issue(float (&v)[2][GM_RCH], const GmGeo g, const float* __restrict__ x, int64_t ldx,
                                              int rows_valid, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int k = 0; k < GM_RCH; ++k) {
            const int row = wave + GM_WAVES * rr, c = lane + 64 * k;
            v[rr][k] = 0.0f;
            if (64 * k < g.bands && row < rows_valid && c < g.bands) v[rr][k] = x[(int64_t)row * ldx + c];
        }
}
__device__ __forceinline__ void gm_rows_store(float* img, const GmGeo g, const float (&v)[2][GM_RCH], int tid) {
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int k = 0; k < GM_RCH; ++k) {
            const int row = wave + GM_WAVES * rr, c = lane + 64 * k;
            if (64 * k < g.bp && c < g.bp) img[row * g.pitch + c] = v[rr][k];
        }
}

// ---- activations kept for the backward pass ---------------------------------------------------------------------------
// A forward pass whose backward follows (hypel_gan_generator_fwd_keep) leaves what gm_forward<KEEP> holds in registers
// -- the layer outputs n_1.. in MFMA C layout, the leaky-ReLU branch bits, the tanh output -- in a caller-provided buffer,
// lane-native: float4 (slot, m) of thread `tid` of row tile t at ((t * GM_KEEP_V4(ENC) + slot * GM_MAXT + m) * 512 + tid)
// float4s, two uint4 of branch bits behind them.  The backward kernel then starts from that copy instead of recomputing
// the forward (29 % of its time at B = 360): 188 KB per 16 samples written and read once at HBM speed.
// Several applications of generators of ONE band count in one launch (hypel_gan_generator_*_apps): application g owns rows
// [g * n, (g + 1) * n) of every row-indexed operand, the variables at w + g * w_stride / bias + g * b_stride -- CycleGAN's
// G_x2y and G_y2x -- and gridDim.x / n_apps consecutive blocks (and gradient slabs); kept activations are indexed by the
// global tile number g * tiles + t.  At the Gulfport size (2048 x 64: 128 row tiles) a launch is one block's latency chain:
// two applications cost what one does.
// TAP (full generator only): a separate instantiation, so that the plain kernels keep their register counts
template <bool ENC, bool STASH, bool TAP = false>
__global__ __launch_bounds__(GM_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void gan_generator_fwd_mfma_kernel(const float* __restrict__ x, int64_t ldx,
                                                                            int64_t n, int bands,
                                                                            const float* __restrict__ w,
                                                                            const float* __restrict__ bias,
                                                                            float* __restrict__ out, int64_t ldo,
                                                                            float* __restrict__ stash,
                                                                            float* __restrict__ enc_out, int64_t ld_enc,
                                                                            GmApps apps) {
    extern __shared__ __attribute__((aligned(16))) float gm_lds[];
    const GmGeo g = gm_geo(bands);
    const int tid = threadIdx.x;
#if GM_DIAG == 7
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        for (int i = 0; i < 8; ++i) gm_fdbg[i] = 0;
        gm_fmark = clock64();
    }
#endif
    const int bpa = gridDim.x / apps.n_apps, app = blockIdx.x / bpa, blk = blockIdx.x - app * bpa;
    w += app * apps.w_stride;
    bias += app * apps.b_stride;
    x += (int64_t)app * n * ldx;
    out += (int64_t)app * n * ldo;
    if constexpr (TAP) enc_out += (int64_t)app * n * ld_enc;
    float* const bufs[3] = {gm_lds, gm_lds + GM_ROWS * g.pitch, gm_lds + 2 * GM_ROWS * g.pitch};
    constexpr bool ONE = GM_FWD_ONE_TABLE != 0;
    float* wz0 = gm_lds + 3 * GM_ROWS * g.pitch;
    float* wz1 = ONE ? wz0 : wz0 + 3 * g.bp;
    float* raw = wz1 + 3 * g.bp;
    static_assert(GM_WAVES == 8 && GM_ROWS == 16, "gm_rows_issue: two rows per wave");
    const int64_t tiles = (n + GM_ROWS - 1) / GM_ROWS;
    // Block set-up.  Wide spectra: the taps, the biases and the first row tile are REQUESTED first, the zero fill of the
    // images runs under those loads, their values go to LDS behind the barrier (before: three serial round trips for the
    // taps + a dozen for the rows; forward 46.2 -> 44.1 us at 4096 x 360).  Narrow ones (one or two 64-column chunks) keep
    // the plain order, which measured 0.7 us faster at 64 bands.
    const bool early = g.bands > 128;
    if (early) {
        const int wtotal = gm_woff(bands, 7);
        constexpr int GM_WCH = (GM_MAX_BANDS * 29 / 8 + GM_THREADS - 1) / GM_THREADS;  // sum k = 3.625 B taps
        float wv[GM_WCH], xr[2][GM_RCH];
#pragma unroll
        for (int k = 0; k < GM_WCH; ++k) {
            const int i = tid + GM_THREADS * k;
            wv[k] = i < wtotal ? w[i] : (i < wtotal + 7 ? bias[i - wtotal] : 0.0f);
        }
        if (blk < tiles)
            gm_rows_issue(xr, g, x + (int64_t)blk * GM_ROWS * ldx, ldx, (int)min((int64_t)GM_ROWS, n - (int64_t)blk * GM_ROWS), tid);
        gm_zero(gm_lds, 3 * GM_ROWS * g.pitch + (ONE ? 3 : 6) * g.bp, tid);  // image padding and tap margins stay zero
        __syncthreads();
#pragma unroll
        for (int k = 0; k < GM_WCH; ++k) {
            const int i = tid + GM_THREADS * k;
            if (i < wtotal + 8) raw[i] = wv[k];
        }
        if (blk < tiles) gm_rows_store(bufs[0], g, xr, tid);
        __syncthreads();  // the LDS copy of the taps is complete before the first layer builds its table
    } else {
        gm_zero(gm_lds, 3 * GM_ROWS * g.pitch + (ONE ? 3 : 6) * g.bp, tid);
        gm_stage_raw(raw, bands, w, bias, tid);
        __syncthreads();
        if (blk < tiles) gm_load_rows(bufs[0], g, x + (int64_t)blk * GM_ROWS * ldx, ldx, (int)min((int64_t)GM_ROWS, n - (int64_t)blk * GM_ROWS), tid);
    }
    GM_FMARK(0)  // block set-up
    float keep[6][GM_MAXT][4];
    unsigned mask[7];
    for (int64_t t = blk; t < tiles; t += bpa) {
        const int64_t r0 = t * GM_ROWS;
        const int rows_valid = (int)min((int64_t)GM_ROWS, n - r0);
        if (t != blk) gm_load_rows(bufs[0], g, x + r0 * ldx, ldx, rows_valid, tid);
        gm_f32x4* sp = nullptr;
        if constexpr (STASH)
            sp = reinterpret_cast<gm_f32x4*>(stash) + (size_t)(app * tiles + t) * GmKeep<ENC>::V4 * GM_THREADS + tid;
        gm_forward<ENC, false, GM_FWD_ROLLED != 0, STASH, ONE, TAP>(gm_lds, wz0, wz1, g, raw, raw + gm_woff(bands, 7),
                                                                    out + r0 * ldo, ldo, rows_valid, tid, keep, mask, sp,
                                                                    TAP ? enc_out + r0 * ld_enc : nullptr, ld_enc);
        __syncthreads();  // the next row tile overwrites bufs[0]
        GM_FMARK(5)  // last epilogue .. tile end
    }
#if GM_DIAG == 7
    if (threadIdx.x == 0 && blockIdx.x == 0)
        for (int i = 0; i < 8; ++i) out[i] = (float)gm_fdbg[i];
#endif
}

// Sum over the 64 lanes of a wave in registers (row_shr 1 / 2 / 4 / 8 inside the rows of 16, row_bcast15 / 31 across them:
// lane 63 ends up with the total), returned wave-uniform.  Six ds_bpermute round trips (__shfl_xor) per layer otherwise.
__device__ __forceinline__ float gm_wave_sum(float v) {
#define GM_DPP_ADD(ctrl, rows)                                                                                        \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rows, 0xf, false))
    GM_DPP_ADD(0x111, 0xf);
    GM_DPP_ADD(0x112, 0xf);
    GM_DPP_ADD(0x114, 0xf);
    GM_DPP_ADD(0x118, 0xf);
    GM_DPP_ADD(0x142, 0xa);
    GM_DPP_ADD(0x143, 0xc);
#undef GM_DPP_ADD
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Sample rows of the filter-gradient products: k-step s, k-slot kq multiply row gm_brow(kq) + s of the X and Z images.  The
// two k-slots a ds_read_b32 lane group serves (kq = 0, 1 / 2, 3) are 8 rows apart: with the images' pitch of 18 mod 32
// (chosen for the convolution tiles' fragments) that is a bank distance of 16, so the 2 x 16 consecutive columns of a
// group hit 32 banks (rows 4 s + kq collided in two banks per group: every fragment read took two LDS cycles more).
#ifndef GM_BROW
#define GM_BROW 1
#endif
__device__ __forceinline__ int gm_brow(int kq) { return GM_BROW ? 4 * (kq >> 1) + 8 * (kq & 1) : 4 * kq; }

// ---- filter gradient of one layer (step B of the backward kernels) ---------------------------------------------------
// G_a = sum_j X_{j+a}^T . Z_j for every tile offset a in [a_lo, a_hi] (X_t, Z_t: the 16-column tiles of the [16 x B] LDS
// images; tap t = d + pad is the sum of diagonal d of the G_a that contain it).  Work items are (offset, half of the
// column tiles) -- layers with fewer offsets than waves (k <= B / 4 at 360 bands: 5 - 7 offsets) cut the j axis at nt / 2,
// so that they still feed 8 waves -- in half-major, offset-minor order; the list is cut into GM_WAVES contiguous runs of
// nearly equal work (gm_wgrad_schedule, once per block) and a wave sweeps its run in sub-runs of <= GM_RUN CONSECUTIVE
// offsets: at column tile j the offsets a0 .. a0 + C - 1 multiply X tiles j + a0 .. j + a0 + C - 1 by the SAME Z tile, and
// the next j needs one new X tile and one new Z tile -- 8 fragment words for 4 C MFMAs where one (offset, j) pair at a
// time took 8 for 4 (round-4 phase stamps: this step 112 k cycles per row tile against 64 k of MFMA issue on its busiest
// SIMD).  An X tile outside [0, nt) enters as zeros (the ragged ends of a sub-run: C (C - 1) / 2 idle tile pairs).  Fixed
// order of the sums: j ascending inside an item, k-steps 0..3 (two chains 0,2 / 1,3 when C <= 2), halves added in order
// by gm_diag_sum.
#ifndef GM_RUN
#define GM_RUN 4
#endif
#ifndef GM_BSTEP
#define GM_BSTEP 2  // 2: pair at a time with gm_conv_tile's loop shape (gm_wgrad_tiles_v2, adopted: 186 -> 180 us at 8192 x 360);
                    // 0: the round-3 loop (gm_wgrad_tiles_v0, same sums bit for bit); 1: balanced runs of consecutive offsets (slower)
#endif
#ifndef GM_BSCHED
#define GM_BSCHED 1
#endif
#ifndef GM_BDIAG
#define GM_BDIAG 0  // timing diagnostics of the filter-gradient step (results garbage): 1 = no fragment reads in the j loop,
                    // 2 = no MFMAs
#endif
struct GmItems {
    int a_lo, n_off, halves, n_items, jmid;  // half 0: column tiles [0, jmid], half 1: (jmid, nt)
};
__device__ __forceinline__ GmItems gm_items(const GmGeo g, int ksz, int pad) {
    GmItems it;
    it.a_lo = -((pad + 15) / 16);
    const int a_hi = (ksz - 1 - pad + 15) / 16;
    it.n_off = a_hi - it.a_lo + 1;
    it.halves = (GM_BSTEP == 1 && it.n_off < GM_WAVES && 2 * it.n_off <= gm_atiles(g.bands) && g.nt >= 2) ? 2 : 1;
    it.n_items = it.n_off * it.halves;
    it.jmid = it.halves == 2 ? (g.nt - 1) / 2 : g.nt - 1;
    return it;
}
// column tiles [lo, hi] of item i (empty: lo > hi)
__device__ __forceinline__ void gm_item_range(const GmItems& it, int nt, int i, int& a, int& lo, int& hi) {
    const int h = i >= it.n_off ? 1 : 0;
    a = it.a_lo + (i - h * it.n_off);
    lo = max(h ? it.jmid + 1 : 0, -a);
    hi = min(h || it.halves == 1 ? nt - 1 : it.jmid, nt - 1 - a);
}
// sched[l * GM_WAVES + w] = first | last << 8: the items [first, last) of layer l that wave w sweeps.  Item i goes to wave
// floor((work before i + work_i / 2) * GM_WAVES / total) -- a running comparison, no division.
__device__ __forceinline__ void gm_wgrad_schedule(int* sched, const GmGeo g, int L, int tid) {
    if (tid >= L * GM_WAVES) return;
    const int l = tid / GM_WAVES, w = tid % GM_WAVES;
    const int ksz = gm_ksz(g.bands, l), pad = (ksz - 1) / 2;
    const GmItems it = gm_items(g, ksz, pad);
    int total = 0;
    for (int i = 0; i < it.n_items; ++i) {
        int a, lo, hi;
        gm_item_range(it, g.nt, i, a, lo, hi);
        total += max(0, hi - lo + 1);
    }
    int first = it.n_items, last = 0, cum = 0, owner = 0;
    for (int i = 0; i < it.n_items; ++i) {
        int a, lo, hi;
        gm_item_range(it, g.nt, i, a, lo, hi);
        const int wk = max(0, hi - lo + 1);
        if (total > 0) {
            while (owner < GM_WAVES - 1 && (owner + 1) * 2 * total <= (2 * cum + wk) * GM_WAVES) ++owner;
        } else {
            owner = i % GM_WAVES;
        }
        if (owner == w) {
            first = min(first, i);
            last = max(last, i + 1);
        }
        cum += wk;
    }
    if (first > last) first = last = 0;
    sched[tid] = first | (last << 8);
}

template <int C>
__device__ __forceinline__ void gm_wgrad_run(const float* __restrict__ X, const float* __restrict__ Z, float* __restrict__ G,
                                             const GmGeo g, const GmItems& it, int i0, int lane) {
    constexpr int NCH = C <= 2 ? 2 : 1;
    const int col = lane & 15, rg = lane >> 4;
    int a0, jA, jB;
    gm_item_range(it, g.nt, i0, a0, jA, jB);
    {
        int a, lo, hi;
        gm_item_range(it, g.nt, i0 + C - 1, a, lo, hi);
        jA = min(jA, lo);  // the union of the items' ranges: they differ only by the clipping of j + a to [0, nt)
        jB = max(jB, hi);
    }
    gm_f32x4 acc[C][NCH];
#pragma unroll
    for (int i = 0; i < C; ++i)
#pragma unroll
        for (int c = 0; c < NCH; ++c) acc[i][c] = gm_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (jA <= jB) {
        const float* xp = X + gm_brow(rg) * g.pitch + col;
        const float* zp = Z + gm_brow(rg) * g.pitch + col;
        // raw fragment of X tile t (address clamped); the caller replaces it by zeros where `ok` is false (wave-uniform)
        auto xtile = [&](int t, float (&f)[4]) -> bool {
            const bool ok = t >= 0 && t < g.nt;
            const int tc = ok ? t : 0;
#pragma unroll
            for (int s = 0; s < 4; ++s) f[s] = xp[16 * tc + s * g.pitch];
            return ok;
        };
        float xf[C][4], zf[4], xn[4], zn[4];
#pragma unroll
        for (int i = 0; i < C; ++i) {
            const bool ok = xtile(jA + a0 + i, xf[i]);
#pragma unroll
            for (int s = 0; s < 4; ++s) xf[i][s] = ok ? xf[i][s] : 0.0f;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) zf[s] = zp[16 * jA + s * g.pitch];
        for (int j = jA; j <= jB; ++j) {
            // the next column tile's two new fragments, requested ahead of this tile's products (the last trip
            // re-reads tile jB: no branch in the loop body)
            const int jn = min(j + 1, jB);
#if GM_BDIAG == 1
            const bool okn = true;
#pragma unroll
            for (int s = 0; s < 4; ++s) { xn[s] = xf[0][s] + 1.0f; zn[s] = zf[s] + 1.0f; }
#else
            const bool okn = xtile(jn + a0 + C - 1, xn);
#pragma unroll
            for (int s = 0; s < 4; ++s) zn[s] = zp[16 * jn + s * g.pitch];
#endif
#if GM_BSCHED
            __builtin_amdgcn_sched_barrier(0);  // hipcc otherwise sinks the eight reads below the MFMAs, right in front of their wait
#endif
#if GM_BDIAG == 2
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < C; ++i) acc[i][s % NCH][0] += xf[i][s] * zf[s];
#else
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < C; ++i)
                    acc[i][s % NCH] = __builtin_amdgcn_mfma_f32_16x16x4f32(xf[i][s], zf[s], acc[i][s % NCH], 0, 0, 0);
#endif
#if GM_BSCHED
            __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int i = 0; i + 1 < C; ++i) xf[i][s] = xf[i + 1][s];
                xf[C - 1][s] = okn ? xn[s] : 0.0f;
                zf[s] = zn[s];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < C; ++i) {
        if constexpr (NCH == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][0][e] += acc[i][1][e];
        }
        float* gt = G + (i0 + i) * 16 * GM_GP;  // [i_local = 4 rg + e][j_local = col]
#pragma unroll
        for (int e = 0; e < 4; ++e) gt[(4 * rg + e) * GM_GP + col] = acc[i][0][e];
    }
}

__device__ __forceinline__ void gm_wgrad_tiles(const float* __restrict__ X, const float* __restrict__ Z, float* __restrict__ G,
                                               const GmGeo g, int ksz, int pad, int run, int lane) {
    const GmItems it = gm_items(g, ksz, pad);
    int i0 = __builtin_amdgcn_readfirstlane(run & 0xff);
    const int last = __builtin_amdgcn_readfirstlane(run >> 8);
    while (i0 < last) {
        // a sub-run: <= GM_RUN consecutive offsets of ONE half
        const int h_end = i0 >= it.n_off ? it.n_items : it.n_off;
        const int c = min(GM_RUN, min(last, h_end) - i0);
        if (GM_RUN >= 4 && c >= 4) gm_wgrad_run<4>(X, Z, G, g, it, i0, lane);
        else if (GM_RUN >= 3 && c == 3) gm_wgrad_run<3>(X, Z, G, g, it, i0, lane);
        else if (c == 2) gm_wgrad_run<2>(X, Z, G, g, it, i0, lane);
        else gm_wgrad_run<1>(X, Z, G, g, it, i0, lane);
        i0 += c;
    }
}

// GM_BSTEP=0: the round-3 form of the step (one (offset, column tile) pair at a time, offsets dealt to the waves round
// robin), kept for A/B runs
__device__ __forceinline__ void gm_wgrad_tiles_v0(const float* __restrict__ X, const float* __restrict__ Z,
                                                  float* __restrict__ G, const GmGeo g, int ksz, int pad, int wave, int lane) {
    const int col = lane & 15, rg = lane >> 4;
    const int a_lo = -((pad + 15) / 16), a_hi = (ksz - 1 - pad + 15) / 16;
    for (int a = a_lo + wave; a <= a_hi; a += GM_WAVES) {
        gm_f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f}, accb = {0.0f, 0.0f, 0.0f, 0.0f};  // two chains (see gm_conv_tile)
        const int j_lo = max(0, -a), j_hi = min(g.nt - 1, g.nt - 1 - a);
        // A[i_local][n] = X[n][16 (jt + a) + i_local], B[n][j_local] = Z[n][16 jt + j_local]; sample row n of k-step s,
        // k-slot kq: gm_brow(kq) + s
        const float* ap = X + gm_brow(rg) * g.pitch + 16 * a + col;
        const float* bp = Z + gm_brow(rg) * g.pitch + col;
        float fa[4], fb[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            fa[s] = ap[16 * j_lo + s * g.pitch];
            fb[s] = bp[16 * j_lo + s * g.pitch];
        }
        for (int jt = j_lo; jt < j_hi; ++jt) {
            float na[4], nb[4];
#if GM_BDIAG == 1
#pragma unroll
            for (int s = 0; s < 4; ++s) { na[s] = fa[s] + 1.0f; nb[s] = fb[s] + 1.0f; }
#else
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                na[s] = ap[16 * (jt + 1) + s * g.pitch];
                nb[s] = bp[16 * (jt + 1) + s * g.pitch];
            }
#endif
#if GM_BDIAG == 2
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[s] += fa[s] * fb[s];
#else
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[0], fb[0], acc, 0, 0, 0);
            accb = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[1], fb[1], accb, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[2], fb[2], acc, 0, 0, 0);
            accb = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[3], fb[3], accb, 0, 0, 0);
#endif
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                fa[s] = na[s];
                fb[s] = nb[s];
            }
        }
        if (j_lo <= j_hi) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[0], fb[0], acc, 0, 0, 0);
            accb = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[1], fb[1], accb, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[2], fb[2], acc, 0, 0, 0);
            accb = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[3], fb[3], accb, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += accb[e];
        float* gt = G + (a - a_lo) * 16 * GM_GP;  // [i_local = 4 rg + e][j_local = col]
#pragma unroll
        for (int e = 0; e < 4; ++e) gt[(4 * rg + e) * GM_GP + col] = acc[e];
    }
}

#ifndef GM_V2_SCHED
#define GM_V2_SCHED 1  // the order fetch / mac / fetch / mac pinned (hipcc otherwise merges both fetches behind the first MFMAs
#endif                 // and waits for the second right in front of its use)
#if GM_V2_SCHED
#define GM_V2_PIN __builtin_amdgcn_sched_barrier(0);
#else
#define GM_V2_PIN
#endif
// GM_BSTEP=2: the same products in the same order as gm_wgrad_tiles_v0 (bit-identical), with the loop shape of gm_conv_tile:
// two fragment sets used alternately (no register copies), the next pair's eight reads requested ahead of this pair's MFMAs
// by a scalar two-pair software pipeline, one LDS base per sample-row step
__device__ __forceinline__ void gm_wgrad_tiles_v2(const float* __restrict__ X, const float* __restrict__ Z,
                                                  float* __restrict__ G, const GmGeo g, int ksz, int pad, int wave, int lane) {
    const int col = lane & 15, rg = lane >> 4;
    const int a_lo = -((pad + 15) / 16), a_hi = (ksz - 1 - pad + 15) / 16;
    const float* xb = X + gm_brow(rg) * g.pitch + col;
    const float* zb = Z + gm_brow(rg) * g.pitch + col;
    for (int a = a_lo + wave; a <= a_hi; a += GM_WAVES) {
        gm_f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f}, accb = {0.0f, 0.0f, 0.0f, 0.0f};
        const int j_lo = max(0, -a), j_hi = min(g.nt - 1, g.nt - 1 - a);
        const int nch = j_hi - j_lo + 1;
        if (nch > 0) {
            const float* ap = xb + 16 * (a + j_lo);
            const float* bp = zb + 16 * j_lo;
            float a0[4], b0[4], a1[4], b1[4];
            auto fetch = [&](int c, float (&fa)[4], float (&fb)[4]) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    fa[s] = ap[16 * c + s * g.pitch];
                    fb[s] = bp[16 * c + s * g.pitch];
                }
            };
            auto mac = [&](const float (&fa)[4], const float (&fb)[4]) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[0], fb[0], acc, 0, 0, 0);
                accb = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[1], fb[1], accb, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[2], fb[2], acc, 0, 0, 0);
                accb = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[3], fb[3], accb, 0, 0, 0);
            };
            fetch(0, a0, b0);
            int c = 0;
            for (; c + 2 < nch; c += 2) {
                fetch(c + 1, a1, b1);
                GM_V2_PIN
                mac(a0, b0);
                GM_V2_PIN
                fetch(c + 2, a0, b0);
                GM_V2_PIN
                mac(a1, b1);
                GM_V2_PIN
            }
            if (c + 1 < nch) {  // two pairs left
                fetch(c + 1, a1, b1);
                mac(a0, b0);
                mac(a1, b1);
            } else {
                mac(a0, b0);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += accb[e];
        float* gt = G + (a - a_lo) * 16 * GM_GP;  // [i_local = 4 rg + e][j_local = col]
#pragma unroll
        for (int e = 0; e < 4; ++e) gt[(4 * rg + e) * GM_GP + col] = acc[e];
    }
}

// tap `tap` of the layer: the sum of diagonal d = tap - pad of G, halves ascending, tiles ascending, rows ascending
__device__ __forceinline__ float gm_diag_sum(const float* __restrict__ G, const GmGeo g, int ksz, int pad, int tap) {
    const GmItems it = gm_items(g, ksz, pad);
    const int a_hi = it.a_lo + it.n_off - 1;
    const int d = tap - pad;
    const int a0 = (d + 15 >= 0 ? (d + 15) / 16 : -((-(d + 15) + 15) / 16));  // floor((d + 15) / 16)
    float total = 0.0f;
#pragma unroll 1
    for (int h = 0; h < it.halves; ++h) {
        // Diagonal d crosses the tiles a0 - 1 (rows dl0 .. 15, dl0 = d - 16 (a0 - 1) in [1, 16]) and a0 (rows 0 .. dl0 - 1):
        // 16 elements, row il from tile a0 when il < dl0.  All 16 are fetched before the first add (as nested loops with
        // data-dependent bounds this was <= 32 dependent LDS round trips per thread and layer); an element of an absent
        // tile is read from G's first tile and replaced by 0 (volatile: hipcc otherwise sinks every load under its
        // predicate -- a branch and an LDS round trip each).
        const int dl0 = d - 16 * (a0 - 1);
        const bool ok0 = a0 - 1 >= it.a_lo && a0 - 1 <= a_hi, ok1 = a0 >= it.a_lo && a0 <= a_hi;
        // element il of tile t sits at gt_t[il * (GM_GP + 1) - dl_t]
        const float* b0 = G + (ok0 ? h * it.n_off + a0 - 1 - it.a_lo : 0) * 16 * GM_GP - (ok0 ? dl0 : 0);
        const float* b1 = G + (ok1 ? h * it.n_off + a0 - it.a_lo : 0) * 16 * GM_GP - (ok1 ? dl0 - 16 : 0);
        typedef const volatile __attribute__((address_space(3))) float* gm_lds_vptr;
        // (two loops: with the select next to its load hipcc waited for every load before it issued the next one -- 16
        // serial LDS round trips, 1.8 k cycles per layer whatever the band count)
        float v[16];
#pragma unroll
        for (int il = 0; il < 16; ++il) v[il] = ((gm_lds_vptr)(il < dl0 ? b1 : b0))[il * (GM_GP + 1)];
        float s = 0.0f;
#pragma unroll
        for (int il = 0; il < 16; ++il) s += ((il < dl0 ? ok1 : ok0) ? v[il] : 0.0f);  // rows ascending
        total = h == 0 ? s : total + s;
    }
    return total;
}

// ---- backward ------------------------------------------------------------------------------------------------------
// pw[blocks][sum k], pb[blocks][8]: this block's partial filter / bias gradients (summed over its row tiles).
template <bool ENC, bool STASH, bool TAP = false>
__global__ __launch_bounds__(GM_THREADS) void gan_generator_bwd_mfma_kernel(
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ dout, int64_t lddo, int64_t n, int bands,
    const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ dx, int64_t lddx, int accumulate_dx,
    float* __restrict__ pw, float* __restrict__ pb, int wtotal, int slabs, const float* __restrict__ stash,
    const float* __restrict__ d_enc, int64_t ld_denc, GmApps apps) {
    constexpr int L = ENC ? 4 : 7;
    extern __shared__ __attribute__((aligned(16))) float gm_lds[];
    const GmGeo g = gm_geo(bands);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bpa = gridDim.x / apps.n_apps, app = blockIdx.x / bpa, blk = blockIdx.x - app * bpa;
    w += app * apps.w_stride;
    bias += app * apps.b_stride;
    x += (int64_t)app * n * ldx;
    dout += (int64_t)app * n * lddo;
    if (dx != nullptr) dx += (int64_t)app * n * lddx;
    if constexpr (TAP) d_enc += (int64_t)app * n * ld_denc;
    const int col = lane & 15, rg = lane >> 4;
    const int img = GM_ROWS * g.pitch;
    float* const bufs[3] = {gm_lds, gm_lds + img, gm_lds + 2 * img};  // forward images, then the gradient ring
    float* Z = gm_lds + 3 * img;   // dz_l
    float* X = gm_lds + 4 * img;   // n_{l-1}
    float* wz = gm_lds + 5 * img;
    float* wz2 = wz + 3 * g.bp;    // second tap table of the forward recompute
    float* G = wz2 + 3 * g.bp;     // [a][16][GM_GP]
    float* red = G + gm_atiles(bands) * 16 * GM_GP;  // [GM_WAVES] bias-gradient partials, [GM_WAVES + l] their sums
    int* sched = reinterpret_cast<int*>(red + 64);    // [L][GM_WAVES] filter-gradient runs (gm_wgrad_schedule)
    float* raw = red + 128;                           // LDS copy of every layer's taps + biases
    gm_zero(gm_lds, 5 * img + 6 * g.bp, tid);
    gm_stage_raw(raw, bands, w, bias, tid);
    if (GM_BSTEP == 1) gm_wgrad_schedule(sched, g, L, tid);
    __syncthreads();

    float dwacc[7];  // thread t owns tap t of every layer; red[GM_WAVES + l] collects the bias gradients
#pragma unroll
    for (int l = 0; l < 7; ++l) dwacc[l] = 0.0f;
    if (tid < 7) red[GM_WAVES + tid] = 0.0f;

#if GM_DIAG == 5
    long long dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tmark = clock64();
#define GM_MARK(i) { const long long now_ = clock64(); dbg[i] += now_ - tmark; tmark = now_; }
#else
#define GM_MARK(i)
#endif
    const int64_t tiles = (n + GM_ROWS - 1) / GM_ROWS;
    for (int64_t t = blk; t < tiles; t += bpa) {
        const int64_t r0 = t * GM_ROWS;
        const int rows_valid = (int)min((int64_t)GM_ROWS, n - r0);
        GM_MARK(0)
        float keep[6][GM_MAXT][4];
        unsigned mask[7];
        [[maybe_unused]] float ylast[GM_MAXT][4];  // STASH: the tanh output of this lane's elements
        int res = 0;
        if constexpr (STASH) {  // the forward pass left its activations behind (see GmKeep)
            const gm_f32x4* sp =
                reinterpret_cast<const gm_f32x4*>(stash) + (size_t)(app * tiles + t) * GmKeep<ENC>::V4 * GM_THREADS + tid;
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int m = 0; m < GM_MAXT; ++m) {
                    gm_f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (q < (ENC ? 3 : 6) && wave + GM_WAVES * m < g.nt) v = sp[(q * GM_MAXT + m) * GM_THREADS];
#pragma unroll
                    for (int e = 0; e < 4; ++e) keep[q][m][e] = v[e];
                }
            if constexpr (!ENC) {
#pragma unroll
                for (int m = 0; m < GM_MAXT; ++m) {
                    gm_f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (wave + GM_WAVES * m < g.nt) v = sp[(6 * GM_MAXT + m) * GM_THREADS];
#pragma unroll
                    for (int e = 0; e < 4; ++e) ylast[m][e] = v[e];
                }
            }
            typedef unsigned gm_u32x4 __attribute__((ext_vector_type(4)));
            const gm_u32x4* mp = reinterpret_cast<const gm_u32x4*>(sp + GmKeep<ENC>::SLOTS * GM_MAXT * GM_THREADS);
            const gm_u32x4 m0 = mp[0];
            mask[0] = m0[0]; mask[1] = m0[1]; mask[2] = m0[2]; mask[3] = m0[3];
            mask[4] = mask[5] = mask[6] = 0u;
            if constexpr (!ENC) {
                const gm_u32x4 m1 = mp[GM_THREADS];
                mask[4] = m1[0]; mask[5] = m1[1];
            }
        } else {
            gm_load_rows(bufs[0], g, x + r0 * ldx, ldx, rows_valid, tid);
            res = gm_forward<ENC, true>(gm_lds, wz, wz2, g, raw, raw + gm_woff(bands, 7), nullptr, 0, rows_valid, tid, keep,
                                        mask);
        }
        __syncthreads();
        // gradient ring: Da = dn_l (complete), Db = partial dn_{l-1}, Dc = dn_{l-2} being initialised
        float* Da = bufs[(res + 1) % 3];
        float* Db = bufs[(res + 2) % 3];
        float* Dc = bufs[res];  // still holds the forward result until the top layer has read it
        const float* fwd_out = bufs[res];
        GM_MARK(1)  // forward recompute
        gm_load_rows(Da, g, dout + r0 * lddo, lddo, rows_valid, tid);
        __syncthreads();
        GM_MARK(2)  // dout load
        int woff = gm_woff(bands, L);
        // One rolled loop over the layers (the kept activations / branch bits of layer l are picked by wave-uniform
        // selects): unrolled, the seven bodies needed > 200 scalar and > 200 vector spill slots.
#pragma unroll 1
        for (int l = L - 1; l >= 0; --l) {
            const int ksz = gm_ksz(bands, l), pad = (ksz - 1) / 2;
            woff -= ksz;
            const bool top_tanh = !ENC && l == 6;
            const bool init_b = l == L - 1 || (!ENC && l == 5);  // the top skip layer initialises dn_{l-1}
            unsigned mk = 0;
#pragma unroll
            for (int q = 0; q < 7; ++q) mk = q == l ? mask[q] : mk;
            // encoder tap: the gradient that reached the encoder-only application's output joins dn_4 (complete in Da when
            // layer 3 begins) -- a pass of its own over the tile: inside step A the extra operand cost 34 registers
            if constexpr (TAP && !ENC) {
                if (l == 3) {
                    for (int i = tid; i < GM_ROWS * g.bp; i += GM_THREADS) {
                        const int row = i / g.bp, c = i - row * g.bp;
                        if (row < rows_valid && c < g.bands) Da[row * g.pitch + c] += d_enc[(r0 + row) * ld_denc + c];
                    }
                    __syncthreads();
                }
            }
            // ---- step A: dz_l, skip gradients, bias gradient; n_{l-1} from the registers (or x) into X ----
            gm_fill_taps(wz, g, raw + woff, ksz, tid);
            float dbl = 0.0f;
#pragma unroll
            for (int m = 0; m < GM_MAXT; ++m) {
                const int jt = wave + GM_WAVES * m;
                if (jt >= g.nt) break;
                const int c = 16 * jt + col;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int o = (4 * rg + e) * g.pitch + c;
                    float xin = 0.0f;
#pragma unroll
                    for (int q = 0; q < 6; ++q) xin = q == l - 1 ? keep[q][m][e] : xin;
                    if (c < g.bands) {
                        const float gd = Da[o];
                        float f;
                        if (top_tanh) {
                            float y;
                            if constexpr (STASH) y = ylast[m][e];
                            else y = fwd_out[o];
                            f = 1.0f - y * y;
                        } else {
                            f = ((mk >> (4 * m + e)) & 1u) ? 1.0f : 0.1f;
                        }
                        const float z = gd * f;
                        Z[o] = z;
                        dbl += z;
                        if (top_tanh) {
                            Db[o] = 0.0f;  // n6 only feeds the last convolution
                        } else {
                            // n_l = c_l + n_{l-1} (+ n_{l-2})
                            Db[o] = init_b ? gd : Db[o] + gd;
                            if (l >= 1) Dc[o] = gd;
                        }
                        if (l >= 1) X[o] = xin;
                    }
                }
            }
            if (l == 0) gm_load_rows(X, g, x + r0 * ldx, ldx, rows_valid, tid);
            // bias gradient: lanes -> wave -> block, fixed order
            dbl = gm_wave_sum(dbl);
            if (lane == 0) red[wave] = dbl;
            __syncthreads();
            if (tid == 0) {
                float s = 0.0f;
#pragma unroll
                for (int wv = 0; wv < GM_WAVES; ++wv) s += red[wv];
                red[GM_WAVES + l] += s;
            }
            GM_MARK(3)  // step A
            // ---- step B: filter gradient (gm_wgrad_tiles: balanced runs of consecutive tile offsets) ----
#if GM_BSTEP == 1
            gm_wgrad_tiles(X, Z, G, g, ksz, pad, sched[l * GM_WAVES + wave], lane);
#elif GM_BSTEP == 2
            gm_wgrad_tiles_v2(X, Z, G, g, ksz, pad, wave, lane);
#else
            gm_wgrad_tiles_v0(X, Z, G, g, ksz, pad, wave, lane);
#endif
            __syncthreads();
            GM_MARK(4)  // step B products
#ifndef GM_NO_DIAGSUM
#define GM_NO_DIAGSUM 0
#endif
            if (!GM_NO_DIAGSUM && tid < ksz) {  // thread t owns tap t
                const float s = gm_diag_sum(G, g, ksz, pad, tid);
#pragma unroll
                for (int q = 0; q < 7; ++q)
                    if (q == l) dwacc[q] += s;
            }
            GM_MARK(5)  // diagonal sums
            // ---- step C: data gradient dn_{l-1} += dz_l . T^T ----
            if (l > 0 || dx != nullptr) {
#pragma unroll
                for (int m = 0; m < GM_MAXT; ++m) {
                    const int jt = wave + GM_WAVES * m;
                    if (jt >= g.nt) break;
                    const gm_f32x4 acc = gm_conv_tile<true>(Z, wz, g, 16 * jt, ksz, pad, lane);
                    const int c = 16 * jt + col;
                    if (c < g.bands) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) Db[(4 * rg + e) * g.pitch + c] += acc[e];
                    }
                }
            }
            __syncthreads();
            GM_MARK(6)  // step C
            float* old = Da;
            Da = Db;
            Db = Dc;
            Dc = old;
        }
        if (dx != nullptr) {
            for (int i = tid; i < GM_ROWS * g.bp; i += GM_THREADS) {
                const int row = i / g.bp, c = i - row * g.bp;
                if (row < rows_valid && c < g.bands) {
                    float* p = dx + (r0 + row) * lddx + c;
                    const float v = Da[row * g.pitch + c];
                    *p = accumulate_dx ? *p + v : v;
                }
            }
        }
        __syncthreads();
    }
    // this block's partials; the slabs of the blocks beyond the grid (the planner's reduce sums `slabs` of them) are zeros
    for (int sb = blockIdx.x + gridDim.x; sb < slabs; sb += gridDim.x) {  // (single application only: slabs == gridDim.x else)
        for (int i = tid; i < wtotal; i += GM_THREADS) pw[(size_t)sb * wtotal + i] = 0.0f;
        if (tid < 8) pb[(size_t)sb * 8 + tid] = 0.0f;
    }
    const int64_t slab = apps.pw_stride ? (int64_t)blk : (int64_t)blockIdx.x;
    pw += app * apps.pw_stride;
    pb += app * apps.pb_stride;
    float* pwb = pw + (size_t)slab * wtotal;
    int woff = 0;
#pragma unroll
    for (int l = 0; l < 7; ++l) {
        const int ksz = gm_ksz(bands, l);
        if (tid < ksz) pwb[woff + tid] = l < L ? dwacc[l] : 0.0f;
        woff += ksz;
    }
    if (tid < 8) pb[(size_t)slab * 8 + tid] = tid < L ? red[GM_WAVES + tid] : 0.0f;
#if GM_DIAG == 5
    GM_MARK(7)
    if (tid == 0 && blockIdx.x == 0)
        for (int i = 0; i < 8; ++i) pb[i] = (float)dbg[i];
#endif
}

// Block barrier of the register-ring backward kernel's layer loop: every wave's LDS accesses are complete, its global
// loads (the next layer's kept activations, taps and branch bits, requested a layer ahead) stay in flight --
// __syncthreads() waits for vmcnt(0) too, which put one exposed HBM round trip into every layer.
#ifndef GM_LDS_BARRIER
#define GM_LDS_BARRIER 1
#endif
#ifndef GM_KEEP_B128
#define GM_KEEP_B128 1  // 16-byte loads of the kept activations (0: four dword buffer loads; 3 % slower)
#endif
__device__ __forceinline__ void gm_lds_barrier() {
#if GM_LDS_BARRIER
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
    __syncthreads();
#endif
}

// ---- backward from kept activations, two resident blocks per CU (round 4) -----------------------------------------------
// The round-3 kernel above keeps five [16 x B] images (three of them the gradient ring dn_l / dn_{l-1} / dn_{l-2}), two tap
// tables and every layer's taps in LDS: 161 KB at 360 bands, ONE block per CU, and its per-layer bookkeeping (step A: 12
// elements per lane, each an LDS read-modify-write chain inside a divergent branch; the sub-run set-up of step B; the
// diagonal sums) is ~60 % of its cycles with the matrix pipe idle (phase stamps, NOTES 4.J).  This kernel serves the form
// the train ops run (activations kept by the forward pass) with the ring in REGISTERS: a lane owns the same 12 elements
// (MFMA C layout) in every layer -- step A reads dn_l from registers, step C adds the data-gradient tile to registers --
// and of the kept activations only the layer at hand is resident (12 floats, requested one layer ahead, like the taps and
// the branch bits).  LDS: dz_l, n_{l-1}, ONE tap table, the G tiles = 80.6 KB at 360 bands, <= 128 registers: two
// blocks per CU, the bookkeeping of one under the products of the other.  Same sums in the same order as the kernel
// above (dn_{l-1} = (dn_{l-1} + dn_l) + dz_l . T^T), so the two stay bit-identical.
__host__ __device__ inline size_t gm_bwd2_lds(int bands) {
    const GmGeo g = gm_geo(bands);
    return sizeof(float) * (2 * (size_t)GM_ROWS * g.pitch + 3 * (size_t)g.bp + (size_t)gm_atiles(bands) * 16 * GM_GP + 128);
}

template <bool ENC, bool TAP = false>
__global__ __launch_bounds__(GM_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void gan_generator_bwd2_mfma_kernel(
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ dout, int64_t lddo, int64_t n, int bands,
    const float* __restrict__ w, float* __restrict__ dx, int64_t lddx, int accumulate_dx, float* __restrict__ pw,
    float* __restrict__ pb, int wtotal, int slabs, const float* __restrict__ stash, const float* __restrict__ d_enc,
    int64_t ld_denc, GmApps apps) {
    constexpr int L = ENC ? 4 : 7;
    extern __shared__ __attribute__((aligned(16))) float gm_lds[];
    const GmGeo g = gm_geo(bands);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bpa = gridDim.x / apps.n_apps, app = blockIdx.x / bpa, blk = blockIdx.x - app * bpa;
    w += app * apps.w_stride;
    x += (int64_t)app * n * ldx;
    dout += (int64_t)app * n * lddo;
    if (dx != nullptr) dx += (int64_t)app * n * lddx;
    if constexpr (TAP) d_enc += (int64_t)app * n * ld_denc;
    const int col = lane & 15, rg = lane >> 4;
    const int img = GM_ROWS * g.pitch;
    float* Z = gm_lds;        // dz_l
    float* X = gm_lds + img;  // n_{l-1}
    float* wz = gm_lds + 2 * img;
    float* G = wz + 3 * g.bp;  // [item][16][GM_GP]
    float* red = G + gm_atiles(bands) * 16 * GM_GP;  // [GM_WAVES] bias-gradient partials, [GM_WAVES + l] their sums
    int* sched = reinterpret_cast<int*>(red + 64);   // [L][GM_WAVES] filter-gradient runs
    gm_zero(gm_lds, 2 * img + 3 * g.bp, tid);  // image padding and tap margins stay zero
    if (GM_BSTEP == 1) gm_wgrad_schedule(sched, g, L, tid);
    if (tid < 7) red[GM_WAVES + tid] = 0.0f;
    __syncthreads();

    float dwacc[7];  // thread t owns tap t of every layer
#pragma unroll
    for (int l = 0; l < 7; ++l) dwacc[l] = 0.0f;
    const int64_t tiles = (n + GM_ROWS - 1) / GM_ROWS;
    constexpr int kOOB = 0x7fffffff;
#if GM_DIAG == 5
    long long dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tmark = clock64();
#endif
#if GM_DIAG == 6  // occupancy timeline: every block reports its start / end (100 MHz wall clock) and where it ran
    const long long wall0 = wall_clock64();
#endif
    for (int64_t t = blk; t < tiles; t += bpa) {
        const int64_t r0 = t * GM_ROWS;
        GM_MARK(0)
        const int rows_valid = (int)min((int64_t)GM_ROWS, n - r0);
        // (opaque copies: hipcc otherwise hoists the ~50 element addresses of the tile and layer loops out of them and
        // spills them)
        int colt = col, rgt = rg, tidt = tid;
        asm volatile("" : "+v"(colt), "+v"(rgt), "+v"(tidt));
        // Row-indexed operands through raw buffer accesses (scalar descriptor per tile, one 32-bit offset per element, rows
        // beyond the batch and columns beyond the bands out of range: loads return 0, stores are dropped) -- with 64-bit
        // addresses for 12 elements per lane the kernel does not fit 128 registers.
        auto tile_rsrc = [&](const float* base, int64_t ld) {
            return __builtin_amdgcn_make_buffer_rsrc((void*)(base + r0 * ld), 0,
                                                     __builtin_amdgcn_readfirstlane(((rows_valid - 1) * (int)ld + g.bands) * 4),
                                                     0x00020000);
        };
        auto elem_off = [&](int m, int e, int ld) {  // byte offset of element (row 4 rg + e, column tile wave + 8 m)
            const int c = 16 * (wave + GM_WAVES * m) + colt;
            return c < g.bands ? ((4 * rgt + e) * ld + c) * 4 : kOOB;
        };
        // kept activations of this row tile: float4 (slot q, m) of thread tid at ((q * GM_MAXT + m) * 512 + tid) * 16 bytes
        const __amdgpu_buffer_rsrc_t keep_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(stash + (size_t)(app * tiles + t) * GmKeep<ENC>::V4 * GM_THREADS * 4), 0,
            GmKeep<ENC>::V4 * GM_THREADS * 16, 0x00020000);
        auto load_slot = [&](int q, float (&v)[GM_MAXT][4]) {  // tiles beyond nt hold nothing
#if GM_KEEP_B128
            // one 16-byte load per column tile: scalar base + a 32-bit lane offset (a dwordx4 RAW BUFFER load under the
            // dword descriptor returned only its first component)
            const char* kb = reinterpret_cast<const char*>(stash) +
                             ((size_t)(app * tiles + t) * GmKeep<ENC>::V4 + (size_t)q * GM_MAXT) * GM_THREADS * 16;
#pragma unroll
            for (int m = 0; m < GM_MAXT; ++m) {
                gm_f32x4 f = {0.0f, 0.0f, 0.0f, 0.0f};
                if (wave + GM_WAVES * m < g.nt)
                    f = *reinterpret_cast<const gm_f32x4*>(kb + (unsigned)(tidt * 16 + m * GM_THREADS * 16));
#pragma unroll
                for (int e = 0; e < 4; ++e) v[m][e] = f[e];
            }
#else
            // (dword loads: a dwordx4 raw buffer load under this dword descriptor returned only its first component)
            const int so = __builtin_amdgcn_readfirstlane(q * GM_MAXT * GM_THREADS * 16);
#pragma unroll
            for (int m = 0; m < GM_MAXT; ++m) {
                const int vo = wave + GM_WAVES * m < g.nt ? tidt * 16 + m * GM_THREADS * 16 : kOOB;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    v[m][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(keep_rsrc, vo + 4 * e, so, 0));
            }
#endif
        };
        auto load_mask = [&](int l) {  // branch bits of layer l: word l of the two uint4 behind the slots
            return (unsigned)__builtin_amdgcn_raw_buffer_load_b32(
                keep_rsrc, tid * 16 + (l & 3) * 4,
                __builtin_amdgcn_readfirstlane((GmKeep<ENC>::SLOTS * GM_MAXT + (l >> 2)) * GM_THREADS * 16), 0);
        };
        float da[GM_MAXT][4], db[GM_MAXT][4], xin[GM_MAXT][4];
        {  // dn_L = dout, straight into the lanes that own it; zero outside the real rows / bands
            const __amdgpu_buffer_rsrc_t rs = tile_rsrc(dout, lddo);
#pragma unroll
            for (int m = 0; m < GM_MAXT; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    da[m][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, elem_off(m, e, (int)lddo), 0, 0));
                    db[m][e] = 0.0f;
                }
        }
        load_slot(ENC ? 2 : 6, xin);  // the top layer's operand: n_3's input n_2 (encoder) / the tanh output (full stack)
        int woff = gm_woff(bands, L);
        float wreg = 0.0f;  // tap `tid` of the layer at hand, requested one layer ahead
        {
            const int k0 = gm_ksz(bands, L - 1);
            if (tid < k0) wreg = w[woff - k0 + tid];
        }
        unsigned mk = ENC ? load_mask(3) : 0u;  // branch bits of the layer at hand (the tanh layer has none)
        GM_MARK(2)  // tile set-up: dout and the top layer's operands requested
#pragma unroll 1
        for (int l = L - 1; l >= 0; --l) {
            const int ksz = gm_ksz(bands, l), pad = (ksz - 1) / 2;
            woff -= ksz;
            const bool top_tanh = !ENC && l == 6;
            const bool init_b = l == L - 1 || (!ENC && l == 5);  // the top skip layer initialises dn_{l-1}
            int lbase = 4 * rg * g.pitch + 16 * wave + col;  // this lane's element (m = 0, e = 0) in a [16 x pitch] image
            int lane_l = lane, tid_l = tid;  // (opaque per layer: nothing derived from them is hoisted out of the layer loop)
            asm volatile("" : "+v"(lbase), "+v"(lane_l), "+v"(tid_l));
            if constexpr (TAP && !ENC) {
                // encoder tap: the gradient that reached the encoder-only application's output joins dn_4
                if (l == 3) {
                    const __amdgpu_buffer_rsrc_t rs = tile_rsrc(d_enc, ld_denc);
#pragma unroll
                    for (int m = 0; m < GM_MAXT; ++m)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            da[m][e] += __builtin_bit_cast(
                                float, __builtin_amdgcn_raw_buffer_load_b32(rs, elem_off(m, e, (int)ld_denc), 0, 0));
                }
            }
            // ---- step A: dz_l and n_{l-1} into LDS, skip gradients in registers, bias gradient ----
            for (int i = tid; i < g.bands; i += GM_THREADS) wz[g.bp + i] = i < ksz ? wreg : 0.0f;
            // (branch-free element loop: as per-element `if (top_tanh) / if (init_b) / if (l >= 1)` hipcc compiled three to four
            // scalar branches and a vmcnt(0) wait around every one of the 12 elements)
            float dbl = 0.0f;
#pragma unroll
            for (int m = 0; m < GM_MAXT; ++m) {
                const int jt = wave + GM_WAVES * m;
                if (jt >= g.nt) break;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int o = lbase + e * g.pitch + 16 * GM_WAVES * m;
                    const float gd = da[m][e];
                    const float y = xin[m][e];  // the tanh output under the top layer (else n_{l-1}, unused here)
                    const float ft = 1.0f - y * y;
                    const float fl = ((mk >> (4 * m + e)) & 1u) ? 1.0f : 0.1f;
                    const float z = gd * (top_tanh ? ft : fl);
                    Z[o] = z;
                    dbl += z;
                    // n_l = c_l + n_{l-1} (+ n_{l-2}); n_6 only feeds the last convolution
                    const float nb = init_b ? gd : db[m][e] + gd;
                    db[m][e] = gd;                     // dn_{l-2} starts from dn_l (unused below layer 1 and under the tanh layer)
                    da[m][e] = top_tanh ? 0.0f : nb;   // dn_{l-1} so far; step C adds dz_l . T^T
                }
            }
            if (top_tanh) load_slot(5, xin);  // the tanh layer's input is n_5: its own slot, requested only now (the registers held y)
            if (l >= 1) {
#pragma unroll
                for (int m = 0; m < GM_MAXT; ++m) {
                    const int jt = wave + GM_WAVES * m;
                    if (jt >= g.nt) break;
#pragma unroll
                    for (int e = 0; e < 4; ++e) X[lbase + e * g.pitch + 16 * GM_WAVES * m] = xin[m][e];
                }
            }
            GM_MARK(1)  // (diagnostics) step A up to the element loop's end
            if (l == 0) gm_load_rows(X, g, x + r0 * ldx, ldx, rows_valid, tid);
            // next layer's operands: its input n_{l-2}, its taps, its branch bits
#ifndef GM_ADIAG
#define GM_ADIAG 0  // timing diagnostics (results garbage): 1 = no requests for the next layer's operands
#endif
            if (!GM_ADIAG && l >= 2) load_slot(l - 2, xin);
            if (!GM_ADIAG && l >= 1) {
                const int k1 = gm_ksz(bands, l - 1);
                wreg = tid < k1 ? w[woff - k1 + tid] : 0.0f;
                mk = load_mask(l - 1);
            }
            // bias gradient: lanes -> wave -> block, fixed order
            dbl = gm_wave_sum(dbl);
            if (lane == 0) red[wave] = dbl;
            gm_lds_barrier();
            if (tid == 0) {
                float s = 0.0f;
#pragma unroll
                for (int wv = 0; wv < GM_WAVES; ++wv) s += red[wv];
                red[GM_WAVES + l] += s;
            }
            GM_MARK(3)  // step A
            // ---- step C: data gradient dn_{l-1} += dz_l . T^T (this wave's column tiles, into its registers) ----
            // Steps C and B only READ dz_l / n_{l-1} / the tap table and are independent of each other: no barrier between
            // them (a wave's MFMA-dense data-gradient tiles run beside other waves' latency-bound filter-gradient items);
            // the diagonal sums behind the barrier read G only, so the next layer's step A may overwrite the images beside
            // them -- two block barriers per layer
            if (l > 0 || dx != nullptr) {
#pragma unroll
                for (int m = 0; m < GM_MAXT; ++m) {
                    const int jt = wave + GM_WAVES * m;
                    if (jt >= g.nt) break;
                    const gm_f32x4 acc = gm_conv_tile<true>(Z, wz, g, 16 * jt, ksz, pad, lane_l);
                    if (16 * jt + (lane_l & 15) < g.bands) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) da[m][e] += acc[e];
                    }
                }
            }
            GM_MARK(6)  // step C
            // ---- step B: filter gradient ----
#if GM_BSTEP == 1
            gm_wgrad_tiles(X, Z, G, g, ksz, pad, sched[l * GM_WAVES + wave], lane_l);
#elif GM_BSTEP == 2
            gm_wgrad_tiles_v2(X, Z, G, g, ksz, pad, wave, lane_l);
#else
            gm_wgrad_tiles_v0(X, Z, G, g, ksz, pad, wave, lane_l);
#endif
            GM_MARK(7)  // (diagnostics) step B: wave 0's own products
            gm_lds_barrier();  // every wave has left Z / X / wz; G is complete
            GM_MARK(4)  // step B products
            if (tid_l < ksz) {
                const float s = gm_diag_sum(G, g, ksz, pad, tid_l);
#pragma unroll
                for (int q = 0; q < 7; ++q)
                    if (q == l) dwacc[q] += s;
            }
            GM_MARK(5)  // diagonal sums
        }
        if (dx != nullptr) {
            const __amdgpu_buffer_rsrc_t rs = tile_rsrc(dx, lddx);
#pragma unroll
            for (int m = 0; m < GM_MAXT; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int o = elem_off(m, e, (int)lddx);
                    float v = da[m][e];
                    if (accumulate_dx) v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, o, 0, 0));
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), rs, o, 0, 0);
                }
        }
    }
    // this block's partials; the slabs of the blocks beyond the grid (the planner's reduce sums `slabs` of them) are zeros
    for (int sb = blockIdx.x + gridDim.x; sb < slabs; sb += gridDim.x) {  // (single application only: slabs == gridDim.x else)
        for (int i = tid; i < wtotal; i += GM_THREADS) pw[(size_t)sb * wtotal + i] = 0.0f;
        if (tid < 8) pb[(size_t)sb * 8 + tid] = 0.0f;
    }
    const int64_t slab = apps.pw_stride ? (int64_t)blk : (int64_t)blockIdx.x;
    pw += app * apps.pw_stride;
    pb += app * apps.pb_stride;
    float* pwb = pw + (size_t)slab * wtotal;
    int woff = 0;
#pragma unroll
    for (int l = 0; l < 7; ++l) {
        const int ksz = gm_ksz(bands, l);
        if (tid < ksz) pwb[woff + tid] = l < L ? dwacc[l] : 0.0f;
        woff += ksz;
    }
    if (tid < 8) pb[(size_t)slab * 8 + tid] = tid < L ? red[GM_WAVES + tid] : 0.0f;
#if GM_DIAG == 5
    GM_MARK(7)
    if (tid == 0 && blockIdx.x == 0)
        for (int i = 0; i < 8; ++i) pb[i] = (float)dbg[i];
#endif
#if GM_DIAG == 6
    __syncthreads();
    if (tid == 0) {
        int* o = reinterpret_cast<int*>(pb + (size_t)slab * 8);
        o[0] = (int)(unsigned)wall0;
        o[1] = (int)(unsigned)wall_clock64();
        o[2] = (int)__builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID
        o[3] = (int)__builtin_amdgcn_s_getreg((31 << 11) | 20);  // XCC_ID
    }
#endif
}

}  // namespace

// Entry points used by gan.hip's dispatch (same contracts as the VALU kernels there).
bool hypel_gm_supported(int bands) {
    // HYPEL_GAN_MFMA_MIN: smallest band count that runs here.  Default 16: also the narrow Gulfport stacks (B = 64:
    // CycleGAN step 0.922 -> 0.887 ms, generator launches 31 -> 26 us) -- gan.hip's kernels keep B < 16 and B > 384
    constexpr int min_bands = 16;
    return bands >= min_bands && bands >= 16 && bands <= GM_MAX_BANDS && gm_bwd_lds(bands) <= 160 * 1024;
}

// floats of the kept-activation buffer of one forward pass over n samples (0: this band count runs on gan.hip's kernels)
int64_t hypel_gm_keep_floats(int64_t n, int bands, int only_encoder) {
    if (!hypel_gm_supported(bands)) return 0;
    const int64_t tiles = (n + GM_ROWS - 1) / GM_ROWS;
    return tiles * (only_encoder ? GmKeep<true>::V4 : GmKeep<false>::V4) * GM_THREADS * 4;
}

#define GM_LAUNCH(K, ...)                                                                                         \
    do {                                                                                                          \
        (void)hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);          \
        hipLaunchKernelGGL(K, dim3(grid), dim3(GM_THREADS), lds, st, __VA_ARGS__);                                \
    } while (0)

int hypel_gm_fwd(const float* x, int64_t ldx, int64_t n, int bands, const float* w, const float* b, int only_encoder,
                 float* out, int64_t ldo, int blocks, hipStream_t st, float* keep, float* enc_out, int64_t ld_enc,
                 int n_apps, int64_t w_stride, int64_t b_stride) {
    const size_t lds = gm_fwd_lds(bands);
    const int64_t tiles = (n + GM_ROWS - 1) / GM_ROWS;  // per application
    const GmApps apps{n_apps, w_stride, b_stride, 0, 0};
    // n_apps > 1: `blocks` is the whole grid, a multiple of n_apps (hypel_gan_generator_blocks_apps)
    const int grid = n_apps > 1 ? blocks : (int)(tiles < blocks ? tiles : blocks);
    if (only_encoder) {
        if (keep) GM_LAUNCH((gan_generator_fwd_mfma_kernel<true, true>), x, ldx, n, bands, w, b, out, ldo, keep, enc_out, ld_enc, apps);
        else GM_LAUNCH((gan_generator_fwd_mfma_kernel<true, false>), x, ldx, n, bands, w, b, out, ldo, keep, enc_out, ld_enc, apps);
    } else if (enc_out) {
        if (keep) GM_LAUNCH((gan_generator_fwd_mfma_kernel<false, true, true>), x, ldx, n, bands, w, b, out, ldo, keep, enc_out, ld_enc, apps);
        else GM_LAUNCH((gan_generator_fwd_mfma_kernel<false, false, true>), x, ldx, n, bands, w, b, out, ldo, keep, enc_out, ld_enc, apps);
    } else {
        if (keep) GM_LAUNCH((gan_generator_fwd_mfma_kernel<false, true>), x, ldx, n, bands, w, b, out, ldo, keep, enc_out, ld_enc, apps);
        else GM_LAUNCH((gan_generator_fwd_mfma_kernel<false, false>), x, ldx, n, bands, w, b, out, ldo, keep, enc_out, ld_enc, apps);
    }
    return 0;
}

int hypel_gm_bwd(const float* x, int64_t ldx, const float* dout, int64_t lddo, int64_t n, int bands, const float* w,
                 const float* b, int only_encoder, float* dx, int64_t lddx, int accumulate_dx, float* pw, float* pb,
                 int blocks, hipStream_t st, const float* keep, const float* d_enc, int64_t ld_denc, int n_apps,
                 int64_t w_stride, int64_t b_stride, int64_t pw_stride, int64_t pb_stride) {
    const size_t lds = gm_bwd_lds(bands);
    const GmApps apps{n_apps, w_stride, b_stride, pw_stride, pb_stride};
    int wtotal = 0;
    for (int l = 0; l < 7; ++l) wtotal += gm_ksz(bands, l);
    // every one of the `blocks` partial slabs is written (the planner's reduce sums all of them)
    const int64_t tiles = (n + GM_ROWS - 1) / GM_ROWS;
    const int grid = n_apps > 1 ? blocks : (int)(tiles < blocks ? tiles : blocks);
    // from kept activations: the register-ring kernel (two resident blocks per CU); HYPEL_GAN_BWD2=0 = the round-3 kernel
    static const int use_bwd2 = getenv("HYPEL_GAN_BWD2") ? atoi(getenv("HYPEL_GAN_BWD2")) : 1;
    if (keep && use_bwd2) {
        const size_t lds2 = gm_bwd2_lds(bands);
#define GM_BWD2(K)                                                                                                     \
    do {                                                                                                               \
        (void)hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);              \
        hipLaunchKernelGGL(K, dim3(grid), dim3(GM_THREADS), lds2, st, x, ldx, dout, lddo, n, bands, w, dx, lddx,       \
                           accumulate_dx, pw, pb, wtotal, blocks, keep, d_enc, ld_denc, apps);                         \
    } while (0)
        if (only_encoder) GM_BWD2((gan_generator_bwd2_mfma_kernel<true, false>));
        else if (d_enc) GM_BWD2((gan_generator_bwd2_mfma_kernel<false, true>));
        else GM_BWD2((gan_generator_bwd2_mfma_kernel<false, false>));
#undef GM_BWD2
        return 0;
    }
#define GM_BWD_ARGS x, ldx, dout, lddo, n, bands, w, b, dx, lddx, accumulate_dx, pw, pb, wtotal, blocks, keep, d_enc, ld_denc, apps
    if (only_encoder) {
        if (keep) GM_LAUNCH((gan_generator_bwd_mfma_kernel<true, true>), GM_BWD_ARGS);
        else GM_LAUNCH((gan_generator_bwd_mfma_kernel<true, false>), GM_BWD_ARGS);
    } else if (d_enc) {
        if (keep) GM_LAUNCH((gan_generator_bwd_mfma_kernel<false, true, true>), GM_BWD_ARGS);
        else GM_LAUNCH((gan_generator_bwd_mfma_kernel<false, false, true>), GM_BWD_ARGS);
    } else {
        if (keep) GM_LAUNCH((gan_generator_bwd_mfma_kernel<false, true>), GM_BWD_ARGS);
        else GM_LAUNCH((gan_generator_bwd_mfma_kernel<false, false>), GM_BWD_ARGS);
    }
#undef GM_BWD_ARGS
    return 0;
}
#undef GM_LAUNCH
