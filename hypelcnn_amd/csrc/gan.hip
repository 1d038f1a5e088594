// Shadow-GAN kernels (gan/shadow_data_models.py, gan/wrappers/*).  The stacks are 1-channel 1-D convolutions
// over <= 360 bands and tiny MLPs: <= 0.4 MFLOP per sample, i.e. launch/latency and HBM bound, not MFMA work
// (SURVEY F8.iii).  The reference executes ~30 TensorFlow ops per generator application; here ONE wavefront
// runs the whole 7-layer generator of a sample out of LDS (activations, skip sums and weights never leave the
// CU), and the backward kernel recomputes that forward instead of storing 13 intermediate tensors.
#include "common.h"

namespace {

constexpr float GEN_ALPHA = 0.1f;  // leaky_relu(alpha=0.1), shadow_data_models.py:53
constexpr int GEN_WAVES = 4;

struct GenLayout {
    int k[7], pl[7], woff[7], wtotal, layers;
};

__host__ __device__ inline GenLayout gen_layout(int bands, int only_encoder) {
    GenLayout g;
    const int ks[7] = {bands, bands / 2, bands / 4, bands / 8, bands / 4, bands / 2, bands};
    int off = 0;
    for (int i = 0; i < 7; ++i) {
        g.k[i] = ks[i];
        g.pl[i] = (ks[i] - 1) / 2;  // SAME: pad_left = floor((k-1)/2), the rest on the right
        g.woff[i] = off;
        off += ks[i];
    }
    g.wtotal = off;
    g.layers = only_encoder ? 4 : 7;
    return g;
}

// out[p] = bias + sum_j w[j] * in[p + j - pl]   (zero outside [0, B))
__device__ __forceinline__ float conv_at(const float* __restrict__ in, const float* __restrict__ w, int k, int pl,
                                         int bands, int p, float bias) {
    const int j0 = max(0, pl - p), j1 = min(k, bands + pl - p);
    float acc = bias;
    for (int j = j0; j < j1; ++j) acc += w[j] * in[p + j - pl];
    return acc;
}

// One wave: forward of one sample.  a: [7][bands] activations (a[0] = input), slope: [6][bands] (may be null).
// Returns through `last`: for the full generator the pre-tanh c7 in tmp[bands]; for the encoder a[4].
__device__ __forceinline__ void gen_forward_wave(const GenLayout& g, int bands, int lane, const float* __restrict__ ws,
                                                 const float* __restrict__ bs, float* __restrict__ a,
                                                 float* __restrict__ slope, float* __restrict__ tmp) {
    const int hidden = g.layers == 7 ? 6 : 4;
    for (int i = 1; i <= hidden; ++i) {
        const float* in = a + (i - 1) * bands;
        float* out = a + i * bands;
        for (int p = lane; p < bands; p += 64) {
            const float c = conv_at(in, ws + g.woff[i - 1], g.k[i - 1], g.pl[i - 1], bands, p, bs[i - 1]);
            const float s = c > 0.0f ? 1.0f : GEN_ALPHA;
            if (slope) slope[(i - 1) * bands + p] = s;
            float v = c * s + in[p];
            if (i >= 2) v += a[(i - 2) * bands + p];
            out[p] = v;
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (g.layers == 7) {
        const float* in = a + 6 * bands;
        for (int p = lane; p < bands; p += 64) tmp[p] = conv_at(in, ws + g.woff[6], g.k[6], g.pl[6], bands, p, bs[6]);
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ __launch_bounds__(64 * GEN_WAVES) void gan_generator_fwd_kernel(const float* __restrict__ x, int64_t ldx,
                                                                            int64_t n, int bands,
                                                                            const float* __restrict__ w,
                                                                            const float* __restrict__ b,
                                                                            int only_encoder, float* __restrict__ out,
                                                                            int64_t ldo) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ GenLayout g_lds;  // layer table in LDS, not in per-lane scratch memory (see the tiled kernels)
    if (threadIdx.x == 0) g_lds = gen_layout(bands, only_encoder);
    __syncthreads();
    const GenLayout& g = g_lds;
    float* ws = smem;                 // [wtotal]
    float* bs = ws + g.wtotal;        // [8]
    float* wave_base = bs + 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < g.wtotal; i += blockDim.x) ws[i] = w[i];
    if (threadIdx.x < 7) bs[threadIdx.x] = b[threadIdx.x];
    __syncthreads();
    float* a = wave_base + (size_t)wave * 8 * bands;  // 7 activation rows + 1 tmp row
    float* tmp = a + 7 * bands;
    for (int64_t s = (int64_t)blockIdx.x * GEN_WAVES + wave; s < n; s += (int64_t)gridDim.x * GEN_WAVES) {
        for (int p = lane; p < bands; p += 64) a[p] = x[s * ldx + p];
        __builtin_amdgcn_wave_barrier();
        gen_forward_wave(g, bands, lane, ws, bs, a, nullptr, tmp);
        for (int p = lane; p < bands; p += 64)
            out[s * ldo + p] = only_encoder ? a[4 * bands + p] : tanhf(tmp[p]);
        __builtin_amdgcn_wave_barrier();
    }
}

// Backward: per sample recompute the forward, then walk the layers in reverse.
//   a_i = h_i + a_{i-1} + a_{i-2}  =>  da_{i-1} += da_i, da_{i-2} += da_i, dc_i = da_i * slope_i
//   dW_i[j] += sum_p dc_i[p] * a_{i-1}[p + j - pl],  db_i += sum_p dc_i[p],  da_{i-1}[q] += sum_j w_i[j] dc_i[q - j + pl]
// Weight gradients are accumulated per wave in LDS (lane l owns taps l, l+64, ...), combined per block, and
// written to pw[block][wtotal] / pb[block][8]; hypel_reduce_splits_f32 sums the blocks in fixed order.
__global__ __launch_bounds__(64 * GEN_WAVES) void gan_generator_bwd_kernel(
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ dout, int64_t lddo, int64_t n, int bands,
    const float* __restrict__ w, const float* __restrict__ b, int only_encoder, float* __restrict__ dx, int64_t lddx,
    int accumulate_dx, float* __restrict__ pw, float* __restrict__ pb) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ GenLayout g_lds;
    if (threadIdx.x == 0) g_lds = gen_layout(bands, only_encoder);
    __syncthreads();
    const GenLayout& g = g_lds;
    float* ws = smem;
    float* bs = ws + g.wtotal;
    float* wave_base = bs + 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < g.wtotal; i += blockDim.x) ws[i] = w[i];
    if (threadIdx.x < 7) bs[threadIdx.x] = b[threadIdx.x];
    // per-wave regions: a[7][B], tmp[B], slope[6][B], da[7][B], dc[B], dw[wtotal], db[8]
    const size_t per_wave = (size_t)(7 + 1 + 6 + 7 + 1) * bands + g.wtotal + 8;
    float* a = wave_base + wave * per_wave;
    float* tmp = a + 7 * bands;
    float* slope = tmp + bands;
    float* da = slope + 6 * bands;
    float* dc = da + 7 * bands;
    float* dw = dc + bands;
    float* db = dw + g.wtotal;
    for (int i = lane; i < g.wtotal; i += 64) dw[i] = 0.0f;
    if (lane < 8) db[lane] = 0.0f;
    __syncthreads();
    const int hidden = g.layers == 7 ? 6 : 4;

    for (int64_t s = (int64_t)blockIdx.x * GEN_WAVES + wave; s < n; s += (int64_t)gridDim.x * GEN_WAVES) {
        for (int p = lane; p < bands; p += 64) a[p] = x[s * ldx + p];
        for (int i = lane; i < 7 * bands; i += 64) da[i] = 0.0f;
        __builtin_amdgcn_wave_barrier();
        gen_forward_wave(g, bands, lane, ws, bs, a, slope, tmp);

        auto layer_bwd = [&](int li /*0-based layer*/, const float* in /*a_{i-1}*/, float* din /*da_{i-1}*/) {
            const int k = g.k[li], pl = g.pl[li];
            const float* wl = ws + g.woff[li];
            // bias gradient: wave sum of dc
            float sb = 0.0f;
            for (int p = lane; p < bands; p += 64) sb += dc[p];
            for (int o = 32; o > 0; o >>= 1) sb += __shfl_down(sb, o, 64);
            if (lane == 0) db[li] += sb;
            // weight gradient: lane owns taps j = lane, lane+64, ...
            for (int j = lane; j < k; j += 64) {
                const int p0 = max(0, pl - j), p1 = min(bands, bands + pl - j);
                float acc = 0.0f;
                for (int p = p0; p < p1; ++p) acc += dc[p] * in[p + j - pl];
                dw[g.woff[li] + j] += acc;
            }
            // input gradient: din[q] += sum_j w[j] * dc[q - j + pl]
            for (int q = lane; q < bands; q += 64) {
                const int j0 = max(0, q + pl - (bands - 1)), j1 = min(k, q + pl + 1);
                float acc = 0.0f;
                for (int j = j0; j < j1; ++j) acc += wl[j] * dc[q - j + pl];
                din[q] += acc;
            }
            __builtin_amdgcn_wave_barrier();
        };

        if (g.layers == 7) {
            for (int p = lane; p < bands; p += 64) {
                const float t = tanhf(tmp[p]);
                dc[p] = dout[s * lddo + p] * (1.0f - t * t);
            }
            __builtin_amdgcn_wave_barrier();
            layer_bwd(6, a + 6 * bands, da + 6 * bands);
        } else {
            for (int p = lane; p < bands; p += 64) da[4 * bands + p] = dout[s * lddo + p];
            __builtin_amdgcn_wave_barrier();
        }
        for (int i = hidden; i >= 1; --i) {
            for (int p = lane; p < bands; p += 64) {
                const float gi = da[i * bands + p];
                da[(i - 1) * bands + p] += gi;
                if (i >= 2) da[(i - 2) * bands + p] += gi;
                dc[p] = gi * slope[(i - 1) * bands + p];
            }
            __builtin_amdgcn_wave_barrier();
            layer_bwd(i - 1, a + (i - 1) * bands, da + (i - 1) * bands);
        }
        if (dx) {
            for (int p = lane; p < bands; p += 64) {
                float* d = dx + s * lddx + p;
                *d = accumulate_dx ? *d + da[p] : da[p];
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // combine the waves of this block (fixed order) and publish the block's partial sums
    float* pwb = pw + (size_t)blockIdx.x * g.wtotal;
    for (int i = threadIdx.x; i < g.wtotal; i += blockDim.x) {
        float t = 0.0f;
        for (int wv = 0; wv < GEN_WAVES; ++wv) t += (wave_base + wv * per_wave + (7 + 1 + 6 + 7 + 1) * bands)[i];
        pwb[i] = t;
    }
    if (threadIdx.x < 8) {
        float t = 0.0f;
        for (int wv = 0; wv < GEN_WAVES; ++wv)
            t += (wave_base + wv * per_wave + (7 + 1 + 6 + 7 + 1) * bands + g.wtotal)[threadIdx.x];
        pb[(size_t)blockIdx.x * 8 + threadIdx.x] = t;
    }
}

// ===================================================================================================================
// Register-tiled generator for wide spectra (bands > GEN_TILED_MIN; AVON: 360 bands = 384 k MAC per sample, which is
// VALU work, not launch latency).  Same math, different schedule:
//   * activation rows live in LDS with zero margins of H floats on both sides, so no tap needs a bounds check;
//   * weights are re-laid in LDS in "offset space" d = j - pad_left, start rounded down to a multiple of 4 and zero
//     filled, once as they are (forward conv, filter gradient) and once mirrored (input gradient);
//   * a lane owns 4 CONTIGUOUS outputs (or 4 contiguous taps for the filter gradient) and walks the other index in
//     steps of 4: two aligned ds_read_b128 (an 8-float window) + one broadcast b128 feed 16 FMAs, i.e. 0.19 LDS
//     reads per FMA instead of 2, and 4 independent accumulators instead of one dependent chain.
constexpr int GEN_TILED_MIN = 128;
constexpr int GT_WAVES = 4;  // samples a block works on at a time ("slots"; each has its own LDS region)
// Wavefronts that share ONE sample (round 2).  The backward kernel needs ~35 KB of LDS per sample at 360 bands, so a CU
// holds four samples: with one wavefront each that was one wave per SIMD, nothing to hide an LDS round trip or an
// FMA chain behind, and 90 output quads on 64 lanes took two passes (64 + 26).  Two waves split the quads / taps of
// every phase (ranges are disjoint, phases end in a block barrier instead of a wave barrier): 8 waves per CU.
constexpr int GT_COOP = 2;
constexpr int GT_SLOT_THREADS = 64 * GT_COOP;
constexpr int GT_QSTEP = 4 * GT_SLOT_THREADS;  // outputs a slot's threads cover per pass (4 per thread)

struct GenTiled {
    int k[7], pl[7], woff[7];
    int dmin[7], nch[7], wpoff[7];   // offset space of the taps: d in [dmin, dmin + 4 nch)
    int emin[7], nchf[7], wfoff[7];  // mirrored taps: e = -d
    int wtotal, wptotal, wftotal, layers, H, W, B4;
};

__host__ __device__ inline GenTiled gen_tiled(int bands, int only_encoder) {
    GenTiled g;
    const int ks[7] = {bands, bands / 2, bands / 4, bands / 8, bands / 4, bands / 2, bands};
    int off = 0, offp = 0, offf = 0;
    for (int i = 0; i < 7; ++i) {
        const int k = ks[i], pl = (k - 1) / 2;
        g.k[i] = k;
        g.pl[i] = pl;
        g.woff[i] = off;
        off += k;
        g.dmin[i] = (-pl) & ~3;  // floor to a multiple of 4 (two's complement)
        g.nch[i] = ((k - 1 - pl) - g.dmin[i] + 4) / 4;
        g.wpoff[i] = offp;
        offp += 4 * g.nch[i];
        g.emin[i] = (pl - k + 1) & ~3;
        g.nchf[i] = (pl - g.emin[i] + 4) / 4;
        g.wfoff[i] = offf;
        offf += 4 * g.nchf[i];
    }
    g.wtotal = off;
    g.wptotal = offp;
    g.wftotal = offf;
    g.layers = only_encoder ? 4 : 7;
    g.B4 = (bands + 3) & ~3;
    g.H = (bands / 2 + 4 + 3) & ~3;
    g.W = g.H + g.B4 + g.H;
    return g;
}

// block-wide: lay the filters out in offset space (zero filled)
__device__ __forceinline__ void gt_stage_weights(const GenTiled& g, const float* __restrict__ w, float* __restrict__ wp,
                                                 float* __restrict__ wf) {
    for (int li = 0; li < 7; ++li) {
        for (int c = threadIdx.x; c < 4 * g.nch[li]; c += blockDim.x) {
            const int j = c + g.dmin[li] + g.pl[li];
            wp[g.wpoff[li] + c] = (j >= 0 && j < g.k[li]) ? w[g.woff[li] + j] : 0.0f;
        }
        if (wf)
            for (int c = threadIdx.x; c < 4 * g.nchf[li]; c += blockDim.x) {
                const int j = g.pl[li] - (c + g.emin[li]);
                wf[g.wfoff[li] + c] = (j >= 0 && j < g.k[li]) ? w[g.woff[li] + j] : 0.0f;
            }
    }
}

// acc[t] += sum_u wv[u] * win[t + u] over all 4-tap chunks: the quad of outputs at p0 of a SAME correlation of the
// padded row `in` (element q at in[H + q]) with filters `wq` in offset space starting at `dmin`.
__device__ __forceinline__ void gt_quad(const float* __restrict__ in, int H, int p0, const float* __restrict__ wq,
                                        int dmin, int nch, float acc[4]) {
    const float4* wv4 = reinterpret_cast<const float4*>(wq);
    const float4* base = reinterpret_cast<const float4*>(in + H + p0 + dmin);
    float4 lo = base[0];
#pragma unroll 2
    for (int c = 0; c < nch; ++c) {
        const float4 hi = base[c + 1];
        const float4 w4 = wv4[c];
        const float win[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] += wv[u] * win[t + u];
        lo = hi;
    }
}

// forward of one sample by the GT_COOP waves of a slot (st = thread index inside the slot).  rows: padded activation
// rows, row(i) = rows + slot(i) * W.  FULL = keep all 7 rows (backward recompute, slot(i) = i), else 3 rolling rows
// (slot(i) = i % 3).  The last full-generator layer leaves the pre-tanh values in `pre` (padded row, element q at
// pre[H + q]).  Every phase ends in a block barrier: all threads of the block must call this together.
template <bool FULL>
__device__ __forceinline__ void gt_forward_wave(const GenTiled& g, int bands, int st, const float* __restrict__ wp,
                                                const float* __restrict__ bs, float* __restrict__ rows,
                                                uint8_t* __restrict__ slope, float* __restrict__ pre) {
    const int hidden = g.layers == 7 ? 6 : 4;
    const int H = g.H, W = g.W;
    auto row = [&](int i) { return rows + (FULL ? i : i % 3) * W; };
    for (int i = 1; i <= hidden; ++i) {
        const float* in = row(i - 1);
        const float* in2 = i >= 2 ? row(i - 2) : nullptr;
        float* out = row(i);
        for (int p0 = 4 * st; p0 < bands; p0 += GT_QSTEP) {
            float acc[4] = {bs[i - 1], bs[i - 1], bs[i - 1], bs[i - 1]};
            gt_quad(in, H, p0, wp + g.wpoff[i - 1], g.dmin[i - 1], g.nch[i - 1], acc);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int p = p0 + t;
                if (p < bands) {
                    const float c = acc[t];
                    const bool pos = c > 0.0f;
                    if (slope) slope[(i - 1) * bands + p] = pos ? 1 : 0;
                    float v = (pos ? c : c * GEN_ALPHA) + in[H + p];
                    if (in2) v += in2[H + p];
                    out[H + p] = v;
                }
            }
        }
        __syncthreads();
    }
    if (g.layers == 7) {
        const float* in = row(6);
        for (int p0 = 4 * st; p0 < bands; p0 += GT_QSTEP) {
            float acc[4] = {bs[6], bs[6], bs[6], bs[6]};
            gt_quad(in, H, p0, wp + g.wpoff[6], g.dmin[6], g.nch[6], acc);
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (p0 + t < bands) pre[H + p0 + t] = acc[t];
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(GT_SLOT_THREADS * GT_WAVES) void gan_generator_fwd_tiled_kernel(
    const float* __restrict__ x, int64_t ldx, int64_t n, int bands, const float* __restrict__ w,
    const float* __restrict__ b, int only_encoder, float* __restrict__ out, int64_t ldo) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // the layer table lives in LDS: as a per-thread struct its dynamically indexed arrays went to scratch memory (284
    // bytes per lane) and every layer of every sample started with a round of private-memory loads
    __shared__ GenTiled g_lds;
    if (threadIdx.x == 0) g_lds = gen_tiled(bands, only_encoder);
    __syncthreads();
    const GenTiled& g = g_lds;
    float* wp = smem;                 // [wptotal]
    float* bs = wp + g.wptotal;       // [8]
    float* wave_base = bs + 8;
    const int st = threadIdx.x % GT_SLOT_THREADS, slot = threadIdx.x / GT_SLOT_THREADS;
    gt_stage_weights(g, w, wp, nullptr);
    if (threadIdx.x < 7) bs[threadIdx.x] = b[threadIdx.x];
    float* rows = wave_base + (size_t)slot * 4 * g.W;  // 3 rolling rows + the pre-tanh row
    float* pre = rows + 3 * g.W;
    for (int i = st; i < 4 * g.W; i += GT_SLOT_THREADS) rows[i] = 0.0f;  // margins stay zero for the whole kernel
    __syncthreads();
    // block-uniform trip count (the phases end in block barriers); a slot past the end works on zeros and stores nothing
    const int64_t first = (int64_t)blockIdx.x * GT_WAVES, stride = (int64_t)gridDim.x * GT_WAVES;
    for (int64_t s0 = first; s0 < n; s0 += stride) {
        const int64_t s = s0 + slot;
        const bool active = s < n;
        for (int p = st; p < bands; p += GT_SLOT_THREADS) rows[g.H + p] = active ? x[s * ldx + p] : 0.0f;
        __syncthreads();
        gt_forward_wave<false>(g, bands, st, wp, bs, rows, nullptr, pre);
        const float* a4 = rows + (4 % 3) * g.W;
        if (active)
            for (int p = st; p < bands; p += GT_SLOT_THREADS)
                out[s * ldo + p] = only_encoder ? a4[g.H + p] : tanhf(pre[g.H + p]);
        __syncthreads();
    }
}

// Backward (recompute + reverse walk), tiled.  Per slot in LDS: 7 padded activation rows, one padded dc row,
// 3 rolling da rows, 6 x bands slope flags (bytes), the filter gradients in offset space, bias gradients + 2 partials.
__global__ __launch_bounds__(GT_SLOT_THREADS * GT_WAVES) void gan_generator_bwd_tiled_kernel(
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ dout, int64_t lddo, int64_t n, int bands,
    const float* __restrict__ w, const float* __restrict__ b, int only_encoder, float* __restrict__ dx, int64_t lddx,
    int accumulate_dx, float* __restrict__ pw, float* __restrict__ pb) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ GenTiled g_lds;  // see the forward kernel
    if (threadIdx.x == 0) g_lds = gen_tiled(bands, only_encoder);
    __syncthreads();
    const GenTiled& g = g_lds;
    const int H = g.H, W = g.W, B4 = g.B4;
    float* wp = smem;
    float* wf = wp + g.wptotal;
    float* bs = wf + g.wftotal;
    float* wave_base = bs + 8;
    const int slope_words = (6 * bands + 15) / 16 * 4;  // bytes rounded up to 16, in floats
    const size_t per_wave = (size_t)8 * W + 3 * B4 + slope_words + g.wptotal + 8;
    const int st = threadIdx.x % GT_SLOT_THREADS, slot = threadIdx.x / GT_SLOT_THREADS;
    const int lane = threadIdx.x & 63, cw = st >> 6;  // lane, wave inside the slot
    gt_stage_weights(g, w, wp, wf);
    if (threadIdx.x < 7) bs[threadIdx.x] = b[threadIdx.x];
    float* rows = wave_base + slot * per_wave;  // a[0..6]
    float* dc = rows + 7 * W;                   // padded
    float* da = dc + W;                         // 3 rolling rows of B4
    uint8_t* slope = reinterpret_cast<uint8_t*>(da + 3 * B4);
    float* dwp = da + 3 * B4 + slope_words;     // [wptotal] offset-space filter gradients
    float* db = dwp + g.wptotal;                // [7] bias gradients
    __shared__ float dbp_all[GT_WAVES][GT_COOP];  // per-wave partial of the bias-gradient sum of the current layer
    float* dbp = dbp_all[slot];
    for (int i = st; i < 8 * W; i += GT_SLOT_THREADS) rows[i] = 0.0f;
    for (int i = st; i < g.wptotal; i += GT_SLOT_THREADS) dwp[i] = 0.0f;
    if (st < 8) db[st] = 0.0f;
    __syncthreads();
    const int hidden = g.layers == 7 ? 6 : 4;
    auto da_row = [&](int i) { return da + (i % 3) * B4; };

    // block-uniform trip count (every phase ends in a block barrier); a slot past the end works on zeros: its dc rows
    // are zero, so it adds nothing to the filter / bias gradients, and it stores nothing
    const int64_t first = (int64_t)blockIdx.x * GT_WAVES, stride = (int64_t)gridDim.x * GT_WAVES;
    for (int64_t s0 = first; s0 < n; s0 += stride) {
        const int64_t s = s0 + slot;
        const bool active = s < n;
        for (int p = st; p < bands; p += GT_SLOT_THREADS) rows[H + p] = active ? x[s * ldx + p] : 0.0f;
        __syncthreads();
        gt_forward_wave<true>(g, bands, st, wp, bs, rows, slope, dc);

        // given dc (padded row): db_li, dW_li (offset space), din[q] (+)= sum_e wf[e] dc[q + e]
        auto layer_bwd = [&](int li, const float* in, float* din, bool assign) {
            float sb = 0.0f;
            for (int p = st; p < bands; p += GT_SLOT_THREADS) sb += dc[H + p];
            for (int o = 32; o > 0; o >>= 1) sb += __shfl_down(sb, o, 64);
            if (lane == 0) dbp[cw] = sb;
            // filter gradient: the thread owns the 4 taps c0..c0+3 of offset space and walks the positions
            for (int c0 = 4 * st; c0 < 4 * g.nch[li]; c0 += GT_QSTEP) {
                float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                const float4* base = reinterpret_cast<const float4*>(in + H + g.dmin[li] + c0);
                const float4* dc4 = reinterpret_cast<const float4*>(dc + H);
                float4 lo = base[0];
#pragma unroll 2
                for (int pc = 0; pc < B4 / 4; ++pc) {
                    const float4 hi = base[pc + 1];
                    const float4 d4 = dc4[pc];
                    const float win[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                    const float dv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int u = 0; u < 4; ++u) acc[u] += dv[t] * win[t + u];
                    lo = hi;
                }
                float4* o4 = reinterpret_cast<float4*>(dwp + g.wpoff[li] + c0);
                float4 cur = *o4;
                cur.x += acc[0];
                cur.y += acc[1];
                cur.z += acc[2];
                cur.w += acc[3];
                *o4 = cur;
            }
            // input gradient
            for (int q0 = 4 * st; q0 < bands; q0 += GT_QSTEP) {
                float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                gt_quad(dc, H, q0, wf + g.wfoff[li], g.emin[li], g.nchf[li], acc);
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (q0 + t < bands) din[q0 + t] = assign ? acc[t] : din[q0 + t] + acc[t];
            }
            __syncthreads();
            if (st == 0) {  // the slot's waves in fixed order (dbp is next written behind another barrier)
                float t = dbp[0];
#pragma unroll
                for (int k = 1; k < GT_COOP; ++k) t += dbp[k];
                db[li] += t;
            }
        };

        if (g.layers == 7) {
            for (int p = st; p < bands; p += GT_SLOT_THREADS) {
                const float t = tanhf(dc[H + p]);
                dc[H + p] = (active ? dout[s * lddo + p] : 0.0f) * (1.0f - t * t);
            }
            __syncthreads();
            layer_bwd(6, rows + 6 * W, da_row(6), true);
        } else {
            float* d4 = da_row(4);
            for (int p = st; p < bands; p += GT_SLOT_THREADS) d4[p] = active ? dout[s * lddo + p] : 0.0f;
            __syncthreads();
        }
        for (int i = hidden; i >= 1; --i) {
            const float* gi_row = da_row(i);
            float* d1 = da_row(i - 1);
            float* d2 = i >= 2 ? da_row(i - 2) : nullptr;
            for (int p = st; p < bands; p += GT_SLOT_THREADS) {
                const float gi = gi_row[p];
                d1[p] = (i == hidden) ? gi : d1[p] + gi;  // first touch of da[hidden-1]
                if (d2) d2[p] = gi;                        // first touch of da[i-2]
                dc[H + p] = slope[(i - 1) * bands + p] ? gi : gi * GEN_ALPHA;
            }
            __syncthreads();
            layer_bwd(i - 1, rows + (i - 1) * W, d1, false);
        }
        if (dx && active) {
            const float* d0 = da_row(0);
            for (int p = st; p < bands; p += GT_SLOT_THREADS) {
                float* d = dx + s * lddx + p;
                *d = accumulate_dx ? *d + d0[p] : d0[p];
            }
        }
        __syncthreads();
    }
    __syncthreads();
    // combine the slots (fixed order), map offset space back to tap order, publish the block's partial sums
    float* pwb = pw + (size_t)blockIdx.x * g.wtotal;
    const size_t dw_at = (size_t)8 * W + 3 * B4 + slope_words;
    for (int li = 0; li < 7; ++li)
        for (int j = threadIdx.x; j < g.k[li]; j += blockDim.x) {
            const int c = j - g.pl[li] - g.dmin[li];
            float t = 0.0f;
            for (int wv = 0; wv < GT_WAVES; ++wv) t += (wave_base + wv * per_wave + dw_at)[g.wpoff[li] + c];
            pwb[g.woff[li] + j] = t;
        }
    if (threadIdx.x < 8) {
        float t = 0.0f;
        for (int wv = 0; wv < GT_WAVES; ++wv) t += (wave_base + wv * per_wave + dw_at + g.wptotal)[threadIdx.x];
        pb[(size_t)blockIdx.x * 8 + threadIdx.x] = t;
    }
}

// ------------------------------------------------------------------------------------- losses
__device__ __forceinline__ float block_sum_256(float v, float* sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = v;
    __syncthreads();
    float t = 0.0f;
    if (threadIdx.x == 0) t = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return t;
}

// mode 0: weight*mean((a-target)^2)   mode 1: weight*mean(|a-b|)   mode 2: weight*mean(a)
__global__ __launch_bounds__(256) void gan_loss_partial_kernel(int mode, const float* __restrict__ a, int64_t lda,
                                                                const float* __restrict__ b, int64_t ldb,
                                                                int64_t rows, int c, float target, float gcoef,
                                                                float* da, int64_t ldda, int acc_da,
                                                                float* db, int64_t lddb, int acc_db,
                                                                float* __restrict__ ws, float pscale) {
    __shared__ float sh[4];
    float s = 0.0f;
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x)
        for (int col = threadIdx.x; col < c; col += 256) {
            const float av = a[row * lda + col];
            float val, ga, gb = 0.0f;
            if (mode == 0) {
                const float d = av - target;
                val = d * d;
                ga = 2.0f * d;
            } else if (mode == 1) {
                const float d = av - b[row * ldb + col];
                val = fabsf(d);
                ga = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
                gb = -ga;
            } else {
                val = av;
                ga = 1.0f;
            }
            s += val;
            if (da) {
                float* p = da + row * ldda + col;
                *p = (acc_da ? *p : 0.0f) + gcoef * ga;
            }
            if (db) {
                float* p = db + row * lddb + col;
                *p = (acc_db ? *p : 0.0f) + gcoef * gb;
            }
        }
    const float t = block_sum_256(s, sh);
    if (threadIdx.x == 0) ws[blockIdx.x] = t * pscale;
}

// One block of 1024 threads: eight partials per thread requested before the first add (as `for (i = tid; i < n; i += 256)`
// the ~7 k slot values of a train op were 28 dependent global round trips per thread: 7.2 us), fp64 sums -- lanes by
// shuffles, the 16 waves through LDS in wave order.
constexpr int LOSS_FIN_THREADS = 1024;
__global__ __launch_bounds__(LOSS_FIN_THREADS) void loss_finalize_kernel(const float* __restrict__ ws, int n, double scale,
                                                                          float* __restrict__ loss, int accumulate) {
    __shared__ double sh[LOSS_FIN_THREADS / 64];
    double s = 0.0;
    for (int i0 = threadIdx.x; i0 < n; i0 += 8 * LOSS_FIN_THREADS) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int i = i0 + q * LOSS_FIN_THREADS;
            v[q] = i < n ? ws[i] : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) s += (double)v[q];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < LOSS_FIN_THREADS / 64; ++k) t += sh[k];
        loss[0] = (accumulate ? loss[0] : 0.0f) + (float)(t * scale);
    }
}

// l2 regulariser: loss += scale/2 * sum w^2, dw += scale * w
__global__ __launch_bounds__(256) void l2_reg_kernel(const float* __restrict__ w, int64_t count, float scale,
                                                      float* __restrict__ dw, float* __restrict__ ws, float pscale) {
    __shared__ float sh[4];
    float s = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
        const float v = w[i];
        s += v * v;
        if (dw) dw[i] += scale * v;
    }
    const float t = block_sum_256(s, sh);
    if (threadIdx.x == 0) ws[blockIdx.x] = t * pscale;
}

// tf.math.l2_normalize(x) with axis=None: the norm of the WHOLE [rows x c] tensor (shadow_data_models.py:147).
// One block of 1024 threads (the tensor is [N x E] with E = 2: 8 k elements); stat[0] = sum x^2, stat[1] = rsqrt(max(sum, 1e-12)).
__global__ __launch_bounds__(1024) void l2norm_fwd_kernel(const float* __restrict__ x, int64_t ldx, int64_t rows, int c,
                                                          float* __restrict__ y, int64_t ldy,
                                                          float* __restrict__ stat) {
    __shared__ double sh[1024];
    __shared__ float inv_s;
    // block (p, g) normalises part p -- columns [p*c, (p+1)*c) -- of row segment g (rows [g*rows, (g+1)*rows): one
    // application of a row-concatenated batch; gridDim.y = 1 for a single application)
    x += (int64_t)blockIdx.x * c + (int64_t)blockIdx.y * rows * ldx;
    y += (int64_t)blockIdx.x * c + (int64_t)blockIdx.y * rows * ldy;
    stat += 2 * (blockIdx.x + blockIdx.y * gridDim.x);
    const int64_t total = rows * c;
    double s = 0.0;
#pragma unroll 4
    for (int64_t i = threadIdx.x; i < total; i += 1024) {
        const float v = x[(i / c) * ldx + (i % c)];
        s += (double)v * v;
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double ss = sh[0];
        const float inv = (float)(1.0 / sqrt(ss > 1e-12 ? ss : 1e-12));
        stat[0] = (float)ss;
        stat[1] = inv;
        inv_s = inv;
    }
    __syncthreads();
    const float inv = inv_s;
    for (int64_t i = threadIdx.x; i < total; i += 1024) y[(i / c) * ldy + (i % c)] = x[(i / c) * ldx + (i % c)] * inv;
}

// dx = g*inv - x * (sum g.x) * inv^3   (dx = g*inv when the clamp is active)
__global__ __launch_bounds__(1024) void l2norm_bwd_kernel(const float* __restrict__ x, int64_t ldx,
                                                          const float* __restrict__ dy, int64_t lddy, int64_t rows,
                                                          int c, const float* __restrict__ stat,
                                                          float* __restrict__ dx, int64_t lddx, int accumulate) {
    __shared__ double sh[1024];
    __shared__ float dot_s;
    x += (int64_t)blockIdx.x * c + (int64_t)blockIdx.y * rows * ldx;
    dy += (int64_t)blockIdx.x * c + (int64_t)blockIdx.y * rows * lddy;
    dx += (int64_t)blockIdx.x * c + (int64_t)blockIdx.y * rows * lddx;
    stat += 2 * (blockIdx.x + blockIdx.y * gridDim.x);
    const int64_t total = rows * c;
    double s = 0.0;
#pragma unroll 4
    for (int64_t i = threadIdx.x; i < total; i += 1024)
        s += (double)x[(i / c) * ldx + (i % c)] * (double)dy[(i / c) * lddy + (i % c)];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) dot_s = (float)sh[0];
    __syncthreads();
    const float inv = stat[1];
    const float coef = stat[0] > 1e-12f ? dot_s * inv * inv * inv : 0.0f;
    for (int64_t i = threadIdx.x; i < total; i += 1024) {
        const float g = dy[(i / c) * lddy + (i % c)] * inv - x[(i / c) * ldx + (i % c)] * coef;
        float* p = dx + (i / c) * lddx + (i % c);
        *p = accumulate ? *p + g : g;
    }
}

// The same two kernels for parts of at most 16 K elements (the CUT embeddings: 4096 x 2 per part): a thread's <= 16
// elements stay in registers between the reduction and the scaling pass, 32-bit index arithmetic, wave shuffles + one
// 16-entry LDS hop instead of a 10-level LDS tree (forward 8.9 -> ~5 us, backward 14.7 -> ~6 us at 4096 x 12).
constexpr int L2N_R = 16;

__device__ __forceinline__ double l2n_block_sum(double s, double* sh16) {
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) sh16[threadIdx.x >> 6] = s;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += sh16[w];  // every thread, fixed order
    return t;
}

__global__ __launch_bounds__(1024) void l2norm_fwd_small_kernel(const float* __restrict__ x, int ldx, int rows, int c,
                                                                float* __restrict__ y, int ldy,
                                                                float* __restrict__ stat) {
    __shared__ double sh16[16];
    x += blockIdx.x * c + (int64_t)blockIdx.y * rows * ldx;
    y += blockIdx.x * c + (int64_t)blockIdx.y * rows * ldy;
    stat += 2 * (blockIdx.x + blockIdx.y * gridDim.x);
    const int total = rows * c;
    float v[L2N_R];
    int off_y[L2N_R];
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < L2N_R; ++k) {
        const int i = threadIdx.x + 1024 * k;
        const int r = i / c, q = i - r * c;
        v[k] = i < total ? x[r * ldx + q] : 0.0f;
        off_y[k] = r * ldy + q;
    }
#pragma unroll
    for (int k = 0; k < L2N_R; ++k) s += (double)v[k] * v[k];
    const double ss = l2n_block_sum(s, sh16);
    const float inv = (float)(1.0 / sqrt(ss > 1e-12 ? ss : 1e-12));
    if (threadIdx.x == 0) {
        stat[0] = (float)ss;
        stat[1] = inv;
    }
#pragma unroll
    for (int k = 0; k < L2N_R; ++k)
        if ((int)threadIdx.x + 1024 * k < total) y[off_y[k]] = v[k] * inv;
}

__global__ __launch_bounds__(1024) void l2norm_bwd_small_kernel(const float* __restrict__ x, int ldx,
                                                                const float* __restrict__ dy, int lddy, int rows, int c,
                                                                const float* __restrict__ stat, float* __restrict__ dx,
                                                                int lddx, int accumulate) {
    __shared__ double sh16[16];
    x += blockIdx.x * c + (int64_t)blockIdx.y * rows * ldx;
    dy += blockIdx.x * c + (int64_t)blockIdx.y * rows * lddy;
    dx += blockIdx.x * c + (int64_t)blockIdx.y * rows * lddx;
    stat += 2 * (blockIdx.x + blockIdx.y * gridDim.x);
    const int total = rows * c;
    const float inv = stat[1], ss = stat[0];
    float xv[L2N_R], gv[L2N_R], old[L2N_R];
    int off_d[L2N_R];
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < L2N_R; ++k) {
        const int i = threadIdx.x + 1024 * k;
        const int r = i / c, q = i - r * c;
        const bool in = i < total;
        xv[k] = in ? x[r * ldx + q] : 0.0f;
        gv[k] = in ? dy[r * lddy + q] : 0.0f;
        off_d[k] = r * lddx + q;
        old[k] = in && accumulate ? dx[off_d[k]] : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < L2N_R; ++k) s += (double)xv[k] * (double)gv[k];
    const float dot = (float)l2n_block_sum(s, sh16);
    const float coef = ss > 1e-12f ? dot * inv * inv * inv : 0.0f;
#pragma unroll
    for (int k = 0; k < L2N_R; ++k)
        if ((int)threadIdx.x + 1024 * k < total) {
            const float g = gv[k] * inv - xv[k] * coef;
            dx[off_d[k]] = accumulate ? old[k] + g : g;
        }
}

// patch-NCE (cut_wrapper.py:360-420): per sample logits[p][q] = <g_p, r_q>/tau over e, labels = eye(P) flattened:
//   loss_n = P * logsumexp(all P^2 logits) - sum_p logits[p][p];  d logits = P*softmax - eye.
__global__ void nce_kernel(const float* __restrict__ g, int64_t ldg, const float* __restrict__ r, int64_t ldr,
                           int64_t n, int p, int e, float inv_tau, float gcoef, float* __restrict__ loss_ps,
                           float* __restrict__ dg, int64_t lddg, int acc_dg, float* __restrict__ dr, int64_t lddr,
                           int acc_dr) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const float* gs = g + s * ldg;
    const float* rs = r + s * ldr;
    auto logit = [&](int a, int b) {
        float d = 0.0f;
        for (int k = 0; k < e; ++k) d += gs[a * e + k] * rs[b * e + k];
        return d * inv_tau;
    };
    float mx = -3.4e38f;
    for (int a = 0; a < p; ++a)
        for (int b = 0; b < p; ++b) mx = fmaxf(mx, logit(a, b));
    float se = 0.0f, diag = 0.0f;
    for (int a = 0; a < p; ++a)
        for (int b = 0; b < p; ++b) {
            const float l = logit(a, b);
            se += expf(l - mx);
            if (a == b) diag += l;
        }
    const float lse = mx + logf(se);
    loss_ps[s] = (float)p * lse - diag;
    if (!dg && !dr) return;
    const float inv_se = 1.0f / se;
    if (dg)
        for (int a = 0; a < p; ++a)
            for (int k = 0; k < e; ++k) {
                float acc = 0.0f;
                for (int b = 0; b < p; ++b) {
                    const float dl = (float)p * expf(logit(a, b) - mx) * inv_se - (a == b ? 1.0f : 0.0f);
                    acc += dl * rs[b * e + k];
                }
                float* q = dg + s * lddg + a * e + k;
                *q = (acc_dg ? *q : 0.0f) + gcoef * inv_tau * acc;
            }
    if (dr)
        for (int b = 0; b < p; ++b)
            for (int k = 0; k < e; ++k) {
                float acc = 0.0f;
                for (int a = 0; a < p; ++a) {
                    const float dl = (float)p * expf(logit(a, b) - mx) * inv_se - (a == b ? 1.0f : 0.0f);
                    acc += dl * gs[a * e + k];
                }
                float* q = dr + s * lddr + b * e + k;
                *q = (acc_dr ? *q : 0.0f) + gcoef * inv_tau * acc;
            }
}

// The same with P and E known at compile time (the reference's defaults: 6 or 7 slices of the band axis, embedding size
// 2): both operand rows and the P^2 logits live in registers, every logit and every exponential is computed ONCE (the
// generic kernel above walks the logits four times with its operands re-read from memory inside the loops: 57 us for
// 4096 samples, this one ~6).  Same arithmetic order per logit; the sums run in the same (a, b) order.
template <int P, int E>
__global__ void nce_fixed_kernel(const float* __restrict__ g, int64_t ldg, const float* __restrict__ r, int64_t ldr,
                                 int64_t n, float inv_tau, float gcoef, float* __restrict__ loss_ps,
                                 float* __restrict__ dg, int64_t lddg, int acc_dg, float* __restrict__ dr, int64_t lddr,
                                 int acc_dr) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    float gv[P * E], rv[P * E], l[P][P];
#pragma unroll
    for (int i = 0; i < P * E; ++i) {
        gv[i] = g[s * ldg + i];
        rv[i] = r[s * ldr + i];
    }
    float mx = -3.4e38f;
#pragma unroll
    for (int a = 0; a < P; ++a)
#pragma unroll
        for (int b = 0; b < P; ++b) {
            float d = 0.0f;
#pragma unroll
            for (int k = 0; k < E; ++k) d += gv[a * E + k] * rv[b * E + k];
            l[a][b] = d * inv_tau;
            mx = fmaxf(mx, l[a][b]);
        }
    float se = 0.0f, diag = 0.0f;
#pragma unroll
    for (int a = 0; a < P; ++a)
#pragma unroll
        for (int b = 0; b < P; ++b) {
            const float ex = expf(l[a][b] - mx);
            se += ex;
            if (a == b) diag += l[a][b];
            l[a][b] = ex;  // from here on: the exponentials
        }
    loss_ps[s] = (float)P * (mx + logf(se)) - diag;
    if (!dg && !dr) return;
    const float inv_se = 1.0f / se;
#pragma unroll
    for (int a = 0; a < P; ++a)
#pragma unroll
        for (int b = 0; b < P; ++b) l[a][b] = (float)P * l[a][b] * inv_se - (a == b ? 1.0f : 0.0f);  // d logits
    if (dg) {
#pragma unroll
        for (int a = 0; a < P; ++a)
#pragma unroll
            for (int k = 0; k < E; ++k) {
                float acc = 0.0f;
#pragma unroll
                for (int b = 0; b < P; ++b) acc += l[a][b] * rv[b * E + k];
                float* q = dg + s * lddg + a * E + k;
                *q = (acc_dg ? *q : 0.0f) + gcoef * inv_tau * acc;
            }
    }
    if (dr) {
#pragma unroll
        for (int b = 0; b < P; ++b)
#pragma unroll
            for (int k = 0; k < E; ++k) {
                float acc = 0.0f;
#pragma unroll
                for (int a = 0; a < P; ++a) acc += l[a][b] * gv[a * E + k];
                float* q = dr + s * lddr + b * E + k;
                *q = (acc_dr ? *q : 0.0f) + gcoef * inv_tau * acc;
            }
    }
}

}  // namespace

#define ST ((hipStream_t)stream)

static size_t gen_fwd_lds(int bands, int only_encoder) {
    const GenLayout g = gen_layout(bands, only_encoder);
    return sizeof(float) * ((size_t)g.wtotal + 8 + (size_t)GEN_WAVES * 8 * bands);
}
static size_t gen_bwd_lds(int bands, int only_encoder) {
    const GenLayout g = gen_layout(bands, only_encoder);
    return sizeof(float) * ((size_t)g.wtotal + 8 + (size_t)GEN_WAVES * ((size_t)22 * bands + g.wtotal + 8));
}

static size_t gt_fwd_lds(int bands, int only_encoder) {
    const GenTiled g = gen_tiled(bands, only_encoder);
    return sizeof(float) * ((size_t)g.wptotal + 8 + (size_t)GT_WAVES * 4 * g.W);
}

static size_t gt_bwd_lds(int bands, int only_encoder) {
    const GenTiled g = gen_tiled(bands, only_encoder);
    const size_t slope_words = (size_t)(6 * bands + 15) / 16 * 4;
    return sizeof(float) * ((size_t)g.wptotal + g.wftotal + 8 +
                            (size_t)GT_WAVES * ((size_t)8 * g.W + 3 * g.B4 + slope_words + g.wptotal + 8));
}

// gan_mfma.hip: the generator on the matrix cores for wide spectra (bands > 128).  HYPEL_GAN_MFMA=0 keeps the
// register-tiled VALU kernels below (experiments, parity cross-checks).
bool hypel_gm_supported(int bands);
int64_t hypel_gm_keep_floats(int64_t n, int bands, int only_encoder);
int hypel_gm_fwd(const float* x, int64_t ldx, int64_t n, int bands, const float* w, const float* b, int only_encoder,
                 float* out, int64_t ldo, int blocks, hipStream_t st, float* keep, float* enc_out = nullptr,
                 int64_t ld_enc = 0, int n_apps = 1, int64_t w_stride = 0, int64_t b_stride = 0);
int hypel_gm_bwd(const float* x, int64_t ldx, const float* dout, int64_t lddo, int64_t n, int bands, const float* w,
                 const float* b, int only_encoder, float* dx, int64_t lddx, int accumulate_dx, float* pw, float* pb,
                 int blocks, hipStream_t st, const float* keep, const float* d_enc = nullptr, int64_t ld_denc = 0,
                 int n_apps = 1, int64_t w_stride = 0, int64_t b_stride = 0, int64_t pw_stride = 0, int64_t pb_stride = 0);

static bool gan_use_mfma(int bands) {
    static const int on = getenv("HYPEL_GAN_MFMA") ? atoi(getenv("HYPEL_GAN_MFMA")) : 1;
    return on && hypel_gm_supported(bands);
}

extern "C" int hypel_gan_generator_blocks(int64_t n) {
    int64_t b = (n + GEN_WAVES - 1) / GEN_WAVES;
    if (b < 1) b = 1;
    if (b > 512) b = 512;
    return (int)b;
}

static int gan_generator_fwd_impl(const float* x, int64_t ldx, int64_t n, int32_t bands, const float* w, const float* b,
                                  int32_t only_encoder, float* out, int64_t ldo, float* keep, hypel_stream_t stream) {
    HYPEL_REQUIRE(x && w && b && out && n > 0 && bands >= 8, "hypel_gan_generator_fwd");
    if (gan_use_mfma(bands)) {
        hypel_gm_fwd(x, ldx, n, bands, w, b, only_encoder, out, ldo, 2 * hypel_gan_generator_blocks(n), ST, keep);
        HYPEL_CHECK_LAUNCH("hypel_gan_generator_fwd");
        return 0;
    }
    if (bands > GEN_TILED_MIN && gt_fwd_lds(bands, only_encoder) <= 160 * 1024 - 512) {  // 512: the static layer table
        const size_t tl = gt_fwd_lds(bands, only_encoder);
        if (tl > 64 * 1024)
            (void)hipFuncSetAttribute((const void*)gan_generator_fwd_tiled_kernel,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)tl);
        hipLaunchKernelGGL(gan_generator_fwd_tiled_kernel, dim3(hypel_gan_generator_blocks(n)), dim3(GT_SLOT_THREADS * GT_WAVES), tl,
                           ST, x, ldx, n, bands, w, b, only_encoder, out, ldo);
        HYPEL_CHECK_LAUNCH("hypel_gan_generator_fwd");
        return 0;
    }
    const size_t lds = gen_fwd_lds(bands, only_encoder);
    HYPEL_REQUIRE(lds <= 160 * 1024 - 512, "hypel_gan_generator_fwd");
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)gan_generator_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
    hipLaunchKernelGGL(gan_generator_fwd_kernel, dim3(hypel_gan_generator_blocks(n)), dim3(64 * GEN_WAVES), lds, ST, x,
                       ldx, n, bands, w, b, only_encoder, out, ldo);
    HYPEL_CHECK_LAUNCH("hypel_gan_generator_fwd");
    return 0;
}

extern "C" int hypel_gan_generator_fwd(const float* x, int64_t ldx, int64_t n, int32_t bands, const float* w,
                                       const float* b, int32_t only_encoder, float* out, int64_t ldo,
                                       hypel_stream_t stream) {
    return gan_generator_fwd_impl(x, ldx, n, bands, w, b, only_encoder, out, ldo, nullptr, stream);
}

extern "C" int64_t hypel_gan_generator_keep_floats(int64_t n, int32_t bands, int32_t only_encoder) {
    return n > 0 && gan_use_mfma(bands) ? hypel_gm_keep_floats(n, bands, only_encoder) : 0;
}

extern "C" int hypel_gan_generator_fwd_keep(const float* x, int64_t ldx, int64_t n, int32_t bands, const float* w,
                                            const float* b, int32_t only_encoder, float* out, int64_t ldo, float* keep,
                                            hypel_stream_t stream) {
    HYPEL_REQUIRE(keep == nullptr || hypel_gan_generator_keep_floats(n, bands, only_encoder) > 0,
                  "hypel_gan_generator_fwd_keep: no kept activations for this band count");
    return gan_generator_fwd_impl(x, ldx, n, bands, w, b, only_encoder, out, ldo, keep, stream);
}

/* The encoder tap (include/hypel.h): the full generator also leaves n_4 -- the value an encoder-only application on the same
 * input would compute, bit for bit -- resp. takes the gradient that reached that value.  Matrix-core kernels only. */
extern "C" int hypel_gan_generator_tap_supported(int32_t bands) { return gan_use_mfma(bands) ? 1 : 0; }

extern "C" int hypel_gan_generator_fwd_tap(const float* x, int64_t ldx, int64_t n, int32_t bands, const float* w,
                                           const float* b, float* out, int64_t ldo, float* enc_out, int64_t ld_enc,
                                           float* keep, hypel_stream_t stream) {
    HYPEL_REQUIRE(x && w && b && out && enc_out && n > 0 && gan_use_mfma(bands), "hypel_gan_generator_fwd_tap");
    hypel_gm_fwd(x, ldx, n, bands, w, b, 0, out, ldo, 2 * hypel_gan_generator_blocks(n), ST, keep, enc_out, ld_enc);
    HYPEL_CHECK_LAUNCH("hypel_gan_generator_fwd_tap");
    return 0;
}

extern "C" int hypel_gan_generator_bwd_tap(const float* x, int64_t ldx, const float* dout, int64_t lddo,
                                           const float* d_enc, int64_t ld_denc, int64_t n, int32_t bands, const float* w,
                                           const float* b, float* dx, int64_t lddx, int32_t accumulate_dx, float* pw,
                                           float* pb, const float* keep, hypel_stream_t stream) {
    HYPEL_REQUIRE(x && dout && d_enc && w && b && pw && pb && n > 0 && gan_use_mfma(bands), "hypel_gan_generator_bwd_tap");
    hypel_gm_bwd(x, ldx, dout, lddo, n, bands, w, b, 0, dx, lddx, accumulate_dx, pw, pb, hypel_gan_generator_blocks(n), ST,
                 keep, d_enc, ld_denc);
    HYPEL_CHECK_LAUNCH("hypel_gan_generator_bwd_tap");
    return 0;
}

/* Several generators of one band count in one launch (include/hypel.h): CycleGAN's G_x2y(x) and G_y2x(y). */
extern "C" int hypel_gan_generator_blocks_apps(int64_t n, int32_t n_apps) {
    if (n_apps < 1) n_apps = 1;
    int64_t t = (n + 15) / 16, per = 512 / n_apps;
    if (per < 1) per = 1;
    if (t < 1) t = 1;
    if (t > per) t = per;
    return (int)(t * n_apps);
}

extern "C" int hypel_gan_generator_fwd_apps(const float* x, int64_t ldx, int64_t n, int32_t n_apps, int64_t w_stride,
                                            int64_t b_stride, int32_t bands, const float* w, const float* b,
                                            int32_t only_encoder, float* out, int64_t ldo, float* keep,
                                            hypel_stream_t stream) {
    HYPEL_REQUIRE(x && w && b && out && n > 0 && n_apps >= 1 && n_apps <= 16 && gan_use_mfma(bands),
                  "hypel_gan_generator_fwd_apps");
    hypel_gm_fwd(x, ldx, n, bands, w, b, only_encoder, out, ldo, hypel_gan_generator_blocks_apps(n, n_apps), ST, keep, nullptr,
                 0, n_apps, w_stride, b_stride);
    HYPEL_CHECK_LAUNCH("hypel_gan_generator_fwd_apps");
    return 0;
}

extern "C" int hypel_gan_generator_bwd_apps(const float* x, int64_t ldx, const float* dout, int64_t lddo, int64_t n,
                                            int32_t n_apps, int64_t w_stride, int64_t b_stride, int64_t pw_stride,
                                            int64_t pb_stride, int32_t bands,
                                            const float* w, const float* b, int32_t only_encoder, float* dx, int64_t lddx,
                                            int32_t accumulate_dx, float* pw, float* pb, const float* keep,
                                            hypel_stream_t stream) {
    HYPEL_REQUIRE(x && dout && w && b && pw && pb && n > 0 && n_apps >= 1 && n_apps <= 16 && gan_use_mfma(bands),
                  "hypel_gan_generator_bwd_apps");
    hypel_gm_bwd(x, ldx, dout, lddo, n, bands, w, b, only_encoder, dx, lddx, accumulate_dx, pw, pb,
                 hypel_gan_generator_blocks_apps(n, n_apps), ST, keep, nullptr, 0, n_apps, w_stride, b_stride, pw_stride,
                 pb_stride);
    HYPEL_CHECK_LAUNCH("hypel_gan_generator_bwd_apps");
    return 0;
}

static int gan_generator_bwd_impl(const float* x, int64_t ldx, const float* dout, int64_t lddo, int64_t n, int32_t bands,
                                  const float* w, const float* b, int32_t only_encoder, float* dx, int64_t lddx,
                                  int32_t accumulate_dx, float* pw, float* pb, const float* keep, hypel_stream_t stream) {
    HYPEL_REQUIRE(x && dout && w && b && pw && pb && n > 0 && bands >= 8, "hypel_gan_generator_bwd");
    if (gan_use_mfma(bands)) {
        hypel_gm_bwd(x, ldx, dout, lddo, n, bands, w, b, only_encoder, dx, lddx, accumulate_dx, pw, pb,
                     hypel_gan_generator_blocks(n), ST, keep);
        HYPEL_CHECK_LAUNCH("hypel_gan_generator_bwd");
        return 0;
    }
    if (bands > GEN_TILED_MIN && gt_bwd_lds(bands, only_encoder) <= 160 * 1024 - 512) {  // 512: the static layer table
        const size_t tl = gt_bwd_lds(bands, only_encoder);
        if (tl > 64 * 1024)
            (void)hipFuncSetAttribute((const void*)gan_generator_bwd_tiled_kernel,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)tl);
        hipLaunchKernelGGL(gan_generator_bwd_tiled_kernel, dim3(hypel_gan_generator_blocks(n)), dim3(GT_SLOT_THREADS * GT_WAVES), tl,
                           ST, x, ldx, dout, lddo, n, bands, w, b, only_encoder, dx, lddx, accumulate_dx, pw, pb);
        HYPEL_CHECK_LAUNCH("hypel_gan_generator_bwd");
        return 0;
    }
    const size_t lds = gen_bwd_lds(bands, only_encoder);
    HYPEL_REQUIRE(lds <= 160 * 1024 - 512, "hypel_gan_generator_bwd");
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)gan_generator_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
    hipLaunchKernelGGL(gan_generator_bwd_kernel, dim3(hypel_gan_generator_blocks(n)), dim3(64 * GEN_WAVES), lds, ST, x,
                       ldx, dout, lddo, n, bands, w, b, only_encoder, dx, lddx, accumulate_dx, pw, pb);
    HYPEL_CHECK_LAUNCH("hypel_gan_generator_bwd");
    return 0;
}

extern "C" int hypel_gan_generator_bwd(const float* x, int64_t ldx, const float* dout, int64_t lddo, int64_t n,
                                       int32_t bands, const float* w, const float* b, int32_t only_encoder, float* dx,
                                       int64_t lddx, int32_t accumulate_dx, float* pw, float* pb,
                                       hypel_stream_t stream) {
    return gan_generator_bwd_impl(x, ldx, dout, lddo, n, bands, w, b, only_encoder, dx, lddx, accumulate_dx, pw, pb,
                                  nullptr, stream);
}

extern "C" int hypel_gan_generator_bwd_kept(const float* x, int64_t ldx, const float* dout, int64_t lddo, int64_t n,
                                            int32_t bands, const float* w, const float* b, int32_t only_encoder,
                                            float* dx, int64_t lddx, int32_t accumulate_dx, float* pw, float* pb,
                                            const float* keep, hypel_stream_t stream) {
    HYPEL_REQUIRE(keep == nullptr || hypel_gan_generator_keep_floats(n, bands, only_encoder) > 0,
                  "hypel_gan_generator_bwd_kept: no kept activations for this band count");
    return gan_generator_bwd_impl(x, ldx, dout, lddo, n, bands, w, b, only_encoder, dx, lddx, accumulate_dx, pw, pb, keep,
                                  stream);
}

extern "C" int hypel_gan_loss(int32_t mode, const float* a, int64_t lda, const float* b, int64_t ldb, int64_t rows,
                              int32_t c, float target, float weight, float* loss, int32_t accumulate_loss, float* da,
                              int64_t ldda, int32_t acc_da, float* db, int64_t lddb, int32_t acc_db, float* ws,
                              hypel_stream_t stream) {
    HYPEL_REQUIRE(a && loss && ws && rows > 0 && c > 0 && mode >= 0 && mode <= 2, "hypel_gan_loss");
    HYPEL_REQUIRE(mode != 1 || b != nullptr, "hypel_gan_loss");
    const int grid = (int)(rows < 1024 ? rows : 1024);
    const double count = (double)rows * c;
    hipLaunchKernelGGL(gan_loss_partial_kernel, dim3(grid), dim3(256), 0, ST, mode, a, lda, b, ldb, rows, c, target,
                       (float)(weight / count), da, ldda, acc_da, db, lddb, acc_db, ws, 1.0f);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(LOSS_FIN_THREADS), 0, ST, ws, grid, (double)weight / count, loss,
                       accumulate_loss);
    HYPEL_CHECK_LAUNCH("hypel_gan_loss");
    return 0;
}

// ---- loss terms with a DEFERRED sum: every term of a train op leaves its (already weighted) block partials in its own
// 1024-float slot; ONE hypel_loss_finalize_slots at the end of the op adds the slots up in index order, and terms that
// write different gradient buffers share one launch.  A CycleGAN generator op has six terms + regularisers: thirteen
// launches of ~4.7 us become three.
constexpr int LOSS_SLOT = 1024;

// Several loss terms in ONE launch (blockIdx.y = term; every operand addressed relative to one base pointer, like the
// merged filter-gradient products): modes 0-2 as hypel_gan_loss, mode 3 = l2 regulariser (a = w, rows = count, c = 1,
// da = dw, always accumulated, gcoef = scale).  A term's weighted block partials go to its slot.
__global__ __launch_bounds__(256) void loss_terms_kernel(const float* __restrict__ base,
                                                          const hypel_loss_term_t* __restrict__ terms,
                                                          float* __restrict__ slots) {
    __shared__ float sh[4];
    const hypel_loss_term_t t = terms[blockIdx.y];
    const float* __restrict__ a = base + t.a_off;
    const float* __restrict__ b = t.b_off == HYPEL_LOSS_NONE ? nullptr : base + t.b_off;
    float* da = t.da_off == HYPEL_LOSS_NONE ? nullptr : const_cast<float*>(base) + t.da_off;
    float* db = t.db_off == HYPEL_LOSS_NONE ? nullptr : const_cast<float*>(base) + t.db_off;
    float s = 0.0f;
    if (t.mode == 3) {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < t.rows; i += (int64_t)gridDim.x * 256) {
            const float v = a[i];
            s += v * v;
            if (da) da[i] += t.gcoef * v;
        }
    } else {
        // a block walks 256 / cp rows at a time (cp = the column count rounded up to a power of two): the [2048 x 32..64]
        // critic outputs of a train op are one pass of 128-256 blocks instead of two dependent passes of 1024 blocks with
        // 3/4 - 7/8 of their threads idle
        int cp = 1;
        while (cp < t.c && cp < 256) cp <<= 1;
        const int rpb = 256 / cp, ty = threadIdx.x / cp, tx = threadIdx.x & (cp - 1);
        for (int64_t row = (int64_t)blockIdx.x * rpb + ty; row < t.rows; row += (int64_t)gridDim.x * rpb)
            for (int col = tx; col < t.c; col += cp) {
                const float av = a[row * t.lda + col];
                float val, ga, gb = 0.0f;
                if (t.mode == 0) {
                    const float d = av - t.target;
                    val = d * d;
                    ga = 2.0f * d;
                } else if (t.mode == 1) {
                    const float d = av - b[row * t.ldb + col];
                    val = fabsf(d);
                    ga = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
                    gb = -ga;
                } else {
                    val = av;
                    ga = 1.0f;
                }
                s += val;
                if (da) {
                    float* p = da + row * t.ldda + col;
                    *p = (t.acc_da ? *p : 0.0f) + t.gcoef * ga;
                }
                if (db) {
                    float* p = db + row * t.lddb + col;
                    *p = (t.acc_db ? *p : 0.0f) + t.gcoef * gb;
                }
            }
    }
    const float tot = block_sum_256(s, sh);
    if (threadIdx.x == 0) slots[(int64_t)t.slot * LOSS_SLOT + blockIdx.x] = tot * t.pscale;
}

extern "C" int hypel_loss_terms_slots(const float* base, const hypel_loss_term_t* terms, int32_t n_terms, float* slots,
                                      hypel_stream_t stream) {
    HYPEL_REQUIRE(base && terms && slots && n_terms > 0, "hypel_loss_terms_slots");
    hipLaunchKernelGGL(loss_terms_kernel, dim3(LOSS_SLOT, n_terms), dim3(256), 0, ST, base, terms, slots);
    HYPEL_CHECK_LAUNCH("hypel_loss_terms_slots");
    return 0;
}

extern "C" int hypel_loss_finalize_slots(const float* slots, int32_t n_slots, float* loss, int32_t accumulate_loss,
                                         hypel_stream_t stream) {
    HYPEL_REQUIRE(slots && loss && n_slots > 0, "hypel_loss_finalize_slots");
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(LOSS_FIN_THREADS), 0, ST, slots, n_slots * LOSS_SLOT, 1.0, loss,
                       accumulate_loss);
    HYPEL_CHECK_LAUNCH("hypel_loss_finalize_slots");
    return 0;
}

extern "C" int hypel_l2_reg(const float* w, int64_t count, float scale, float* loss, int32_t accumulate_loss, float* dw,
                            float* ws, hypel_stream_t stream) {
    HYPEL_REQUIRE(w && loss && ws && count > 0, "hypel_l2_reg");
    const int grid = hypel_grid_1d(count, 256, 1024);
    hipLaunchKernelGGL(l2_reg_kernel, dim3(grid), dim3(256), 0, ST, w, count, scale, dw, ws, 1.0f);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(LOSS_FIN_THREADS), 0, ST, ws, grid, 0.5 * (double)scale, loss,
                       accumulate_loss);
    HYPEL_CHECK_LAUNCH("hypel_l2_reg");
    return 0;
}

extern "C" int hypel_l2norm_segs_fwd(const float* x, int64_t ldx, int64_t rows, int32_t c, int32_t parts, int32_t segs,
                                     float* y, int64_t ldy, float* stat, hypel_stream_t stream) {
    HYPEL_REQUIRE(x && y && stat && rows > 0 && c > 0 && parts > 0 && segs > 0, "hypel_l2norm_segs_fwd");
    const bool small = rows * c <= L2N_R * 1024 && rows * segs * (ldx > ldy ? ldx : ldy) < (1ll << 30);
    if (small)
        hipLaunchKernelGGL(l2norm_fwd_small_kernel, dim3(parts, segs), dim3(1024), 0, ST, x, (int)ldx, (int)rows, c, y,
                           (int)ldy, stat);
    else
        hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(parts, segs), dim3(1024), 0, ST, x, ldx, rows, c, y, ldy, stat);
    HYPEL_CHECK_LAUNCH("hypel_l2norm_segs_fwd");
    return 0;
}

extern "C" int hypel_l2norm_segs_bwd(const float* x, int64_t ldx, const float* dy, int64_t lddy, int64_t rows,
                                     int32_t c, int32_t parts, int32_t segs, const float* stat, float* dx, int64_t lddx,
                                     int32_t accumulate, hypel_stream_t stream) {
    HYPEL_REQUIRE(x && dy && dx && stat && rows > 0 && c > 0 && parts > 0 && segs > 0, "hypel_l2norm_segs_bwd");
    const int64_t ldmax = ldx > lddy ? (ldx > lddx ? ldx : lddx) : (lddy > lddx ? lddy : lddx);
    if (rows * c <= L2N_R * 1024 && rows * segs * ldmax < (1ll << 30))
        hipLaunchKernelGGL(l2norm_bwd_small_kernel, dim3(parts, segs), dim3(1024), 0, ST, x, (int)ldx, dy, (int)lddy,
                           (int)rows, c, stat, dx, (int)lddx, accumulate);
    else
        hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(parts, segs), dim3(1024), 0, ST, x, ldx, dy, lddy, rows, c, stat, dx,
                           lddx, accumulate);
    HYPEL_CHECK_LAUNCH("hypel_l2norm_segs_bwd");
    return 0;
}

extern "C" int hypel_l2norm_parts_fwd(const float* x, int64_t ldx, int64_t rows, int32_t c, int32_t parts, float* y,
                                      int64_t ldy, float* stat, hypel_stream_t stream) {
    return hypel_l2norm_segs_fwd(x, ldx, rows, c, parts, 1, y, ldy, stat, stream);
}

extern "C" int hypel_l2norm_parts_bwd(const float* x, int64_t ldx, const float* dy, int64_t lddy, int64_t rows,
                                      int32_t c, int32_t parts, const float* stat, float* dx, int64_t lddx,
                                      int32_t accumulate, hypel_stream_t stream) {
    return hypel_l2norm_segs_bwd(x, ldx, dy, lddy, rows, c, parts, 1, stat, dx, lddx, accumulate, stream);
}

extern "C" int hypel_l2norm_fwd(const float* x, int64_t ldx, int64_t rows, int32_t c, float* y, int64_t ldy,
                                float* stat, hypel_stream_t stream) {
    return hypel_l2norm_parts_fwd(x, ldx, rows, c, 1, y, ldy, stat, stream);
}

extern "C" int hypel_l2norm_bwd(const float* x, int64_t ldx, const float* dy, int64_t lddy, int64_t rows, int32_t c,
                                const float* stat, float* dx, int64_t lddx, int32_t accumulate,
                                hypel_stream_t stream) {
    return hypel_l2norm_parts_bwd(x, ldx, dy, lddy, rows, c, 1, stat, dx, lddx, accumulate, stream);
}

extern "C" int hypel_nce_loss(const float* g, int64_t ldg, const float* r, int64_t ldr, int64_t n, int32_t p,
                              int32_t e, float tau, float weight, float* loss, int32_t accumulate_loss, float* dg,
                              int64_t lddg, int32_t acc_dg, float* dr, int64_t lddr, int32_t acc_dr, float* ws,
                              hypel_stream_t stream) {
    HYPEL_REQUIRE(g && r && loss && ws && n > 0 && p > 0 && e > 0 && tau > 0.0f, "hypel_nce_loss");
    // ws: [n] per-sample losses followed by 1024 floats of reduction scratch
    const dim3 grid((unsigned)((n + 63) / 64));
    const float gcoef = (float)(weight / (double)n);
#define HYPEL_NCE(P_, E_)                                                                                              \
    hipLaunchKernelGGL((nce_fixed_kernel<P_, E_>), grid, dim3(64), 0, ST, g, ldg, r, ldr, n, 1.0f / tau, gcoef, ws, dg, \
                       lddg, acc_dg, dr, lddr, acc_dr)
    if (p == 6 && e == 2) HYPEL_NCE(6, 2);
    else if (p == 7 && e == 2) HYPEL_NCE(7, 2);
    else
        hipLaunchKernelGGL(nce_kernel, grid, dim3(64), 0, ST, g, ldg, r, ldr, n, p, e, 1.0f / tau, gcoef, ws, dg, lddg,
                           acc_dg, dr, lddr, acc_dr);
#undef HYPEL_NCE
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(LOSS_FIN_THREADS), 0, ST, ws, (int)n, (double)weight / (double)n, loss,
                       accumulate_loss);
    HYPEL_CHECK_LAUNCH("hypel_nce_loss");
    return 0;
}
