// HBM-bound kernels of the classifier path: layout conversion, batch-norm statistics, the fused
// normalise/activate/dropout/residual post-op and its two-pass backward, channel-map gradient,
// losses, optimisers, dropout masks, metrics, LRN.  Every reduction is two-stage with a fixed
// combination order (no float atomics) so that results are run-to-run deterministic -- the
// "per-pixel class labels bit-exact" requirement of BASELINE.json needs that.
#include <stdlib.h>

#include "common.h"


namespace {

constexpr int STAT_TX = 64;  // threads along channels
constexpr int STAT_TY = 4;   // row lanes

// Elementwise kernels map a 256-thread block onto [TY rows x TX column-vectors] (TX = power of two >= the
// number of column vectors, capped at 256), so (row, col) needs no division, consecutive lanes touch
// consecutive 16-byte vectors of a row, and the grid (up to 8192 blocks) keeps every CU's queues full.
struct EwShape {
    int tx_log2;  // log2(TX)
    int grid;
};
static inline EwShape ew_shape(int64_t rows, int cvec) {
    int l = 0;
    while ((1 << l) < cvec && l < 8) ++l;
    const int ty = 256 >> l;
    int64_t g = (rows + ty - 1) / ty;
    if (g < 1) g = 1;
    constexpr int cap = 8192;
    if (g > cap) g = cap;
    return EwShape{l, (int)g};
}
static inline bool aligned16(const void* p) { return p == nullptr || (((uintptr_t)p) & 15) == 0; }

template <int VEC>
struct Vec;
template <>
struct Vec<1> {
    float v[1];
    __device__ __forceinline__ void load(const float* p) { v[0] = p[0]; }
    __device__ __forceinline__ void store(float* p) const { p[0] = v[0]; }
    __device__ __forceinline__ void load_param(const float* p) { v[0] = p[0]; }
};
template <>
struct Vec<4> {
    float v[4];
    __device__ __forceinline__ void load(const float* p) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    __device__ __forceinline__ void store(float* p) const {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
    // per-channel parameter vectors live at arbitrary offsets of the flat parameter buffer: dword loads
    __device__ __forceinline__ void load_param(const float* p) {
        v[0] = p[0]; v[1] = p[1]; v[2] = p[2]; v[3] = p[3];
    }
};

template <int VEC, class F>
__device__ __forceinline__ void ew_loop(int64_t rows, int c, int tx_log2, F body) {
    const int TX = 1 << tx_log2;
    const int TY = 256 >> tx_log2;
    const int tx = threadIdx.x & (TX - 1);
    const int ty = threadIdx.x >> tx_log2;
    const int cv = c / VEC;
    // columns outside, rows inside: a thread's column is fixed while it walks down the rows, so the per-channel
    // operands of the body (statistics, beta, channel-map indices) are loop invariant and get hoisted
    for (int cvi = tx; cvi < cv; cvi += TX)
        for (int64_t row = (int64_t)blockIdx.x * TY + ty; row < rows; row += (int64_t)gridDim.x * TY) body(row, cvi * VEC);
}

// ------------------------------------------------------------------------------------- layout
// One wavefront per output row (pixel pp, sample nn): the row's c floats are contiguous on both sides, and the index
// arithmetic (two divisions) is paid once per row instead of four 64-bit divisions per element.
__global__ __launch_bounds__(256) void nhwc_to_pnc_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n,
                                                          int p, int c, int64_t ld) {
    constexpr int ROWS = 4;  // rows a wave moves per pass: all their loads are in flight before the first store
    const int lane = threadIdx.x & 63;
    const int64_t n_rows = (int64_t)p * n;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
    for (int64_t r0 = wave * ROWS; r0 < n_rows; r0 += n_waves * ROWS) {
        for (int c0 = 0; c0 < (int)ld; c0 += 64 * 3) {
            float v[ROWS][3];
#pragma unroll
            for (int k = 0; k < ROWS; ++k) {
                const int64_t rn = min(r0 + k, n_rows - 1);
                const int64_t pp = rn / n, nn = rn - pp * n;
                const float* __restrict__ src = x + (nn * p + pp) * c;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int cc = c0 + lane + 64 * j;
                    v[k][j] = cc < c ? src[cc] : 0.0f;
                }
            }
#pragma unroll
            for (int k = 0; k < ROWS; ++k) {
                if (r0 + k >= n_rows) break;
                float* __restrict__ dst = out + (r0 + k) * ld;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int cc = c0 + lane + 64 * j;
                    if (cc < (int)ld) dst[cc] = v[k][j];
                }
            }
        }
    }
}

__global__ void pnc_to_nhwc_kernel(const float* __restrict__ in, int64_t ld, float* __restrict__ x, int64_t n, int p,
                                   int c) {
    const int64_t total = n * p * c;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cc = (int)(i % c);
        const int64_t np_ = i / c;
        const int pp = (int)(np_ % p);
        const int64_t nn = np_ / p;
        x[i] = in[((int64_t)pp * n + nn) * ld + cc];
    }
}

__global__ void fill_kernel(float* __restrict__ dst, int64_t count, float v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = v;
}

// Many slabs, few outputs (the per-block filter-gradient partials of the fused generator, bias gradients): one
// wavefront per output, lanes stride over the slabs, fixed-order butterfly -> deterministic, and ~n_splits/64
// dependent loads per lane instead of n_splits.
__global__ void reduce_splits_wave_kernel(const float* __restrict__ partial, int64_t stride, int n_splits,
                                          float* __restrict__ out, int64_t count, int accumulate,
                                          const float* __restrict__ bias, int n, int64_t ldc) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t i = wave; i < count; i += n_waves) {
        const int64_t o = ldc > 0 ? (i / n) * ldc + (i % n) : i;
        float s = 0.0f;
        for (int k = lane; k < n_splits; k += 64) s += partial[(int64_t)k * stride + o];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) {
            if (bias) s += bias[(int)(i % n)];
            out[o] = accumulate ? out[o] + s : s;
        }
    }
}

// Two reductions over the same number of slabs in one launch (the filter and the bias gradients of a fused stack:
// generator, narrow discriminator): output i < count0 belongs to the first, the rest to the second; per output the same
// lanes-over-slabs butterfly as reduce_splits_wave_kernel (bit-identical to two separate launches).
__global__ void reduce_splits_wave_pair_kernel(const float* __restrict__ p0, int64_t stride0, int64_t count0,
                                               float* __restrict__ out0, const float* __restrict__ p1, int64_t stride1,
                                               int64_t count1, float* __restrict__ out1, int n_splits, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t i = wave; i < count0 + count1; i += n_waves) {
        const bool first = i < count0;
        const float* __restrict__ p = first ? p0 : p1;
        const int64_t stride = first ? stride0 : stride1, o = first ? i : i - count0;
        float* __restrict__ out = first ? out0 : out1;
        float s = 0.0f;
        for (int k = lane; k < n_splits; k += 64) s += p[(int64_t)k * stride + o];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) out[o] = accumulate ? out[o] + s : s;
    }
}

// Several slab reductions in one launch, one WAVE per output element (many slabs, few outputs: the per-block filter /
// bias gradient slabs of the fused generator and dense-stack kernels of a whole GAN phase): the outputs of all entries are
// numbered consecutively, a wave finds its entry by walking the (short) table; per output the same lanes-over-slabs
// butterfly as reduce_splits_wave_kernel, the old value added last.
// (round 4, second form) One block = 64 consecutive outputs x 16 slab groups: the 64 lanes of a wave read 64 CONSECUTIVE
// outputs of one slab (one 256-byte row segment per load; with a wave per output every lane touched its own cache line --
// 64 sectors per load instruction), thread (tx, ty) sums the slabs ty, ty + 16, .. of output tx in ascending order, the 16
// group sums are added in group order through LDS, the old value last (two contributions that are exact negatives of each
// other still cancel before it is added).  Deterministic; 10.9 -> ~7 us for the 256 x 10.5 k slabs of a CycleGAN train op.
constexpr int RSW_TX = 64, RSW_TY = 16;
__global__ __launch_bounds__(RSW_TX * RSW_TY) void reduce_splits_wave_multi_kernel(
    const float* __restrict__ base, const hypel_reduce_entry_t* __restrict__ entries, int n_entries, int64_t total) {
    __shared__ float sh[RSW_TY][RSW_TX];
    const int tx = threadIdx.x & (RSW_TX - 1), ty = threadIdx.x / RSW_TX;
    for (int64_t c0 = (int64_t)blockIdx.x * RSW_TX; c0 < total; c0 += (int64_t)gridDim.x * RSW_TX) {
        const int64_t i = c0 + tx;
        float s = 0.0f;
        int64_t o = i;
        int e = 0;
        const bool live = i < total;
        if (live) {
            while (e + 1 < n_entries && o >= entries[e].count) {
                o -= entries[e].count;
                ++e;
            }
            const hypel_reduce_entry_t en = entries[e];
            const float* __restrict__ p = base + en.partial_off + o;
            int k = ty;
            for (; k + 7 * RSW_TY < en.n_splits; k += 8 * RSW_TY) {  // eight slabs in flight
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = p[(int64_t)(k + q * RSW_TY) * en.stride];
#pragma unroll
                for (int q = 0; q < 8; ++q) s += v[q];
            }
            for (; k < en.n_splits; k += RSW_TY) s += p[(int64_t)k * en.stride];
        }
        sh[ty][tx] = s;
        __syncthreads();
        if (ty == 0 && live) {
            float t = sh[0][tx];
#pragma unroll
            for (int g = 1; g < RSW_TY; ++g) t += sh[g][tx];
            const hypel_reduce_entry_t en = entries[e];
            float* out = const_cast<float*>(base) + en.out_off + o;
            *out = (en.flags & 1) ? *out + t : t;
        }
        __syncthreads();
    }
}

// float4 variant (count, n, ldc, stride multiples of 4; 16-byte aligned bases)
__global__ void reduce_splits_v4_kernel(const float* __restrict__ partial, int64_t stride, int n_splits,
                                        float* __restrict__ out, int64_t count4, int accumulate,
                                        const float* __restrict__ bias, int n, int64_t ldc) {
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < count4; q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = q * 4;
        const int64_t o = ldc > 0 ? (i / n) * ldc + (i % n) : i;
        // the slabs are summed FIRST and the old value added last, as in the wave kernel: two contributions that are
        // exact negatives of each other (the last bias of a Wasserstein critic: +1/N and -1/N per sample) then cancel
        // exactly whichever kernel the slab count selects
        const float4 old = accumulate ? *reinterpret_cast<const float4*>(out + o) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias) {
            const float* bp = bias + (int)(i % n);
            s.x += bp[0]; s.y += bp[1]; s.z += bp[2]; s.w += bp[3];
        }
#pragma unroll 8
        for (int k = 0; k < n_splits; ++k) {
            const float4 v = *reinterpret_cast<const float4*>(partial + (int64_t)k * stride + o);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        s.x += old.x; s.y += old.y; s.z += old.z; s.w += old.w;
        *reinterpret_cast<float4*>(out + o) = s;
    }
}

__global__ void reduce_splits_kernel(const float* __restrict__ partial, int64_t stride, int n_splits,
                                     float* __restrict__ out, int64_t count, int accumulate,
                                     const float* __restrict__ bias, int n, int64_t ldc) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        // ldc > 0: element i is (row i / n, column i % n) of an [rows x n] window with leading dimension ldc,
        // on both the partial slabs and the output
        const int64_t o = ldc > 0 ? (i / n) * ldc + (i % n) : i;
        const float old = accumulate ? out[o] : 0.0f;
        float s = bias ? bias[(int)(i % n)] : 0.0f;
#pragma unroll 8
        for (int k = 0; k < n_splits; ++k) s += partial[(int64_t)k * stride + o];
        out[o] = s + old;  // slabs first, the old value last (see the float4 variant)
    }
}

// Several reductions in one launch: blockIdx.y = entry, blockIdx.x strides over the entry's elements (float4 when
// the entry's count / stride / addresses allow it).  Same summation order as reduce_splits_kernel.
__global__ void reduce_splits_multi_kernel(const float* __restrict__ base,
                                           const hypel_reduce_entry_t* __restrict__ entries) {
    const hypel_reduce_entry_t e = entries[blockIdx.y];
    const float* __restrict__ partial = base + e.partial_off;
    float* __restrict__ out = const_cast<float*>(base) + e.out_off;
    const bool acc = e.flags & 1;
    const bool v4 = (e.count % 4 == 0) && (e.stride % 4 == 0) && ((((uintptr_t)partial) | ((uintptr_t)out)) & 15) == 0;
    if (v4) {
        const int64_t count4 = e.count / 4;
        for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < count4;
             q += (int64_t)gridDim.x * blockDim.x) {
            const int64_t o = q * 4;
            float4 s = acc ? *reinterpret_cast<const float4*>(out + o) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
            for (int k = 0; k < e.n_splits; ++k) {
                const float4 v = *reinterpret_cast<const float4*>(partial + (int64_t)k * e.stride + o);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            *reinterpret_cast<float4*>(out + o) = s;
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < e.count;
             i += (int64_t)gridDim.x * blockDim.x) {
            float s = acc ? out[i] : 0.0f;
#pragma unroll 8
            for (int k = 0; k < e.n_splits; ++k) s += partial[(int64_t)k * e.stride + i];
            out[i] = s;
        }
    }
}

// ------------------------------------------------------------------------------------- BN statistics
// grid = (n_chunks, ceil(c/64)); block = 64 x 4.  Shifted sums (shift = first row of the chunk) keep the
// fp32 accumulation well conditioned; the chunk's (mean, M2) pair is then exact enough to Chan-merge in fp64.

// float4 variant: block = 16 column quads x 16 row lanes.  Four times the bytes in flight per block: narrow layers
// (c <= 128) launch only ~256-512 blocks, and at 8 KB in flight per CU the scalar kernel is latency bound.
constexpr int STAT_V4_TY = 16;

// Combine the chunk partials: mean = sum n_k mean_k / N, M2 = sum (M2_k + n_k (mean_k - mean)^2), both as fp64
// sums with a FIXED association (16 strided lanes per channel, then an xor tree) -> deterministic, and ~20x
// shorter than a serial Chan chain over ~200 chunks.  block = 16 channels x 16 lanes.
__device__ __forceinline__ double lanes16_sum(double v, double* sh) {
    // threads: ch = tid & 15, lane = tid >> 4 (0..15); wave holds 4 lanes of each channel
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    const int w = threadIdx.x >> 6, ch = threadIdx.x & 15;
    if ((threadIdx.x & 63) < 16) sh[w * 16 + ch] = v;
    __syncthreads();
    const double t = (sh[ch] + sh[16 + ch]) + (sh[32 + ch] + sh[48 + ch]);
    __syncthreads();
    return t;
}

// body of the statistics finaliser for the 16 channels [blk16*16, blk16*16+16); all 256 threads of a block call it
// The chunk partials are read in bursts of 16 per lane with clamped (always valid) addresses, so the loads of a
// burst are all in flight together; a plain `for k` loop made this kernel a chain of ~2 x n_chunks/16 dependent
// memory round trips (14 us for 256 chunks).
template <bool M2OUT = false>  // M2OUT: `rstd` receives the merged sum of squared deviations instead
__device__ __forceinline__ void bn_finalize_body(int blk16, const float* __restrict__ partial, int n_chunks,
                                                 int chunk_rows, int64_t rows, int c, float eps,
                                                 float* __restrict__ mean, float* __restrict__ rstd,
                                                 float* __restrict__ moving_mean, float* __restrict__ moving_var,
                                                 float decay, double* sh) {
    const int ch = threadIdx.x & 15, lane = threadIdx.x >> 4;
    const int col = blk16 * 16 + ch;
    const bool ok = col < c;
    const int colc = ok ? col : c - 1;
    const double n_total = (double)rows;
    const int64_t last_rows = rows - (int64_t)(n_chunks - 1) * chunk_rows;
    // ONE pass over the partials (round 2: the two-pass form -- mean first, then deviations from it -- was two chains of
    // dependent load bursts): deviations are taken from the first chunk's mean instead, which every lane loads along
    // with its burst; sum n_k d_k and sum (M2_k + n_k d_k^2) then give mean and M2 without cancellation (fp64, and the
    // chunk means of one channel lie within a few batch standard deviations of each other).
    const double shift = (double)*(partial + colc);
    // requested up front: behind the reductions' barriers these would be one more memory round trip
    const float mm0 = (!M2OUT && moving_mean) ? moving_mean[colc] : 0.0f;
    const float mv0 = (!M2OUT && moving_mean) ? moving_var[colc] : 0.0f;
    double s = 0.0, m2 = 0.0;
    for (int k0 = 0; k0 < n_chunks; k0 += 256) {
        float m[16], q[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int k = min(k0 + lane + 16 * j, n_chunks - 1);
            m[j] = *(partial + (int64_t)k * 2 * c + colc);
            q[j] = *(partial + (int64_t)k * 2 * c + c + colc);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int k = k0 + lane + 16 * j;
            const double n_k = k < n_chunks - 1 ? (double)chunk_rows : (k == n_chunks - 1 ? (double)last_rows : 0.0);
            const double d = (double)m[j] - shift;
            s += n_k * d;
            m2 += k < n_chunks ? (double)q[j] + n_k * d * d : 0.0;
        }
    }
    const double s_a = lanes16_sum(s, sh);
    const double mean_a = shift + s_a / n_total;
    double m2_a = lanes16_sum(m2, sh) - s_a * s_a / n_total;
    if (m2_a < 0.0) m2_a = 0.0;
    if (M2OUT) {
        if (ok && lane == 0) {
            mean[col] = (float)mean_a;
            rstd[col] = (float)m2_a;
        }
        return;
    }
    if (ok && lane == 0) {
        const double var = m2_a / n_total;  // biased: what the fused batch norm normalises with
        mean[col] = (float)mean_a;
        rstd[col] = (float)(1.0 / sqrt(var + (double)eps));
        if (moving_mean) {
            const double unbiased = n_total > 1.0 ? m2_a / (n_total - 1.0) : var;  // Bessel-corrected
            moving_mean[col] = (float)((double)mm0 * decay + mean_a * (1.0 - (double)decay));
            moving_var[col] = (float)((double)mv0 * decay + unbiased * (1.0 - (double)decay));
        }
    }
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ partial, int n_chunks,
                                                           int chunk_rows, int64_t rows, int c, float eps,
                                                           float* __restrict__ mean, float* __restrict__ rstd,
                                                           float* __restrict__ moving_mean,
                                                           float* __restrict__ moving_var, float decay) {
    __shared__ double sh[64];
    bn_finalize_body<>(blockIdx.x, partial, n_chunks, chunk_rows, rows, c, eps, mean, rstd, moving_mean,
                            moving_var, decay, sh);
}

// Synchronised batch norm (data parallel, optional: SURVEY 8e): a rank merges its chunk partials into ONE
// (mean, M2, rows) record -- out[0..c) mean, out[c..2c) M2, out[2c] = rows -- the records of all ranks are
// all-gathered and bn_finalize_ranks merges them in rank order, so every rank normalises with the statistics of the
// GLOBAL batch and holds bit-identical mean / rstd / moving averages.
__global__ __launch_bounds__(256) void bn_merge_partials_kernel(const float* __restrict__ partial, int n_chunks,
                                                                 int chunk_rows, int64_t rows, int c,
                                                                 float* __restrict__ out) {
    __shared__ double sh[64];
    bn_finalize_body<true>(blockIdx.x, partial, n_chunks, chunk_rows, rows, c, 0.0f, out, out + c, nullptr,
                                  nullptr, 0.0f, sh);
    if (blockIdx.x == 0 && threadIdx.x == 0) out[2 * c] = (float)rows;
}

__global__ void bn_finalize_ranks_kernel(const float* __restrict__ gathered, int world, int c, float eps,
                                         float* __restrict__ mean, float* __restrict__ rstd,
                                         float* __restrict__ moving_mean, float* __restrict__ moving_var, float decay) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= c) return;
    const int64_t rec = 2 * (int64_t)c + 1;
    double n_total = 0.0, s = 0.0;
    for (int k = 0; k < world; ++k) {
        const double n_k = (double)gathered[k * rec + 2 * c];
        n_total += n_k;
        s += n_k * (double)gathered[k * rec + col];
    }
    const double mean_a = s / n_total;
    double m2 = 0.0;
    for (int k = 0; k < world; ++k) {
        const double n_k = (double)gathered[k * rec + 2 * c];
        const double d = (double)gathered[k * rec + col] - mean_a;
        m2 += (double)gathered[k * rec + c + col] + n_k * d * d;
    }
    const double var = m2 / n_total;
    mean[col] = (float)mean_a;
    rstd[col] = (float)(1.0 / sqrt(var + (double)eps));
    if (moving_mean) {
        const double unbiased = n_total > 1.0 ? m2 / (n_total - 1.0) : var;
        moving_mean[col] = (float)((double)moving_mean[col] * decay + mean_a * (1.0 - (double)decay));
        moving_var[col] = (float)((double)moving_var[col] * decay + unbiased * (1.0 - (double)decay));
    }
}

__global__ __launch_bounds__(256) void col_stats_partial_kernel(const float* __restrict__ x, int64_t ld, int64_t rows,
                                                                 int c, int chunk_rows, float* __restrict__ partial) {
    __shared__ float sh[2][STAT_TY][STAT_TX];
    const int tx = threadIdx.x & (STAT_TX - 1), ty = threadIdx.x / STAT_TX;
    const int col = blockIdx.y * STAT_TX + tx;
    const int64_t r0 = (int64_t)blockIdx.x * chunk_rows;
    const int64_t r1 = min(rows, r0 + (int64_t)chunk_rows);
    float s = 0.0f, ss = 0.0f, shift = 0.0f;
    if (col < c) {
        shift = x[r0 * ld + col];
#pragma unroll 4
        for (int64_t r = r0 + ty; r < r1; r += STAT_TY) {
            const float d = x[r * ld + col] - shift;
            s += d;
            ss += d * d;
        }
    }
    sh[0][ty][tx] = s;
    sh[1][ty][tx] = ss;
    __syncthreads();
    if (ty == 0 && col < c) {
        float ts = 0.0f, tss = 0.0f;
#pragma unroll
        for (int k = 0; k < STAT_TY; ++k) {
            ts += sh[0][k][tx];
            tss += sh[1][k][tx];
        }
        const float cnt = (float)(r1 - r0);
        const float mean_d = ts / cnt;
        float m2 = tss - ts * mean_d;
        if (m2 < 0.0f) m2 = 0.0f;
        float* po = partial + (int64_t)blockIdx.x * 2 * c;
        po[col] = shift + mean_d;
        po[c + col] = m2;
    }
}

__global__ __launch_bounds__(256) void col_stats_partial_v4_kernel(const float* __restrict__ x, int64_t ld,
                                                                    int64_t rows, int c, int chunk_rows,
                                                                    float* __restrict__ partial) {
    __shared__ float sh[2][STAT_V4_TY][STAT_TX];
    const int tq = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int col = blockIdx.y * STAT_TX + tq * 4;
    const int64_t r0 = (int64_t)blockIdx.x * chunk_rows;
    const int64_t r1 = min(rows, r0 + (int64_t)chunk_rows);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), ss = s, shift = s;
    if (col < c) {  // c % 4 == 0: a quad is entirely inside or outside
        shift = *reinterpret_cast<const float4*>(x + r0 * ld + col);
#pragma unroll 4
        for (int64_t r = r0 + ty; r < r1; r += STAT_V4_TY) {
            const float4 v = *reinterpret_cast<const float4*>(x + r * ld + col);
            const float dx = v.x - shift.x, dy = v.y - shift.y, dz = v.z - shift.z, dw = v.w - shift.w;
            s.x += dx; s.y += dy; s.z += dz; s.w += dw;
            ss.x += dx * dx; ss.y += dy * dy; ss.z += dz * dz; ss.w += dw * dw;
        }
    }
    *reinterpret_cast<float4*>(&sh[0][ty][tq * 4]) = s;
    *reinterpret_cast<float4*>(&sh[1][ty][tq * 4]) = ss;
    __syncthreads();
    if (threadIdx.x < STAT_TX) {
        const int tx = threadIdx.x, cc = blockIdx.y * STAT_TX + tx;
        if (cc < c) {
            float ts = 0.0f, tss = 0.0f;
#pragma unroll
            for (int k = 0; k < STAT_V4_TY; ++k) {
                ts += sh[0][k][tx];
                tss += sh[1][k][tx];
            }
            const float cnt = (float)(r1 - r0);
            const float mean_d = ts / cnt;
            float m2 = tss - ts * mean_d;
            if (m2 < 0.0f) m2 = 0.0f;
            float* po = partial + (int64_t)blockIdx.x * 2 * c;
            po[cc] = x[r0 * ld + cc] + mean_d;
            po[c + cc] = m2;
        }
    }
}

__global__ void rstd_from_var_kernel(const float* __restrict__ var, int c, float eps, float* __restrict__ rstd) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < c) rstd[i] = (float)(1.0 / sqrt((double)var[i] + (double)eps));
}

// ------------------------------------------------------------------------------------- fused post-op
template <int VEC>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(
    const float* __restrict__ y, int64_t ldy, int64_t rows, int c, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ beta, int act, float alpha,
    const float* __restrict__ mask, int64_t ldm, const float* __restrict__ res1, int64_t ld1,
    const int32_t* __restrict__ idx1, const float* __restrict__ res2, int64_t ld2, const int32_t* __restrict__ idx2,
    float* __restrict__ z, int64_t ldz, int tx_log2) {
    ew_loop<VEC>(rows, c, tx_log2, [&](int64_t row, int col) {
        Vec<VEC> v;
        v.load(y + row * ldy + col);
        if (mean) {
            Vec<VEC> m, r, b;
            m.load_param(mean + col); r.load_param(rstd + col); b.load_param(beta + col);
#pragma unroll
            for (int k = 0; k < VEC; ++k) v.v[k] = hypel_bn_pre(hypel_bn_xhat(v.v[k], m.v[k], r.v[k]), b.v[k]);
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) v.v[k] = hypel_act(v.v[k], act, alpha);
        if (mask) {
            Vec<VEC> mk;
            mk.load(mask + row * ldm + col);
#pragma unroll
            for (int k = 0; k < VEC; ++k) v.v[k] *= mk.v[k];
        }
        if (res1) {
            if (idx1) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) v.v[k] += res1[row * ld1 + idx1[col + k]];
            } else {
                Vec<VEC> t;
                t.load(res1 + row * ld1 + col);
#pragma unroll
                for (int k = 0; k < VEC; ++k) v.v[k] += t.v[k];
            }
        }
        if (res2) {
            if (idx2) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) v.v[k] += res2[row * ld2 + idx2[col + k]];
            } else {
                Vec<VEC> t;
                t.load(res2 + row * ld2 + col);
#pragma unroll
                for (int k = 0; k < VEC; ++k) v.v[k] += t.v[k];
            }
        }
        v.store(z + row * ldz + col);
    });
}

__device__ __forceinline__ void bwd_elem(const float* dz, int64_t lddz, const float* __restrict__ y,
                                         int64_t ldy, int64_t row, int col, const float* __restrict__ mean,
                                         const float* __restrict__ rstd, const float* __restrict__ beta, int act,
                                         float alpha, const float* __restrict__ mask, int64_t ldm, float& dyh,
                                         float& xhat) {
    float v = y[row * ldy + col];
    float pre = v;
    if (mean) {
        xhat = hypel_bn_xhat(v, mean[col], rstd[col]);
        pre = hypel_bn_pre(xhat, beta[col]);
    } else {
        xhat = v;
    }
    float g = dz[row * lddz + col];
    if (mask) g *= mask[row * ldm + col];
    dyh = g * hypel_act_grad(pre, act, alpha);
}


// float4 x 16 row lanes (see col_stats_partial_v4_kernel)

__device__ __forceinline__ void bwd_finalize_body(int blk16, const float* __restrict__ partial, int n_chunks, int c,
                                                  float* __restrict__ sums, float* __restrict__ dparam,
                                                  int accumulate, double* sh) {
    const int ch = threadIdx.x & 15, lane = threadIdx.x >> 4;
    const int col = blk16 * 16 + ch;
    const bool ok = col < c;
    const int colc = ok ? col : c - 1;
    const float dp0 = (dparam && accumulate) ? dparam[colc] : 0.0f;  // requested before the reductions' barriers
    double a = 0.0, b = 0.0;
    for (int k0 = 0; k0 < n_chunks; k0 += 256) {  // bursts of 16 independent loads per lane (see bn_finalize_body)
        float pa[16], pb[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int k = min(k0 + lane + 16 * j, n_chunks - 1);
            pa[j] = *(partial + (int64_t)k * 2 * c + colc);
            pb[j] = *(partial + (int64_t)k * 2 * c + c + colc);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const bool in = k0 + lane + 16 * j < n_chunks;
            a += in ? (double)pa[j] : 0.0;
            b += in ? (double)pb[j] : 0.0;
        }
    }
    a = lanes16_sum(a, sh);
    b = lanes16_sum(b, sh);
    if (ok && lane == 0) {
        sums[col] = (float)a;
        sums[c + col] = (float)b;
        if (dparam) dparam[col] = dp0 + (float)a;
    }
}

__global__ __launch_bounds__(256) void bwd_reduce_finalize_kernel(const float* __restrict__ partial, int n_chunks,
                                                                   int c, float* __restrict__ sums,
                                                                   float* __restrict__ dparam, int accumulate) {
    __shared__ double sh[64];
    bwd_finalize_body(blockIdx.x, partial, n_chunks, c, sums, dparam, accumulate, sh);
}

// DY (layers WITHOUT batch norm only): the pass also writes dY = dZ * act'(y) (* mask) -- there the input gradient does
// not depend on the column sums, so the reduction pass of the bias gradient delivers dY as well and the separate
// hypel_bn_act_bwd_apply launch (one more read of dZ and Y) goes away (hypel_act_bias_bwd_reduce).
// dy may alias dz (in-place post-op backward, include/hypel.h): neither pointer is __restrict__
template <bool DY>
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_kernel(
    const float* dz, int64_t lddz, const float* __restrict__ y, int64_t ldy, int64_t rows, int c,
    const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ beta, int act,
    float alpha, const float* __restrict__ mask, int64_t ldm, int chunk_rows, float* __restrict__ partial,
    float* dy, int64_t lddy) {
    __shared__ float sh[2][STAT_TY][STAT_TX];
    const int tx = threadIdx.x & (STAT_TX - 1), ty = threadIdx.x / STAT_TX;
    const int col = blockIdx.y * STAT_TX + tx;
    const int64_t r0 = (int64_t)blockIdx.x * chunk_rows;
    const int64_t r1 = min(rows, r0 + (int64_t)chunk_rows);
    float s0 = 0.0f, s1 = 0.0f;
    if (col < c) {
#pragma unroll 4
        for (int64_t r = r0 + ty; r < r1; r += STAT_TY) {
            float dyh, xhat;
            bwd_elem(dz, lddz, y, ldy, r, col, mean, rstd, beta, act, alpha, mask, ldm, dyh, xhat);
            if constexpr (DY) dy[r * lddy + col] = dyh;
            s0 += dyh;
            s1 += dyh * xhat;
        }
    }
    sh[0][ty][tx] = s0;
    sh[1][ty][tx] = s1;
    __syncthreads();
    if (ty == 0 && col < c) {
        float t0 = 0.0f, t1 = 0.0f;
#pragma unroll
        for (int k = 0; k < STAT_TY; ++k) {
            t0 += sh[0][k][tx];
            t1 += sh[1][k][tx];
        }
        float* po = partial + (int64_t)blockIdx.x * 2 * c;
        po[col] = t0;
        po[c + col] = t1;
    }
}

template <bool DY>
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_v4_kernel(
    const float* dz, int64_t lddz, const float* __restrict__ y, int64_t ldy, int64_t rows, int c,
    const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ beta, int act,
    float alpha, const float* __restrict__ mask, int64_t ldm, int chunk_rows, float* __restrict__ partial,
    float* dy, int64_t lddy) {
    __shared__ float sh[2][STAT_V4_TY][STAT_TX];
    const int tq = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int col = blockIdx.y * STAT_TX + tq * 4;
    const int64_t r0 = (int64_t)blockIdx.x * chunk_rows;
    const int64_t r1 = min(rows, r0 + (int64_t)chunk_rows);
    float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
    if (col < c) {
        float mu[4] = {0.f, 0.f, 0.f, 0.f}, rs[4] = {1.f, 1.f, 1.f, 1.f}, be[4] = {0.f, 0.f, 0.f, 0.f};
        if (mean) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                mu[q] = mean[col + q];
                rs[q] = rstd[col + q];
                be[q] = beta[col + q];
            }
        }
#pragma unroll 4
        for (int64_t r = r0 + ty; r < r1; r += STAT_V4_TY) {
            const float4 yv4 = *reinterpret_cast<const float4*>(y + r * ldy + col);
            const float4 gv4 = *reinterpret_cast<const float4*>(dz + r * lddz + col);
            float4 mv4 = make_float4(1.f, 1.f, 1.f, 1.f);
            if (mask) mv4 = *reinterpret_cast<const float4*>(mask + r * ldm + col);
            const float yv[4] = {yv4.x, yv4.y, yv4.z, yv4.w}, gv[4] = {gv4.x, gv4.y, gv4.z, gv4.w};
            const float mv[4] = {mv4.x, mv4.y, mv4.z, mv4.w};
            [[maybe_unused]] float dv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float xhat = yv[q], pre = yv[q];
                if (mean) {
                    xhat = hypel_bn_xhat(yv[q], mu[q], rs[q]);
                    pre = hypel_bn_pre(xhat, be[q]);
                }
                float g = gv[q];
                if (mask) g *= mv[q];
                const float dyh = g * hypel_act_grad(pre, act, alpha);
                if constexpr (DY) dv[q] = dyh;
                s0[q] += dyh;
                s1[q] += dyh * xhat;
            }
            if constexpr (DY) *reinterpret_cast<float4*>(dy + r * lddy + col) = make_float4(dv[0], dv[1], dv[2], dv[3]);
        }
    }
    *reinterpret_cast<float4*>(&sh[0][ty][tq * 4]) = make_float4(s0[0], s0[1], s0[2], s0[3]);
    *reinterpret_cast<float4*>(&sh[1][ty][tq * 4]) = make_float4(s1[0], s1[1], s1[2], s1[3]);
    __syncthreads();
    if (threadIdx.x < STAT_TX) {
        const int tx = threadIdx.x, cc = blockIdx.y * STAT_TX + tx;
        if (cc < c) {
            float t0 = 0.0f, t1 = 0.0f;
#pragma unroll
            for (int k = 0; k < STAT_V4_TY; ++k) {
                t0 += sh[0][k][tx];
                t1 += sh[1][k][tx];
            }
            float* po = partial + (int64_t)blockIdx.x * 2 * c;
            po[cc] = t0;
            po[c + cc] = t1;
        }
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(
    const float* dz, int64_t lddz, const float* __restrict__ y, int64_t ldy, int64_t rows, int c,
    const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ beta, int act,
    float alpha, const float* __restrict__ mask, int64_t ldm, const float* __restrict__ sums, float* dy,
    int64_t lddy, int tx_log2, int64_t stat_rows) {
    const float inv_m = 1.0f / (float)stat_rows;  // rows the sums run over: `rows`, or the global batch (synchronised BN)
    ew_loop<VEC>(rows, c, tx_log2, [&](int64_t row, int col) {
        Vec<VEC> yv, g;
        yv.load(y + row * ldy + col);
        g.load(dz + row * lddz + col);
        if (mask) {
            Vec<VEC> mk;
            mk.load(mask + row * ldm + col);
#pragma unroll
            for (int k = 0; k < VEC; ++k) g.v[k] *= mk.v[k];
        }
        if (mean) {
            Vec<VEC> m, r, b, s0, s1;
            m.load_param(mean + col); r.load_param(rstd + col); b.load_param(beta + col);
            s0.load_param(sums + col); s1.load_param(sums + c + col);
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const float xhat = hypel_bn_xhat(yv.v[k], m.v[k], r.v[k]);
                const float dyh = g.v[k] * hypel_act_grad(hypel_bn_pre(xhat, b.v[k]), act, alpha);
                g.v[k] = r.v[k] * (dyh - s0.v[k] * inv_m - xhat * (s1.v[k] * inv_m));
            }
        } else {
#pragma unroll
            for (int k = 0; k < VEC; ++k) g.v[k] *= hypel_act_grad(yv.v[k], act, alpha);
        }
        g.store(dy + row * lddy + col);
    });
}

template <int VEC>
__global__ __launch_bounds__(256) void chanmap_bwd_kernel(const float* __restrict__ dz, int64_t lddz, int64_t rows,
                                                           int c, float* __restrict__ dr, int64_t lddr, int cin,
                                                           const int32_t* __restrict__ start, int accumulate,
                                                           int tx_log2) {
    ew_loop<VEC>(rows, cin, tx_log2, [&](int64_t row, int col) {
        Vec<VEC> s;
        if (start) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const int a = start[col + k];
                const int b = start[col + k + 1];
                float t = 0.0f;
                for (int q = a; q < b; ++q) t += dz[row * lddz + q];
                s.v[k] = t;
            }
        } else {
            s.load(dz + row * lddz + col);
        }
        float* p = dr + row * lddr + col;
        if (accumulate) {
            Vec<VEC> o;
            o.load(p);
#pragma unroll
            for (int k = 0; k < VEC; ++k) s.v[k] += o.v[k];
        }
        s.store(p);
    });
}

// ------------------------------------------------------------------------------------- losses
__global__ void softmax_xent_kernel(const float* __restrict__ logits, int64_t ld, int64_t n, int c,
                                    const float* __restrict__ labels, int64_t ldl, float* __restrict__ loss,
                                    float* __restrict__ dlogits, int64_t lddl, float gscale) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* z = logits + i * ld;
    const float* lab = labels + i * ldl;
    float zmax = z[0];
    for (int j = 1; j < c; ++j) zmax = fmaxf(zmax, z[j]);
    float se = 0.0f;
    for (int j = 0; j < c; ++j) se += expf(z[j] - zmax);
    const float lse = logf(se);
    float l = 0.0f, lsum = 0.0f;
    for (int j = 0; j < c; ++j) {
        l -= lab[j] * (z[j] - zmax - lse);
        lsum += lab[j];
    }
    if (loss) loss[i] = l;
    if (dlogits) {
        float* d = dlogits + i * lddl;
        const float inv = 1.0f / se;
        for (int j = 0; j < c; ++j) d[j] = gscale * (expf(z[j] - zmax) * inv * lsum - lab[j]);
    }
}

constexpr int RED_BLOCKS = 1024;

__device__ __forceinline__ float block_sum_256(float v, float* sh) {
    // wave64 shuffle reduce, then 4 waves through LDS
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = v;
    __syncthreads();
    float t = 0.0f;
    if (threadIdx.x == 0) t = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(256) void mse_partial_kernel(const float* __restrict__ a, int64_t lda,
                                                           const float* __restrict__ b, int64_t ldb, int64_t rows,
                                                           int c, float* __restrict__ da, int64_t ldda, float gcoef,
                                                           float* __restrict__ ws) {
    __shared__ float sh[4];
    const int64_t total = rows * c;
    float s = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / c;
        const int col = (int)(i - row * c);
        const float d = a[row * lda + col] - b[row * ldb + col];
        s += d * d;
        if (da) da[row * ldda + col] = gcoef * d;
    }
    const float t = block_sum_256(s, sh);
    if (threadIdx.x == 0) ws[blockIdx.x] = t;
}

// contiguous operands (lda == ldb == ldda == c, 16-byte aligned, count % 4 == 0): three flat float4 streams instead of
// a 64-bit division per element
__global__ __launch_bounds__(256) void mse_partial_flat4_kernel(const float4* __restrict__ a, const float4* __restrict__ b,
                                                                 int64_t count4, float4* __restrict__ da, float gcoef,
                                                                 float* __restrict__ ws) {
    __shared__ float sh[4];
    float s = 0.0f;
#pragma unroll 2
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 x = a[i], y = b[i];
        const float4 d = make_float4(x.x - y.x, x.y - y.y, x.z - y.z, x.w - y.w);
        s += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
        if (da) da[i] = make_float4(gcoef * d.x, gcoef * d.y, gcoef * d.z, gcoef * d.w);
    }
    const float t = block_sum_256(s, sh);
    if (threadIdx.x == 0) ws[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void sum_partial_kernel(const float* __restrict__ x, int64_t count,
                                                           float* __restrict__ ws) {
    __shared__ float sh[4];
    float s = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
        s += x[i];
    const float t = block_sum_256(s, sh);
    if (threadIdx.x == 0) ws[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void sum_finalize_kernel(const float* __restrict__ ws, int n, double scale,
                                                            float* __restrict__ out) {
    __shared__ double sh[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)ws[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(sh[0] * scale);
}

// The tail of the classifier loss in ONE launch (it was five: two-stage sum of the per-row cross entropies, second stage
// of the reconstruction MSE, the non-finite guard, the dropout step counter): mean of loss_rows[n_rows] -> out_ce,
// sum of mse_ws[n_mse] x mse_scale -> out_mse (nullable), flag = either non-finite (nullable), step += 1 (nullable).
// fp64, fixed association (thread-strided, then a tree over the 256 threads).
__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* __restrict__ loss_rows, int n_rows,
                                                             const float* __restrict__ mse_ws, int n_mse,
                                                             double mse_scale, float* __restrict__ out_ce,
                                                             float* __restrict__ out_mse, float* __restrict__ flag,
                                                             uint64_t* __restrict__ step) {
    __shared__ double sh[2][256];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < n_rows; i += 256) a += (double)loss_rows[i];
    if (mse_ws)
        for (int i = threadIdx.x; i < n_mse; i += 256) b += (double)mse_ws[i];
    sh[0][threadIdx.x] = a;
    sh[1][threadIdx.x] = b;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float ce = (float)(sh[0][0] / (double)n_rows);
        const float mse = (float)(sh[1][0] * mse_scale);
        out_ce[0] = ce;
        if (out_mse) out_mse[0] = mse;
        if (flag) flag[0] = (isfinite(ce) && (!mse_ws || isfinite(mse))) ? 0.0f : 1.0f;
        if (step) step[0] += 1;
    }
}

// ------------------------------------------------------------------------------------- optimisers
// `skip` (nullable): a device flag written by hypel_loss_guard_f32 (non-zero = the step's loss was not finite).  The
// optimiser then leaves parameters and slots untouched -- create_train_op's check_numerics refuses the update the
// same way (common_nn_ops.py:232) -- without the host having to look at the loss before launching it.
__global__ void adam_tf1_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                float* __restrict__ v, int64_t count, float lr_t, float b1, float b2, float eps,
                                const float* __restrict__ skip) {
    if (skip && *skip != 0.0f) return;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}

__global__ void momentum_tf1_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ a,
                                    int64_t count, float lr, float mu, const float* __restrict__ skip) {
    if (skip && *skip != 0.0f) return;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const float ai = mu * a[i] + g[i];
        a[i] = ai;
        p[i] -= lr * ai;
    }
}

// ------------------------------------------------------------------------------------- dropout mask
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// flag[0] = 1 if any of the (up to two) scalar losses is NaN/Inf, else 0.  The flag lives in the element behind the
// flat gradient buffer, so the data-parallel gradient all-reduce (sum) hands every rank the same verdict.
__global__ void loss_guard_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ flag) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const bool ok = isfinite(a[0]) && (b == nullptr || isfinite(b[0]));
        flag[0] = ok ? 0.0f : 1.0f;
    }
}

__global__ void step_inc_kernel(uint64_t* step) {
    if (threadIdx.x == 0 && blockIdx.x == 0) step[0] += 1;
}

__global__ void dropout_mask_kernel(float* __restrict__ mask, int64_t count, float keep, uint64_t seed,
                                    const uint64_t* __restrict__ step_dev) {
    const float scale = 1.0f / keep;
    const uint64_t step = step_dev[0];
    const int64_t groups = (count + 3) / 4;
    for (int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gi < groups;
         gi += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t ctr = (uint64_t)gi;
        uint32_t rnd[4];
        philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)step, (uint32_t)(step >> 32), (uint32_t)seed,
                      (uint32_t)(seed >> 32), rnd);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t i = gi * 4 + k;
            if (i < count) {
                const float u = (float)(rnd[k] >> 8) * (1.0f / 16777216.0f);  // [0,1)
                mask[i] = u < keep ? scale : 0.0f;
            }
        }
    }
}

// ------------------------------------------------------------------------------------- small-rows BN
// Fully-connected tail of the models: activations are [N x C] with N = the batch (1024 rows), so every channel's
// statistics fit in ONE block.  A block owns a 32-channel stripe for ALL rows: statistics, finaliser and the
// normalise/activate pass (forward), or the two reductions and the gradient (backward), in a single launch instead
// of three -- no cross-block dependency, hence none of the cross-XCD coherence cost of the ticket scheme.
constexpr int SMALL_TX = 32;    // channels per block
constexpr int SMALL_TY = 32;    // row lanes (block = 1024 threads)
constexpr int SMALL_R = 32;     // rows per thread, kept in registers between the reduction and the apply pass
constexpr int SMALL_MAX_ROWS = SMALL_TY * SMALL_R;

// Both column sums of a kernel in ONE exchange: a wave holds two row lanes of its 32 columns, which one cross-lane add
// combines (fixed order); the 16 wave partials go through LDS as float2 and every thread adds them up in wave order
// -- one barrier and 16 ds_read_b64 per thread instead of four barriers and 64 ds_read_b32 (the two separate
// 32-deep exchanges were 3.7 of the 7 us such a kernel spends, phase-stamped copy in tools/exp/probe).
__device__ __forceinline__ void small_lane_sum2(float& a, float& b, float2 (*sh)[SMALL_TX]) {
    const int tx = threadIdx.x & (SMALL_TX - 1), wave = threadIdx.x >> 6;
    a += __shfl_xor(a, 32, 64);
    b += __shfl_xor(b, 32, 64);
    if ((threadIdx.x & 63) < SMALL_TX) sh[wave][tx] = make_float2(a, b);
    __syncthreads();
    float ta = 0.0f, tb = 0.0f;
#pragma unroll
    for (int k = 0; k < SMALL_TY / 2; ++k) {
        const float2 t = sh[k][tx];
        ta += t.x;
        tb += t.y;
    }
    a = ta;
    b = tb;
}

// All loads of a thread's rows are issued back to back (one memory round trip instead of rows/TY dependent ones:
// a single-block-per-stripe kernel is otherwise pure latency) and the values stay in registers for the second phase.
// Raw buffer accesses: the array base sits in a scalar descriptor, a thread holds ONE 32-bit offset per array and the
// row step is a scalar offset, so the 64-96 loads in flight need no address registers (with 64-bit pointers the kernels
// spilled 28 / 67 VGPRs to scratch at the 128-register budget of a 1024-thread block).  A column or row outside the
// tensor is addressed beyond the descriptor's range: the load returns 0 and the store is dropped.
constexpr uint32_t SMALL_OOB = 0x7fffffffu;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t small_rsrc(const float* p) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7ffffff0, 0x00020000);
}

// Row validity without compare masks (32 of them cost the ragged variant 64 scalar registers and 52-76 spills): a row
// past the matrix turns the offset into 0xffffffff arithmetically, which the descriptor rejects like SMALL_OOB; the
// row step is one running scalar offset (see seg_gemm.hip stage()).
__device__ __forceinline__ uint32_t small_row_off(uint32_t v0, int last_row_minus_ty, int i) {
    return v0 | (uint32_t)((last_row_minus_ty - i * SMALL_TY) >> 31);
}

template <bool FULL>
__device__ __forceinline__ void small_load_rows(const float* p, int ld, int rows, int col, bool ok, float (&v)[SMALL_R]) {
    const int ty = threadIdx.x / SMALL_TX;
    const __amdgpu_buffer_rsrc_t rs = small_rsrc(p);
    const uint32_t v0 = ok ? (uint32_t)(ty * ld + col) * 4u : SMALL_OOB;
    const int last = rows - 1 - ty;
    const int step = SMALL_TY * ld * 4;
    int so = 0;
#pragma unroll
    for (int i = 0; i < SMALL_R; ++i) {
        const uint32_t vo = FULL ? v0 : small_row_off(v0, last, i);
        v[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo, so, 0));
        asm volatile("s_add_u32 %0, %0, %1" : "+s"(so) : "s"(step) : "scc");
    }
}

template <bool FULL>
__device__ __forceinline__ void small_store_rows(float* p, int ld, int rows, int col, bool ok, const float (&v)[SMALL_R]) {
    const int ty = threadIdx.x / SMALL_TX;
    const __amdgpu_buffer_rsrc_t rs = small_rsrc(p);
    const uint32_t v0 = ok ? (uint32_t)(ty * ld + col) * 4u : SMALL_OOB;
    const int last = rows - 1 - ty;
    const int step = SMALL_TY * ld * 4;
    int so = 0;
#pragma unroll
    for (int i = 0; i < SMALL_R; ++i) {
        const uint32_t vo = FULL ? v0 : small_row_off(v0, last, i);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[i]), rs, vo, so, 0);
        asm volatile("s_add_u32 %0, %0, %1" : "+s"(so) : "s"(step) : "scc");
    }
}

template <bool FULL>  // FULL: rows == SMALL_MAX_ROWS, no row guards at all
__global__ __launch_bounds__(1024) void bn_act_small_fwd_kernel(
    const float* __restrict__ y, int ldy, int rows, int c, float eps, const float* __restrict__ beta, int act,
    float alpha, const float* __restrict__ mask, int ldm, float* __restrict__ mean_out,
    float* __restrict__ rstd_out, float* __restrict__ moving_mean, float* __restrict__ moving_var, float decay,
    float* __restrict__ z, int ldz) {
    __shared__ float2 sh[SMALL_TY / 2][SMALL_TX];
    const int tx = threadIdx.x & (SMALL_TX - 1), ty = threadIdx.x / SMALL_TX;
    const int col = blockIdx.x * SMALL_TX + tx;
    const bool ok = col < c;
    const int colc = ok ? col : c - 1;
    float v[SMALL_R], mk[SMALL_R];
    small_load_rows<FULL>(y, ldy, rows, col, ok, v);
    if (mask) small_load_rows<FULL>(mask, ldm, rows, col, ok, mk);
    const float shift = y[colc];  // first row: keeps the fp32 sums well conditioned
    // everything the tail needs is requested now: a load behind the block barriers below would be one more memory
    // round trip in a kernel that is nothing but latency
    const float be = beta[colc];
    const float mm0 = moving_mean ? moving_mean[colc] : 0.0f, mv0 = moving_mean ? moving_var[colc] : 0.0f;
    float s = 0.0f, ss = 0.0f;
#pragma unroll
    for (int i = 0; i < SMALL_R; ++i) {
        const float d = (FULL || ty + i * SMALL_TY < rows) ? v[i] - shift : 0.0f;
        s += d;
        ss += d * d;
    }
    small_lane_sum2(s, ss, sh);
    const double ts = (double)s, tss = (double)ss;
    const double n = (double)rows;
    const double mean = (double)shift + ts / n;
    double m2 = tss - ts * ts / n;
    if (m2 < 0.0) m2 = 0.0;
    const double var = m2 / n;
    const float mu = (float)mean, rs = (float)(1.0 / sqrt(var + (double)eps));
    if (ok && ty == 0) {
        mean_out[col] = mu;
        rstd_out[col] = rs;
        if (moving_mean) {
            const double unbiased = n > 1.0 ? m2 / (n - 1.0) : var;
            moving_mean[col] = (float)((double)mm0 * decay + mean * (1.0 - (double)decay));
            moving_var[col] = (float)((double)mv0 * decay + unbiased * (1.0 - (double)decay));
        }
    }
    if (act == HYPEL_ACT_LRELU) {  // the common case without the per-element activation switch
#pragma unroll
        for (int i = 0; i < SMALL_R; ++i) {
            const float p = hypel_bn_pre(hypel_bn_xhat(v[i], mu, rs), be);
            v[i] = p > 0.0f ? p : p * alpha;
        }
    } else {
#pragma unroll
        for (int i = 0; i < SMALL_R; ++i) v[i] = hypel_act(hypel_bn_pre(hypel_bn_xhat(v[i], mu, rs), be), act, alpha);
    }
    if (mask) {
#pragma unroll
        for (int i = 0; i < SMALL_R; ++i) v[i] *= mk[i];
    }
    small_store_rows<FULL>(z, ldz, rows, col, ok, v);
}

template <bool FULL>
__global__ __launch_bounds__(1024) void bn_act_small_bwd_kernel(
    const float* dz, int lddz, const float* __restrict__ y, int ldy, int rows, int c,
    const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ beta, int act,
    float alpha, const float* __restrict__ mask, int ldm, float* dy, int lddy,
    float* __restrict__ dparam, int accumulate) {
    __shared__ float2 sh[SMALL_TY / 2][SMALL_TX];
    const int tx = threadIdx.x & (SMALL_TX - 1), ty = threadIdx.x / SMALL_TX;
    const int col = blockIdx.x * SMALL_TX + tx;
    const bool ok = col < c;
    const int colc = ok ? col : c - 1;
    float g[SMALL_R], xh[SMALL_R];  // dyh and xhat of this thread's rows
    const float mu = mean[colc], rs = rstd[colc], be = beta[colc];
    const float dp0 = (dparam && accumulate) ? dparam[colc] : 0.0f;  // requested before the barriers (see forward)
    small_load_rows<FULL>(y, ldy, rows, col, ok, xh);
    small_load_rows<FULL>(dz, lddz, rows, col, ok, g);
    if (mask) {
        float mk[SMALL_R];
        small_load_rows<FULL>(mask, ldm, rows, col, ok, mk);
#pragma unroll
        for (int i = 0; i < SMALL_R; ++i) g[i] *= mk[i];
    }
    float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
    for (int i = 0; i < SMALL_R; ++i) {
        xh[i] = hypel_bn_xhat(xh[i], mu, rs);
        const float p = hypel_bn_pre(xh[i], be);
        const float slope = act == HYPEL_ACT_LRELU ? (p > 0.0f ? 1.0f : alpha) : hypel_act_grad(p, act, alpha);
        g[i] = (FULL || ty + i * SMALL_TY < rows) ? g[i] * slope : 0.0f;
        s0 += g[i];
        s1 += g[i] * xh[i];
    }
    small_lane_sum2(s0, s1, sh);
    const float t0 = s0, t1 = s1;
    if (ok && ty == 0 && dparam) dparam[col] = dp0 + t0;
    const float inv_m = 1.0f / (float)rows;
    const float m0 = t0 * inv_m, m1 = t1 * inv_m;
#pragma unroll
    for (int i = 0; i < SMALL_R; ++i) g[i] = rs * (g[i] - m0 - xh[i] * m1);
    small_store_rows<FULL>(dy, lddy, rows, col, ok, g);
}

// ------------------------------------------------------------------------------------- 2-D block copies
// (packed weight image of a merged multi-kernel level <-> the HWIO variables / their gradient slots)
__global__ __launch_bounds__(256) void copy_blocks_kernel(const float* __restrict__ base,
                                                           const hypel_copy_block_t* __restrict__ entries) {
    const hypel_copy_block_t e = entries[blockIdx.x];
    const float* src = base + e.src_off;
    float* dst = const_cast<float*>(base) + e.dst_off;
    const int64_t total = (int64_t)e.rows * e.cols;
    const bool acc = (e.flags & 1) != 0;
    // 16-byte path: whole rows of float4 on both sides (the [nb x bands] blocks of a batched GAN application)
    if ((e.cols & 3) == 0 && (e.src_ld & 3) == 0 && (e.dst_ld & 3) == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
        const int c4 = e.cols >> 2;
        const int64_t total4 = total >> 2;
        for (int64_t i = threadIdx.x + (int64_t)blockIdx.y * 256; i < total4; i += 256 * (int64_t)gridDim.y) {
            const int64_t r = i / c4;
            const int c = (int)(i - r * c4) * 4;
            const float4 v = *reinterpret_cast<const float4*>(src + r * e.src_ld + c);
            float4* d = reinterpret_cast<float4*>(dst + r * e.dst_ld + c);
            if (acc) {
                const float4 o = *d;
                *d = make_float4(o.x + v.x, o.y + v.y, o.z + v.z, o.w + v.w);
            } else {
                *d = v;
            }
        }
        return;
    }
    for (int64_t i = threadIdx.x + (int64_t)blockIdx.y * 256; i < total; i += 256 * (int64_t)gridDim.y) {
        const int64_t r = i / e.cols;
        const int c = (int)(i - r * e.cols);
        const float v = src[r * e.src_ld + c];
        float* d = dst + r * e.dst_ld + c;
        *d = acc ? *d + v : v;
    }
}

// ------------------------------------------------------------------------------------- metrics
__global__ void argmax_confusion_kernel(const float* __restrict__ logits, int64_t ld, int64_t n, int c,
                                        const int32_t* __restrict__ labels, int32_t* __restrict__ pred,
                                        int32_t* __restrict__ confusion) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* z = logits + i * ld;
    int best = 0;
    float bv = z[0];
    for (int j = 1; j < c; ++j)
        if (z[j] > bv) {  // strict: first maximum wins, as tf.argmax
            bv = z[j];
            best = j;
        }
    if (pred) pred[i] = best;
    if (confusion && labels) {
        const int lab = labels[i];
        if (lab >= 0 && lab < c) atomicAdd(&confusion[lab * c + best], 1);
    }
}

// ------------------------------------------------------------------------------------- LRN
__global__ void lrn_fwd_kernel(const float* __restrict__ x, int64_t ldx, int64_t rows, int c, int radius, float bias,
                               float alpha, float beta, float* __restrict__ y, int64_t ldy) {
    const int64_t total = rows * c;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / c;
        const int col = (int)(i - row * c);
        const float* xr = x + row * ldx;
        const int lo = max(0, col - radius), hi = min(c - 1, col + radius);
        float s = 0.0f;
        for (int j = lo; j <= hi; ++j) s += xr[j] * xr[j];
        y[row * ldy + col] = xr[col] * powf(bias + alpha * s, -beta);
    }
}

// dx_j = dy_j * s_j^-beta - 2 alpha beta x_j * sum_{|i-j|<=r} dy_i x_i s_i^(-beta-1)
__global__ void lrn_bwd_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ dy, int64_t lddy,
                               int64_t rows, int c, int radius, float bias, float alpha, float beta,
                               float* __restrict__ dx, int64_t lddx, int accumulate) {
    const int64_t total = rows * c;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / c;
        const int col = (int)(i - row * c);
        const float* xr = x + row * ldx;
        const float* gr = dy + row * lddy;
        const int lo = max(0, col - radius), hi = min(c - 1, col + radius);
        float win = 0.0f, s_self = 0.0f;
        for (int k = lo; k <= hi; ++k) {
            const int l2 = max(0, k - radius), h2 = min(c - 1, k + radius);
            float s = 0.0f;
            for (int j = l2; j <= h2; ++j) s += xr[j] * xr[j];
            s = bias + alpha * s;
            if (k == col) s_self = s;
            win += gr[k] * xr[k] * powf(s, -beta - 1.0f);
        }
        const float g = gr[col] * powf(s_self, -beta) - 2.0f * alpha * beta * xr[col] * win;
        float* p = dx + row * lddx + col;
        *p = accumulate ? *p + g : g;
    }
}

}  // namespace

// ======================================================================================= C-ABI
#define ST ((hipStream_t)stream)

extern "C" int hypel_nhwc_to_pnc(const float* x, float* out, int64_t n, int32_t p, int32_t c, int64_t ld,
                                 hypel_stream_t stream) {
    HYPEL_REQUIRE(x && out && n > 0 && p > 0 && c > 0 && ld >= c, "hypel_nhwc_to_pnc");
    const int64_t row_blocks = ((int64_t)p * n + 15) / 16;  // 4 waves x 4 rows per block and pass
    hipLaunchKernelGGL(nhwc_to_pnc_kernel, dim3((unsigned)(row_blocks < 65536 * 4 ? row_blocks : 65536 * 4)), dim3(256), 0,
                       ST, x, out, n, p, c, ld);
    HYPEL_CHECK_LAUNCH("hypel_nhwc_to_pnc");
    return 0;
}

extern "C" int hypel_pnc_to_nhwc(const float* in, int64_t ld, float* x, int64_t n, int32_t p, int32_t c,
                                 hypel_stream_t stream) {
    HYPEL_REQUIRE(x && in && n > 0 && p > 0 && c > 0 && ld >= c, "hypel_pnc_to_nhwc");
    hipLaunchKernelGGL(pnc_to_nhwc_kernel, dim3(hypel_grid_1d(n * p * c, 256)), dim3(256), 0, ST, in, ld, x, n, p, c);
    HYPEL_CHECK_LAUNCH("hypel_pnc_to_nhwc");
    return 0;
}

extern "C" int hypel_fill_f32(float* dst, int64_t count, float value, hypel_stream_t stream) {
    HYPEL_REQUIRE(dst && count >= 0, "hypel_fill_f32");
    if (count == 0) return 0;
    hipLaunchKernelGGL(fill_kernel, dim3(hypel_grid_1d(count, 256)), dim3(256), 0, ST, dst, count, value);
    HYPEL_CHECK_LAUNCH("hypel_fill_f32");
    return 0;
}

static int red_max_blocks() {
    constexpr int v = 8192;
    return v;
}

extern "C" int hypel_reduce_splits_f32(const float* partial, int64_t stride, int32_t n_splits, float* out,
                                       int64_t count, int32_t accumulate, const float* bias, int32_t n,
                                       int64_t ldc, hypel_stream_t stream) {
    HYPEL_REQUIRE(partial && out && n_splits >= 1 && count >= 0, "hypel_reduce_splits_f32");
    HYPEL_REQUIRE((bias == nullptr && ldc <= 0) || n > 0, "hypel_reduce_splits_f32");
    if (count == 0) return 0;
    if (n_splits >= 32 && count <= 65536)
        hipLaunchKernelGGL(reduce_splits_wave_kernel, dim3(hypel_grid_1d(count * 64, 256)), dim3(256), 0, ST, partial,
                           stride, n_splits, out, count, accumulate, bias, n, ldc);
    else if ((count % 4 == 0) &&
             (stride % 4 == 0) && aligned16(partial) && aligned16(out) &&
             ((ldc <= 0 && (!bias || n % 4 == 0)) || (ldc > 0 && ldc % 4 == 0 && n % 4 == 0)))
        hipLaunchKernelGGL(reduce_splits_v4_kernel, dim3(hypel_grid_1d(count / 4, 256, red_max_blocks())), dim3(256), 0, ST,
                           partial, stride, n_splits, out, count / 4, accumulate, bias, n, ldc);
    else
        hipLaunchKernelGGL(reduce_splits_kernel, dim3(hypel_grid_1d(count, 256, red_max_blocks())), dim3(256), 0, ST,
                           partial, stride, n_splits, out, count, accumulate, bias, n, ldc);
    HYPEL_CHECK_LAUNCH("hypel_reduce_splits_f32");
    return 0;
}

extern "C" int hypel_reduce_splits_wave_multi_f32(const float* base, const hypel_reduce_entry_t* entries,
                                                  int32_t n_entries, int64_t total_count, hypel_stream_t stream) {
    HYPEL_REQUIRE(base && entries && n_entries >= 0 && total_count >= 0, "hypel_reduce_splits_wave_multi_f32");
    if (n_entries == 0 || total_count == 0) return 0;
    const int64_t chunks = (total_count + RSW_TX - 1) / RSW_TX;
    hipLaunchKernelGGL(reduce_splits_wave_multi_kernel, dim3((unsigned)(chunks < 4096 ? chunks : 4096)), dim3(RSW_TX * RSW_TY), 0, ST,
                       base, entries, n_entries, total_count);
    HYPEL_CHECK_LAUNCH("hypel_reduce_splits_wave_multi_f32");
    return 0;
}

extern "C" int hypel_copy_blocks_f32(const float* base, const hypel_copy_block_t* entries, int32_t n_entries,
                                     int64_t max_block_elems, hypel_stream_t stream) {
    HYPEL_REQUIRE(base && entries && n_entries >= 0 && max_block_elems >= 0, "hypel_copy_blocks_f32");
    if (n_entries == 0) return 0;
    // blocks per entry: ~2048 elements (512 float4) per block, so that a few large entries still fill the device
    const int64_t per = (max_block_elems + 2047) / 2048;
    const int gy = (int)(per < 1 ? 1 : (per > 1024 ? 1024 : per));
    hipLaunchKernelGGL(copy_blocks_kernel, dim3(n_entries, gy), dim3(256), 0, ST, base, entries);
    HYPEL_CHECK_LAUNCH("hypel_copy_blocks_f32");
    return 0;
}

// two flat copies in one launch (blockIdx.y = which); float4 when both ends and the count allow
__global__ __launch_bounds__(256) void copy_pair_kernel(float* dst0, const float* src0, int64_t n0, float* dst1,
                                                        const float* src1, int64_t n1) {
    float* dst = blockIdx.y ? dst1 : dst0;
    const float* src = blockIdx.y ? src1 : src0;
    const int64_t n = blockIdx.y ? n1 : n0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (((n & 3) | (reinterpret_cast<uintptr_t>(dst) & 15) | (reinterpret_cast<uintptr_t>(src) & 15)) == 0) {
        const float4* s4 = reinterpret_cast<const float4*>(src);
        float4* d4 = reinterpret_cast<float4*>(dst);
        for (; i < (n >> 2); i += stride) d4[i] = s4[i];
    } else {
        for (; i < n; i += stride) dst[i] = src[i];
    }
}

extern "C" int hypel_copy_pair_f32(float* dst0, const float* src0, int64_t n0, float* dst1, const float* src1, int64_t n1,
                                   hypel_stream_t stream) {
    HYPEL_REQUIRE(dst0 && src0 && n0 >= 0 && (n1 == 0 || (dst1 && src1)) && n1 >= 0, "hypel_copy_pair_f32");
    const int64_t big = n0 > n1 ? n0 : n1;
    if (big == 0) return 0;
    hipLaunchKernelGGL(copy_pair_kernel, dim3(hypel_grid_1d((big + 3) / 4, 256, 2048), n1 > 0 ? 2 : 1), dim3(256), 0, ST, dst0,
                       src0, n0, dst1, src1, n1);
    HYPEL_CHECK_LAUNCH("hypel_copy_pair_f32");
    return 0;
}

extern "C" int hypel_reduce_splits_pair_f32(const float* partial0, int64_t stride0, int64_t count0, float* out0,
                                            const float* partial1, int64_t stride1, int64_t count1, float* out1,
                                            int32_t n_splits, int32_t accumulate, hypel_stream_t stream) {
    HYPEL_REQUIRE(partial0 && out0 && partial1 && out1 && n_splits >= 1 && count0 > 0 && count1 > 0,
                  "hypel_reduce_splits_pair_f32");
    if (n_splits >= 32 && count0 <= 65536 && count1 <= 65536) {  // the form hypel_reduce_splits_f32 would pick for both
        hipLaunchKernelGGL(reduce_splits_wave_pair_kernel, dim3(hypel_grid_1d((count0 + count1) * 64, 256)), dim3(256), 0, ST,
                           partial0, stride0, count0, out0, partial1, stride1, count1, out1, n_splits, accumulate);
        HYPEL_CHECK_LAUNCH("hypel_reduce_splits_pair_f32");
        return 0;
    }
    int rc = hypel_reduce_splits_f32(partial0, stride0, n_splits, out0, count0, accumulate, nullptr, 0, 0, stream);
    if (rc == 0) rc = hypel_reduce_splits_f32(partial1, stride1, n_splits, out1, count1, accumulate, nullptr, 0, 0, stream);
    return rc;
}

extern "C" int hypel_reduce_splits_multi_f32(const float* base, const hypel_reduce_entry_t* entries, int32_t n_entries,
                                             hypel_stream_t stream) {
    HYPEL_REQUIRE(base && entries && n_entries >= 0, "hypel_reduce_splits_multi_f32");
    if (n_entries == 0) return 0;
    // blocks per entry: the big entries (FC weights: 3 M elements x 2-3 slabs) want many, the many-slab entries of the
    // 1x1 convolutions few; 768 measured best on the H13 step (192: +35 us, 1536: +8, 3072: +20; HYPEL_RED_GX)
    constexpr int gx = 768;
    hipLaunchKernelGGL(reduce_splits_multi_kernel, dim3(gx, n_entries), dim3(256), 0, ST, base, entries);
    HYPEL_CHECK_LAUNCH("hypel_reduce_splits_multi_f32");
    return 0;
}

// The same table when the caller knows the largest entry (the K-slice partials of a GEMM launch: a hundred entries of one
// 128-row tile each -- with 768 blocks per entry 98 % of the grid found nothing to do: 26 us for 20 MB).
extern "C" int hypel_reduce_splits_multi_sized_f32(const float* base, const hypel_reduce_entry_t* entries, int32_t n_entries,
                                                   int64_t max_count, hypel_stream_t stream) {
    HYPEL_REQUIRE(base && entries && n_entries >= 0 && max_count >= 0, "hypel_reduce_splits_multi_sized_f32");
    if (n_entries == 0 || max_count == 0) return 0;
    const int64_t want = (max_count + 4 * 256 - 1) / (4 * 256);  // one float4 per thread
    const int gx = (int)(want < 1 ? 1 : (want > 768 ? 768 : want));
    hipLaunchKernelGGL(reduce_splits_multi_kernel, dim3(gx, n_entries), dim3(256), 0, ST, base, entries);
    HYPEL_CHECK_LAUNCH("hypel_reduce_splits_multi_sized_f32");
    return 0;
}

static int launch_col_stats(const float* x, int64_t ld, int64_t rows, int32_t c, int32_t chunk_rows, float* partial,
                            hypel_stream_t stream) {
    const int n_chunks = (int)((rows + chunk_rows - 1) / chunk_rows);
    constexpr bool v4_on = true;
    const bool v4 = v4_on && (c % 4 == 0) && (ld % 4 == 0) && (((uintptr_t)x & 15) == 0);
    if (v4)
        hipLaunchKernelGGL(col_stats_partial_v4_kernel, dim3(n_chunks, (c + STAT_TX - 1) / STAT_TX), dim3(256), 0, ST,
                           x, ld, rows, c, chunk_rows, partial);
    else
        hipLaunchKernelGGL(col_stats_partial_kernel, dim3(n_chunks, (c + STAT_TX - 1) / STAT_TX), dim3(256), 0, ST, x,
                           ld, rows, c, chunk_rows, partial);
    return 0;
}

extern "C" int hypel_col_stats_partial(const float* x, int64_t ld, int64_t rows, int32_t c, int32_t chunk_rows,
                                       float* partial, hypel_stream_t stream) {
    HYPEL_REQUIRE(x && partial && rows > 0 && c > 0 && chunk_rows > 0, "hypel_col_stats_partial");
    launch_col_stats(x, ld, rows, c, chunk_rows, partial,
                     stream);
    HYPEL_CHECK_LAUNCH("hypel_col_stats_partial");
    return 0;
}

extern "C" int hypel_bn_finalize(const float* partial, int32_t n_chunks, int32_t chunk_rows, int64_t rows, int32_t c,
                                 float eps, float* mean, float* rstd, float* moving_mean, float* moving_var,
                                 float decay, hypel_stream_t stream) {
    HYPEL_REQUIRE(partial && mean && rstd && n_chunks > 0 && c > 0, "hypel_bn_finalize");
    HYPEL_REQUIRE((moving_mean == nullptr) == (moving_var == nullptr), "hypel_bn_finalize");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((c + 15) / 16), dim3(256), 0, ST, partial, n_chunks, chunk_rows, rows,
                       c, eps, mean, rstd, moving_mean, moving_var, decay);
    HYPEL_CHECK_LAUNCH("hypel_bn_finalize");
    return 0;
}

extern "C" int hypel_bn_merge_partials(const float* partial, int32_t n_chunks, int32_t chunk_rows, int64_t rows,
                                       int32_t c, float* out, hypel_stream_t stream) {
    HYPEL_REQUIRE(partial && out && n_chunks > 0 && c > 0 && rows > 0, "hypel_bn_merge_partials");
    hipLaunchKernelGGL(bn_merge_partials_kernel, dim3((c + 15) / 16), dim3(256), 0, ST, partial, n_chunks, chunk_rows,
                       rows, c, out);
    HYPEL_CHECK_LAUNCH("hypel_bn_merge_partials");
    return 0;
}

extern "C" int hypel_bn_finalize_ranks(const float* gathered, int32_t world, int32_t c, float eps, float* mean,
                                       float* rstd, float* moving_mean, float* moving_var, float decay,
                                       hypel_stream_t stream) {
    HYPEL_REQUIRE(gathered && mean && rstd && world > 0 && c > 0, "hypel_bn_finalize_ranks");
    HYPEL_REQUIRE((moving_mean == nullptr) == (moving_var == nullptr), "hypel_bn_finalize_ranks");
    hipLaunchKernelGGL(bn_finalize_ranks_kernel, dim3((c + 63) / 64), dim3(64), 0, ST, gathered, world, c, eps, mean,
                       rstd, moving_mean, moving_var, decay);
    HYPEL_CHECK_LAUNCH("hypel_bn_finalize_ranks");
    return 0;
}

extern "C" int hypel_rstd_from_var(const float* var, int32_t c, float eps, float* rstd, hypel_stream_t stream) {
    HYPEL_REQUIRE(var && rstd && c > 0, "hypel_rstd_from_var");
    hipLaunchKernelGGL(rstd_from_var_kernel, dim3((c + 63) / 64), dim3(64), 0, ST, var, c, eps, rstd);
    HYPEL_CHECK_LAUNCH("hypel_rstd_from_var");
    return 0;
}

extern "C" int hypel_bn_act_fwd(const float* y, int64_t ldy, int64_t rows, int32_t c, const float* mean,
                                const float* rstd, const float* beta, int32_t act, float alpha, const float* mask,
                                int64_t ldm, const float* res1, int64_t ld1, const int32_t* idx1, const float* res2,
                                int64_t ld2, const int32_t* idx2, float* z, int64_t ldz, hypel_stream_t stream) {
    HYPEL_REQUIRE(y && z && rows > 0 && c > 0, "hypel_bn_act_fwd");
    HYPEL_REQUIRE((mean == nullptr) == (rstd == nullptr) && (mean == nullptr) == (beta == nullptr), "hypel_bn_act_fwd");
    const bool v4 = (c % 4 == 0) && aligned16(y) && aligned16(z) && aligned16(mask) && ldy % 4 == 0 &&
                    ldz % 4 == 0 && (!mask || ldm % 4 == 0) &&
                    (!res1 || idx1 || (aligned16(res1) && ld1 % 4 == 0)) &&
                    (!res2 || idx2 || (aligned16(res2) && ld2 % 4 == 0));
    const EwShape sh = ew_shape(rows, v4 ? c / 4 : c);
    if (v4)
        hipLaunchKernelGGL(bn_act_fwd_kernel<4>, dim3(sh.grid), dim3(256), 0, ST, y, ldy, rows, c, mean, rstd, beta,
                           act, alpha, mask, ldm, res1, ld1, idx1, res2, ld2, idx2, z, ldz, sh.tx_log2);
    else
        hipLaunchKernelGGL(bn_act_fwd_kernel<1>, dim3(sh.grid), dim3(256), 0, ST, y, ldy, rows, c, mean, rstd, beta,
                           act, alpha, mask, ldm, res1, ld1, idx1, res2, ld2, idx2, z, ldz, sh.tx_log2);
    HYPEL_CHECK_LAUNCH("hypel_bn_act_fwd");
    return 0;
}

static int launch_bwd_reduce(const float* dz, int64_t lddz, const float* y, int64_t ldy, int64_t rows, int32_t c,
                             const float* mean, const float* rstd, const float* beta, int32_t act, float alpha,
                             const float* mask, int64_t ldm, int32_t chunk_rows, float* partial,
                             hypel_stream_t stream, float* dy = nullptr, int64_t lddy = 0) {
    const int n_chunks = (int)((rows + chunk_rows - 1) / chunk_rows);
    constexpr bool v4_on = true;
    const bool v4 = v4_on && (c % 4 == 0) && (lddz % 4 == 0) && (ldy % 4 == 0) && (((uintptr_t)dz & 15) == 0) &&
                    (((uintptr_t)y & 15) == 0) && (!mask || ((ldm % 4 == 0) && (((uintptr_t)mask & 15) == 0))) &&
                    (!dy || ((lddy % 4 == 0) && (((uintptr_t)dy & 15) == 0)));
    const dim3 grid(n_chunks, (c + STAT_TX - 1) / STAT_TX);
#define HYPEL_BWD_REDUCE(K, D)                                                                                          \
    hipLaunchKernelGGL(K<D>, grid, dim3(256), 0, ST, dz, lddz, y, ldy, rows, c, mean, rstd, beta, act, alpha, mask, ldm, \
                       chunk_rows, partial, dy, lddy)
    if (v4) {
        if (dy) HYPEL_BWD_REDUCE(bn_act_bwd_reduce_v4_kernel, true);
        else HYPEL_BWD_REDUCE(bn_act_bwd_reduce_v4_kernel, false);
    } else {
        if (dy) HYPEL_BWD_REDUCE(bn_act_bwd_reduce_kernel, true);
        else HYPEL_BWD_REDUCE(bn_act_bwd_reduce_kernel, false);
    }
#undef HYPEL_BWD_REDUCE
    return 0;
}

extern "C" int hypel_bn_act_bwd_reduce(const float* dz, int64_t lddz, const float* y, int64_t ldy, int64_t rows,
                                       int32_t c, const float* mean, const float* rstd, const float* beta, int32_t act,
                                       float alpha, const float* mask, int64_t ldm, int32_t chunk_rows, float* partial,
                                       hypel_stream_t stream) {
    HYPEL_REQUIRE(dz && y && partial && rows > 0 && c > 0 && chunk_rows > 0, "hypel_bn_act_bwd_reduce");
    launch_bwd_reduce(dz, lddz, y, ldy, rows, c, mean, rstd, beta, act, alpha, mask, ldm, chunk_rows, partial,
                      stream);
    HYPEL_CHECK_LAUNCH("hypel_bn_act_bwd_reduce");
    return 0;
}

extern "C" int hypel_act_bias_bwd_reduce(const float* dz, int64_t lddz, const float* y, int64_t ldy, int64_t rows,
                                         int32_t c, int32_t act, float alpha, const float* mask, int64_t ldm,
                                         int32_t chunk_rows, float* partial, float* dy, int64_t lddy,
                                         hypel_stream_t stream) {
    HYPEL_REQUIRE(dz && y && partial && dy && rows > 0 && c > 0 && chunk_rows > 0 && lddy >= c, "hypel_act_bias_bwd_reduce");
    launch_bwd_reduce(dz, lddz, y, ldy, rows, c, nullptr, nullptr, nullptr, act, alpha, mask, ldm, chunk_rows, partial,
                      stream, dy, lddy);
    HYPEL_CHECK_LAUNCH("hypel_act_bias_bwd_reduce");
    return 0;
}

// byte offsets of the small-rows kernels are 32-bit buffer offsets
static inline bool small_span_ok(int64_t rows, int64_t ld) { return ld > 0 && rows * ld * 4 < 0x7ffffff0ll; }

extern "C" int hypel_bn_act_small_fwd(const float* y, int64_t ldy, int64_t rows, int32_t c, float eps,
                                      const float* beta, int32_t act, float alpha, const float* mask, int64_t ldm,
                                      float* mean, float* rstd, float* moving_mean, float* moving_var, float decay,
                                      float* z, int64_t ldz, hypel_stream_t stream) {
    HYPEL_REQUIRE(y && beta && mean && rstd && z && rows > 0 && rows <= SMALL_MAX_ROWS && c > 0,
                  "hypel_bn_act_small_fwd");
    HYPEL_REQUIRE((moving_mean == nullptr) == (moving_var == nullptr), "hypel_bn_act_small_fwd");
    HYPEL_REQUIRE(small_span_ok(rows, ldy) && small_span_ok(rows, ldz) && (!mask || small_span_ok(rows, ldm)),
                  "hypel_bn_act_small_fwd");
    const dim3 grid((c + SMALL_TX - 1) / SMALL_TX);
    if (rows == SMALL_MAX_ROWS)
        hipLaunchKernelGGL(bn_act_small_fwd_kernel<true>, grid, dim3(1024), 0, ST, y, (int)ldy, (int)rows, c, eps, beta,
                           act, alpha, mask, (int)ldm, mean, rstd, moving_mean, moving_var, decay, z, (int)ldz);
    else
        hipLaunchKernelGGL(bn_act_small_fwd_kernel<false>, grid, dim3(1024), 0, ST, y, (int)ldy, (int)rows, c, eps, beta,
                           act, alpha, mask, (int)ldm, mean, rstd, moving_mean, moving_var, decay, z, (int)ldz);
    HYPEL_CHECK_LAUNCH("hypel_bn_act_small_fwd");
    return 0;
}

extern "C" int hypel_bn_act_small_bwd(const float* dz, int64_t lddz, const float* y, int64_t ldy, int64_t rows,
                                      int32_t c, const float* mean, const float* rstd, const float* beta, int32_t act,
                                      float alpha, const float* mask, int64_t ldm, float* dy, int64_t lddy,
                                      float* dparam, int32_t accumulate, hypel_stream_t stream) {
    HYPEL_REQUIRE(dz && y && mean && rstd && beta && dy && rows > 0 && rows <= SMALL_MAX_ROWS && c > 0,
                  "hypel_bn_act_small_bwd");
    HYPEL_REQUIRE(small_span_ok(rows, lddz) && small_span_ok(rows, ldy) && small_span_ok(rows, lddy) &&
                      (!mask || small_span_ok(rows, ldm)),
                  "hypel_bn_act_small_bwd");
    const dim3 grid((c + SMALL_TX - 1) / SMALL_TX);
    if (rows == SMALL_MAX_ROWS)
        hipLaunchKernelGGL(bn_act_small_bwd_kernel<true>, grid, dim3(1024), 0, ST, dz, (int)lddz, y, (int)ldy, (int)rows,
                           c, mean, rstd, beta, act, alpha, mask, (int)ldm, dy, (int)lddy, dparam, accumulate);
    else
        hipLaunchKernelGGL(bn_act_small_bwd_kernel<false>, grid, dim3(1024), 0, ST, dz, (int)lddz, y, (int)ldy,
                           (int)rows, c, mean, rstd, beta, act, alpha, mask, (int)ldm, dy, (int)lddy, dparam, accumulate);
    HYPEL_CHECK_LAUNCH("hypel_bn_act_small_bwd");
    return 0;
}

extern "C" int hypel_bwd_reduce_finalize(const float* partial, int32_t n_chunks, int32_t c, float* sums, float* dparam,
                                         int32_t accumulate, hypel_stream_t stream) {
    HYPEL_REQUIRE(partial && sums && n_chunks > 0 && c > 0, "hypel_bwd_reduce_finalize");
    hipLaunchKernelGGL(bwd_reduce_finalize_kernel, dim3((c + 15) / 16), dim3(256), 0, ST, partial, n_chunks, c, sums,
                       dparam, accumulate);
    HYPEL_CHECK_LAUNCH("hypel_bwd_reduce_finalize");
    return 0;
}

static int bn_act_bwd_apply_launch(const float* dz, int64_t lddz, const float* y, int64_t ldy, int64_t rows, int32_t c,
                                   const float* mean, const float* rstd, const float* beta, int32_t act, float alpha,
                                   const float* mask, int64_t ldm, const float* sums, int64_t stat_rows, float* dy,
                                   int64_t lddy, hypel_stream_t stream, const char* what) {
    HYPEL_REQUIRE(dz && y && dy && rows > 0 && c > 0 && stat_rows >= rows, what);
    HYPEL_REQUIRE(mean == nullptr || sums != nullptr, what);
    const bool v4 = (c % 4 == 0) && aligned16(y) && aligned16(dz) && aligned16(dy) && aligned16(mask) &&
                    ldy % 4 == 0 && lddz % 4 == 0 && lddy % 4 == 0 && (!mask || ldm % 4 == 0);
    const EwShape sh = ew_shape(rows, v4 ? c / 4 : c);
    if (v4)
        hipLaunchKernelGGL(bn_act_bwd_apply_kernel<4>, dim3(sh.grid), dim3(256), 0, ST, dz, lddz, y, ldy, rows, c, mean,
                           rstd, beta, act, alpha, mask, ldm, sums, dy, lddy, sh.tx_log2, stat_rows);
    else
        hipLaunchKernelGGL(bn_act_bwd_apply_kernel<1>, dim3(sh.grid), dim3(256), 0, ST, dz, lddz, y, ldy, rows, c, mean,
                           rstd, beta, act, alpha, mask, ldm, sums, dy, lddy, sh.tx_log2, stat_rows);
    HYPEL_CHECK_LAUNCH(what);
    return 0;
}

extern "C" int hypel_bn_act_bwd_apply(const float* dz, int64_t lddz, const float* y, int64_t ldy, int64_t rows,
                                      int32_t c, const float* mean, const float* rstd, const float* beta, int32_t act,
                                      float alpha, const float* mask, int64_t ldm, const float* sums, float* dy,
                                      int64_t lddy, hypel_stream_t stream) {
    return bn_act_bwd_apply_launch(dz, lddz, y, ldy, rows, c, mean, rstd, beta, act, alpha, mask, ldm, sums, rows, dy,
                                   lddy, stream, "hypel_bn_act_bwd_apply");
}

extern "C" int hypel_bn_act_bwd_apply_global(const float* dz, int64_t lddz, const float* y, int64_t ldy, int64_t rows,
                                             int32_t c, const float* mean, const float* rstd, const float* beta,
                                             int32_t act, float alpha, const float* mask, int64_t ldm,
                                             const float* sums, int64_t stat_rows, float* dy, int64_t lddy,
                                             hypel_stream_t stream) {
    return bn_act_bwd_apply_launch(dz, lddz, y, ldy, rows, c, mean, rstd, beta, act, alpha, mask, ldm, sums, stat_rows,
                                   dy, lddy, stream, "hypel_bn_act_bwd_apply_global");
}

extern "C" int hypel_chanmap_bwd(const float* dz, int64_t lddz, int64_t rows, int32_t c, float* dr, int64_t lddr,
                                 int32_t cin, const int32_t* start, int32_t accumulate, hypel_stream_t stream) {
    HYPEL_REQUIRE(dz && dr && rows > 0 && c > 0 && cin > 0, "hypel_chanmap_bwd");
    HYPEL_REQUIRE(start != nullptr || cin == c, "hypel_chanmap_bwd");
    const bool v4 = (cin % 4 == 0) && aligned16(dr) && lddr % 4 == 0 && (start || (aligned16(dz) && lddz % 4 == 0));
    const EwShape sh = ew_shape(rows, v4 ? cin / 4 : cin);
    if (v4)
        hipLaunchKernelGGL(chanmap_bwd_kernel<4>, dim3(sh.grid), dim3(256), 0, ST, dz, lddz, rows, c, dr, lddr, cin,
                           start, accumulate, sh.tx_log2);
    else
        hipLaunchKernelGGL(chanmap_bwd_kernel<1>, dim3(sh.grid), dim3(256), 0, ST, dz, lddz, rows, c, dr, lddr, cin,
                           start, accumulate, sh.tx_log2);
    HYPEL_CHECK_LAUNCH("hypel_chanmap_bwd");
    return 0;
}

extern "C" int hypel_softmax_xent(const float* logits, int64_t ld, int64_t n, int32_t c, const float* labels,
                                  int64_t ldl, float* loss, float* dlogits, int64_t lddl, float gscale,
                                  hypel_stream_t stream) {
    HYPEL_REQUIRE(logits && labels && n > 0 && c > 0, "hypel_softmax_xent");
    hipLaunchKernelGGL(softmax_xent_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, ST, logits, ld, n, c,
                       labels, ldl, loss, dlogits, lddl, gscale);
    HYPEL_CHECK_LAUNCH("hypel_softmax_xent");
    return 0;
}

static void mse_partial_launch(const float* a, int64_t lda, const float* b, int64_t ldb, int64_t rows, int32_t c,
                               float* da, int64_t ldda, float gscale, float* ws, int grid, hypel_stream_t stream);

extern "C" int hypel_mse(const float* a, int64_t lda, const float* b, int64_t ldb, int64_t rows, int32_t c, float* out,
                         float* da, int64_t ldda, float gscale, float* ws, hypel_stream_t stream) {
    HYPEL_REQUIRE(a && b && out && ws && rows > 0 && c > 0, "hypel_mse");
    const int64_t total = rows * c;
    const int grid = hypel_grid_1d(total, 256, RED_BLOCKS);
    mse_partial_launch(a, lda, b, ldb, rows, c, da, ldda, gscale, ws, grid, stream);
    hipLaunchKernelGGL(sum_finalize_kernel, dim3(1), dim3(256), 0, ST, ws, grid, 1.0 / (double)total, out);
    HYPEL_CHECK_LAUNCH("hypel_mse");
    return 0;
}

static void mse_partial_launch(const float* a, int64_t lda, const float* b, int64_t ldb, int64_t rows, int32_t c,
                               float* da, int64_t ldda, float gscale, float* ws, int grid, hypel_stream_t stream) {
    const int64_t total = rows * c;
    const float gcoef = gscale * 2.0f / (float)total;
    const bool flat = lda == c && ldb == c && (!da || ldda == c) && total % 4 == 0 && aligned16(a) && aligned16(b) &&
                      aligned16(da);
    if (flat)
        hipLaunchKernelGGL(mse_partial_flat4_kernel, dim3(grid), dim3(256), 0, ST, reinterpret_cast<const float4*>(a),
                           reinterpret_cast<const float4*>(b), total / 4, reinterpret_cast<float4*>(da), gcoef, ws);
    else
        hipLaunchKernelGGL(mse_partial_kernel, dim3(grid), dim3(256), 0, ST, a, lda, b, ldb, rows, c, da, ldda, gcoef, ws);
}

extern "C" int hypel_mse_partial_f32(const float* a, int64_t lda, const float* b, int64_t ldb, int64_t rows, int32_t c,
                                     float* da, int64_t ldda, float gscale, float* ws, hypel_stream_t stream) {
    HYPEL_REQUIRE(a && b && ws && rows > 0 && c > 0, "hypel_mse_partial_f32");
    // always HYPEL_MSE_PARTIALS blocks (a block without elements publishes 0): the finaliser's count is a constant
    mse_partial_launch(a, lda, b, ldb, rows, c, da, ldda, gscale, ws, HYPEL_MSE_PARTIALS, stream);
    HYPEL_CHECK_LAUNCH("hypel_mse_partial_f32");
    return 0;
}

extern "C" int hypel_loss_finalize_f32(const float* loss_rows, int32_t n_rows, const float* mse_ws, double mse_scale,
                                       float* out_ce, float* out_mse, float* flag, uint64_t* step,
                                       hypel_stream_t stream) {
    HYPEL_REQUIRE(loss_rows && n_rows > 0 && out_ce && ((mse_ws == nullptr) == (out_mse == nullptr)),
                  "hypel_loss_finalize_f32");
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, ST, loss_rows, n_rows, mse_ws, HYPEL_MSE_PARTIALS,
                       mse_scale, out_ce, out_mse, flag, step);
    HYPEL_CHECK_LAUNCH("hypel_loss_finalize_f32");
    return 0;
}

extern "C" int hypel_sum_f32(const float* x, int64_t count, float scale, float* out, float* ws,
                             hypel_stream_t stream) {
    HYPEL_REQUIRE(x && out && ws && count > 0, "hypel_sum_f32");
    const int grid = hypel_grid_1d(count, 256, RED_BLOCKS);
    hipLaunchKernelGGL(sum_partial_kernel, dim3(grid), dim3(256), 0, ST, x, count, ws);
    hipLaunchKernelGGL(sum_finalize_kernel, dim3(1), dim3(256), 0, ST, ws, grid, (double)scale, out);
    HYPEL_CHECK_LAUNCH("hypel_sum_f32");
    return 0;
}

extern "C" int hypel_adam_tf1_guarded(float* p, const float* g, float* m, float* v, int64_t count, float lr_t,
                                      float beta1, float beta2, float eps, const float* skip, hypel_stream_t stream) {
    HYPEL_REQUIRE(p && g && m && v && count >= 0, "hypel_adam_tf1");
    if (count == 0) return 0;
    hipLaunchKernelGGL(adam_tf1_kernel, dim3(hypel_grid_1d(count, 256)), dim3(256), 0, ST, p, g, m, v, count, lr_t,
                       beta1, beta2, eps, skip);
    HYPEL_CHECK_LAUNCH("hypel_adam_tf1");
    return 0;
}

extern "C" int hypel_adam_tf1(float* p, const float* g, float* m, float* v, int64_t count, float lr_t, float beta1,
                              float beta2, float eps, hypel_stream_t stream) {
    return hypel_adam_tf1_guarded(p, g, m, v, count, lr_t, beta1, beta2, eps, nullptr, stream);
}

extern "C" int hypel_momentum_tf1_guarded(float* p, const float* g, float* a, int64_t count, float lr, float mu,
                                          const float* skip, hypel_stream_t stream) {
    HYPEL_REQUIRE(p && g && a && count >= 0, "hypel_momentum_tf1");
    if (count == 0) return 0;
    hipLaunchKernelGGL(momentum_tf1_kernel, dim3(hypel_grid_1d(count, 256)), dim3(256), 0, ST, p, g, a, count, lr, mu,
                       skip);
    HYPEL_CHECK_LAUNCH("hypel_momentum_tf1");
    return 0;
}

extern "C" int hypel_momentum_tf1(float* p, const float* g, float* a, int64_t count, float lr, float mu,
                                  hypel_stream_t stream) {
    return hypel_momentum_tf1_guarded(p, g, a, count, lr, mu, nullptr, stream);
}

extern "C" int hypel_loss_guard_f32(const float* loss_a, const float* loss_b, float* flag, hypel_stream_t stream) {
    HYPEL_REQUIRE(loss_a && flag, "hypel_loss_guard_f32");
    hipLaunchKernelGGL(loss_guard_kernel, dim3(1), dim3(64), 0, ST, loss_a, loss_b, flag);
    HYPEL_CHECK_LAUNCH("hypel_loss_guard_f32");
    return 0;
}

extern "C" int hypel_step_inc(uint64_t* step_dev, hypel_stream_t stream) {
    HYPEL_REQUIRE(step_dev, "hypel_step_inc");
    hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(64), 0, ST, step_dev);
    HYPEL_CHECK_LAUNCH("hypel_step_inc");
    return 0;
}

extern "C" int hypel_dropout_mask(float* mask, int64_t count, float keep_prob, uint64_t seed, const uint64_t* step_dev,
                                  hypel_stream_t stream) {
    HYPEL_REQUIRE(mask && step_dev && count > 0 && keep_prob > 0.0f && keep_prob <= 1.0f, "hypel_dropout_mask");
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(hypel_grid_1d((count + 3) / 4, 256)), dim3(256), 0, ST, mask, count,
                       keep_prob, seed, step_dev);
    HYPEL_CHECK_LAUNCH("hypel_dropout_mask");
    return 0;
}

extern "C" int hypel_argmax_confusion(const float* logits, int64_t ld, int64_t n, int32_t c, const int32_t* labels,
                                      int32_t* pred, int32_t* confusion, hypel_stream_t stream) {
    HYPEL_REQUIRE(logits && n > 0 && c > 0, "hypel_argmax_confusion");
    hipLaunchKernelGGL(argmax_confusion_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, ST, logits, ld, n, c,
                       labels, pred, confusion);
    HYPEL_CHECK_LAUNCH("hypel_argmax_confusion");
    return 0;
}

extern "C" int hypel_lrn_fwd(const float* x, int64_t ldx, int64_t rows, int32_t c, int32_t radius, float bias,
                             float alpha, float beta, float* y, int64_t ldy, hypel_stream_t stream) {
    HYPEL_REQUIRE(x && y && rows > 0 && c > 0 && radius >= 0, "hypel_lrn_fwd");
    hipLaunchKernelGGL(lrn_fwd_kernel, dim3(hypel_grid_1d(rows * c, 256)), dim3(256), 0, ST, x, ldx, rows, c, radius,
                       bias, alpha, beta, y, ldy);
    HYPEL_CHECK_LAUNCH("hypel_lrn_fwd");
    return 0;
}

extern "C" int hypel_lrn_bwd(const float* x, int64_t ldx, const float* dy, int64_t lddy, int64_t rows, int32_t c,
                             int32_t radius, float bias, float alpha, float beta, float* dx, int64_t lddx,
                             int32_t accumulate, hypel_stream_t stream) {
    HYPEL_REQUIRE(x && dy && dx && rows > 0 && c > 0 && radius >= 0, "hypel_lrn_bwd");
    hipLaunchKernelGGL(lrn_bwd_kernel, dim3(hypel_grid_1d(rows * c, 256)), dim3(256), 0, ST, x, ldx, dy, lddy, rows, c,
                       radius, bias, alpha, beta, dx, lddx, accumulate);
    HYPEL_CHECK_LAUNCH("hypel_lrn_bwd");
    return 0;
}
