// A short stack of narrow fully-connected layers as ONE launch per direction (gfx950, fp32 MFMA).
//
// The reference's discriminator (gan/shadow_data_models.py:95-121: flatten -> FC B->B -> FC B->B -> FC B->B/2, biases,
// leaky-ReLU 0.1 on the first two) is 3 tf_slim.fully_connected layers.  At the Gulfport band count (B = 64) a layer is 8
// MFLOP per 2048 samples: as separate launches (GEMM, activation; backward: data gradient, filter gradient, bias-gradient
// reduction + finaliser, activation backward) one application costs 5 launches forward and ~12 backward of ~4.7 us each
// -- 90 of the 130 launches of a CycleGAN step.  Here one block owns 16 samples (one v_mfma_f32_16x16x4_f32 row tile) and
// walks the whole stack out of LDS: activations never leave the CU, every layer's weights are staged once per block
// (the shape is supported when all of it fits the 160 KB: widths up to ~88 for the 3-layer discriminator).
//
//   forward : A_{l+1} = act_l(A_l W_l + b_l)                                    A_0 = x
//   backward: recompute A_0..A_L, then for l = L-1..0:  dZ = dA_{l+1} * act_l'(A_{l+1}),  db_l += colsum(dZ),
//             dW_l += A_l^T dZ,  dA_l = dZ W_l^T
// act' is taken from the OUTPUT's sign (leaky-ReLU with alpha > 0 keeps the sign of its argument; zero counts as the
// negative branch, as in the element-wise kernels).  Filter / bias gradients leave the block as partial slabs
// pw[blocks][sum cin*cout], pb[blocks][sum cout] (one per block, its row tiles summed in order), reduced in slab order
// by hypel_reduce_splits_f32 like the generator's: deterministic.
//
// LDS images: activations [16 x PA] with PA = 18 mod 32 (the 16 rows x 2 k of a half-wave's A fragment hit 32 banks),
// a layer's weights [cin x CP] with CP = 18 mod 32 (conflict-free as the transposed operand of dA = dZ W^T, 2 of 32
// banks doubled as the plain operand of the forward product).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

typedef float ds_f32x4 __attribute__((ext_vector_type(4)));
#define ST ((hipStream_t)stream)

namespace {

#ifndef DS_DIAG
#define DS_DIAG 0  // timing diagnostics (results wrong): 1 = no weight staging, 2 = no layer products, 3 = no zero fill
#endif
#ifndef DS_REGW
#define DS_REGW 1  // forward of stacks with every width <= 64: weights in registers (0: the LDS-staged kernel for every shape)
#endif
constexpr int DS_ROWS = 16;
constexpr int DS_THREADS = 256;
constexpr int DS_WAVES = 4;
constexpr int DS_MAXL = 4;
constexpr int DS_MAXW = 128;  // widest layer

struct DsShape {
    int n_layers;
    int width[DS_MAXL + 1];  // width[l] -> width[l + 1]
    int act[DS_MAXL];        // HYPEL_ACT_* of layer l (NONE or LRELU)
    float alpha;
    int pa, cp;              // LDS pitches
    int woff[DS_MAXL + 1], boff[DS_MAXL + 1];
    int wlds[DS_MAXL + 1];   // float offset of layer l's weight image inside the LDS weight area (all layers resident)
};

__host__ __device__ inline int ds_pitch(int w) {  // smallest p >= w with p = 18 (mod 32)
    int p = (w / 32) * 32 + 18;
    return p >= w ? p : p + 32;
}

__device__ __forceinline__ float ds_act(float v, int act, float alpha) {
    return act == HYPEL_ACT_LRELU ? (v > 0.0f ? v : alpha * v) : v;
}

__device__ __forceinline__ void ds_zero(float* p, int n, int tid) {
    if (DS_DIAG == 3) return;
    for (int i = tid; i < n; i += DS_THREADS) p[i] = 0.0f;
}

// Global -> LDS copies are LDS-DMA (global_load_lds_dword: the wave's 64 lanes land in 64 consecutive LDS words, no
// register in between), so that EVERY row of every operand is in flight before the one wait in front of the block
// barrier.  As load / store loops these copies were ~50 dependent memory round trips: 12 of the forward kernel's 15 us.
// DMA never writes the zero margins (rows / columns beyond the operand): they are zeroed once per block.
__device__ __forceinline__ void ds_dma_row(const float* g, float* l, int count, int lane) {  // count <= DS_MAXW
    for (int c0 = 0; c0 < count; c0 += 64)
        if (c0 + lane < count)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + c0 + lane),
                                             (__attribute__((address_space(3))) void*)(l + c0), 4, 0, 0);
}

// rows [0, rows_valid) of a [16 x width] tile; rows beyond (a ragged last tile) are zeroed with ordinary stores
__device__ __forceinline__ void ds_load_rows(float* img, int pa, int width, const float* __restrict__ x, int64_t ldx,
                                             int rows_valid, int tid, int lane, int wave) {
    for (int row = wave; row < rows_valid; row += DS_WAVES) ds_dma_row(x + (int64_t)row * ldx, img + row * pa, width, lane);
    if (rows_valid < DS_ROWS) {
        const int row = tid >> 4;
        if (row >= rows_valid)
            for (int c = tid & 15; c < width; c += 16) img[row * pa + c] = 0.0f;
    }
}

// The parts of a weight image the DMA does not write but a reduction loop walks: rows [cin, cin4) (k of the forward
// product) and columns [cout, cout4) (k of the data gradient)
__device__ __forceinline__ void ds_zero_w_margins(float* wl, int cin, int cout, int cp, int tid) {
    const int cin4 = (cin + 3) & ~3, cout4 = (cout + 3) & ~3, cout16 = (cout + 15) & ~15;
    for (int i = tid; i < (cin4 - cin) * cout16; i += DS_THREADS) wl[(cin + i / cout16) * cp + i % cout16] = 0.0f;
    if (cout4 > cout)
        for (int i = tid; i < cin * 4; i += DS_THREADS)
            if (cout + (i & 3) < cout4) wl[(i >> 2) * cp + cout + (i & 3)] = 0.0f;
}

// W_l [cin][cout] -> LDS [cin][cp] (margins: ds_zero_w_margins)
__device__ __forceinline__ void ds_stage_w(float* wl, const float* __restrict__ w, int cin, int cout, int cp, int lane,
                                           int wave) {
    if (DS_DIAG == 1) return;
    for (int r = wave; r < cin; r += DS_WAVES) ds_dma_row(w + r * cout, wl + r * cp, cout, lane);
}

// acc += sum_s a[s * sa] * b[s * sb] over `ksteps` MFMA k-steps: four steps' fragments are read before their MFMAs (a plain
// loop waits for two LDS reads in front of every MFMA of the dependent chain)
__device__ __forceinline__ ds_f32x4 ds_dot(const float* ap, int sa, const float* bp, int sb, int ksteps, ds_f32x4 acc) {
    int s = 0;
    for (; s + 4 <= ksteps; s += 4) {
        float a[4], b[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            a[q] = ap[(s + q) * sa];
            b[q] = bp[(s + q) * sb];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], b[q], acc, 0, 0, 0);
    }
    for (; s < ksteps; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[s * sa], bp[s * sb], acc, 0, 0, 0);
    return acc;
}

// one layer forward: dst = act(src W + b); with out != nullptr the result also goes to global memory.  `bias`: LDS copy.
__device__ __forceinline__ void ds_layer_fwd(const float* src, float* dst, const float* wl, const float* bias,
                                             int cin, int cout, int pa, int cp, int act, float alpha,
                                             float* __restrict__ out, int64_t ldo, int rows_valid, int lane, int wave) {
    const int r = lane & 15, kq = lane >> 4;
    const int ksteps = (cin + 3) >> 2;
    for (int jt = wave; 16 * jt < cout; jt += DS_WAVES) {
        ds_f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
        const float* ap = src + r * pa + kq;
        const float* bp = wl + kq * cp + 16 * jt + r;
        if (DS_DIAG != 2) acc = ds_dot(ap, 4, bp, 4 * cp, ksteps, acc);
        const int c = 16 * jt + r;
        if (c < cout) {
            const float bv = bias[c];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = 4 * kq + e;
                const float v = ds_act(acc[e] + bv, act, alpha);
                dst[row * pa + c] = v;
                if (out != nullptr && row < rows_valid) out[(int64_t)row * ldo + c] = v;
            }
        }
    }
}

// Several applications of stacks of ONE shape in one launch (hypel_dense_stack_*_apps): application g owns rows
// [g * n, (g + 1) * n) of x / out (/ dout / dx) and the weights at w + g * w_stride, b + g * b_stride -- the two critics of a
// CycleGAN step (same layer widths, different variables) -- and gridDim.x / n_apps consecutive blocks (and gradient slabs).
struct DsApps {
    int n_apps;
    int64_t w_stride, b_stride;    // element distance of application g + 1's variables from application g's
    int64_t pw_stride, pb_stride;  // ... of its first gradient slab (0: the slabs of all applications are consecutive)
};

__global__ __launch_bounds__(DS_THREADS) void dense_stack_fwd_kernel(const float* __restrict__ x, int64_t ldx, int64_t n,
                                                                     DsShape sh, const float* __restrict__ w,
                                                                     const float* __restrict__ b, float* __restrict__ out,
                                                                     int64_t ldo, DsApps apps) {
    extern __shared__ __attribute__((aligned(16))) float ds_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bpa = gridDim.x / apps.n_apps, app = blockIdx.x / bpa, blk = blockIdx.x - app * bpa;
    w += app * apps.w_stride;
    b += app * apps.b_stride;
    x += (int64_t)app * n * ldx;
    out += (int64_t)app * n * ldo;
    const int img = DS_ROWS * sh.pa;
    float* X = ds_lds;  // the input tile (only ever written by its DMA: the margins stay zero)
    float* P0 = ds_lds + img;
    float* P1 = ds_lds + 2 * img;
    float* wl = ds_lds + 3 * img;
    float* bl = wl + sh.wlds[sh.n_layers];  // LDS copy of the biases
    ds_zero(ds_lds, 3 * img, tid);  // the images; of the weight area only the margins (the DMA fills the rest)
    for (int l = 0; l < sh.n_layers; ++l) ds_zero_w_margins(wl + sh.wlds[l], sh.width[l], sh.width[l + 1], sh.cp, tid);
    __syncthreads();
    // every layer's weights and biases are staged ONCE per block, together with the first row tile
    for (int l = 0; l < sh.n_layers; ++l) ds_stage_w(wl + sh.wlds[l], w + sh.woff[l], sh.width[l], sh.width[l + 1], sh.cp, lane, wave);
    if (wave == 0) ds_dma_row(b, bl, min(sh.boff[sh.n_layers], DS_MAXW), lane);
    if (wave == 1 && sh.boff[sh.n_layers] > DS_MAXW)
        for (int o = DS_MAXW; o < sh.boff[sh.n_layers]; o += DS_MAXW) ds_dma_row(b + o, bl + o, min(sh.boff[sh.n_layers] - o, DS_MAXW), lane);
    const int64_t tiles = (n + DS_ROWS - 1) / DS_ROWS;
    for (int64_t t = blk; t < tiles; t += bpa) {
        const int64_t r0 = t * DS_ROWS;
        const int rows_valid = (int)min((int64_t)DS_ROWS, n - r0);
        ds_load_rows(X, sh.pa, sh.width[0], x + r0 * ldx, ldx, rows_valid, tid, lane, wave);
        __syncthreads();  // (hipcc drains the DMA queue -- vmcnt(0) -- in front of the barrier)
        const float* src = X;
        float* dst = P0;
        for (int l = 0; l < sh.n_layers; ++l) {
            const int cin = sh.width[l], cout = sh.width[l + 1];
            ds_layer_fwd(src, dst, wl + sh.wlds[l], bl + sh.boff[l], cin, cout, sh.pa, sh.cp, sh.act[l], sh.alpha,
                         l == sh.n_layers - 1 ? out + r0 * ldo : nullptr, ldo, rows_valid, lane, wave);
            // columns [cout, next multiple of 4) of dst must read as zero for the next layer's k loop
            if (tid < DS_ROWS * 4) {
                const int c = cout + (tid & 3);
                if (c < ((cout + 3) & ~3)) dst[(tid >> 2) * sh.pa + c] = 0.0f;
            }
            __syncthreads();
            src = dst;
            dst = dst == P0 ? P1 : P0;
        }
    }
}

// Forward, widths <= 64 (the Gulfport critic 64-64-64-32, the feature-discriminator slices): every layer has at most four 16-column
// tiles -- one per wave -- and at most 16 k-steps, so a lane's share of ALL layers' weights is <= 4 x 16 fragments.  They are
// requested straight from global memory (L2: every block reads the same 40 KB) into registers at kernel entry, all layers at once,
// under the input tile's DMA; no weight image in LDS, no margin zeroing, none of the 160 dword LDS-DMA instructions and their drain in
// front of the first barrier (round-4 diagnostic DS_DIAG=1: 4.3 us of the forward launch's 11.2).  Same k-step order as
// ds_layer_fwd, i.e. the same fmaf chain.
constexpr int DS_RW = 64;             // widest layer of the register-weight path
constexpr int DS_RK = DS_RW / 4;      // k-steps of its widest layer
__global__ __launch_bounds__(DS_THREADS) void dense_stack_fwd_regw_kernel(const float* __restrict__ x, int64_t ldx, int64_t n,
                                                                          DsShape sh, const float* __restrict__ w,
                                                                          const float* __restrict__ b, float* __restrict__ out,
                                                                          int64_t ldo, DsApps apps) {
    extern __shared__ __attribute__((aligned(16))) float ds_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bpa = gridDim.x / apps.n_apps, app = blockIdx.x / bpa, blk = blockIdx.x - app * bpa;
    w += app * apps.w_stride;
    b += app * apps.b_stride;
    x += (int64_t)app * n * ldx;
    out += (int64_t)app * n * ldo;
    const int r = lane & 15, kq = lane >> 4;
    const int col = 16 * wave + r;  // this lane's output column in every layer
    const int img = DS_ROWS * sh.pa;
    float* X = ds_lds;
    float* P0 = ds_lds + img;
    float* P1 = ds_lds + 2 * img;
    // this lane's weight fragments of every layer: wf[l][s] = W_l[4 s + kq][col] (0 outside the matrix), and its bias
    float wf[DS_MAXL][DS_RK], bv[DS_MAXL];
#pragma unroll
    for (int l = 0; l < DS_MAXL; ++l) {
        const int cin = l < sh.n_layers ? sh.width[l] : 0, cout = l < sh.n_layers ? sh.width[l + 1] : 0;
        const float* wl = w + sh.woff[l];
#pragma unroll
        for (int s = 0; s < DS_RK; ++s) {
            const int k = 4 * s + kq;
            wf[l][s] = (k < cin && col < cout) ? wl[k * cout + col] : 0.0f;
        }
        bv[l] = col < cout ? b[sh.boff[l] + col] : 0.0f;
    }
    ds_zero(ds_lds, 3 * img, tid);
    __syncthreads();
    const int64_t tiles = (n + DS_ROWS - 1) / DS_ROWS;
    for (int64_t t = blk; t < tiles; t += bpa) {
        const int64_t r0 = t * DS_ROWS;
        const int rows_valid = (int)min((int64_t)DS_ROWS, n - r0);
        ds_load_rows(X, sh.pa, sh.width[0], x + r0 * ldx, ldx, rows_valid, tid, lane, wave);
        __syncthreads();
        const float* src = X;
        float* dst = P0;
#pragma unroll
        for (int l = 0; l < DS_MAXL; ++l) {
            if (l < sh.n_layers) {
                const int cin = sh.width[l], cout = sh.width[l + 1];
                const int ksteps = (cin + 3) >> 2;
                if (16 * wave < cout) {  // wave-uniform: this wave owns a column tile of the layer
                    const float* ap = src + r * sh.pa + kq;
                    float a[DS_RK];
#pragma unroll
                    for (int s = 0; s < DS_RK; ++s) a[s] = s < ksteps ? ap[4 * s] : 0.0f;
                    ds_f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int s = 0; s < DS_RK; ++s)
                        if (s < ksteps) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], wf[l][s], acc, 0, 0, 0);
                    if (col < cout) {
                        float* o = l == sh.n_layers - 1 ? out + r0 * ldo : nullptr;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int row = 4 * kq + e;
                            const float v = ds_act(acc[e] + bv[l], sh.act[l], sh.alpha);
                            dst[row * sh.pa + col] = v;
                            if (o != nullptr && row < rows_valid) o[(int64_t)row * ldo + col] = v;
                        }
                    }
                }
                if (tid < DS_ROWS * 4) {  // columns [cout, next multiple of 4) of dst must read as zero for the next layer's k loop
                    const int c = cout + (tid & 3);
                    if (c < ((cout + 3) & ~3)) dst[(tid >> 2) * sh.pa + c] = 0.0f;
                }
                __syncthreads();
                src = dst;
                dst = dst == P0 ? P1 : P0;
            }
        }
    }
}

// Backward.  LDS: A_0..A_L (L + 1 images), the output-gradient tile, two gradient images, every layer's weights + biases.
__global__ __launch_bounds__(DS_THREADS) void dense_stack_bwd_kernel(
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ dout, int64_t lddo, int64_t n, DsShape sh,
    const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ dx, int64_t lddx, int accumulate_dx,
    float* __restrict__ pw, float* __restrict__ pb, DsApps apps) {
    extern __shared__ __attribute__((aligned(16))) float ds_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bpa = gridDim.x / apps.n_apps, app = blockIdx.x / bpa, blk = blockIdx.x - app * bpa;
    w += app * apps.w_stride;
    b += app * apps.b_stride;
    x += (int64_t)app * n * ldx;
    dout += (int64_t)app * n * lddo;
    if (dx != nullptr) dx += (int64_t)app * n * lddx;
    const int r = lane & 15, kq = lane >> 4;
    const int trow = tid >> 4, tcol = tid & 15;  // element-wise passes: thread -> (row, column + 16 k)
    const int L = sh.n_layers, pa = sh.pa, cp = sh.cp;
    const int img = DS_ROWS * pa;
    float* A = ds_lds;  // A_l = A + l * img (A_0: only ever written by its DMA)
    float* G = ds_lds + (DS_MAXL + 1) * img;  // the output gradient tile (DMA only)
    float* D0 = G + img;
    float* D1 = D0 + img;
    float* wl = D1 + img;
    float* bl = wl + sh.wlds[L];  // LDS copy of the biases
    ds_zero(ds_lds, (DS_MAXL + 4) * img, tid);  // the images; of the weight area only the margins
    for (int l = 0; l < L; ++l) ds_zero_w_margins(wl + sh.wlds[l], sh.width[l], sh.width[l + 1], cp, tid);
    __syncthreads();
    for (int l = 0; l < L; ++l) ds_stage_w(wl + sh.wlds[l], w + sh.woff[l], sh.width[l], sh.width[l + 1], cp, lane, wave);
    if (wave == 0) ds_dma_row(b, bl, min(sh.boff[L], DS_MAXW), lane);
    if (wave == 1 && sh.boff[L] > DS_MAXW)
        for (int o = DS_MAXW; o < sh.boff[L]; o += DS_MAXW) ds_dma_row(b + o, bl + o, min(sh.boff[L] - o, DS_MAXW), lane);
    const int wtotal = sh.woff[L], btotal = sh.boff[L];
    // gradient slabs: this application's share starts `pw_stride` floats behind the previous application's (the planner
    // appends the slabs of every application of one weight set to that set's region: PhasePlan._defer_slab_reduce)
    const int64_t slab = apps.pw_stride ? (int64_t)blk : (int64_t)blockIdx.x;
    pw += app * apps.pw_stride;
    pb += app * apps.pb_stride;
    float* my_pw = pw + slab * wtotal;
    float dbacc[DS_MAXL] = {0.0f, 0.0f, 0.0f, 0.0f};  // thread t owns column t of every layer's bias gradient
    bool first_tile = true;
    const int64_t tiles = (n + DS_ROWS - 1) / DS_ROWS;
    for (int64_t t = blk; t < tiles; t += bpa) {
        const int64_t r0 = t * DS_ROWS;
        const int rows_valid = (int)min((int64_t)DS_ROWS, n - r0);
        if (!first_tile) __syncthreads();  // the previous tile's gradients have been consumed
        ds_load_rows(A, pa, sh.width[0], x + r0 * ldx, ldx, rows_valid, tid, lane, wave);
        ds_load_rows(G, pa, sh.width[L], dout + r0 * lddo, lddo, rows_valid, tid, lane, wave);
        __syncthreads();  // (hipcc drains the DMA queue -- vmcnt(0) -- in front of the barrier)
        // ---- forward recompute: every layer's input stays in its own image ----
        for (int l = 0; l < L; ++l) {
            ds_layer_fwd(A + l * img, A + (l + 1) * img, wl + sh.wlds[l], bl + sh.boff[l], sh.width[l], sh.width[l + 1], pa, cp,
                         sh.act[l], sh.alpha, nullptr, 0, rows_valid, lane, wave);
            __syncthreads();
        }
        float* D = G;
        float* Dn = D0;
#pragma unroll 1
        for (int l = L - 1; l >= 0; --l) {
            const int cin = sh.width[l], cout = sh.width[l + 1];
            const float* Ain = A + l * img;
            const float* Aout = A + (l + 1) * img;
            const float* wcur = wl + sh.wlds[l];
            // dZ = dA_{l+1} * act'(A_{l+1}), in place
            if (sh.act[l] == HYPEL_ACT_LRELU) {
                for (int c = tcol; c < cout; c += 16)
                    if (!(Aout[trow * pa + c] > 0.0f)) D[trow * pa + c] *= sh.alpha;
                __syncthreads();
            }
            // bias gradient: column sums over the 16 rows, rows ascending
            if (tid < cout) {
                float s = 0.0f;
#pragma unroll
                for (int row = 0; row < DS_ROWS; ++row) s += D[row * pa + tid];
#pragma unroll
                for (int q = 0; q < DS_MAXL; ++q) dbacc[q] += q == l ? s : 0.0f;
            }
            // filter gradient tiles: dW[16 it .. +16][16 jt .. +16] (+)= A_l^T dZ  (reduction over the 16 rows)
            const int nit = (cin + 15) >> 4, njt = (cout + 15) >> 4;
            for (int tile = wave; tile < nit * njt; tile += DS_WAVES) {
                const int it = tile / njt, jt = tile - it * njt;
                ds_f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
                const float* ap = Ain + kq * pa + 16 * it + r;  // A^T fragment: element (i = r, n = 4 s + kq)
                const float* bp = D + kq * pa + 16 * jt + r;    // dZ fragment:  element (n = 4 s + kq, j = r)
                // all eight fragment words before the first MFMA (hipcc otherwise reads two, waits, multiplies, four times over)
                float fa[4], fb[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    fa[s] = ap[4 * s * pa];
                    fb[s] = bp[4 * s * pa];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[s], fb[s], acc, 0, 0, 0);
                const int j = 16 * jt + r;
                if (j < cout) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int i = 16 * it + 4 * kq + e;
                        if (i < cin) {
                            float* p = my_pw + sh.woff[l] + i * cout + j;
                            *p = first_tile ? acc[e] : *p + acc[e];
                        }
                    }
                }
            }
            // data gradient dA_l = dZ W_l^T (not needed below the first layer unless dx is wanted)
            if (l > 0 || dx != nullptr) {
                const int ksteps = (cout + 3) >> 2;
                for (int it = wave; 16 * it < cin; it += DS_WAVES) {
                    ds_f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
                    const float* ap = D + r * pa + kq;                 // dZ[row = r][k = 4 s + kq]
                    const float* bp = wcur + (16 * it + r) * cp + kq;  // W[i = 16 it + r][k = 4 s + kq]
                    acc = ds_dot(ap, 4, bp, 4, ksteps, acc);
                    const int c = 16 * it + r;
                    if (c < cin) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) Dn[(4 * kq + e) * pa + c] = acc[e];
                    }
                }
            }
            __syncthreads();
            D = Dn;
            Dn = Dn == D0 ? D1 : D0;
        }
        if (dx != nullptr && trow < rows_valid) {
            float* p = dx + (r0 + trow) * lddx;
            for (int c = tcol; c < sh.width[0]; c += 16) p[c] = accumulate_dx ? p[c] + D[trow * pa + c] : D[trow * pa + c];
        }
        first_tile = false;
    }
    // a block without a row tile still owns a slab: zero it (every slab is summed by the reduce)
    if (first_tile) {
        for (int i = tid; i < wtotal; i += DS_THREADS) my_pw[i] = 0.0f;
    }
    for (int l = 0; l < L; ++l)
        if (tid < sh.width[l + 1]) {
            float v = 0.0f;
#pragma unroll
            for (int q = 0; q < DS_MAXL; ++q) v = q == l ? dbacc[q] : v;
            pb[slab * btotal + sh.boff[l] + tid] = v;
        }
}

bool ds_shape(int n_layers, const int32_t* widths, int32_t act_mask, float alpha, DsShape& sh) {
    if (n_layers < 1 || n_layers > DS_MAXL) return false;
    int wmax = 0;
    for (int l = 0; l <= n_layers; ++l) {
        if (widths[l] < 1 || widths[l] > DS_MAXW) return false;
        wmax = widths[l] > wmax ? widths[l] : wmax;
    }
    sh.n_layers = n_layers;
    sh.alpha = alpha;
    sh.woff[0] = sh.boff[0] = 0;
    for (int l = 0; l <= DS_MAXL; ++l) sh.width[l] = l <= n_layers ? widths[l] : 0;
    for (int l = 0; l < DS_MAXL; ++l) {
        sh.act[l] = (l < n_layers && ((act_mask >> l) & 1)) ? HYPEL_ACT_LRELU : HYPEL_ACT_NONE;
        sh.woff[l + 1] = sh.woff[l] + (l < n_layers ? widths[l] * widths[l + 1] : 0);
        sh.boff[l + 1] = sh.boff[l] + (l < n_layers ? widths[l + 1] : 0);
    }
    const int w16 = (wmax + 15) & ~15;
    sh.pa = ds_pitch(w16);
    sh.cp = ds_pitch(w16);
    sh.wlds[0] = 0;
    for (int l = 0; l < DS_MAXL; ++l) sh.wlds[l + 1] = sh.wlds[l] + (l < n_layers ? ((widths[l] + 3) & ~3) * sh.cp : 0);
    return true;
}

// The register-weight forward pays a fixed price (4 x 16 predicated fragment loads, a 16-step predicated product loop) and saves the
// weight image's staging, so it is chosen by the size of that image -- same-box A/B, round 5 (tools/exp/dense_regw_ab.sh):
// 64-64-64-32 (10 240 floats) 11.3 -> 8.8 us, 64-64 (4 096) 5.9 -> 6.5, 16-16-16-8 (640) 6.2 -> 8.1.
constexpr int DS_REGW_MIN_WEIGHTS = 8192;
bool ds_regw(const DsShape& sh) {
    for (int l = 0; l <= sh.n_layers; ++l)
        if (sh.width[l] > DS_RW) return false;
    return sh.woff[sh.n_layers] >= DS_REGW_MIN_WEIGHTS;
}
size_t ds_fwd_regw_lds(const DsShape& sh) { return (size_t)(3 * DS_ROWS * sh.pa) * sizeof(float); }
size_t ds_fwd_lds(const DsShape& sh) {
    return (size_t)(3 * DS_ROWS * sh.pa + sh.wlds[sh.n_layers] + sh.boff[sh.n_layers]) * sizeof(float);
}
size_t ds_bwd_lds(const DsShape& sh) {
    return (size_t)((DS_MAXL + 4) * DS_ROWS * sh.pa + sh.wlds[sh.n_layers] + sh.boff[sh.n_layers]) * sizeof(float);
}

}  // namespace

extern "C" int hypel_dense_stack_blocks(int64_t n) {
    int64_t t = (n + DS_ROWS - 1) / DS_ROWS;
    if (t < 1) t = 1;
    if (t > 256) t = 256;
    return (int)t;
}

/* blocks (= gradient slabs) of a launch over n_apps applications of n rows each: n_apps equal shares */
extern "C" int hypel_dense_stack_blocks_apps(int64_t n, int32_t n_apps) {
    if (n_apps < 1) n_apps = 1;
    int64_t t = (n + DS_ROWS - 1) / DS_ROWS;
    int64_t per = 256 / n_apps;
    if (per < 1) per = 1;
    if (t < 1) t = 1;
    if (t > per) t = per;
    return (int)(t * n_apps);
}

extern "C" int hypel_dense_stack_supported(int32_t n_layers, int32_t w0, int32_t w1, int32_t w2, int32_t w3, int32_t w4) {
    const int32_t widths[DS_MAXL + 1] = {w0, w1, w2, w3, w4};
    DsShape sh;
    return ds_shape(n_layers, widths, 0, 0.0f, sh) && ds_bwd_lds(sh) <= 160 * 1024 ? 1 : 0;
}

extern "C" int hypel_dense_stack_fwd(const float* x, int64_t ldx, int64_t n, int32_t n_layers, int32_t w0, int32_t w1,
                                     int32_t w2, int32_t w3, int32_t w4, int32_t act_mask, float alpha, const float* w,
                                     const float* b, float* out, int64_t ldo, hypel_stream_t stream) {
    const int32_t widths[DS_MAXL + 1] = {w0, w1, w2, w3, w4};
    DsShape sh;
    HYPEL_REQUIRE(x && w && b && out && n > 0 && ds_shape(n_layers, widths, act_mask, alpha, sh), "hypel_dense_stack_fwd");
    if (DS_REGW && ds_regw(sh)) {
        hipLaunchKernelGGL(dense_stack_fwd_regw_kernel, dim3(hypel_dense_stack_blocks(n)), dim3(DS_THREADS), ds_fwd_regw_lds(sh), ST,
                           x, ldx, n, sh, w, b, out, ldo, DsApps{1, 0, 0, 0, 0});
        HYPEL_CHECK_LAUNCH("hypel_dense_stack_fwd");
        return 0;
    }
    const size_t lds = ds_fwd_lds(sh);
    HYPEL_REQUIRE(lds <= 160 * 1024, "hypel_dense_stack_fwd");
    (void)hipFuncSetAttribute((const void*)dense_stack_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(dense_stack_fwd_kernel, dim3(hypel_dense_stack_blocks(n)), dim3(DS_THREADS), lds, ST, x, ldx, n, sh, w,
                       b, out, ldo, DsApps{1, 0, 0, 0, 0});
    HYPEL_CHECK_LAUNCH("hypel_dense_stack_fwd");
    return 0;
}

extern "C" int hypel_dense_stack_fwd_apps(const float* x, int64_t ldx, int64_t n, int32_t n_apps, int64_t w_stride,
                                          int64_t b_stride, int32_t n_layers, int32_t w0, int32_t w1, int32_t w2, int32_t w3,
                                          int32_t w4, int32_t act_mask, float alpha, const float* w, const float* b,
                                          float* out, int64_t ldo, hypel_stream_t stream) {
    const int32_t widths[DS_MAXL + 1] = {w0, w1, w2, w3, w4};
    DsShape sh;
    HYPEL_REQUIRE(x && w && b && out && n > 0 && n_apps >= 1 && n_apps <= 16 && ds_shape(n_layers, widths, act_mask, alpha, sh),
                  "hypel_dense_stack_fwd_apps");
    if (DS_REGW && ds_regw(sh)) {
        hipLaunchKernelGGL(dense_stack_fwd_regw_kernel, dim3(hypel_dense_stack_blocks_apps(n, n_apps)), dim3(DS_THREADS),
                           ds_fwd_regw_lds(sh), ST, x, ldx, n, sh, w, b, out, ldo, DsApps{n_apps, w_stride, b_stride, 0, 0});
        HYPEL_CHECK_LAUNCH("hypel_dense_stack_fwd_apps");
        return 0;
    }
    const size_t lds = ds_fwd_lds(sh);
    HYPEL_REQUIRE(lds <= 160 * 1024, "hypel_dense_stack_fwd_apps");
    (void)hipFuncSetAttribute((const void*)dense_stack_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(dense_stack_fwd_kernel, dim3(hypel_dense_stack_blocks_apps(n, n_apps)), dim3(DS_THREADS), lds, ST, x,
                       ldx, n, sh, w, b, out, ldo, DsApps{n_apps, w_stride, b_stride, 0, 0});
    HYPEL_CHECK_LAUNCH("hypel_dense_stack_fwd_apps");
    return 0;
}

extern "C" int hypel_dense_stack_bwd(const float* x, int64_t ldx, const float* dout, int64_t lddo, int64_t n,
                                     int32_t n_layers, int32_t w0, int32_t w1, int32_t w2, int32_t w3, int32_t w4,
                                     int32_t act_mask, float alpha, const float* w, const float* b, float* dx,
                                     int64_t lddx, int32_t accumulate_dx, float* pw, float* pb, hypel_stream_t stream) {
    const int32_t widths[DS_MAXL + 1] = {w0, w1, w2, w3, w4};
    DsShape sh;
    HYPEL_REQUIRE(x && dout && w && b && pw && pb && n > 0 && ds_shape(n_layers, widths, act_mask, alpha, sh),
                  "hypel_dense_stack_bwd");
    const size_t lds = ds_bwd_lds(sh);
    HYPEL_REQUIRE(lds <= 160 * 1024, "hypel_dense_stack_bwd");
    (void)hipFuncSetAttribute((const void*)dense_stack_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(dense_stack_bwd_kernel, dim3(hypel_dense_stack_blocks(n)), dim3(DS_THREADS), lds, ST, x, ldx, dout, lddo,
                       n, sh, w, b, dx, lddx, accumulate_dx, pw, pb, DsApps{1, 0, 0, 0, 0});
    HYPEL_CHECK_LAUNCH("hypel_dense_stack_bwd");
    return 0;
}

extern "C" int hypel_dense_stack_bwd_apps(const float* x, int64_t ldx, const float* dout, int64_t lddo, int64_t n,
                                          int32_t n_apps, int64_t w_stride, int64_t b_stride, int64_t pw_stride,
                                          int64_t pb_stride, int32_t n_layers, int32_t w0,
                                          int32_t w1, int32_t w2, int32_t w3, int32_t w4, int32_t act_mask, float alpha,
                                          const float* w, const float* b, float* dx, int64_t lddx, int32_t accumulate_dx,
                                          float* pw, float* pb, hypel_stream_t stream) {
    const int32_t widths[DS_MAXL + 1] = {w0, w1, w2, w3, w4};
    DsShape sh;
    HYPEL_REQUIRE(x && dout && w && b && pw && pb && n > 0 && n_apps >= 1 && n_apps <= 16 &&
                      ds_shape(n_layers, widths, act_mask, alpha, sh),
                  "hypel_dense_stack_bwd_apps");
    const size_t lds = ds_bwd_lds(sh);
    HYPEL_REQUIRE(lds <= 160 * 1024, "hypel_dense_stack_bwd_apps");
    (void)hipFuncSetAttribute((const void*)dense_stack_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(dense_stack_bwd_kernel, dim3(hypel_dense_stack_blocks_apps(n, n_apps)), dim3(DS_THREADS), lds, ST, x,
                       ldx, dout, lddo, n, sh, w, b, dx, lddx, accumulate_dx, pw, pb,
                       DsApps{n_apps, w_stride, b_stride, pw_stride, pb_stride});
    HYPEL_CHECK_LAUNCH("hypel_dense_stack_bwd_apps");
    return 0;
}
