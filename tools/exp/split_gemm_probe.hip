// EXPERIMENT (round 4, NOTES 4.G): fp32 GEMM on the bf16 matrix cores with operands split three ways.
//   x = hi + mid + lo exactly (three round-to-nearest bf16 parts cover the 24-bit significand); every partial product
//   part_a * part_b is exact in the fp32 accumulator; NPROD = 6 drops mid*lo, lo*mid, lo*lo (<= 2^-24 of |a*b|: below
//   the rounding unit of the fp32 accumulate that follows), NPROD = 9 keeps all (the exact product), NPROD = 3 is the
//   usual "bf16x3" (~2^-16).  v_mfma_f32_32x32x16_bf16 runs at 16x the rate of v_mfma_f32_32x32x2_f32, so six products
//   per multiply-add still leave 2.67x the fp32 matrix peak (157.3 -> 419 TFLOP/s), nine 1.78x.
// Build: hipcc --offload-arch=gfx950 -O3 tools/exp/split_gemm_probe.hip -o tools/exp/split_gemm_probe
// Run:   tools/exp/split_gemm_probe [M N K]   (prints TFLOP/s and the error against fp64 next to a plain fp32 chain)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                   \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int PITCH = BK + 8;            // bf16 elements per LDS row: 80 bytes, 8 consecutive rows cover the 32 banks once
constexpr int PLANE = BM * PITCH;        // elements of one part's image

// two floats -> their (hi, mid, lo) bf16 parts, element 0 in the low half of each word
__device__ __forceinline__ void split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    bf16x2 p = {(__bf16)x0, (__bf16)x1};
    h = __builtin_bit_cast(unsigned, p);
    float r0 = x0 - __builtin_bit_cast(float, h << 16), r1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
    p = bf16x2{(__bf16)r0, (__bf16)r1};
    m = __builtin_bit_cast(unsigned, p);
    r0 -= __builtin_bit_cast(float, m << 16);
    r1 -= __builtin_bit_cast(float, m & 0xffff0000u);
    p = bf16x2{(__bf16)r0, (__bf16)r1};
    l = __builtin_bit_cast(unsigned, p);
}

template <int NPROD>
__global__ __launch_bounds__(256, 2) void split_gemm(const float* __restrict__ A, int lda, const float* __restrict__ B,
                                                     int ldb, float* __restrict__ C, int ldc, int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) unsigned short As[3 * PLANE], Bs[3 * PLANE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lhi = lane >> 5;
    const int n_nt = (N + BN - 1) / BN;
    const int m0 = (blockIdx.x / n_nt) * BM, n0 = (blockIdx.x % n_nt) * BN;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // A [m][k]: thread -> 4 x (row, k quad); B [k][n]: thread -> column n, 4 k quads (rows of the same column: k-contiguous in LDS)
    float4 ra[4];
    float rb[4][4];
    const int a_kq = tid & 7, a_r0 = tid >> 3;  // rows a_r0 + 32 i
    const int b_n = tid & 127, b_q0 = tid >> 7;  // k quads b_q0 + 2 i
    auto load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = m0 + a_r0 + 32 * i;
            ra[i] = r < M ? *reinterpret_cast<const float4*>(A + (size_t)r * lda + k0 + 4 * a_kq) : float4{0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = k0 + 4 * (b_q0 + 2 * i) + e;
                rb[i][e] = n0 + b_n < N ? B[(size_t)k * ldb + n0 + b_n] : 0.0f;
            }
    };
    load(0);
    const int a_rd = (wm * 64 + l31) * PITCH + 8 * lhi, b_rd = (wn * 64 + l31) * PITCH + 8 * lhi;
    for (int k0 = 0; k0 < K; k0 += BK) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint2 h, m, l;
            split2(ra[i].x, ra[i].y, h.x, m.x, l.x);
            split2(ra[i].z, ra[i].w, h.y, m.y, l.y);
            const int o = (a_r0 + 32 * i) * PITCH + 4 * a_kq;
            *reinterpret_cast<uint2*>(&As[o]) = h;
            *reinterpret_cast<uint2*>(&As[PLANE + o]) = m;
            *reinterpret_cast<uint2*>(&As[2 * PLANE + o]) = l;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint2 h, m, l;
            split2(rb[i][0], rb[i][1], h.x, m.x, l.x);
            split2(rb[i][2], rb[i][3], h.y, m.y, l.y);
            const int o = b_n * PITCH + 4 * (b_q0 + 2 * i);
            *reinterpret_cast<uint2*>(&Bs[o]) = h;
            *reinterpret_cast<uint2*>(&Bs[PLANE + o]) = m;
            *reinterpret_cast<uint2*>(&Bs[2 * PLANE + o]) = l;
        }
        __syncthreads();
        if (k0 + BK < K) load(k0 + BK);
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            bf16x8 a[2][3], b[2][3];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    a[i][p] = *reinterpret_cast<const bf16x8*>(&As[p * PLANE + a_rd + i * 32 * PITCH + 16 * s]);
                    b[i][p] = *reinterpret_cast<const bf16x8*>(&Bs[p * PLANE + b_rd + i * 32 * PITCH + 16 * s]);
                }
            // smallest partial products first
            constexpr int PA[9] = {2, 1, 2, 1, 2, 0, 1, 0, 0};
            constexpr int PB[9] = {2, 2, 1, 1, 0, 2, 0, 1, 0};
#pragma unroll
            for (int q = 9 - NPROD; q < 9; ++q)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA[q]], b[j][PB[q]], acc[i][j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = m0 + wm * 64 + i * 32 + (e >> 2) * 8 + lhi * 4 + (e & 3), c = n0 + wn * 64 + j * 32 + l31;
                if (r < M && c < N) C[(size_t)r * ldc + c] = acc[i][j][e];
            }
}

// v2: A fragments straight from global memory (each lane loads the 8 consecutive reduction columns of its row that the
// 32x32x16 fragment wants, splits them in registers; one k-tile ahead), B through a double-buffered LDS image: one barrier
// per k-tile, half the LDS traffic.
template <int NPROD>
__global__ __launch_bounds__(256, 2) void split_gemm_v2(const float* __restrict__ A, int lda, const float* __restrict__ B,
                                                        int ldb, float* __restrict__ C, int ldc, int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) unsigned short Bs[2][3 * PLANE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lhi = lane >> 5;
    const int n_nt = (N + BN - 1) / BN;
    const int m0 = (blockIdx.x / n_nt) * BM, n0 = (blockIdx.x % n_nt) * BN;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    float4 ra[2][2][2], rn[2][2][2];  // [row tile][k step][half]
    float rb[4][4];
    const int b_n = tid & 127, b_q0 = tid >> 7;
    const float* arow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) arow[i] = A + (size_t)min(m0 + wm * 64 + i * 32 + l31, M - 1) * lda + 8 * lhi;
    auto load_a = [&](int k0, float4 (&r)[2][2][2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                r[i][s][0] = *reinterpret_cast<const float4*>(arow[i] + k0 + 16 * s);
                r[i][s][1] = *reinterpret_cast<const float4*>(arow[i] + k0 + 16 * s + 4);
            }
    };
    auto load_b = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = k0 + 4 * (b_q0 + 2 * i) + e;
                rb[i][e] = n0 + b_n < N ? B[(size_t)k * ldb + n0 + b_n] : 0.0f;
            }
    };
    auto store_b = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint2 h, m, l;
            split2(rb[i][0], rb[i][1], h.x, m.x, l.x);
            split2(rb[i][2], rb[i][3], h.y, m.y, l.y);
            const int o = b_n * PITCH + 4 * (b_q0 + 2 * i);
            *reinterpret_cast<uint2*>(&Bs[buf][o]) = h;
            *reinterpret_cast<uint2*>(&Bs[buf][PLANE + o]) = m;
            *reinterpret_cast<uint2*>(&Bs[buf][2 * PLANE + o]) = l;
        }
    };
    load_a(0, ra);
    load_b(0);
    store_b(0);
    __syncthreads();
    const int b_rd = (wn * 64 + l31) * PITCH + 8 * lhi;
    int buf = 0;
    for (int k0 = 0; k0 < K; k0 += BK) {
        const bool more = k0 + BK < K;
        if (more) {
            load_a(k0 + BK, rn);
            load_b(k0 + BK);
        }
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            bf16x8 a[2][3], b[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                uint4 h, m, l;
                split2(ra[i][s][0].x, ra[i][s][0].y, h.x, m.x, l.x);
                split2(ra[i][s][0].z, ra[i][s][0].w, h.y, m.y, l.y);
                split2(ra[i][s][1].x, ra[i][s][1].y, h.z, m.z, l.z);
                split2(ra[i][s][1].z, ra[i][s][1].w, h.w, m.w, l.w);
                a[i][0] = __builtin_bit_cast(bf16x8, h);
                a[i][1] = __builtin_bit_cast(bf16x8, m);
                a[i][2] = __builtin_bit_cast(bf16x8, l);
            }
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    b[i][p] = *reinterpret_cast<const bf16x8*>(&Bs[buf][p * PLANE + b_rd + i * 32 * PITCH + 16 * s]);
            constexpr int PA[9] = {2, 1, 2, 1, 2, 0, 1, 0, 0};
            constexpr int PB[9] = {2, 2, 1, 1, 0, 2, 0, 1, 0};
#pragma unroll
            for (int q = 9 - NPROD; q < 9; ++q)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA[q]], b[j][PB[q]], acc[i][j], 0, 0, 0);
        }
        if (more) store_b(buf ^ 1);
        __syncthreads();
        buf ^= 1;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                ra[i][s][0] = rn[i][s][0];
                ra[i][s][1] = rn[i][s][1];
            }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = m0 + wm * 64 + i * 32 + (e >> 2) * 8 + lhi * 4 + (e & 3), c = n0 + wn * 64 + j * 32 + l31;
                if (r < M && c < N) C[(size_t)r * ldc + c] = acc[i][j][e];
            }
}

// v3: the B operand (weights: small, constant over a step) arrives PRE-SPLIT -- three bf16 planes [n][k] written once by
// presplit_b -- so that its share of the staging is six 16-byte copies per thread and k-tile (no conversion, no scattered
// 4-byte loads); A as in v1.
__global__ void presplit_b(const float* __restrict__ B, int ldb, unsigned short* __restrict__ P, int N, int K) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, k2 = blockIdx.y;  // one (n, pair of k) per thread
    if (n >= N) return;
    unsigned h, m, l;
    split2(B[(size_t)(2 * k2) * ldb + n], B[(size_t)(2 * k2 + 1) * ldb + n], h, m, l);
    const size_t plane = (size_t)N * K;
    reinterpret_cast<unsigned*>(P)[((size_t)n * K + 2 * k2) / 2] = h;
    reinterpret_cast<unsigned*>(P + plane)[((size_t)n * K + 2 * k2) / 2] = m;
    reinterpret_cast<unsigned*>(P + 2 * plane)[((size_t)n * K + 2 * k2) / 2] = l;
}

template <int NPROD>
__global__ __launch_bounds__(256, 2) void split_gemm_v3(const float* __restrict__ A, int lda,
                                                        const unsigned short* __restrict__ Bp, float* __restrict__ C,
                                                        int ldc, int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) unsigned short As[3 * PLANE], Bs[3 * PLANE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lhi = lane >> 5;
    const int n_nt = (N + BN - 1) / BN;
    const int m0 = (blockIdx.x / n_nt) * BM, n0 = (blockIdx.x % n_nt) * BN;
    const size_t plane = (size_t)N * K;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    float4 ra[4];
    uint4 rb[3][2];  // [plane][half]: 8 bf16 of row b_n0 + 64 h, k chunk b_kc
    const int a_kq = tid & 7, a_r0 = tid >> 3;
    const int b_kc = tid & 3, b_n0 = tid >> 2;  // 4 chunks of 8 k per row, rows b_n0, b_n0 + 64
    auto load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = m0 + a_r0 + 32 * i;
            ra[i] = r < M ? *reinterpret_cast<const float4*>(A + (size_t)r * lda + k0 + 4 * a_kq) : float4{0, 0, 0, 0};
        }
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int n = n0 + b_n0 + 64 * h;
                rb[p][h] = n < N ? *reinterpret_cast<const uint4*>(Bp + p * plane + (size_t)n * K + k0 + 8 * b_kc) : uint4{0, 0, 0, 0};
            }
    };
    load(0);
    const int a_rd = (wm * 64 + l31) * PITCH + 8 * lhi, b_rd = (wn * 64 + l31) * PITCH + 8 * lhi;
    for (int k0 = 0; k0 < K; k0 += BK) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint2 h, m, l;
            split2(ra[i].x, ra[i].y, h.x, m.x, l.x);
            split2(ra[i].z, ra[i].w, h.y, m.y, l.y);
            const int o = (a_r0 + 32 * i) * PITCH + 4 * a_kq;
            *reinterpret_cast<uint2*>(&As[o]) = h;
            *reinterpret_cast<uint2*>(&As[PLANE + o]) = m;
            *reinterpret_cast<uint2*>(&As[2 * PLANE + o]) = l;
        }
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                *reinterpret_cast<uint4*>(&Bs[p * PLANE + (b_n0 + 64 * h) * PITCH + 8 * b_kc]) = rb[p][h];
        __syncthreads();
        if (k0 + BK < K) load(k0 + BK);
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            bf16x8 a[2][3], b[2][3];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    a[i][p] = *reinterpret_cast<const bf16x8*>(&As[p * PLANE + a_rd + i * 32 * PITCH + 16 * s]);
                    b[i][p] = *reinterpret_cast<const bf16x8*>(&Bs[p * PLANE + b_rd + i * 32 * PITCH + 16 * s]);
                }
            constexpr int PA[9] = {2, 1, 2, 1, 2, 0, 1, 0, 0};
            constexpr int PB[9] = {2, 2, 1, 1, 0, 2, 0, 1, 0};
#pragma unroll
            for (int q = 9 - NPROD; q < 9; ++q)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA[q]], b[j][PB[q]], acc[i][j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = m0 + wm * 64 + i * 32 + (e >> 2) * 8 + lhi * 4 + (e & 3), c = n0 + wn * 64 + j * 32 + l31;
                if (r < M && c < N) C[(size_t)r * ldc + c] = acc[i][j][e];
            }
}

template <int NPROD, int VER>
static void run(const float* dA, const float* dB, float* dC, int M, int N, int K, const std::vector<float>& hA,
                const std::vector<float>& hB) {
    const int blocks = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    static unsigned short* dBp = nullptr;
    if (VER == 3 && !dBp) {
        CHECK(hipMalloc(&dBp, (size_t)3 * N * K * 2));
        presplit_b<<<dim3((N + 255) / 256, K / 2), 256>>>(dB, N, dBp, N, K);
    }
    auto go = [&]() {
        if (VER == 3) split_gemm_v3<NPROD><<<blocks, 256>>>(dA, K, dBp, dC, N, M, N, K);
        else if (VER == 2) split_gemm_v2<NPROD><<<blocks, 256>>>(dA, K, dB, N, dC, N, M, N, K);
        else split_gemm<NPROD><<<blocks, 256>>>(dA, K, dB, N, dC, N, M, N, K);
    };
    for (int i = 0; i < 3; ++i) go();
    CHECK(hipEventRecord(e0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) go();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    std::vector<float> hC((size_t)M * N);
    CHECK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    // error against fp64 on a sample of rows, next to a plain fp32 fmaf chain (what an fp32 matrix core leaves)
    double worst = 0, worst32 = 0, scale = 0;
    for (int r = 0; r < M; r += M / 37 + 1)
        for (int c = 0; c < N; c += 7) {
            double ref = 0, mag = 0;
            float f = 0;
            for (int k = 0; k < K; ++k) {
                ref += (double)hA[(size_t)r * K + k] * hB[(size_t)k * N + c];
                mag += fabs((double)hA[(size_t)r * K + k] * hB[(size_t)k * N + c]);
                f = fmaf(hA[(size_t)r * K + k], hB[(size_t)k * N + c], f);
            }
            worst = fmax(worst, fabs(hC[(size_t)r * N + c] - ref) / mag);
            worst32 = fmax(worst32, fabs(f - ref) / mag);
            scale = fmax(scale, mag);
        }
    printf("v%d NPROD %d: %8.1f us  %7.1f TFLOP/s (fp32-equivalent)   max |err| / sum|a b|: %.3e   (fp32 fmaf chain: %.3e)\n", VER, NPROD,
           ms * 1e3, 2.0 * M * N * K / ms / 1e9, worst, worst32);
}

int main(int argc, char** argv) {
    const int M = argc > 3 ? atoi(argv[1]) : 50176, N = argc > 3 ? atoi(argv[2]) : 512, K = argc > 3 ? atoi(argv[3]) : 960;
    std::vector<float> hA((size_t)M * K), hB((size_t)K * N);
    unsigned s = 12345;
    auto rnd = [&]() {
        s = s * 1664525u + 1013904223u;
        return ((s >> 8) / 16777216.0f) * 2.0f - 1.0f;
    };
    for (auto& v : hA) v = rnd() * expf(3.0f * rnd());
    for (auto& v : hB) v = rnd() * expf(3.0f * rnd());
    float *dA, *dB, *dC;
    CHECK(hipMalloc(&dA, hA.size() * 4));
    CHECK(hipMalloc(&dB, hB.size() * 4));
    CHECK(hipMalloc(&dC, (size_t)M * N * 4));
    CHECK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    printf("M %d N %d K %d\n", M, N, K);
    run<3, 1>(dA, dB, dC, M, N, K, hA, hB);
    run<6, 1>(dA, dB, dC, M, N, K, hA, hB);
    run<9, 1>(dA, dB, dC, M, N, K, hA, hB);
    run<3, 3>(dA, dB, dC, M, N, K, hA, hB);
    run<6, 3>(dA, dB, dC, M, N, K, hA, hB);
    run<9, 3>(dA, dB, dC, M, N, K, hA, hB);
    run<3, 2>(dA, dB, dC, M, N, K, hA, hB);
    run<6, 2>(dA, dB, dC, M, N, K, hA, hB);
    run<9, 2>(dA, dB, dC, M, N, K, hA, hB);
    return 0;
}
