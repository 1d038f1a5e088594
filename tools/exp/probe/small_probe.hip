// Phase timestamps inside a copy of bn_act_small_bwd (1 block of 1024 threads): where do 14 us go?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../../hypelcnn_amd/csrc/common.h"
constexpr int SMALL_TX = 32, SMALL_TY = 32, SMALL_R = 32;
constexpr uint32_t SMALL_OOB = 0x7fffffffu;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t small_rsrc(const float* p) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7ffffff0, 0x00020000);
}
__device__ __forceinline__ void load_rows(const float* p, int ld, int col, bool ok, float (&v)[SMALL_R]) {
    const int ty = threadIdx.x / SMALL_TX;
    const __amdgpu_buffer_rsrc_t rs = small_rsrc(p);
    const uint32_t v0 = ok ? (uint32_t)(ty * ld + col) * 4u : SMALL_OOB;
#pragma unroll
    for (int i = 0; i < SMALL_R; ++i)
        v[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, v0, i * SMALL_TY * ld * 4, 0));
}
__device__ __forceinline__ float lane_sum(float v, float (*sh)[SMALL_TX + 1]) {
    const int tx = threadIdx.x & (SMALL_TX - 1), ty = threadIdx.x / SMALL_TX;
    __syncthreads();
    sh[ty][tx] = v;
    __syncthreads();
    float t = 0.0f;
#pragma unroll
    for (int k = 0; k < SMALL_TY; ++k) t += sh[k][tx];
    return t;
}
#define STAMP(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if (threadIdx.x == 0 && blockIdx.x == 0) stamps[k] = wall_clock64(); } while (0)
__global__ __launch_bounds__(1024) void probe_bwd(const float* dz, const float* y, int c, const float* mean, const float* rstd,
                                                  const float* beta, float alpha, float* dy, float* dparam, long long* stamps, int mode) {
    __shared__ float sh[SMALL_TY][SMALL_TX + 1];
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[0] = wall_clock64();
    const int tx = threadIdx.x & (SMALL_TX - 1), ty = threadIdx.x / SMALL_TX;
    const int col = blockIdx.x * SMALL_TX + tx;
    const bool ok = col < c;
    const int colc = ok ? col : c - 1;
    float g[SMALL_R], xh[SMALL_R];
    const float mu = mean[colc], rs = rstd[colc], be = beta[colc];
    load_rows(y, c, col, ok, xh);
    load_rows(dz, c, col, ok, g);
    STAMP(1);
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) stamps[8 + (threadIdx.x >> 6)] = wall_clock64();
    float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
    for (int i = 0; i < SMALL_R; ++i) {
        xh[i] = hypel_bn_xhat(xh[i], mu, rs);
        const float p = hypel_bn_pre(xh[i], be);
        g[i] = g[i] * (p > 0.0f ? 1.0f : alpha);
        s0 += g[i];
        s1 += g[i] * xh[i];
    }
    STAMP(2);
    const float t0 = lane_sum(s0, sh);
    const float t1 = lane_sum(s1, sh);
    STAMP(3);
    if (ok && ty == 0 && dparam) dparam[col] = t0;
    const float m0 = t0 * (1.0f / 1024.0f), m1 = t1 * (1.0f / 1024.0f);
    const __amdgpu_buffer_rsrc_t rsd = small_rsrc(dy);
    const uint32_t v0 = ok ? (uint32_t)(ty * c + col) * 4u : SMALL_OOB;
#pragma unroll
    for (int i = 0; i < SMALL_R; ++i)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, rs * (g[i] - m0 - xh[i] * m1)), rsd, v0, i * SMALL_TY * c * 4, 0);
    STAMP(4);
}
int main() {
    for (int c : {15, 108, 980}) {
        const int rows = 1024;
        float *dz, *y, *dy, *mean, *rstd, *beta, *dparam; long long* st;
        hipMalloc(&dz, rows * c * 4); hipMalloc(&y, rows * c * 4); hipMalloc(&dy, rows * c * 4);
        hipMalloc(&mean, c * 4); hipMalloc(&rstd, c * 4); hipMalloc(&beta, c * 4); hipMalloc(&dparam, c * 4);
        hipMalloc(&st, 256);
        hipMemset(dz, 0, rows * c * 4); hipMemset(y, 0, rows * c * 4); hipMemset(mean, 0, c * 4); hipMemset(rstd, 0, c * 4); hipMemset(beta, 0, c * 4);
        long long h[32];
        for (int rep = 0; rep < 4; ++rep) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a);
            for (int k = 0; k < 20; ++k)
                hipLaunchKernelGGL(probe_bwd, dim3((c + 31) / 32), dim3(1024), 0, 0, dz, y, c, mean, rstd, beta, 0.18f, dy, dparam, st, 0);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            hipMemcpy(h, st, 192, hipMemcpyDeviceToHost);
            if (rep == 3)
                printf("c=%4d  %.2f us/launch (eager chain)  in-kernel (100 MHz ticks -> us): loads %.2f  math %.2f  sums %.2f  stores %.2f  total %.2f\n", c,
                       ms * 1e3 / 20, (h[1] - h[0]) / 100.0, (h[2] - h[1]) / 100.0, (h[3] - h[2]) / 100.0, (h[4] - h[3]) / 100.0, (h[4] - h[0]) / 100.0);
            if (rep == 3) { printf("   per-wave load completion (us after start):"); for (int w = 0; w < 16; ++w) printf(" %.2f", (h[8 + w] - h[0]) / 100.0); printf("\n"); }
        }
    }
    return 0;
}
