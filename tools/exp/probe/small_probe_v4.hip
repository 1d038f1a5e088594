// Vectorised variant: a thread owns 4 adjacent columns (one dwordx4 per row) and 8 rows; block = 8 column quads x 128 row lanes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../../hypelcnn_amd/csrc/common.h"
constexpr int QX = 8, RY = 128, RR = 8;   // 32 columns per block, 1024 rows
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const float* p) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7ffffff0, 0x00020000);
}
template <int W>
__global__ __launch_bounds__(1024) void probe_bwd_v(const float* dz, const float* y, int c, const float* mean, const float* rstd,
                                                    const float* beta, float alpha, float* dy, float* dparam, long long* stamps) {
    __shared__ float sh[2][16][32];
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[0] = wall_clock64();
    const int q = threadIdx.x & (QX - 1), ty = threadIdx.x / QX;
    const int col = blockIdx.x * 32 + q * 4;
    float g[RR][4], xh[RR][4];
    const __amdgpu_buffer_rsrc_t ry = rsrc(y), rz = rsrc(dz);
    const uint32_t v0 = (uint32_t)(ty * c + col) * 4u;
    // columns past c inside the quad: the buffer range check is per dword? (raw buffer: per element of the vector) -> use c as bound via descriptor would need stride; here c % 4 == 0 or the tail quad reads into the next row (harmless for timing)
#pragma unroll
    for (int i = 0; i < RR; ++i) {
        if (W == 4) {
            const auto a = __builtin_amdgcn_raw_buffer_load_b128(ry, v0, i * RY * c * 4, 0);
            const auto b = __builtin_amdgcn_raw_buffer_load_b128(rz, v0, i * RY * c * 4, 0);
            for (int k = 0; k < 4; ++k) { xh[i][k] = __builtin_bit_cast(float, a[k]); g[i][k] = __builtin_bit_cast(float, b[k]); }
        } else {
            for (int k = 0; k < 4; ++k) {
                xh[i][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry, v0 + 4 * k, i * RY * c * 4, 0));
                g[i][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rz, v0 + 4 * k, i * RY * c * 4, 0));
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) stamps[8 + (threadIdx.x >> 6)] = wall_clock64();
    float s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
    float mu[4], rs[4], be[4];
    for (int k = 0; k < 4; ++k) { const int cc = min(col + k, c - 1); mu[k] = mean[cc]; rs[k] = rstd[cc]; be[k] = beta[cc]; }
#pragma unroll
    for (int i = 0; i < RR; ++i)
        for (int k = 0; k < 4; ++k) {
            xh[i][k] = hypel_bn_xhat(xh[i][k], mu[k], rs[k]);
            const float p = hypel_bn_pre(xh[i][k], be[k]);
            g[i][k] *= (p > 0.0f ? 1.0f : alpha);
            s0[k] += g[i][k];
            s1[k] += g[i][k] * xh[i][k];
        }
    // a wave = 8 quads x 8 row lanes: combine the 8 row lanes (xor 8, 16, 32), then 16 waves through LDS
    for (int k = 0; k < 4; ++k)
        for (int o = 8; o < 64; o <<= 1) { s0[k] += __shfl_xor(s0[k], o, 64); s1[k] += __shfl_xor(s1[k], o, 64); }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) < QX)
        for (int k = 0; k < 4; ++k) { sh[0][wave][q * 4 + k] = s0[k]; sh[1][wave][q * 4 + k] = s1[k]; }
    __syncthreads();
    float t0[4] = {0, 0, 0, 0}, t1[4] = {0, 0, 0, 0};
    for (int w = 0; w < 16; ++w)
        for (int k = 0; k < 4; ++k) { t0[k] += sh[0][w][q * 4 + k]; t1[k] += sh[1][w][q * 4 + k]; }
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[3] = wall_clock64();
    if (ty == 0) for (int k = 0; k < 4; ++k) if (col + k < c) dparam[col + k] = t0[k];
    const __amdgpu_buffer_rsrc_t rd = rsrc(dy);
#pragma unroll
    for (int i = 0; i < RR; ++i) {
        f4 o;
        for (int k = 0; k < 4; ++k) o[k] = rs[k] * (g[i][k] - t0[k] * (1.0f / 1024.0f) - xh[i][k] * (t1[k] * (1.0f / 1024.0f)));
        if (W == 4) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, o), rd, v0, i * RY * c * 4, 0);
        else for (int k = 0; k < 4; ++k) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o[k]), rd, v0 + 4 * k, i * RY * c * 4, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[4] = wall_clock64();
}
template <int W>
void run(int c) {
    const int rows = 1024 + 8;
    float *dz, *y, *dy, *mean, *rstd, *beta, *dparam; long long* st;
    hipMalloc(&dz, rows * c * 4); hipMalloc(&y, rows * c * 4); hipMalloc(&dy, rows * c * 4);
    hipMalloc(&mean, c * 4); hipMalloc(&rstd, c * 4); hipMalloc(&beta, c * 4); hipMalloc(&dparam, c * 4 + 64); hipMalloc(&st, 256);
    hipMemset(dz, 0, rows * c * 4); hipMemset(y, 0, rows * c * 4); hipMemset(mean, 0, c * 4); hipMemset(rstd, 0, c * 4); hipMemset(beta, 0, c * 4);
    long long h[32]; float ms = 0;
    for (int rep = 0; rep < 4; ++rep) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        for (int k = 0; k < 20; ++k)
            hipLaunchKernelGGL(probe_bwd_v<W>, dim3((c + 31) / 32), dim3(1024), 0, 0, dz, y, c, mean, rstd, beta, 0.18f, dy, dparam, st);
        hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
    }
    hipMemcpy(h, st, 192, hipMemcpyDeviceToHost);
    printf("W=%d c=%4d  %.2f us/launch  in-kernel: to sums done %.2f  total %.2f   wave load completion:", W, c, ms * 1e3 / 20, (h[3] - h[0]) / 100.0, (h[4] - h[0]) / 100.0);
    for (int w = 0; w < 16; w += 3) printf(" %.2f", (h[8 + w] - h[0]) / 100.0);
    printf("\n");
}
int main() {
    for (int c : {16, 15, 108, 326, 980, 405}) { run<4>(c); run<1>(c); }
    return 0;
}
