// gfx950: 16-byte buffer_load ... lds (LDS-DMA), the facts a plane-staging GEMM loop would rest on.  Standalone:
//   hipcc --offload-arch=gfx950 -O3 -o lds_dma16_probe lds_dma16_probe.hip && ./lds_dma16_probe
// 1. placement: where do the 64 lanes' 16 bytes land (expected: M0 base + 16 * lane, whatever the per-lane SOURCE offset is)
// 2. an out-of-range lane (beyond the descriptor's num_records): zeros written, or the slot left untouched?
// 3. two transfers into different LDS bases inside one wave (M0 handling by the compiler)
// 4. cost: a block streaming a buffer global -> LDS by DMA vs global -> VGPR -> ds_write_b128, same bytes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))

__global__ void place(const float* g, int valid_bytes, float* out) {  // one wave
    __shared__ __attribute__((aligned(16))) float l[2 * 256];
    for (int i = threadIdx.x; i < 512; i += 64) l[i] = -7.0f;  // sentinel
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, valid_bytes, 0x00020000);
    const int lane = threadIdx.x;
    const int src = ((lane ^ 5) * 16);  // permuted source granule: lane i fetches granule i ^ 5
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDSP(l), 16, src, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDSP(l + 256), 16, lane * 16, 0, 0, 0);  // second base, identity
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = l[i];
}

template <bool DMA>
__global__ __launch_bounds__(256) void stream(const float* g, long n_f4, float* sink, int iters) {
    __shared__ __attribute__((aligned(16))) float l[4 * 1024 * 4];  // 64 KB: four 16 KB stages
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, 0x7ffffff0, 0x00020000);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float acc = 0.0f;
    long base = (long)blockIdx.x * 1024;  // f4 index: each block step moves 1024 f4 = 16 KB
    const long stride = (long)gridDim.x * 1024;
    for (int it = 0; it < iters; ++it, base += stride) {
        float* st = l + (it & 3) * 4096;
        const long b = base % n_f4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {  // 4 x (256 threads x 16 B) = 16 KB per block step
            const int idx = j * 256 + threadIdx.x;
            if (DMA) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDSP(st + (j * 256 + wave * 64) * 4), 16, (int)((b + idx) * 16 & 0x7fffffff), 0, 0, 0);
            } else {
                const f4 v = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((b + idx) * 16 & 0x7fffffff), 0, 0));
                *reinterpret_cast<f4*>(st + idx * 4) = v;
            }
        }
        if ((it & 3) == 3) {  // consume a little so nothing is dead: one read per thread per 4 steps
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            acc += l[(threadIdx.x * 17 + it) & 16383];
            __syncthreads();
        }
    }
    (void)lane;
    if (acc == 123.456f) sink[0] = acc;
}

int main() {
    const int N = 64 * 4;
    std::vector<float> h(N);
    for (int i = 0; i < N; ++i) h[i] = (float)i;
    float *g, *o;
    CK(hipMalloc(&g, N * 4)); CK(hipMalloc(&o, 512 * 4));
    CK(hipMemcpy(g, h.data(), N * 4, hipMemcpyHostToDevice));
    std::vector<float> r(512);
    for (int valid : {N * 4, 40 * 16}) {  // all lanes in range; then only source granules 0..39 in range
        hipLaunchKernelGGL(place, dim3(1), dim3(64), 0, 0, g, valid, o);
        CK(hipDeviceSynchronize()); CK(hipMemcpy(r.data(), o, 512 * 4, hipMemcpyDeviceToHost));
        int ok_perm = 0, ok_id = 0, zero = 0, sentinel = 0;
        for (int lane = 0; lane < 64; ++lane) {
            const int sg = lane ^ 5;
            const bool inr = sg * 16 < valid, inr2 = lane * 16 < valid;
            const float a = r[lane * 4], b = r[256 + lane * 4];
            if (inr && a == (float)(sg * 4)) ok_perm++;
            if (inr2 && b == (float)(lane * 4)) ok_id++;
            if (!inr) { if (a == 0.0f) zero++; else if (a == -7.0f) sentinel++; }
        }
        printf("valid %4d bytes: permuted-source transfer placed at 16*lane for %d lanes, second base (identity) right for %d lanes; "
               "out-of-range lanes: %d wrote ZERO, %d left the sentinel\n", valid, ok_perm, ok_id, zero, sentinel);
    }
    // cost
    const long n_f4 = 64l << 20 >> 4;  // 64 MB buffer (L2 / MALL resident after the first pass)
    float *big, *sink;
    CK(hipMalloc(&big, n_f4 * 16)); CK(hipMalloc(&sink, 4)); CK(hipMemset(big, 0, n_f4 * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int dma = 0; dma < 2; ++dma)
        for (int rep = 0; rep < 3; ++rep) {
            const int iters = 512, blocks = 512;
            CK(hipEventRecord(e0));
            if (dma) hipLaunchKernelGGL(stream<true>, dim3(blocks), dim3(256), 0, 0, big, n_f4, sink, iters);
            else hipLaunchKernelGGL(stream<false>, dim3(blocks), dim3(256), 0, 0, big, n_f4, sink, iters);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double bytes = (double)blocks * iters * 16384;
            if (rep) printf("%s: %.1f us for %.1f MB into LDS = %.0f GB/s\n", dma ? "global -> LDS by DMA (dwordx4 lds)     " : "global -> VGPR -> ds_write_b128       ", ms * 1e3, bytes / 1e6, bytes / ms / 1e6);
        }
    return 0;
}
