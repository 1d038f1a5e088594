#include <hip/hip_runtime.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
// 16-byte buffer_load ... lds issued from inline assembly: invisible to hipcc's s_waitcnt insertion
__device__ __forceinline__ void dma16(const void* base, int span, int voff, __attribute__((address_space(3))) float* dst) {
    const unsigned long long b = (unsigned long long)base;
    i32x4 rs;
    rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    rs[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(b >> 32) & 0xffffu));
    rs[2] = __builtin_amdgcn_readfirstlane(span);
    rs[3] = 0x00020000;
    const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)dst);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" :: "v"(voff), "s"(rs), "s"(l) : "memory");
}
__global__ void k(const float* g, float* out, int n) {
    __shared__ __attribute__((aligned(16))) float l[512];
    for (int i = threadIdx.x; i < 512; i += 64) l[i] = -7.0f;
    __syncthreads();
    dma16(g, n * 4, (threadIdx.x ^ 5) * 16, (__attribute__((address_space(3))) float*)l);
    dma16(g, n * 4, threadIdx.x * 16, (__attribute__((address_space(3))) float*)(l + 256));
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = l[i];
}
int main() {
    float h[256], r[512]; for (int i = 0; i < 256; ++i) h[i] = i;
    float *g, *o; hipMalloc(&g, 1024); hipMalloc(&o, 2048); hipMemcpy(g, h, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, g, o, 160);  // 160 floats = 640 bytes valid
    hipDeviceSynchronize(); hipMemcpy(r, o, 2048, hipMemcpyDeviceToHost);
    int okp = 0, oki = 0, z = 0;
    for (int lane = 0; lane < 64; ++lane) {
        int sg = lane ^ 5;
        if (sg * 16 < 640 && r[lane * 4] == (float)(sg * 4)) okp++;
        if (lane * 16 < 640 && r[256 + lane * 4] == (float)(lane * 4)) oki++;
        if (sg * 16 >= 640 && r[lane * 4] == 0.0f) z++;
    }
    printf("asm-issued DMA: permuted placed right for %d / 40 lanes, identity for %d / 40, out-of-range zeros %d / 24\n", okp, oki, z);
    return 0;
}
