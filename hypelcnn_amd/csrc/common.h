// Shared helpers for libhypel_hip.so (gfx950 only; no portability layers).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "hypel.h"

void hypel_set_error(const char* fmt, ...);

#define HYPEL_CHECK_LAUNCH(name)                                              \
    do {                                                                      \
        hipError_t e__ = hipGetLastError();                                   \
        if (e__ != hipSuccess) {                                              \
            hypel_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return -2;                                                        \
        }                                                                     \
    } while (0)

#define HYPEL_REQUIRE(cond, name)                                   \
    do {                                                            \
        if (!(cond)) {                                              \
            hypel_set_error("%s: invalid argument: %s", name, #cond); \
            return -1;                                              \
        }                                                           \
    } while (0)

static inline int hypel_grid_1d(int64_t work_items, int block, int max_blocks = 256 * 8) {
    int64_t g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (int)g;
}

__device__ __forceinline__ float hypel_act(float v, int act, float alpha) {
    switch (act) {
        case HYPEL_ACT_LRELU: return v > 0.0f ? v : v * alpha;
        case HYPEL_ACT_RELU: return v > 0.0f ? v : 0.0f;
        case HYPEL_ACT_SIGMOID: return 1.0f / (1.0f + __expf(-v));
        case HYPEL_ACT_TANH: return tanhf(v);
        default: return v;
    }
}

// Batch-norm pre-activation with separately rounded steps -- xhat = (y - mean) * rstd, pre = xhat + beta -- in EVERY
// kernel that evaluates it.  Forward and backward must take the same leaky-ReLU branch for an element that lands
// within rounding of the kink (and the parity harness must be able to recompute that decision from the buffers);
// left to itself hipcc contracts the expression into an fma in some kernels and not in others.
// (__fmul_rn / __fadd_rn do not stop it: in HIP-Clang they are plain operators; the fp-contract pragma does.)
__device__ __forceinline__ float hypel_bn_xhat(float y, float mean, float rstd) {
#pragma clang fp contract(off)
    const float d = y - mean;
    return d * rstd;
}
__device__ __forceinline__ float hypel_bn_pre(float xhat, float beta) {
#pragma clang fp contract(off)
    return xhat + beta;
}

// derivative of the activation evaluated from its INPUT v
__device__ __forceinline__ float hypel_act_grad(float v, int act, float alpha) {
    switch (act) {
        case HYPEL_ACT_LRELU: return v > 0.0f ? 1.0f : alpha;
        case HYPEL_ACT_RELU: return v > 0.0f ? 1.0f : 0.0f;
        case HYPEL_ACT_SIGMOID: {
            float s = 1.0f / (1.0f + __expf(-v));
            return s * (1.0f - s);
        }
        case HYPEL_ACT_TANH: {
            float t = tanhf(v);
            return 1.0f - t * t;
        }
        default: return 1.0f;
    }
}
