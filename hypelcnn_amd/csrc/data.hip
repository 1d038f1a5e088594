// Data-side kernels of the hot path: everything between "the scene is resident in HBM" and "a batch of
// [N,P,P,C] patches enters the first GEMM", plus the scatter of predictions into the label raster.
//   * gather_patches   : BasicDataSet.get_data_point for a whole batch of targets (common_nn_ops.py:169-185,
//                        InMemoryImporter.py:27-38, GeneratorImporter) -- one launch instead of a Python loop
//   * augment_patches  : the training iterator's map stage (common_nn_ops.py:376-440) fused into the batch
//                        gather: index_select + rot90^k + shadow (per-band ratio, or a pre-computed generator
//                        output) + left/right and up/down flips + per-channel spectral shift, one pass
//   * argmax_scatter   : perform_prediction (common_nn_ops.py:313-327): argmax of the logits written straight
//                        into the uint8 label raster at the target's (x, y)
// All three are pure HBM streaming: one read and one write of every patch element, channel-contiguous so that
// consecutive lanes touch consecutive addresses (C >= 49 floats per pixel in every configuration).
#include "common.h"

#define ST ((hipStream_t)stream)

namespace {

// one block row of 256 lanes walks (sample, pixel) pairs; lanes run over the channels of that pixel
__global__ void gather_patches_kernel(const float* __restrict__ casi, const float* __restrict__ lidar, int64_t wp,
                                      int cc, int cl, const int32_t* __restrict__ points, int64_t n, int p,
                                      float* __restrict__ out) {
    const int c = cc + cl;
    const int npix = p * p;
    const int64_t total = n * npix;
    for (int64_t item = blockIdx.x; item < total; item += gridDim.x) {
        const int64_t s = item / npix;
        const int pix = (int)(item - s * npix);
        const int py = pix / p, px = pix - py * p;
        const int64_t x0 = points[2 * s], y0 = points[2 * s + 1];
        const int64_t src = (y0 + py) * wp + (x0 + px);
        float* o = out + item * c;
        const float* a = casi + src * cc;
        for (int ch = threadIdx.x; ch < cc; ch += blockDim.x) o[ch] = a[ch];
        if (cl > 0) {
            const float* l = lidar + src * cl;
            for (int ch = threadIdx.x; ch < cl; ch += blockDim.x) o[cc + ch] = l[ch];
        }
    }
}

// GRSS2018: the hyperspectral raster has HALF the resolution of the LiDAR raster the targets are given in
// (loader/GRSS2018DataLoader.py:12-44): patch pixel (py, px) takes its spectrum from
// casi[sy + py/2][sx + px/2] with s = coordinate/2 + nb - nb/2 (both rasters are padded by nb) and its height from
// lidar[y + py][x + px].
__global__ void gather_patches_2x_kernel(const float* __restrict__ casi, const float* __restrict__ lidar,
                                         int64_t casi_wp, int64_t lidar_wp, int cc, int cl, int nb,
                                         const int32_t* __restrict__ points, int64_t n, int p,
                                         float* __restrict__ out) {
    const int c = cc + cl;
    const int npix = p * p;
    const int64_t total = n * npix;
    for (int64_t item = blockIdx.x; item < total; item += gridDim.x) {
        const int64_t s = item / npix;
        const int pix = (int)(item - s * npix);
        const int py = pix / p, px = pix - py * p;
        const int64_t x0 = points[2 * s], y0 = points[2 * s + 1];
        const int64_t sx = (x0 >> 1) + nb - (nb >> 1), sy = (y0 >> 1) + nb - (nb >> 1);
        const float* a = casi + ((sy + (py >> 1)) * casi_wp + sx + (px >> 1)) * cc;
        float* o = out + item * c;
        for (int ch = threadIdx.x; ch < cc; ch += blockDim.x) o[ch] = a[ch];
        const float* l = lidar + ((y0 + py) * lidar_wp + x0 + px) * cl;
        for (int ch = threadIdx.x; ch < cl; ch += blockDim.x) o[cc + ch] = l[ch];
    }
}

__global__ void augment_patches_kernel(const float* __restrict__ x, const int64_t* __restrict__ idx, int64_t n, int p,
                                       int c, const int32_t* __restrict__ rot_k,
                                       const uint8_t* __restrict__ shadow_pick, const float* __restrict__ shadow_ratio,
                                       const float* __restrict__ shadow_alt, const uint8_t* __restrict__ flip_lr,
                                       const uint8_t* __restrict__ flip_ud, const float* __restrict__ delta,
                                       float* __restrict__ out) {
    const int npix = p * p;
    const int64_t total = n * npix;
    for (int64_t item = blockIdx.x; item < total; item += gridDim.x) {
        const int64_t s = item / npix;
        const int pix = (int)(item - s * npix);
        int i = pix / p, j = pix - i * p;
        // undo the maps in reverse order of application: up/down flip, left/right flip, then the rotation
        if (flip_ud && flip_ud[s]) i = p - 1 - i;
        if (flip_lr && flip_lr[s]) j = p - 1 - j;
        const int k = rot_k ? rot_k[s] : 0;
        int si = i, sj = j;
        if (k == 1) {  // counter-clockwise quarter turn: out[i][j] = in[j][P-1-i]
            si = j;
            sj = p - 1 - i;
        } else if (k == 2) {
            si = p - 1 - i;
            sj = p - 1 - j;
        } else if (k == 3) {
            si = p - 1 - j;
            sj = i;
        }
        const bool shade = shadow_pick && shadow_pick[s];
        const int64_t src_s = idx ? idx[s] : s;
        // the shadow operators act per pixel spectrum, so they commute with the spatial maps
        const float* src = (shade && shadow_alt) ? shadow_alt + (s * npix + si * p + sj) * c
                                                 : x + (src_s * npix + si * p + sj) * c;
        float* o = out + item * c;
        const float* d = delta ? delta + s * c : nullptr;
        for (int ch = threadIdx.x; ch < c; ch += blockDim.x) {
            float v = src[ch];
            if (shade && shadow_ratio) v = v / shadow_ratio[ch];
            if (d) v = v + d[ch];
            o[ch] = v;
        }
    }
}

__global__ void argmax_scatter_kernel(const float* __restrict__ logits, int64_t ld, int64_t n, int c,
                                      const int32_t* __restrict__ points, uint8_t* __restrict__ raster,
                                      int64_t raster_w) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* z = logits + i * ld;
    int best = 0;
    float bv = z[0];
    for (int j = 1; j < c; ++j)
        if (z[j] > bv) {  // first maximum wins, as tf.argmax
            bv = z[j];
            best = j;
        }
    raster[(int64_t)points[2 * i + 1] * raster_w + points[2 * i]] = (uint8_t)best;
}

// One (normal, shadow) spectrum pair per 64-lane row pass: gather + the regulariser swap of
// perform_shadow_augmentation_random (gan/gan_train_for_shadow.py:171-182).
__global__ void gather_pairs_kernel(const float* __restrict__ normal, const float* __restrict__ shadow,
                                    const int64_t* __restrict__ idx, int64_t n, int bands, const float* __restrict__ ratio,
                                    const float* __restrict__ u1, const float* __restrict__ u2, float rate,
                                    float* __restrict__ out_x, float* __restrict__ out_y) {
    for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
        const int64_t src = idx[i];
        const bool swap_x = ratio != nullptr && u1[i] < rate, swap_y = ratio != nullptr && u2[i] < rate;
        for (int b = threadIdx.x; b < bands; b += blockDim.x) {
            float x = normal[src * bands + b];
            float y = shadow[src * bands + b];
            // the reference draws the two decisions independently and builds the second pair from the ALREADY swapped
            // normal spectrum (normal_images_rand / ratio): reproduced
            if (swap_x) x = y * ratio[b];
            if (swap_y) y = x / ratio[b];
            out_x[i * bands + b] = x;
            out_y[i * bands + b] = y;
        }
    }
}

}  // namespace

extern "C" int hypel_gather_pairs_f32(const float* normal, const float* shadow, const int64_t* idx, int64_t n, int32_t bands,
                                      const float* ratio, const float* u1, const float* u2, float rate, float* out_x,
                                      float* out_y, hypel_stream_t stream) {
    HYPEL_REQUIRE(normal && shadow && idx && out_x && out_y && n > 0 && bands > 0, "hypel_gather_pairs_f32");
    HYPEL_REQUIRE(ratio == nullptr || (u1 && u2), "hypel_gather_pairs_f32");
    const int block = bands >= 192 ? 256 : (bands >= 96 ? 128 : 64);
    hipLaunchKernelGGL(gather_pairs_kernel, dim3(hypel_grid_1d(n, 1, 256 * 32)), dim3(block), 0, ST, normal, shadow, idx, n,
                       bands, ratio, u1, u2, rate, out_x, out_y);
    HYPEL_CHECK_LAUNCH("hypel_gather_pairs_f32");
    return 0;
}

extern "C" int hypel_gather_patches_f32(const float* casi, const float* lidar, int64_t hp, int64_t wp, int32_t cc,
                                        int32_t cl, const int32_t* points, int64_t n, int32_t p, float* out,
                                        hypel_stream_t stream) {
    HYPEL_REQUIRE(casi && points && out && n > 0 && p > 0 && cc > 0 && cl >= 0 && hp >= p && wp >= p,
                  "hypel_gather_patches_f32");
    HYPEL_REQUIRE(cl == 0 || lidar, "hypel_gather_patches_f32");
    const int c = cc + cl;
    const int block = c >= 192 ? 256 : (c >= 96 ? 128 : 64);
    hipLaunchKernelGGL(gather_patches_kernel, dim3(hypel_grid_1d(n * p * p, 1, 256 * 32)), dim3(block), 0, ST, casi,
                       lidar, wp, cc, cl, points, n, p, out);
    HYPEL_CHECK_LAUNCH("hypel_gather_patches_f32");
    return 0;
}

extern "C" int hypel_gather_patches_2x_f32(const float* casi, const float* lidar, int64_t casi_wp, int64_t lidar_wp,
                                           int32_t cc, int32_t cl, int32_t neighborhood, const int32_t* points,
                                           int64_t n, int32_t p, float* out, hypel_stream_t stream) {
    HYPEL_REQUIRE(casi && lidar && points && out && n > 0 && p > 0 && cc > 0 && cl > 0 && neighborhood >= 0,
                  "hypel_gather_patches_2x_f32");
    const int c = cc + cl;
    const int block = c >= 192 ? 256 : (c >= 96 ? 128 : 64);
    hipLaunchKernelGGL(gather_patches_2x_kernel, dim3(hypel_grid_1d(n * p * p, 1, 256 * 32)), dim3(block), 0, ST, casi,
                       lidar, casi_wp, lidar_wp, cc, cl, neighborhood, points, n, p, out);
    HYPEL_CHECK_LAUNCH("hypel_gather_patches_2x_f32");
    return 0;
}

extern "C" int hypel_augment_patches_f32(const float* x, const int64_t* idx, int64_t n, int32_t p, int32_t c,
                                         const int32_t* rot_k, const uint8_t* shadow_pick, const float* shadow_ratio,
                                         const float* shadow_alt, const uint8_t* flip_lr, const uint8_t* flip_ud,
                                         const float* delta, float* out, hypel_stream_t stream) {
    HYPEL_REQUIRE(x && out && x != out && n > 0 && p > 0 && c > 0, "hypel_augment_patches_f32");
    HYPEL_REQUIRE(!shadow_pick || shadow_ratio || shadow_alt, "hypel_augment_patches_f32");
    const int block = c >= 192 ? 256 : (c >= 96 ? 128 : 64);
    hipLaunchKernelGGL(augment_patches_kernel, dim3(hypel_grid_1d(n * p * p, 1, 256 * 32)), dim3(block), 0, ST, x, idx,
                       n, p, c, rot_k, shadow_pick, shadow_ratio, shadow_alt, flip_lr, flip_ud, delta, out);
    HYPEL_CHECK_LAUNCH("hypel_augment_patches_f32");
    return 0;
}

extern "C" int hypel_argmax_scatter(const float* logits, int64_t ld, int64_t n, int32_t c, const int32_t* points,
                                    uint8_t* raster, int64_t raster_w, hypel_stream_t stream) {
    HYPEL_REQUIRE(logits && points && raster && n > 0 && c > 0 && c <= 256 && raster_w > 0, "hypel_argmax_scatter");
    hipLaunchKernelGGL(argmax_scatter_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, ST, logits, ld, n, c,
                       points, raster, raster_w);
    HYPEL_CHECK_LAUNCH("hypel_argmax_scatter");
    return 0;
}
