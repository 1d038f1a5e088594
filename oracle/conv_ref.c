/* ORACLE (test infrastructure; never linked into the product).
 *
 * Literal-definition C loops for the operators the reference's hot path calls into
 * TensorFlow for.  They exist to (a) check the faster numpy restatement in oracle/ops.py
 * and (b) serve as the scalar/OpenMP "port" leg of bench.py's cpu_baseline.
 * Parity unpinned at the TensorFlow boundary: semantics follow
 * tf_slim 1.1.0 / TF 2.9.
 *
 *   conv2d  : tf_slim.conv2d, NHWC x HWIO, stride 1, SAME zero padding
 *             (nnmodel/HYPELCNNModel.py:136,157,177; DUALCNNModel.py:99; CONCNNModel.py:33-35)
 *   conv1d  : tf_slim.convolution1d SAME (gan/shadow_data_models.py:62-86)
 *   lrn     : tf.nn.local_response_normalization defaults (nnmodel/CONCNNModel.py:37,41)
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define IDX4(n, y, x, c, H, W, C) ((((size_t)(n) * (H) + (y)) * (W) + (x)) * (C) + (c))

/* y[N,H,W,Co] = conv(x[N,H,W,Ci], w[kh,kw,Ci,Co]) (+ b).  pad_before = (k-1)/2. */
void orc_conv2d_same_fwd_f32(const float* x, const float* w, const float* b, float* y, int N, int H, int W,
                             int Ci, int Co, int kh, int kw) {
    const int pt = (kh - 1) / 2, pl = (kw - 1) / 2;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int oy = 0; oy < H; ++oy)
            for (int ox = 0; ox < W; ++ox) {
                float* yo = y + IDX4(n, oy, ox, 0, H, W, Co);
                for (int co = 0; co < Co; ++co) yo[co] = b ? b[co] : 0.0f;
                for (int i = 0; i < kh; ++i) {
                    const int iy = oy + i - pt;
                    if (iy < 0 || iy >= H) continue;
                    for (int j = 0; j < kw; ++j) {
                        const int ix = ox + j - pl;
                        if (ix < 0 || ix >= W) continue;
                        const float* xi = x + IDX4(n, iy, ix, 0, H, W, Ci);
                        const float* wt = w + ((size_t)(i * kw + j) * Ci) * Co;
                        for (int ci = 0; ci < Ci; ++ci) {
                            const float xv = xi[ci];
                            const float* wr = wt + (size_t)ci * Co;
                            for (int co = 0; co < Co; ++co) yo[co] += xv * wr[co];
                        }
                    }
                }
            }
}

/* dx[N,H,W,Ci] = sum over taps of dy shifted times w^T. */
void orc_conv2d_same_bwd_input_f32(const float* dy, const float* w, float* dx, int N, int H, int W, int Ci, int Co,
                                   int kh, int kw) {
    const int pt = (kh - 1) / 2, pl = (kw - 1) / 2;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int iy = 0; iy < H; ++iy)
            for (int ix = 0; ix < W; ++ix) {
                float* dxo = dx + IDX4(n, iy, ix, 0, H, W, Ci);
                for (int ci = 0; ci < Ci; ++ci) dxo[ci] = 0.0f;
                for (int i = 0; i < kh; ++i) {
                    const int oy = iy - (i - pt);
                    if (oy < 0 || oy >= H) continue;
                    for (int j = 0; j < kw; ++j) {
                        const int ox = ix - (j - pl);
                        if (ox < 0 || ox >= W) continue;
                        const float* g = dy + IDX4(n, oy, ox, 0, H, W, Co);
                        const float* wt = w + ((size_t)(i * kw + j) * Ci) * Co;
                        for (int ci = 0; ci < Ci; ++ci) {
                            const float* wr = wt + (size_t)ci * Co;
                            float acc = 0.0f;
                            for (int co = 0; co < Co; ++co) acc += g[co] * wr[co];
                            dxo[ci] += acc;
                        }
                    }
                }
            }
}

/* dw[kh,kw,Ci,Co] = sum_n,pixels x^T dy ; db[Co] optional. */
void orc_conv2d_same_bwd_filter_f32(const float* x, const float* dy, float* dw, float* db, int N, int H, int W,
                                    int Ci, int Co, int kh, int kw) {
    const int pt = (kh - 1) / 2, pl = (kw - 1) / 2;
#pragma omp parallel for collapse(2) schedule(dynamic)
    for (int i = 0; i < kh; ++i)
        for (int j = 0; j < kw; ++j) {
            float* wt = dw + ((size_t)(i * kw + j) * Ci) * Co;
            memset(wt, 0, sizeof(float) * (size_t)Ci * Co);
            for (int n = 0; n < N; ++n)
                for (int oy = 0; oy < H; ++oy) {
                    const int iy = oy + i - pt;
                    if (iy < 0 || iy >= H) continue;
                    for (int ox = 0; ox < W; ++ox) {
                        const int ix = ox + j - pl;
                        if (ix < 0 || ix >= W) continue;
                        const float* xi = x + IDX4(n, iy, ix, 0, H, W, Ci);
                        const float* g = dy + IDX4(n, oy, ox, 0, H, W, Co);
                        for (int ci = 0; ci < Ci; ++ci) {
                            const float xv = xi[ci];
                            float* wr = wt + (size_t)ci * Co;
                            for (int co = 0; co < Co; ++co) wr[co] += xv * g[co];
                        }
                    }
                }
        }
    if (db) {
        for (int co = 0; co < Co; ++co) db[co] = 0.0f;
        const size_t rows = (size_t)N * H * W;
        for (size_t r = 0; r < rows; ++r)
            for (int co = 0; co < Co; ++co) db[co] += dy[r * Co + co];
    }
}

/* x[N,L,Ci], w[k,Ci,Co] -> y[N,L,Co]; SAME: pad_left = (k-1)/2 (even k pads the extra on the right). */
void orc_conv1d_same_fwd_f32(const float* x, const float* w, const float* b, float* y, int N, int L, int Ci, int Co,
                             int k) {
    const int pl = (k - 1) / 2;
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n)
        for (int o = 0; o < L; ++o) {
            float* yo = y + ((size_t)n * L + o) * Co;
            for (int co = 0; co < Co; ++co) yo[co] = b ? b[co] : 0.0f;
            for (int j = 0; j < k; ++j) {
                const int i = o + j - pl;
                if (i < 0 || i >= L) continue;
                const float* xi = x + ((size_t)n * L + i) * Ci;
                for (int ci = 0; ci < Ci; ++ci)
                    for (int co = 0; co < Co; ++co) yo[co] += xi[ci] * w[((size_t)j * Ci + ci) * Co + co];
            }
        }
}

/* y_i = x_i / (bias + alpha * sum_{|j-i|<=r} x_j^2)^beta over the last axis. */
void orc_lrn_fwd_f32(const float* x, float* y, size_t rows, int C, int radius, float bias, float alpha, float beta) {
#pragma omp parallel for schedule(static)
    for (long long r = 0; r < (long long)rows; ++r) {
        const float* xr = x + (size_t)r * C;
        float* yr = y + (size_t)r * C;
        for (int i = 0; i < C; ++i) {
            int lo = i - radius < 0 ? 0 : i - radius, hi = i + radius >= C ? C - 1 : i + radius;
            float s = 0.0f;
            for (int j = lo; j <= hi; ++j) s += xr[j] * xr[j];
            yr[i] = xr[i] * powf(bias + alpha * s, -beta);
        }
    }
}

int orc_version(void) { return 1; }
