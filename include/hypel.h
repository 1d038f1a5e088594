/* hypel.h -- C-ABI of libhypel_hip.so: the MI355X (gfx950) compute path behind the
 * NNModel / GAN-wrapper plugin API of aligokalppeker/hypelcnn.
 *
 * The reference has NO native layer: every entry point below replaces a call the reference
 * makes into TensorFlow / tf_slim kernels (reference file:line cited per function).  The
 * binding a maintainer would add on the reference side is a ctypes stub.
 *
 * Conventions
 *  - plain C types only; every pointer is a DEVICE pointer unless it says "host".
 *  - all work is enqueued asynchronously on `stream` (a hipStream_t); no implicit syncs,
 *    no allocation; the caller (PyTorch caching allocator on the Python side) owns every
 *    buffer including workspaces.
 *  - return 0 on success, negative on error; message via hypel_last_error() (thread-local).
 *  - activations are row-major matrices [rows, C] with a leading dimension `ld` (floats).
 *    Internally the host keeps patch tensors PIXEL-MAJOR: [P = H*W][N][C], i.e. the rows
 *    of pixel p for the whole minibatch are contiguous.  That turns a SAME convolution
 *    into a sum over VALID taps of plain GEMMs on contiguous row blocks (no im2col, no
 *    padding taps): see hypel_seg_gemm_f32.  hypel_nhwc_to_pnc converts the loader's
 *    NHWC batches (importer/InMemoryImporter.py:27-38).
 *  - weights keep the TF variable layouts: conv HWIO [kh][kw][Cin][Cout], FC [in][out],
 *    BN vectors [C] -- checkpoints keyed by TF names map 1:1.
 */
#ifndef HYPEL_H
#define HYPEL_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* hypel_stream_t; /* hipStream_t */

#define HYPEL_ABI_VERSION 6  /* bump whenever a prototype, a struct layout or the meaning of a flag changes */

/* activation codes (leaky_relu: HYPELCNNModel.py:39, DUALCNNModel.py:18, shadow_data_models.py:53;
 * relu: tf_slim default, CONCNNModel.py; sigmoid: HYPELCNNModel.py:93; tanh: shadow_data_models.py:86) */
enum { HYPEL_ACT_NONE = 0, HYPEL_ACT_LRELU = 1, HYPEL_ACT_RELU = 2, HYPEL_ACT_SIGMOID = 3, HYPEL_ACT_TANH = 4 };

int hypel_version(void);
const char* hypel_last_error(void);
/* number of compute units / XCDs of the current device (host query, for table building) */
int hypel_device_info(int32_t* n_cu, int32_t* n_xcd);

/* host helper: CRC-32C (Castagnoli) of `n` bytes continued from `crc` (0 to start); used by the TensorFlow checkpoint
 * bundle reader/writer (monitored_session_runner.py:164-171 saves, cycle_gan_wrapper.py:140-147 restores such files) */
uint32_t hypel_crc32c(uint32_t crc, const void* data, uint64_t n);

/* ---- layout ------------------------------------------------------------------------------- */
/* x[N][P][C] (NHWC with P=H*W) -> out[P][N][ld] (pad columns zeroed).  Replaces the
 * feed_dict/prefetch_to_device hand-over (InMemoryImporter.py:80-83, common_nn_ops.py:200). */
int hypel_nhwc_to_pnc(const float* x, float* out, int64_t n, int32_t p, int32_t c, int64_t ld,
                      hypel_stream_t stream);
int hypel_pnc_to_nhwc(const float* in, int64_t ld, float* x, int64_t n, int32_t p, int32_t c,
                      hypel_stream_t stream);
int hypel_fill_f32(float* dst, int64_t count, float value, hypel_stream_t stream);

/* ---- grouped multi-segment GEMM (fp32 MFMA, 32x32x2) ------------------------------------------
 * For every group g:  C_g[rows_g x n] (+)= sum_{s in segs(g)} op(A_s)[rows_g x k_s] * op(B_s)[k_s x n] (+ bias)
 *   trans_a = 0: op(A_s)[i][kk] = A[a_off + i*lda + kk]      trans_a = 1: A[a_off + kk*lda + i]
 *   trans_b = 0: op(B_s)[kk][j] = B[b_off + kk*ldb + j]      trans_b = 1: B[b_off + j*ldb + kk]
 *   C_g[i][j] = C[c_off + i*ldc + j]
 * One kernel covers: tf_slim.conv2d 1x1 and kxk SAME as exact-tap sums (HYPELCNNModel.py:136,
 * 157,177; DUALCNNModel.py:99; CONCNNModel.py:33-60), tf_slim.fully_connected incl. the
 * flatten that precedes it (HYPELCNNModel.py:74-94,121; DUALCNNModel.py:31,48-54), and their
 * data-gradient (trans_b=1) and filter-gradient (trans_a=1, split over rows) passes that
 * tf.gradients derives inside create_train_op (common_nn_ops.py:232).
 * bias (optional) is indexed by the ABSOLUTE output column, (c_off mod ldc) + j, so the branches of a
 * merged multi-kernel level (groups starting at different channel offsets) share one launch.
 * `tiles` lists every 128-row output tile: its group and first row, plus a copy of what the kernel needs to start the
 * tile (the group's rows / segment range / c_off and the first segment), so that a block reads ONE record before its
 * first operand loads instead of chasing tiles -> groups -> segs; tables live on the device. */
typedef struct { int64_t a_off; int64_t b_off; int32_t k; int32_t reserved; } hypel_seg_t;
typedef struct { int64_t c_off; int32_t seg_begin; int32_t seg_count; int32_t rows; int32_t reserved; } hypel_group_t;
typedef struct {
    int32_t group; int32_t m0;                         /* output rows [m0, min(m0 + 128, rows)) of groups[group] */
    int32_t rows; int32_t seg_begin; int32_t seg_count; /* copies of groups[group] */
    int32_t k0; int64_t c_off; int64_t a_off0; int64_t b_off0; /* c_off copy; segs[seg_begin] copy (0 if none) */
    int32_t flags; int32_t n;                          /* HYPEL_TILE_*; n > 0: THIS tile's group has n output columns (<= the launch's n) */
} hypel_tile_t;
#define HYPEL_GEMM_BM 128
/* K-slice records (round 6; tail splitting for the 512-slot split-operand kernels): the segment list of a heavy output tile
 * -- or of the last, partly filled round of a launch -- is cut into slices that run as blocks of their own.  Slice 0 is an
 * ordinary record of the launch (bias, accumulate, shortcut gather apply to it); the other slices carry HYPEL_TILE_PLAIN:
 * their block writes its partial sum to its own c_off (a scratch region, addressed relative to `c` like every c_off)
 * and IGNORES the launch's bias / accumulate bit / shortcut operands.  The caller adds the partials to the output in a
 * fixed order afterwards (hypel_reduce_splits_multi_f32, accumulate flag set): deterministic, no atomics.  Not valid in
 * hypel_seg_gemm_stats_f32 launches and with HYPEL_GEMM_ACT_* (their epilogues need the whole sum). */
#define HYPEL_TILE_PLAIN 1
/* Short segments (data gradients through convolutions with <= 16 filters: K = 15 in the narrowest HYPELCNN level):
 * a segment whose `k` has HYPEL_SEG_PAIR_FLAG set (real k = k & ~flag, <= 16) shares ONE 32-column k-tile with the
 * NEXT segment of its group (k <= 16, flag clear); the tile record's k0 copy carries the flag too.  Launches whose
 * tables contain such pairs pass HYPEL_GEMM_PAIRED_SEGS in `accumulate` (trans_a = 0, trans_b = 1, n > 16 only; the
 * two a_off / b_off of a pair must be less than 2 GB apart).  Results do not depend on pairing. */
#define HYPEL_SEG_PAIR_FLAG 0x40000000
#define HYPEL_GEMM_PAIRED_SEGS 0x400
/* Hint: every group of the launch has exactly ONE segment (a 1x1 convolution or its data gradient).  With 128x32 blocks
 * such launches run on a build of the kernel that keeps 7 instead of 6 blocks resident per CU.  Results do not depend
 * on the hint. */
#define HYPEL_GEMM_SINGLE_SEG 0x800

/* Merged multi-kernel levels (nnmodel/HYPELCNNModel.py:167-183, DUALCNNModel.py:92-104).  The kernel sizes of a level
 * are nested (1 c 3 c 5 c 7 ...): an input offset at ring r = max(|dy|, |dx|) belongs to every branch with
 * k >= 2r + 1, a SUFFIX of the concat order, so its output columns are ONE contiguous range [r * cout, C).  With a
 * packed weight image W_pack[offset][Cin][C] (hypel_copy_blocks_f32 builds it from the HWIO variables) a level is, per
 * output pixel and ring, a plain product on that column range: groups of one launch then differ in their column
 * count, which travels in hypel_tile_t.n (0 = the launch's n; the grid covers the widest group, blocks beyond a
 * narrower group's columns exit).  HYPEL_GEMM_MFMA16X4 (n <= 64, trans_a = trans_b = 0): 128x64 blocks on the 16x16x4
 * MFMA, four 16-column tiles per wave, for levels with <= 16 filters per branch; HYPEL_GEMM_VAR_N: the tile records
 * carry column counts -- 128x64 blocks then skip an accumulator tile that lies outside their group.  Results depend
 * on neither (a launch without them computes the unused tiles on zeros). */
#define HYPEL_GEMM_MFMA16X4 0x2000
#define HYPEL_GEMM_VAR_N 0x4000
/* Activation in the product's epilogue (hypel_seg_gemm_f32, trans_a = trans_b = 0, no accumulate): C = lrelu(product + bias)
 * with the slope the code names -- a tf_slim.fully_connected(activation_fn=leaky_relu) WITHOUT a normaliser in one launch
 * (gan/shadow_data_models.py:95-149: the critics' and feature-discriminator layers, slope 0.1).  The slopes are a closed
 * list so that the constant is exact fp32 (no float travels in `accumulate`).  Backward passes take act' from the sign of
 * the OUTPUT, which a positive slope preserves. */
#define HYPEL_GEMM_ACT_LRELU_0_1 0x10000
#define HYPEL_GEMM_ACT_LRELU_0_18 0x20000
#define HYPEL_GEMM_ACT_LRELU_0_2 0x30000
#define HYPEL_GEMM_ACT_LRELU_0_01 0x40000
/* fp32 products on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, 16x the rate of v_mfma_f32_32x32x2_f32).  Every
 * fp32 operand element is split EXACTLY into three round-to-nearest bf16 parts (x = hi + mid + lo: 8 + 8 + 8 significand
 * bits, fp32's exponent range) on its way into LDS; each k-step issues the six partial products hi hi, hi mid, mid hi,
 * hi lo, lo hi, mid mid (every one exact in the fp32 accumulator) and drops mid lo, lo mid, lo lo, which are below
 * 2^-24 |a b| -- the rounding unit of the fp32 accumulate itself.  Inputs, outputs, tables and epilogues (bias,
 * accumulate, shortcut gather, statistics) are those of the plain launch; the result is as accurate as the fp32 MFMA
 * chain (tests/test_gpu_kernels.py::test_seg_gemm_split6_error_vs_fp32_chain measures max |c - fp64| / sum |a b| of
 * both on every launch shape of the benchmark step) but not bit-identical to it.  Plain products only (no paired
 * segments, MFMA16X4, ACT; HYPEL_GEMM_VAR_N is fine), n > 16, not trans_a = trans_b = 1.  With this flag the tile-width hint reads
 * 1 = 128x32, 2 = 128x64, 3 = 128x128 blocks.  hypel_seg_gemm_multi_f32: OR HYPEL_GEMM_MULTI_SPLIT6 into tile_width
 * (32, 64 or 128 then).
 * Limits of the claim (tests/test_gpu_kernels.py::test_seg_gemm_split6_hard_operands, ..._nonfinite_and_extreme_operands):
 *  - non-finite operands: an Inf splits into hi = Inf, mid = Inf - Inf = NaN, so a product the fp32 chain evaluates to
 *    +-Inf comes out NaN here (a NaN operand gives NaN on both paths).  Non-finite stays non-finite -- the loss guard
 *    (hypel_loss_guard, NanTensorHook's counterpart) fires either way -- but Inf is not distinguished from NaN;
 *  - finite |x| >= 2^128 - 2^119 (above bf16's largest finite value, 3.3895e38: the top 0.4 % of fp32's last binade)
 *    rounds hi to Inf and gives NaN where the fp32 chain would give a huge finite number or Inf;
 *  - |x| < 2^-110: the lo (then the mid) part is a bf16 subnormal, which the matrix cores flush to zero -- the operand
 *    keeps 16 (then 8) significand bits instead of 24; products of such operands are below fp32's own subnormal range
 *    against any operand of ordinary magnitude;
 *  - from 2^-110 up to bf16's maximum the result stays within a few 2^-24 of sum |a b| whatever the mix of signs and
 *    magnitudes in a row: cancellation does not hurt (the partial products are exact), and where ONE product dominates a
 *    row's sum (operands spanning tens of binades) its partial products enter the accumulator as three non-negligible
 *    additions instead of the chain's one -- measured 1.5 x the fp32 chain's error there (7.3e-7 vs 4.7e-7 of sum |a b|),
 *    bounded by 3 x. */
#define HYPEL_GEMM_SPLIT6 0x8000
#define HYPEL_GEMM_MULTI_SPLIT6 0x100

/* `accumulate`: bit 0 = add to C instead of overwriting it; bits 8-9 = optional tile-width hint
 * (0 = library heuristic, 1 = 128x32 blocks, 2 = 128x64 blocks, 3 = 128x96 blocks for n > 64) -- results do not
 * depend on it; bit 10 = HYPEL_GEMM_PAIRED_SEGS; bit 11 = HYPEL_GEMM_SINGLE_SEG;
 * bit 13 = HYPEL_GEMM_MFMA16X4; bit 14 = HYPEL_GEMM_VAR_N; bit 15 = HYPEL_GEMM_SPLIT6; bits 16-18 = HYPEL_GEMM_ACT_*. */
int hypel_seg_gemm_f32(const float* a, int64_t lda, int32_t trans_a, const float* b, int64_t ldb, int32_t trans_b,
                       float* c, int64_t ldc, int32_t n, const hypel_group_t* groups, const hypel_seg_t* segs,
                       const hypel_tile_t* tiles, int32_t n_tiles, const float* bias, int32_t accumulate,
                       hypel_stream_t stream);

/* hypel_seg_gemm_f32 that also leaves the batch-norm statistics of its output (tf_slim.batch_norm under the conv /
 * fully_connected arg_scope, nnmodel/HYPELCNNModel.py:37,40-45): for a launch with ONE group (a 1x1 convolution or
 * a dense layer: C is one [rows x n] matrix, ldc == n) every 128-row tile t writes
 * stats_partial[(2t) * n + col] = mean of the tile's valid rows, stats_partial[(2t+1) * n + col] = their sum of
 * squared deviations -- the chunk format of hypel_col_stats_partial with chunk_rows = 128, ready for
 * hypel_bn_finalize(n_chunks = ceil(rows / 128), chunk_rows = 128).  No accumulate. */
int hypel_seg_gemm_stats_f32(const float* a, int64_t lda, int32_t trans_a, const float* b, int64_t ldb,
                             int32_t trans_b, float* c, int64_t ldc, int32_t n, const hypel_group_t* groups,
                             const hypel_seg_t* segs, const hypel_tile_t* tiles, int32_t n_tiles, const float* bias,
                             int32_t accumulate, float* stats_partial, hypel_stream_t stream);

/* hypel_seg_gemm_f32 with the shortcut gradient folded into the epilogue.  For `net = f(conv(net)) + scale_in_to_out(
 * net)` (nnmodel/HYPELCNNModel.py:160-163,176-183) the gradient of `net` is conv-data-gradient + transpose of the
 * channel map applied to dZ; instead of a separate gather pass plus a read-modify-write of the result, output
 * element (row, col) additionally receives sum_{o in [res_start[col], res_start[col+1])} res[row_abs * ldr + o],
 * row_abs = c_off / ldc + m0 + row (res has the row order of C: pixel-major).  res_start NULL = identity map. */
int hypel_seg_gemm_res_f32(const float* a, int64_t lda, int32_t trans_a, const float* b, int64_t ldb, int32_t trans_b,
                           float* c, int64_t ldc, int32_t n, const hypel_group_t* groups, const hypel_seg_t* segs,
                           const hypel_tile_t* tiles, int32_t n_tiles, const float* bias, int32_t accumulate,
                           const float* res, int64_t ldr, const int32_t* res_start, hypel_stream_t stream);

/* SEVERAL products C_p = A_p^T B_p in ONE launch -- the filter gradients of many layers (tf.gradients w.r.t. the
 * `weights` of every tf_slim.conv2d / fully_connected, common/common_nn_ops.py:232), which are mutually independent
 * and individually too small to fill 256 CUs: every launch pays ~25 us of ramp-up and drain, a step has twenty of
 * them.  `blocks` holds one record per 128 x tile_width output BLOCK: its group (rows, segment range, output offset,
 * first segment, as in hypel_tile_t) plus what used to be per-launch arguments -- column tile n0, the product's n,
 * lda, ldb, ldc -- and flags (bit 0 = accumulate into C).  Every offset (records and `segs`) is in elements relative
 * to `base`, one pointer for A, B and C (the operands live in different allocations; differences of device
 * addresses are exact in int64).  tile_width in {16, 32, 64} (| HYPEL_GEMM_MULTI_SPLIT6: {32, 64, 128}, the width of
 * the column tiles n0 the records were built for); trans_a = 1, trans_b = 0 only. */
typedef struct {
    int64_t c_off; int64_t a_off0; int64_t b_off0;      /* output offset of the group; segs[seg_begin] copy */
    int32_t m0; int32_t rows; int32_t n0; int32_t n;      /* output rows [m0, m0+128) x columns [n0, n0+tile_width) */
    int32_t seg_begin; int32_t seg_count; int32_t k0; int32_t flags;
    int32_t lda; int32_t ldb; int32_t ldc; int32_t reserved;
} hypel_mtile_t;
int hypel_seg_gemm_multi_f32(const float* base, int32_t trans_a, int32_t trans_b, int32_t tile_width,
                             const hypel_seg_t* segs, const hypel_mtile_t* blocks, int32_t n_blocks,
                             hypel_stream_t stream);

/* Several hypel_reduce_splits_f32 in one launch (the second stage of the merged filter-gradient launch):
 * entry e: out_e[i] = (flags_e & 1 ? out_e[i] : 0) + sum_{s < n_splits_e} partial_e[s*stride_e + i], i < count_e;
 * partial_off / out_off in elements relative to `base`. */
typedef struct {
    int64_t partial_off; int64_t out_off; int64_t stride; int64_t count; int32_t n_splits; int32_t flags;
} hypel_reduce_entry_t;
int hypel_reduce_splits_multi_f32(const float* base, const hypel_reduce_entry_t* entries, int32_t n_entries,
                                  hypel_stream_t stream);
/* The same reduction when the caller knows max_count = the largest entry's count: the grid is sized for it (the K-slice
 * partials of a GEMM launch, HYPEL_TILE_PLAIN: many entries of one 128-row tile each).  Same sums, same order. */
int hypel_reduce_splits_multi_sized_f32(const float* base, const hypel_reduce_entry_t* entries, int32_t n_entries,
                                        int64_t max_count, hypel_stream_t stream);

/* The same table for reductions with MANY slabs and few outputs (the per-block gradient slabs that the fused
 * generator / dense-stack backward kernels of one GAN train op leave: all of them in one launch): one wavefront per
 * output element, lanes over the slabs; total_count = sum of the entries' counts (host knowledge: sizes the grid).
 * Two entries of one launch must not write the same output. */
int hypel_reduce_splits_wave_multi_f32(const float* base, const hypel_reduce_entry_t* entries, int32_t n_entries,
                                       int64_t total_count, hypel_stream_t stream);

/* out[o(i)] = (accumulate ? out[o(i)] : 0) + (bias ? bias[i mod n] : 0) + sum_s partial[s*stride + o(i)], s ascending
 * (deterministic second stage of every split launch: filter gradients, FC-shaped products whose output has too
 * few tiles to fill 256 CUs, and the tap-split heavy branches of a multi-kernel level).
 * ldc <= 0: o(i) = i (dense).  ldc > 0: o(i) = (i / n) * ldc + i mod n, a [count/n x n] window of a wider matrix. */
int hypel_reduce_splits_f32(const float* partial, int64_t stride, int32_t n_splits, float* out, int64_t count,
                            int32_t accumulate, const float* bias, int32_t n, int64_t ldc, hypel_stream_t stream);

/* Two hypel_reduce_splits_f32 (dense, no bias) over the same n_splits slabs in one launch -- the filter and the bias
 * gradients a fused stack (hypel_gan_generator_bwd, hypel_dense_stack_bwd) leaves per block; bit-identical to the two
 * separate calls. */
int hypel_reduce_splits_pair_f32(const float* partial0, int64_t stride0, int64_t count0, float* out0,
                                 const float* partial1, int64_t stride1, int64_t count1, float* out1, int32_t n_splits,
                                 int32_t accumulate, hypel_stream_t stream);

/* 2-D block copies in one launch: entry e copies (or adds) a [rows x cols] block, dst[dst_off + r * dst_ld + c]
 * (+)= src[src_off + r * src_ld + c]; offsets in elements relative to `base` (one pointer for every operand, as in
 * hypel_seg_gemm_multi_f32), flags bit 0 = accumulate.  Builds the packed weight image of a merged multi-kernel level
 * from its tf_slim.conv2d HWIO variables (one entry per (branch, tap): the [Cin x cout] slice goes to offset d, columns
 * [branch * cout, (branch + 1) * cout) of W_pack) before the forward pass, and scatters the packed filter gradient back
 * into the variables' gradient slots after the backward pass -- TF names and layouts stay at the boundary.  Also gathers
 * the inputs of same-weight GAN applications into one row-concatenated batch (cut_wrapper.py:301-339) and scatters its
 * gradient back.  max_block_elems = rows * cols of the largest entry (host knowledge: sizes the grid). */
typedef struct {
    int64_t src_off; int64_t dst_off; int32_t rows; int32_t cols; int32_t src_ld; int32_t dst_ld; int32_t flags;
    int32_t reserved;
} hypel_copy_block_t;
int hypel_copy_blocks_f32(const float* base, const hypel_copy_block_t* entries, int32_t n_entries,
                          int64_t max_block_elems, hypel_stream_t stream);
/* Two flat copies in one launch, dst_i[0 .. n_i) = src_i[0 .. n_i) (n1 may be 0): the two batches a GAN train op is fed
 * with (x and y, resp. the two tensor-pool results of the critics' phase; gan/wrappers/gan_common.py run_step) -- each
 * copy launch of a 0.28 ms CycleGAN step is 1.7 % of it. */
int hypel_copy_pair_f32(float* dst0, const float* src0, int64_t n0, float* dst1, const float* src1, int64_t n1,
                        hypel_stream_t stream);

/* ---- batch norm statistics (tf_slim.batch_norm fused, HYPELCNNModel.py:37,43-44) ------------------
 * partial[chunk][0][c] = mean of the chunk's rows, partial[chunk][1][c] = sum of squared deviations.
 * chunk = `chunk_rows` consecutive rows (last may be short). */
int hypel_col_stats_partial(const float* x, int64_t ld, int64_t rows, int32_t c, int32_t chunk_rows, float* partial,
                            hypel_stream_t stream);
/* Chan-merge the chunk partials in chunk order (fp64) -> mean[c], rstd[c] = 1/sqrt(var_biased+eps);
 * optional moving-average update m <- m*decay + batch*(1-decay) with the Bessel-corrected variance. */
int hypel_bn_finalize(const float* partial, int32_t n_chunks, int32_t chunk_rows, int64_t rows, int32_t c, float eps,
                      float* mean, float* rstd, float* moving_mean, float* moving_var, float decay,
                      hypel_stream_t stream);
/* Synchronised batch norm across data-parallel ranks (optional; the reference is single-device, so this is what makes
 * N ranks x nb equal the reference's one device at batch N x nb -- tf_slim.batch_norm, HYPELCNNModel.py:37,43):
 * a rank merges its chunk partials into one record out[0..c) = mean, out[c..2c) = sum of squared deviations,
 * out[2c] = rows; the host all-gathers the records (RCCL); hypel_bn_finalize_ranks merges `world` records
 * ([world][2c+1], rank order, fp64) into mean / rstd / moving averages -- identical on every rank. */
int hypel_bn_merge_partials(const float* partial, int32_t n_chunks, int32_t chunk_rows, int64_t rows, int32_t c,
                            float* out, hypel_stream_t stream);
int hypel_bn_finalize_ranks(const float* gathered, int32_t world, int32_t c, float eps, float* mean, float* rstd,
                            float* moving_mean, float* moving_var, float decay, hypel_stream_t stream);
int hypel_bn_act_small_fwd(const float* y, int64_t ldy, int64_t rows, int32_t c, float eps, const float* beta,
                           int32_t act, float alpha, const float* mask, int64_t ldm, float* mean, float* rstd,
                           float* moving_mean, float* moving_var, float decay, float* z, int64_t ldz,
                           hypel_stream_t stream);
int hypel_bn_act_small_bwd(const float* dz, int64_t lddz, const float* y, int64_t ldy, int64_t rows, int32_t c,
                           const float* mean, const float* rstd, const float* beta, int32_t act, float alpha,
                           const float* mask, int64_t ldm, float* dy, int64_t lddy, float* dparam,
                           int32_t accumulate, hypel_stream_t stream);
/* inference: rstd[c] = 1/sqrt(moving_var[c] + eps) */
int hypel_rstd_from_var(const float* var, int32_t c, float eps, float* rstd, hypel_stream_t stream);

/* ---- fused post-op: z = act((y - mean) * rstd + beta) * mask + res1[:, idx1] + res2[:, idx2] ---------
 * mean/rstd/beta NULL -> no normalisation (bias was added by the GEMM).  mask NULL -> no dropout
 * (tf_slim.dropout, HYPELCNNModel.py:123).  idx NULL -> identity channel map; otherwise the
 * scale_in_to_out gather/repeat index vector (common_nn_ops.py:546-564). */
int hypel_bn_act_fwd(const float* y, int64_t ldy, int64_t rows, int32_t c, const float* mean, const float* rstd,
                     const float* beta, int32_t act, float alpha, const float* mask, int64_t ldm, const float* res1,
                     int64_t ld1, const int32_t* idx1, const float* res2, int64_t ld2, const int32_t* idx2, float* z,
                     int64_t ldz, hypel_stream_t stream);
/* backward pass 1: partial[chunk][0][c] = sum dyh, partial[chunk][1][c] = sum dyh*xhat over the chunk rows,
 * dyh = dz*mask*act'(yh).  (xhat := y when there is no normalisation.) */
int hypel_bn_act_bwd_reduce(const float* dz, int64_t lddz, const float* y, int64_t ldy, int64_t rows, int32_t c,
                            const float* mean, const float* rstd, const float* beta, int32_t act, float alpha,
                            const float* mask, int64_t ldm, int32_t chunk_rows, float* partial, hypel_stream_t stream);
/* sums[0][c] = sum over chunks (fp64, chunk order) of partial[.][0][c]; sums[1][c] likewise;
 * dparam (beta or bias gradient) = sums[0] (+= when accumulate). */
int hypel_bwd_reduce_finalize(const float* partial, int32_t n_chunks, int32_t c, float* sums, float* dparam,
                              int32_t accumulate, hypel_stream_t stream);
/* hypel_bn_act_bwd_reduce for a layer WITHOUT batch norm (tf_slim.fully_connected / conv2d with biases:
 * shadow_data_models.py:95-146, DUALCNNModel.py:48-54,99-100) that ALSO writes dY = dZ * act'(y) (* mask): there dY
 * needs no column sum, so the pass that reduces the bias gradient delivers it and hypel_bn_act_bwd_apply (one more read
 * of dZ and Y, one more launch) is not needed.  partial as hypel_bn_act_bwd_reduce (finish with
 * hypel_bwd_reduce_finalize); dy may alias dz. */
int hypel_act_bias_bwd_reduce(const float* dz, int64_t lddz, const float* y, int64_t ldy, int64_t rows, int32_t c,
                              int32_t act, float alpha, const float* mask, int64_t ldm, int32_t chunk_rows, float* partial,
                              float* dy, int64_t lddy, hypel_stream_t stream);
/* backward pass 2: dy = rstd*(dyh - sums0/M - xhat*sums1/M)   (or dy = dyh without normalisation).
 * dy may alias dz. */
int hypel_bn_act_bwd_apply(const float* dz, int64_t lddz, const float* y, int64_t ldy, int64_t rows, int32_t c,
                           const float* mean, const float* rstd, const float* beta, int32_t act, float alpha,
                           const float* mask, int64_t ldm, const float* sums, float* dy, int64_t lddy,
                           hypel_stream_t stream);
/* the same with M = stat_rows >= rows: the sums were all-reduced over the data-parallel ranks and run over the global
 * batch (synchronised batch norm, see hypel_bn_merge_partials). */
int hypel_bn_act_bwd_apply_global(const float* dz, int64_t lddz, const float* y, int64_t ldy, int64_t rows, int32_t c,
                                  const float* mean, const float* rstd, const float* beta, int32_t act, float alpha,
                                  const float* mask, int64_t ldm, const float* sums, int64_t stat_rows, float* dy,
                                  int64_t lddy, hypel_stream_t stream);
/* gradient of the channel map: dr[row][ci] (+)= sum_{c in [start[ci], start[ci+1])} dz[row][c];
 * start NULL -> identity (cin == c). */
int hypel_chanmap_bwd(const float* dz, int64_t lddz, int64_t rows, int32_t c, float* dr, int64_t lddr, int32_t cin,
                      const int32_t* start, int32_t accumulate, hypel_stream_t stream);

/* ---- losses ------------------------------------------------------------------------------------------ */
/* tf.nn.softmax_cross_entropy_with_logits (HYPELCNNModel.py:102, DUALCNNModel.py:88, cut_wrapper.py:382,413):
 * loss[i] = -sum_j labels[i][j]*log_softmax(logits[i])[j];
 * dlogits[i][j] = gscale*(softmax[i][j]*sum_j labels[i][j] - labels[i][j])  (dlogits NULL -> skipped). */
int hypel_softmax_xent(const float* logits, int64_t ld, int64_t n, int32_t c, const float* labels, int64_t ldl,
                       float* loss, float* dlogits, int64_t lddl, float gscale, hypel_stream_t stream);
/* reconstruction MSE (HYPELCNNModel.py:106-109): out[0] = mean((a-b)^2) over rows x c;
 * da = gscale*2*(a-b)/(rows*c) (da NULL -> skipped).  ws: >= 1024 floats. */
int hypel_mse(const float* a, int64_t lda, const float* b, int64_t ldb, int64_t rows, int32_t c, float* out, float* da,
              int64_t ldda, float gscale, float* ws, hypel_stream_t stream);
/* The classifier's loss tail in two launches instead of six (HYPELCNNModel.py:101-112: softmax cross entropy + weighted
 * reconstruction MSE; NanTensorHook, monitored_session_runner.py:151):
 * hypel_mse_partial_f32 = first stage of hypel_mse -- da as there, ws[HYPEL_MSE_PARTIALS] = per-block sums of squares;
 * hypel_loss_finalize_f32: out_ce = mean(loss_rows[n_rows]) (the per-row values hypel_softmax_xent wrote),
 * out_mse = sum(mse_ws) * mse_scale (both NULL = no reconstruction term), flag = 1.0f if either is not finite else 0
 * (NULL = no guard; same flag as hypel_loss_guard_f32), *step += 1 (NULL = leave; same as hypel_step_inc). */
#define HYPEL_MSE_PARTIALS 1024
int hypel_mse_partial_f32(const float* a, int64_t lda, const float* b, int64_t ldb, int64_t rows, int32_t c, float* da,
                          int64_t ldda, float gscale, float* ws, hypel_stream_t stream);
int hypel_loss_finalize_f32(const float* loss_rows, int32_t n_rows, const float* mse_ws, double mse_scale, float* out_ce,
                            float* out_mse, float* flag, uint64_t* step, hypel_stream_t stream);

/* out[0] = sum(x[0..count)) * scale, deterministic two-stage.  ws >= 1024 floats. */
int hypel_sum_f32(const float* x, int64_t count, float scale, float* out, float* ws, hypel_stream_t stream);

/* ---- optimisers (common_nn_ops.py:223-230; gan_common.py:264-265) ---------------------------------------- */
/* TF1 Adam: m=b1*m+(1-b1)g; v=b2*v+(1-b2)g^2; p -= lr_t*m/(sqrt(v)+eps), lr_t precomputed on the host. */
int hypel_adam_tf1(float* p, const float* g, float* m, float* v, int64_t count, float lr_t, float beta1, float beta2,
                   float eps, hypel_stream_t stream);
/* TF1 Momentum: a = mu*a + g; p -= lr*a. */
int hypel_momentum_tf1(float* p, const float* g, float* a, int64_t count, float lr, float mu, hypel_stream_t stream);
/* Non-finite loss guard -- NanTensorHook (classify/monitored_session_runner.py:151) + check_numerics inside
 * create_train_op (common/common_nn_ops.py:232), moved onto the device so that the step loop never waits for
 * the loss: flag[0] = 1.0f if loss_a[0] (or loss_b[0], nullable) is NaN/Inf, else 0.0f.  The session keeps
 * the flag in the element BEHIND the flat gradient buffer, so the data-parallel all-reduce (sum) of the gradients
 * gives every rank the same verdict.  The *_guarded optimisers return without touching p / slots when
 * skip[0] != 0 (skip NULL = unguarded). */
int hypel_loss_guard_f32(const float* loss_a, const float* loss_b, float* flag, hypel_stream_t stream);
int hypel_adam_tf1_guarded(float* p, const float* g, float* m, float* v, int64_t count, float lr_t, float beta1,
                           float beta2, float eps, const float* skip, hypel_stream_t stream);
int hypel_momentum_tf1_guarded(float* p, const float* g, float* a, int64_t count, float lr, float mu, const float* skip,
                               hypel_stream_t stream);

/* ---- dropout mask (tf_slim.dropout): mask in {0, 1/keep}, Philox4x32-10 counter RNG ----------------------
 * counter = (element group, *step_dev), key = seed: the step lives on the device so that a captured
 * HIP graph draws fresh masks on every replay; hypel_step_inc bumps it once per training step. */
int hypel_dropout_mask(float* mask, int64_t count, float keep_prob, uint64_t seed, const uint64_t* step_dev,
                       hypel_stream_t stream);
int hypel_step_inc(uint64_t* step_dev, hypel_stream_t stream);

/* ---- evaluation (common_nn_ops.py:243-277): pred[i] = argmax logits[i] (first max, as tf.argmax);
 * confusion[label][pred] += 1 (int32 atomics). labels: int32 class ids. pred may be NULL. */
int hypel_argmax_confusion(const float* logits, int64_t ld, int64_t n, int32_t c, const int32_t* labels,
                           int32_t* pred, int32_t* confusion, hypel_stream_t stream);

/* ---- data side: scene -> patches -> augmented batch; predictions -> label raster ---------------------------
 * hypel_gather_patches_f32 replaces the per-target Python loop over BasicDataSet.get_data_point
 * (common_nn_ops.py:169-185; importer/InMemoryImporter.py:27-38; importer/GeneratorImporter.py): casi
 * [hp, wp, cc] and lidar [hp, wp, cl] (NULL when cl == 0) are the symmetric-padded, normalised scene in HBM;
 * points int32 [n, 2] = (x, y) of the target = window origin in the padded scene; out [n, p, p, cc + cl].
 *
 * hypel_augment_patches_f32 replaces the tf.data map stage (common_nn_ops.py:376-440) and the batch gather in
 * front of it: for sample s, source = x[idx[s]] (idx NULL = identity), rotated counter-clockwise by rot_k[s]
 * quarter turns (the reference draws k in {0,1,2}, :402), shadowed when shadow_pick[s] (by division with
 * shadow_ratio[c] -- create_simple_shadow_struct, gan_utilities.py:17-27 -- or by taking the sample from
 * shadow_alt [n,p,p,c], the pre-computed generator output in batch order), flipped left/right and up/down when
 * flip_lr[s] / flip_ud[s], shifted by delta[s, c] (spectral augmentation).  Every selector may be NULL.  The random
 * decisions are drawn by the host iterator (seeded generator) and handed over as small device arrays.
 *
 * hypel_argmax_scatter replaces perform_prediction's per-sample Python loop (common_nn_ops.py:313-327):
 * raster[y * raster_w + x] = argmax(logits[i]) for points[i] = (x, y). */
int hypel_gather_patches_f32(const float* casi, const float* lidar, int64_t hp, int64_t wp, int32_t cc, int32_t cl,
                             const int32_t* points, int64_t n, int32_t p, float* out, hypel_stream_t stream);
/* GRSS2018DataSet.get_data_point (loader/GRSS2018DataLoader.py:12-44): casi [*, casi_wp, cc] has half the
 * resolution of lidar [*, lidar_wp, cl]; both are padded by `neighborhood`; points are LiDAR-grid coordinates. */
int hypel_gather_patches_2x_f32(const float* casi, const float* lidar, int64_t casi_wp, int64_t lidar_wp, int32_t cc,
                                int32_t cl, int32_t neighborhood, const int32_t* points, int64_t n, int32_t p,
                                float* out, hypel_stream_t stream);
int hypel_augment_patches_f32(const float* x, const int64_t* idx, int64_t n, int32_t p, int32_t c,
                              const int32_t* rot_k, const uint8_t* shadow_pick, const float* shadow_ratio,
                              const float* shadow_alt, const uint8_t* flip_lr, const uint8_t* flip_ud,
                              const float* delta, float* out, hypel_stream_t stream);
int hypel_argmax_scatter(const float* logits, int64_t ld, int64_t n, int32_t c, const int32_t* points,
                         uint8_t* raster, int64_t raster_w, hypel_stream_t stream);
/* The GAN trainer's tf.data stage (gan/gan_train_for_shadow.py:147-182: from_tensor_slices -> shuffle_and_repeat ->
 * map(perform_shadow_augmentation_random) -> batch): pair i of the batch = (normal[idx[i]], shadow[idx[i]]), both
 * [pool][bands] resident in HBM; with ratio != NULL the regulariser swap of :171-182 -- u1[i] < rate replaces the normal
 * spectrum by shadow * ratio, u2[i] < rate then replaces the shadow spectrum by (the possibly replaced) normal / ratio;
 * u1 / u2 are the two uniform draws per pair (host RNG, the reference's tf.random.uniform([1], 0.01, 0.99)). */
int hypel_gather_pairs_f32(const float* normal, const float* shadow, const int64_t* idx, int64_t n, int32_t bands,
                           const float* ratio, const float* u1, const float* u2, float rate, float* out_x, float* out_y,
                           hypel_stream_t stream);

/* ---- LRN (tf.nn.local_response_normalization, CONCNNModel.py:37,41) -------------------------------------- */
int hypel_lrn_fwd(const float* x, int64_t ldx, int64_t rows, int32_t c, int32_t radius, float bias, float alpha,
                  float beta, float* y, int64_t ldy, hypel_stream_t stream);
int hypel_lrn_bwd(const float* x, int64_t ldx, const float* dy, int64_t lddy, int64_t rows, int32_t c, int32_t radius,
                  float bias, float alpha, float beta, float* dx, int64_t lddx, int32_t accumulate,
                  hypel_stream_t stream);

/* ---- shadow GAN stacks (gan/shadow_data_models.py, gan/wrappers/) -------------------------------------------
 * Fused generator (shadowdata_generator_model :43-90): x[N][B] -> out[N][B]; seven 1-channel SAME 1-D convolutions
 * over the band axis with kernel sizes B, B/2, B/4, B/8, B/4, B/2, B, leaky-ReLU(0.1), skip sums
 * n_i = c_i + n_{i-1} + n_{i-2}, tanh on the last layer; only_encoder != 0 stops after layer 4 and returns n4.
 * w = the layers' kernels concatenated (TF variables netK/weights [k,1,1]), b = 7 biases.  From 16 to 384 bands the
 * stack runs on the matrix cores (csrc/gan_mfma.hip: a 1-channel SAME convolution over B bands of N samples is
 * X[N x B] . T[B x B] with T banded Toeplitz; 16 samples per block, activations in LDS, the taps as the MFMA B operand
 * straight from a zero-margined table, filter gradient = diagonal sums of X^T dZ tiles accumulated per tile offset);
 * otherwise one wavefront (or two) per sample on the vector ALUs (csrc/gan.hip).  The backward recomputes the forward;
 * weight / bias gradients are written as per-block partial sums pw[blocks][sum k], pb[blocks][8] with
 * blocks = hypel_gan_generator_blocks(n) (reduce with hypel_reduce_splits_f32; every slab is written).  dx may be NULL. */
int hypel_gan_generator_blocks(int64_t n);
int hypel_gan_generator_fwd(const float* x, int64_t ldx, int64_t n, int32_t bands, const float* w, const float* b,
                            int32_t only_encoder, float* out, int64_t ldo, hypel_stream_t stream);
int hypel_gan_generator_bwd(const float* x, int64_t ldx, const float* dout, int64_t lddo, int64_t n, int32_t bands,
                            const float* w, const float* b, int32_t only_encoder, float* dx, int64_t lddx,
                            int32_t accumulate_dx, float* pw, float* pb, hypel_stream_t stream);

/* The same pair with the forward pass's activations KEPT for the backward pass instead of recomputed there (29 % of the
 * matrix-core backward kernel at B = 360): hypel_gan_generator_fwd_keep leaves the layer outputs, leaky-ReLU branch bits and
 * the tanh output of every 16-sample tile in `keep` (hypel_gan_generator_keep_floats(n, bands, only_encoder) floats; an
 * opaque, lane-native layout), hypel_gan_generator_bwd_kept of the SAME (x, w, b, n, bands, only_encoder) starts from it:
 * bit-identical to the recomputing pair.  keep_floats == 0: this band count runs on the VALU kernels, which always
 * recompute -- pass keep = NULL (then the calls are hypel_gan_generator_fwd / _bwd). */
int64_t hypel_gan_generator_keep_floats(int64_t n, int32_t bands, int32_t only_encoder);
/* Encoder tap (round 4).  CUT applies the encoder-only generator to tensors the FULL generator of the same train op also
 * consumes (gan/wrappers/cut_wrapper.py:301-339: gen(x), gen(x, only_encoder) -- and the same for y): the encoder output is
 * the full generator's n_4.  hypel_gan_generator_fwd_tap = hypel_gan_generator_fwd[_keep](only_encoder = 0) that also writes
 * enc_out[n x bands] = what hypel_gan_generator_fwd(only_encoder = 1) would write, bit for bit (keep may be NULL);
 * hypel_gan_generator_bwd_tap = hypel_gan_generator_bwd[_kept](only_encoder = 0) that adds d_enc -- the gradient of enc_out
 * -- to the gradient of n_4: one backward pass yields the input and filter gradients of both applications.
 * Matrix-core kernels only: hypel_gan_generator_tap_supported(bands) != 0. */
int hypel_gan_generator_tap_supported(int32_t bands);
int hypel_gan_generator_fwd_tap(const float* x, int64_t ldx, int64_t n, int32_t bands, const float* w, const float* b,
                                float* out, int64_t ldo, float* enc_out, int64_t ld_enc, float* keep,
                                hypel_stream_t stream);
int hypel_gan_generator_bwd_tap(const float* x, int64_t ldx, const float* dout, int64_t lddo, const float* d_enc,
                                int64_t ld_denc, int64_t n, int32_t bands, const float* w, const float* b, float* dx,
                                int64_t lddx, int32_t accumulate_dx, float* pw, float* pb, const float* keep,
                                hypel_stream_t stream);
int hypel_gan_generator_fwd_keep(const float* x, int64_t ldx, int64_t n, int32_t bands, const float* w, const float* b,
                                 int32_t only_encoder, float* out, int64_t ldo, float* keep, hypel_stream_t stream);
int hypel_gan_generator_bwd_kept(const float* x, int64_t ldx, const float* dout, int64_t lddo, int64_t n, int32_t bands,
                                 const float* w, const float* b, int32_t only_encoder, float* dx, int64_t lddx,
                                 int32_t accumulate_dx, float* pw, float* pb, const float* keep, hypel_stream_t stream);

/* ---- a short stack of narrow fully-connected layers in one launch per direction ------------------------------------
 * shadowdata_discriminator_model (gan/shadow_data_models.py:95-121): flatten -> tf_slim.fully_connected B -> B -> B -> B/2
 * with biases, leaky-ReLU(0.1) on the first two.  When every width is <= 128 (the Gulfport stacks, B = 64) the whole stack
 * of one application runs out of LDS, 16 samples per block, on v_mfma_f32_16x16x4_f32 -- instead of one GEMM + one
 * activation launch per layer forward and ~12 launches backward, each a few microseconds of launch latency.
 *   n_layers <= 4, widths w0 -> w1 -> ... (unused trailing widths 0); bit l of act_mask: leaky-ReLU(alpha) after layer l.
 *   w: the layers' [cin][cout] matrices one after the other (tf_slim `weights`), b: their biases one after the other.
 *   bwd: recomputes the forward; dx may be NULL; pw[blocks][sum cin*cout] / pb[blocks][sum cout] are per-block partial
 *   filter / bias gradients, blocks = hypel_dense_stack_blocks(n) -- every slab is written; sum them in slab order with
 *   hypel_reduce_splits_f32 (deterministic).  hypel_dense_stack_supported: 1 when the shape runs here. */
int hypel_dense_stack_supported(int32_t n_layers, int32_t w0, int32_t w1, int32_t w2, int32_t w3, int32_t w4);
int hypel_dense_stack_blocks(int64_t n);
int hypel_dense_stack_fwd(const float* x, int64_t ldx, int64_t n, int32_t n_layers, int32_t w0, int32_t w1, int32_t w2,
                          int32_t w3, int32_t w4, int32_t act_mask, float alpha, const float* w, const float* b, float* out,
                          int64_t ldo, hypel_stream_t stream);
int hypel_dense_stack_bwd(const float* x, int64_t ldx, const float* dout, int64_t lddo, int64_t n, int32_t n_layers,
                          int32_t w0, int32_t w1, int32_t w2, int32_t w3, int32_t w4, int32_t act_mask, float alpha,
                          const float* w, const float* b, float* dx, int64_t lddx, int32_t accumulate_dx, float* pw,
                          float* pb, hypel_stream_t stream);

/* ---- several applications of same-shaped networks with DIFFERENT variables in one launch -----------------------------
 * CycleGAN applies two generators (x2y, y2x) and two critics (D_x, D_y) of one shape side by side
 * (gan/wrappers/cycle_gan_wrapper.py:82-124; the DCL-GAN pair of CUT models likewise): at the Gulfport sizes every such
 * application is a latency chain that leaves most of the chip idle, so application g of `n_apps` runs on its own share of
 * the blocks of ONE launch.  Rows: application g reads rows [g*n, (g+1)*n) of x / dout / keep and writes the same rows of
 * out / dx (n rows per application).  Variables: application g's are at w + g*w_stride, b + g*b_stride (elements; any
 * sign -- the variables of two models lie a fixed distance apart in the flat parameter buffer).  Gradient slabs (bwd):
 * blocks_apps(n, n_apps) / n_apps slabs per application, application g's first slab at pw + g*pw_stride, pb + g*pb_stride
 * (pw_stride == 0: all slabs consecutive in application order).  Each application's results are bit-identical to the
 * single-application entry point's on its rows, and its slabs sum to the same gradients.
 * Generator: matrix-core shapes only (hypel_gan_generator_tap_supported(bands)); keep (nullable) holds
 * n_apps * hypel_gan_generator_keep_floats(n) floats, application after application. */
int hypel_gan_generator_blocks_apps(int64_t n, int32_t n_apps);
int hypel_gan_generator_fwd_apps(const float* x, int64_t ldx, int64_t n, int32_t n_apps, int64_t w_stride, int64_t b_stride,
                                 int32_t bands, const float* w, const float* b, int32_t only_encoder, float* out,
                                 int64_t ldo, float* keep, hypel_stream_t stream);
int hypel_gan_generator_bwd_apps(const float* x, int64_t ldx, const float* dout, int64_t lddo, int64_t n, int32_t n_apps,
                                 int64_t w_stride, int64_t b_stride, int64_t pw_stride, int64_t pb_stride, int32_t bands,
                                 const float* w, const float* b, int32_t only_encoder, float* dx, int64_t lddx,
                                 int32_t accumulate_dx, float* pw, float* pb, const float* keep, hypel_stream_t stream);
int hypel_dense_stack_blocks_apps(int64_t n, int32_t n_apps);
int hypel_dense_stack_fwd_apps(const float* x, int64_t ldx, int64_t n, int32_t n_apps, int64_t w_stride, int64_t b_stride,
                               int32_t n_layers, int32_t w0, int32_t w1, int32_t w2, int32_t w3, int32_t w4,
                               int32_t act_mask, float alpha, const float* w, const float* b, float* out, int64_t ldo,
                               hypel_stream_t stream);
int hypel_dense_stack_bwd_apps(const float* x, int64_t ldx, const float* dout, int64_t lddo, int64_t n, int32_t n_apps,
                               int64_t w_stride, int64_t b_stride, int64_t pw_stride, int64_t pb_stride, int32_t n_layers,
                               int32_t w0, int32_t w1, int32_t w2, int32_t w3, int32_t w4, int32_t act_mask, float alpha,
                               const float* w, const float* b, float* dx, int64_t lddx, int32_t accumulate_dx, float* pw,
                               float* pb, hypel_stream_t stream);

/* tensorflow_gan losses: mode 0: weight*mean((a-target)^2) (least squares, pass weight/2),
 * mode 1: weight*mean(|a-b|) (cycle consistency / absolute_difference), mode 2: weight*mean(a) (Wasserstein).
 * loss[0] (+)= value; da / db (nullable) (+)= gradient.  ws >= 1024 floats. */
int hypel_gan_loss(int32_t mode, const float* a, int64_t lda, const float* b, int64_t ldb, int64_t rows, int32_t c,
                   float target, float weight, float* loss, int32_t accumulate_loss, float* da, int64_t ldda,
                   int32_t acc_da, float* db, int64_t lddb, int32_t acc_db, float* ws, hypel_stream_t stream);
/* tf_slim.l2_regularizer: loss[0] (+)= scale/2 * sum w^2 ; dw (nullable) += scale * w.  ws >= 1024 floats. */
int hypel_l2_reg(const float* w, int64_t count, float scale, float* loss, int32_t accumulate_loss, float* dw,
                 float* ws, hypel_stream_t stream);

/* The same terms with a DEFERRED sum: every loss term of a train op (tfgan tuple losses, cycle / identity L1, the scope's
 * l2 regularisers) leaves its weighted block partials in its own slot of 1024 floats (hypel_loss_terms_slots); ONE
 * hypel_loss_finalize_slots at the end of the op adds slots [0, n_slots) in index order into the op's loss -- one
 * finaliser launch per op instead of one per term. */
int hypel_loss_finalize_slots(const float* slots, int32_t n_slots, float* loss, int32_t accumulate_loss,
                              hypel_stream_t stream);

/* Several of those terms in ONE launch.  Offsets are in elements relative to `base` (HYPEL_LOSS_NONE = operand absent);
 * mode 0-2 as hypel_gan_loss (gcoef = weight / (rows * c), pscale = the same), mode 3 = hypel_l2_reg (a = w, rows =
 * count, da = dw, gcoef = scale, pscale = scale / 2).  Two terms of one launch must not write the same gradient buffer. */
#define HYPEL_LOSS_NONE INT64_MIN
typedef struct {
    int64_t a_off, b_off, da_off, db_off;
    int64_t lda, ldb, ldda, lddb, rows;
    int32_t mode, c, acc_da, acc_db;
    float target, gcoef, pscale;
    int32_t slot;
} hypel_loss_term_t;
int hypel_loss_terms_slots(const float* base, const hypel_loss_term_t* terms, int32_t n_terms, float* slots,
                           hypel_stream_t stream);
/* tf.math.l2_normalize(axis=None) over the whole [rows x c] tensor (shadow_data_models.py:147).
 * stat[0] = sum x^2, stat[1] = rsqrt(max(sum, 1e-12)). */
/* `parts` adjacent [rows x c] column blocks of one matrix, each normalised by its own whole-block norm, in one
 * launch (the stacked slice embeddings of the feature discriminator); stat holds 2 floats per part. */
int hypel_l2norm_parts_fwd(const float* x, int64_t ldx, int64_t rows, int32_t c, int32_t parts, float* y, int64_t ldy,
                           float* stat, hypel_stream_t stream);
int hypel_l2norm_parts_bwd(const float* x, int64_t ldx, const float* dy, int64_t lddy, int64_t rows, int32_t c,
                           int32_t parts, const float* stat, float* dx, int64_t lddx, int32_t accumulate,
                           hypel_stream_t stream);
/* The same for `segs` row segments of `rows` rows each (the applications of a row-concatenated batch: every
 * application keeps its own whole-tensor norms); stat holds 2 floats per (segment, part), segment-major. */
int hypel_l2norm_segs_fwd(const float* x, int64_t ldx, int64_t rows, int32_t c, int32_t parts, int32_t segs, float* y,
                          int64_t ldy, float* stat, hypel_stream_t stream);
int hypel_l2norm_segs_bwd(const float* x, int64_t ldx, const float* dy, int64_t lddy, int64_t rows, int32_t c,
                          int32_t parts, int32_t segs, const float* stat, float* dx, int64_t lddx, int32_t accumulate,
                          hypel_stream_t stream);
int hypel_l2norm_fwd(const float* x, int64_t ldx, int64_t rows, int32_t c, float* y, int64_t ldy, float* stat,
                     hypel_stream_t stream);
int hypel_l2norm_bwd(const float* x, int64_t ldx, const float* dy, int64_t lddy, int64_t rows, int32_t c,
                     const float* stat, float* dx, int64_t lddx, int32_t accumulate, hypel_stream_t stream);
/* patch-NCE (cut_wrapper.py:360-420): g, r are [n][p][e] embeddings; per sample softmax-CE over the p*p logits
 * <g_a, r_b>/tau against the flattened identity; loss[0] (+)= weight * mean over n.  ws >= n + 1024 floats. */
int hypel_nce_loss(const float* g, int64_t ldg, const float* r, int64_t ldr, int64_t n, int32_t p, int32_t e,
                   float tau, float weight, float* loss, int32_t accumulate_loss, float* dg, int64_t lddg,
                   int32_t acc_dg, float* dr, int64_t lddr, int32_t acc_dr, float* ws, hypel_stream_t stream);

/* ---- graph capture helpers (HIP graphs instead of a tracing compiler) ---------------------------------------- */
int hypel_graph_begin_capture(hypel_stream_t stream);
int hypel_graph_end_capture(hypel_stream_t stream, void** graph_exec_out);
int hypel_graph_launch(void* graph_exec, hypel_stream_t stream);
int hypel_graph_destroy(void* graph_exec);

#ifdef __cplusplus
}
#endif
#endif /* HYPEL_H */
