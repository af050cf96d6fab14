/* libmtp_hip.so -- C ABI of the MI355X-native (gfx950) ViT+RVSA backbone hot path of ViTAE-Transformer/MTP.
 *
 * Drop-in boundary.  The reference's only native-operator precedent is DCNv3's
 * pybind pair dcnv3_forward / dcnv3_backward (Multi-Task_Pretrain/backbone/ops_dcnv3/src/vision.cpp:14-17,
 * dcnv3.h:20-59): contiguous device tensors in, work on the current stream, no hidden sync.  For the ViT/RVSA
 * path the reference has NO native ops -- every entry point below replaces a chain of ATen library calls made by
 * Multi-Task_Pretrain/backbone/vit_win_rvsa_v3_wsz7.py ("VIT"), cited per function.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; every buffer (inputs, outputs, workspaces) is owned by the CALLER
 *     (torch's caching allocator on the Python side) -- nothing is allocated or freed in here;
 *   - every call is asynchronous on `stream`, never synchronises, keeps no global mutable state, and may be
 *     issued concurrently on different streams;
 *   - inputs must be contiguous (leading dimensions are passed where a kernel supports strides) and 16-byte
 *     aligned; token tensors are row-major (T, C) with T = B*Hp*Wp (row t = (b, y, x));
 *   - return value: 0 = launched; MTP_ERR_* (<0) = argument check failed (nothing launched); >0 = hipError_t;
 *   - dtype arguments use mtp_dtype; "ACT" tensors are bf16 (throughput mode) or f32 (parity mode); statistics,
 *     parameters, parameter gradients and the residual stream are always f32.
 */
#ifndef MTP_HIP_H_
#define MTP_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mtp_stream_t; /* hipStream_t */

typedef enum { MTP_F32 = 0, MTP_BF16 = 1, MTP_F64 = 2 /* mtp_dcnv3_fwd / mtp_dcnv3_bwd only: the reference's double dispatch, dcnv3_cuda.cu:69 */ } mtp_dtype;

enum { MTP_OK = 0, MTP_ERR_ARG = -1, MTP_ERR_UNSUPPORTED = -2 };

/* GEMM epilogues (fused elementwise work of VIT:55-62, 506-513, 793-794) */
typedef enum {
    MTP_EPI_BIAS = 0,      /* C = acc + bias                          (bias may be NULL)                   */
    MTP_EPI_BIAS_GELU = 1, /* aux = acc + bias (pre-activation u);  C = gelu(u)            nn.GELU erf   */
    MTP_EPI_BIAS_RES = 2,  /* C(f32) = res[row % res_mod] + rowscale[row / rows_per_sample] * (acc + bias) */
    MTP_EPI_DGELU = 3,     /* C = acc * gelu'(aux)                                                         */
    /* the pair the training engine uses for fc1 / its backward (VIT:55-62): the forward stores gelu'(u) instead of u -- the erf
     * and the exponential are shared with gelu(u), and the backward epilogue is one multiplication instead of another erf */
    MTP_EPI_BIAS_GELU_DG = 4, /* u = acc + bias;  aux = gelu'(u);  C = gelu(u)                              */
    MTP_EPI_MUL = 5           /* C = acc * aux                                                            */
} mtp_epilogue;

typedef struct {
    const void* A;      /* NT: (M, K) row-major, lda.   TN: (Kc, M) row-major, lda                         */
    const void* B;      /* NT: (N, K) row-major, ldb.   TN: (Kc, N) row-major, ldb                         */
    void* C;            /* (M, N) row-major, ldc                                                          */
    int64_t M, N, K;    /* K = contraction length                                                         */
    int64_t lda, ldb, ldc;
    int in_dtype;       /* mtp_dtype of A and B                                                           */
    int out_dtype;      /* mtp_dtype of C (and aux)                                                       */
    int epilogue;       /* mtp_epilogue (NT only)                                                         */
    const float* bias;  /* [bias_mod ? bias_mod : N] or NULL                                              */
    int64_t bias_mod;   /* bias index = n % bias_mod when > 0 (ConvTranspose2d bias repeated per tap)      */
    const float* res;   /* EPI_BIAS_RES: f32 (res rows, N), ld res_ld                                     */
    int64_t res_ld, res_mod;
    const float* rowscale; /* EPI_BIAS_RES: per-sample drop-path factor or NULL                           */
    int64_t rows_per_sample;
    void* aux;          /* EPI_BIAS_GELU: u out;  EPI_DGELU: u in;  EPI_BIAS_GELU_DG: gelu'(u) out;  EPI_MUL: factor in;
                         * (M, N), ld aux_ld, dtype out_dtype.
                         * TN with split_k > 1: optional f32 workspace (split_k * M * N) for the partial tiles,
                         * summed into C by the callee (deterministic); NULL = f32 atomicAdd into zeroed C  */
    int64_t aux_ld;
    int split_k;        /* TN only: >1 = split the contraction over gridDim.y                               */
    int variant;        /* 0 = default kernels and heuristics; other values force A/B choices (debug / tests): bit0 NT register-
                         * staged loads; bits1-2 tile order (1 plain, 2 grouped / M-fastest, 3 row-major); bit3 TN single-stage;
                         * bit4 TN register-transposing; bit5 / bit6 force / forbid the 256x128 8-wave NT tile; bits8-9 the 8-wave
                         * pipelined NT kernel (1 = tile height picked per problem, 2 = 224 x 256 tiles, 3 = 256 x 256 tiles), bit10
                         * forbids it; bit15 / bit16 force / forbid its persistent-tile form; bits20-21 store policy of its epilogue
                         * (0 = by epilogue: nt, sc1 for the f32 residual form; 1 = nt, 2 = sc1 write-through, 3 = plain).  All variants
                         * of one problem give bit-identical results.  Bit 17 forces the strip kernel (round 5; see mtp_gemm_nt_tile), bit 18
                         * forbids it.  TN grouped: bit 19 = the plain phase instead of the read-ahead phase (round 5 A/B).  (Bits 7, 11-14
                         * selected kernels that were removed in round 4 -- tools/ablation/ -- and are ignored.) */
    float* colsum;      /* TN only, optional: colsum[m] += sum_k A[k][m]  (f32, M entries, ACCUMULATES) -- the bias
                         * gradient db = sum_rows dY comes out of the dW = dY^T X GEMM that streams dY anyway  */
    int defer_sum;      /* TN with a split-K workspace: 1 = leave the partial tiles in `aux`; the caller reduces them
                         * later with mtp_sum_partials_batch (one launch for the weight gradients of a whole block) */
    int pad_;
    void* workspace;    /* mtp_gemm_tn_grouped (round 6): optional device float (workspace_bytes >= 4) that receives += sum(C^2) of this
                         * problem -- the clipping step's gradient norm as a by-product; MTP_ERR_UNSUPPORTED with split_k > 1.  Ignored by
                         * every other entry (round 3: scratch of the stream-K NT form, removed in round 4).                          */
    int64_t workspace_bytes;
} mtp_gemm_args;

/* y = x W^T (+epilogue): nn.Linear fwd/dgrad (VIT:50,52,78,87,256,262), patch-embed conv as GEMM (VIT:529),
 * ConvTranspose2d(2,2) as GEMM (VIT:642-649).  C[m][n] = sum_k A[m][k] * B[n][k]. */
int mtp_gemm_nt(const mtp_gemm_args* args, mtp_stream_t stream);
/* Query: the kernel family mtp_gemm_nt runs these arguments on, named by its tile -- 256 = the 8-wave pipelined
 * 256 x 256 x 64 kernel (bf16, K % 128 == 0, M % 8 == 0, N % 8 == 0), 64 = the strip kernel (round 5: 128 x 256 strips of
 * 64 x 64 wave blocks with two accumulator sets, the epilogue of a strip computed under the next strip's K loop; bf16,
 * K % 64 == 0, K >= 704), 128 = the 128-wide kernels (every other case, and f32).
 * All families accumulate in the same k order: results are bit-identical. */
int mtp_gemm_nt_tile(const mtp_gemm_args* args);
/* bytes of `workspace` mtp_gemm_nt wants: 0 since round 4 (kept in the ABI for callers compiled against 0.3) */
int64_t mtp_gemm_nt_workspace_bytes(void);
/* weight gradient: C[m][n] = sum_k A[k][m] * B[k][n]  (dW = dY^T X), f32 output. */
int mtp_gemm_tn(const mtp_gemm_args* args, mtp_stream_t stream);
/* Grouped weight gradients: `count` (<= MTP_MAX_GROUPED_GEMMS) independent problems C_i (M_i, N_i) f32 = A_i (K_i, M_i)^T B_i (K_i, N_i)
 * (+ colsum_i += column sums of A_i) in ONE launch of 256 x 256 output tiles, each workgroup running the whole contraction of
 * its tile through the 8-phase pipeline (no split-K, no partial tiles).  Meant for the weight gradients of several transformer
 * blocks at once (4 ViT-L blocks = 768 tiles = three full rounds of the 256 CUs); results overwrite C_i.  Every problem must be
 * bf16 in / f32 out with M_i % 256 == 0, N_i % 256 == 0, K_i % 128 == 0; otherwise MTP_ERR_UNSUPPORTED and nothing is launched
 * (use mtp_gemm_tn per problem).  Fields epilogue / bias / res / aux / split_k / defer_sum of the entries are ignored. */
#define MTP_MAX_GROUPED_GEMMS 32
int mtp_gemm_tn_grouped(const mtp_gemm_args* args, int count, mtp_stream_t stream);
/* out[i] = sum over `splits[i]` partial tiles of numel[i] f32 each, stored back to back at parts[i] -- the deferred split-K
 * reductions of up to MTP_MAX_SEGMENTS weight-gradient GEMMs (args.defer_sum) in one launch.  Host arrays. */
int mtp_sum_partials_batch(const float* const* parts, float* const* outs, const int64_t* numel, const int* splits, int count,
                           mtp_stream_t stream);

/* nn.LayerNorm(C, eps) over the last dim (VIT:484,496,579,596); optional fused exact GELU (fpn1: Norm2d -> GELU,
 * VIT:643-644).  x: (rows, C) in x_dtype; y in y_dtype; mean/rstd f32 (rows). */
int mtp_layernorm_fwd(const void* x, int x_dtype, const float* gamma, const float* beta, void* y, int y_dtype,
                      float* mean, float* rstd, int64_t rows, int64_t C, float eps, int fuse_gelu, mtp_stream_t stream);
/* dx_out(f32 or ACT) = [dres] + [extra] + LN'(dy);  dx_copy (ACT, optional) = copy_scale[row / rows_per_sample] * dx_out;
 * dgamma/dbeta partials: nblk rows of C f32 each, row stride part_ld (0 = C; 2C when both live in one (nblk, 2C) buffer so
 * that one mtp_reduce_rows_f32 launch finishes both), nblk = mtp_layernorm_bwd_partial_rows(rows). */
int64_t mtp_layernorm_bwd_partial_rows(int64_t rows);
int mtp_layernorm_bwd(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* mean, const float* rstd,
                      const float* gamma, const float* beta, int fuse_gelu,
                      const float* dres, const float* extra, void* dx, int dx_dtype,
                      void* dx_copy, int copy_dtype, const float* copy_scale, int64_t rows_per_sample,
                      float* dgamma_part, float* dbeta_part, int64_t part_ld, int64_t rows, int64_t C, mtp_stream_t stream);
/* The same with a per-window addend of the incoming gradient: dy_eff[row] = dy[row] + win_add[window(row)] where the rows are the tokens
 * (b, y, x) of a (B, Hp, Wp) grid and window(row) its 7 x 7 RVSA window (padding split as VIT:298-303; B * Hp * Wp == rows).  win_add:
 * (B * nh * nw, C) f32 from mtp_rvsa_sampling_bwd_win -- norm1's backward then adds the sampling heads' input gradient while it reads the
 * row anyway, instead of mtp_rvsa_sampling_bwd's read-modify-write pass over (T, C) (round 4).  No GELU form; f32 residual stream. */
int mtp_layernorm_bwd_win(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* mean, const float* rstd,
                          const float* gamma, const float* dres, const float* extra, void* dx, int dx_dtype,
                          void* dx_copy, int copy_dtype, const float* copy_scale, int64_t rows_per_sample,
                          float* dgamma_part, float* dbeta_part, int64_t part_ld, int64_t rows, int64_t C,
                          const float* win_add, int64_t B, int64_t Hp, int64_t Wp, mtp_stream_t stream);
/* InternImage's post-norm residual (intern_image.py:424-426) in one pass each way (round 4):
 *   forward   out (rows, C) f32 = x + sample_scale[row / rows_per_sample] * layer_scale * LayerNorm(h);  out_act = its ACT copy (optional); h in `dtype`
 *   backward  dh (`dtype`) = LN'(s * layer_scale * dout);  part: (mtp_layernorm_bwd_partial_rows(rows), 3 C) f32 = per-workgroup partials of
 *             [d gamma | d beta | d layer_scale], d layer_scale = sum_rows s * dout * LN(h) with LN(h) recomputed from mean / rstd. */
int mtp_layernorm_residual_fwd(const void* h, int dtype, const float* gamma, const float* beta, const float* x, const float* layer_scale,
                               const float* sample_scale, int64_t rows_per_sample, float* out, void* out_act, float* mean, float* rstd,
                               int64_t rows, int64_t C, float eps, mtp_stream_t stream);
int mtp_layernorm_residual_bwd(const float* dout, const void* h, int dtype, const float* mean, const float* rstd, const float* gamma, const float* beta,
                               const float* layer_scale, const float* sample_scale, int64_t rows_per_sample, void* dh, float* part,
                               int64_t rows, int64_t C, mtp_stream_t stream);
/* out[c] (+)= sum_r part[r * ld + c], c < C   (per-workgroup partials -> parameter gradient; ld >= C lets one
 * partial buffer feed several parameters) */
int mtp_reduce_rows_f32(const float* part, int64_t ld, float* out, int64_t rows, int64_t C, int accumulate, mtp_stream_t stream);
/* the same for n <= MTP_REDUCE_BATCH_MAX partial buffers of one shape in ONE launch: outs[i][c] (+)= sum_r parts[i][r * ld + c].
 * parts / outs are HOST arrays of device pointers (the LayerNorm parameter gradients of a burst of blocks: the partial rows of
 * mtp_layernorm_bwd are kept until the burst's weight-gradient group is launched, then reduced together) */
#define MTP_REDUCE_BATCH_MAX 32
int mtp_reduce_rows_batched_f32(const float* const* parts, float* const* outs, int n, int64_t ld, int64_t rows, int64_t C, int accumulate,
                                mtp_stream_t stream);
/* the same, result transposed: part (rows, R*C) f32, column a*C + b is summed into out[b*R + a] (out is (C, R)) */
int mtp_reduce_rows_t_f32(const float* part, int64_t ld, float* out, int64_t rows, int64_t R, int64_t C, int accumulate, mtp_stream_t stream);
/* ... and n <= MTP_REDUCE_BATCH_MAX of those of one shape in one launch (host arrays of device pointers) */
int mtp_reduce_rows_t_batched_f32(const float* const* parts, float* const* outs, int n, int64_t ld, int64_t rows, int64_t R, int64_t C, int accumulate,
                                  mtp_stream_t stream);
/* bias gradient: out[n] = sum_m dY[m][n] */
int mtp_colsum(const void* dY, int dtype, int64_t ld, float* out, int64_t M, int64_t N, mtp_stream_t stream);
/* same, accumulating: out[n] += ... (no clearing pass; used with a gradient buffer that is zeroed once per step) */
int mtp_colsum_acc(const void* dY, int dtype, int64_t ld, float* out, int64_t M, int64_t N, mtp_stream_t stream);

/* ---- layout / elementwise ------------------------------------------------------------------------------- */
/* PatchEmbed im2col (VIT:529,536-539): img f32 NCHW -> cols (B*Hp*Wp, Cin*P*P) ACT, K order (c, ky, kx). */
int mtp_patchify(const float* img, void* cols, int dtype, int64_t B, int64_t Cin, int64_t H, int64_t W, int64_t P, mtp_stream_t stream);
int mtp_unpatchify(const void* cols, int dtype, float* dimg, int64_t B, int64_t Cin, int64_t H, int64_t W, int64_t P, mtp_stream_t stream);
/* Image side of MTP_DataPreprocessor (Multi-Task_Pretrain/preprocessing.py:145-148 -> mmengine ImgDataPreprocessor.forward:
 * channel flip, .float(), (x - mean) / std, pad bottom/right with pad_value to a multiple of pad_size_divisor; configured at
 * models.py:37-41) fused with the PatchEmbed im2col: img (B, H, W, 3) uint8 HWC -> cols (B*Hp*Wp, 3*P*P) ACT, K order
 * (c, ky, kx), Hp = ceil(H / pad_divisor) * pad_divisor / P (likewise Wp).  mean / std: 3 HOST floats each, indexed by the
 * OUTPUT channel (i.e. after the flip). */
int mtp_preprocess_patchify(const uint8_t* img, void* cols, int dtype, int64_t B, int64_t H, int64_t W, int64_t P, int64_t pad_divisor,
                            const float* mean, const float* std, int bgr_to_rgb, float pad_value, mtp_stream_t stream);
int mtp_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, mtp_stream_t stream);
/* dst (C, R) = src (R, C)^T with dtype conversion (weight copies for dgrad) */
int mtp_transpose_cast(const float* src, void* dst, int dst_dtype, int64_t R, int64_t C, mtp_stream_t stream);
/* All GEMM-side images of the f32 master weights in ONE launch per optimizer step (nn.Linear weights of every block,
 * VIT:50-52,78,87,256,262, and the patch-embed conv, VIT:529): for each matrix src (R, C) f32 -> w (R, C) and/or
 * wt (C, R) in `act_dtype` (or f32 when f32_out: the three stacked RVSA head weights/biases, VIT:231-242).
 * `descs` is a DEVICE array of n descriptors built once by the host; tile0 = exclusive prefix sum of
 * ceil(R/64)*ceil(C/64) over the table, total_tiles = its total.  Matrices with C % 4 != 0 (or R % 4 != 0 with a
 * transpose) take an element-wise path. */
typedef struct {
    const float* src;
    void* w;        /* or NULL */
    void* wt;       /* or NULL */
    int64_t R, C;
    int64_t tile0;
    int32_t f32_out; /* images of this entry are f32 whatever act_dtype says */
    float wd;        /* weight decay of this parameter (mtp_adamw_weight_images only; mtp_weight_images ignores it) */
} mtp_wimg_desc;
int mtp_weight_images(const mtp_wimg_desc* descs_dev, int n, int64_t total_tiles, int act_dtype, mtp_stream_t stream);
/* The optimizer step and the images in ONE pass (round 6): torch.optim.AdamW semantics + clip_grad_norm_ scaling exactly as mtp_adamw_flat (main_pretrain.py:424-457,
 * 783-788), applied tile by tile to the parameters the descriptors name -- `src` points INTO the flat parameter buffer that starts at p_base; the gradient and the two
 * moments of a parameter sit at the same offset of g_base / m_base / v_base -- and the updated tile is written to the parameter and to its images w / wt (either may be
 * NULL: parameters that no GEMM reads).  Descriptors must cover every parameter that is to be updated exactly once. */
int mtp_adamw_weight_images(const mtp_wimg_desc* descs_dev, int n, int64_t total_tiles, int act_dtype, float* p_base, const float* g_base, float* m_base, float* v_base,
                            const float* hyper, const float* sqnorm, float max_norm, float grad_scale, mtp_stream_t stream);
/* ConvTranspose2d weight (Cin, Cout, 2, 2) f32 -> GEMM weight wg (4*Cout, Cin) and its transpose wgT (Cin, 4*Cout) */
int mtp_convt_pack(const float* w, void* wg, void* wgT, int dtype, int64_t Cin, int64_t Cout, mtp_stream_t stream);
/* dwg (4*Cout, Cin) f32 -> dw (Cin, Cout, 2, 2) f32 */
int mtp_convt_unpack_grad(const float* dwg, float* dw, int64_t Cin, int64_t Cout, mtp_stream_t stream);
/* rows (b, py, px, q_1..q_L) x C -> NCHW (B, C, Hp*2^L, Wp*2^L)  (VIT:807 + ConvT pixel shuffle) and back */
int mtp_tokens_to_nchw(const void* x, int x_dtype, void* out, int out_dtype, int64_t B, int64_t Hp, int64_t Wp, int64_t C, int levels, mtp_stream_t stream);
int mtp_nchw_to_tokens(const void* f, int f_dtype, void* out, int out_dtype, int64_t B, int64_t Hp, int64_t Wp, int64_t C, int levels, mtp_stream_t stream);
/* fpn4 = MaxPool2d(2,2) (VIT:654) on token-major x (T,C) f32 -> y (B*(Hp/2)*(Wp/2), C) token-major (then mtp_tokens_to_nchw);
 * bwd routes dy to the FIRST maximum of each 2x2 window (torch's tie rule); dx (T,C) f32 */
int mtp_maxpool2_tokens_fwd(const float* x, void* y, int y_dtype, int64_t B, int64_t Hp, int64_t Wp, int64_t C, mtp_stream_t stream);
int mtp_maxpool2_tokens_bwd(const float* x, const void* dy, int dy_dtype, float* dx, int accumulate, int64_t B, int64_t Hp, int64_t Wp, int64_t C, mtp_stream_t stream);
int mtp_axpy_f32(float* y, const float* x, float alpha, int64_t n, mtp_stream_t stream);
/* n <= MTP_MAX_SEGMENTS independent f32 copies dst[i][0..count[i]) = src[i][0..count[i]) in ONE launch (the stacked
 * gradient of the three RVSA 1x1-conv heads, VIT:231-242, goes back to six separate parameters). Host arrays. */
#define MTP_MAX_SEGMENTS 12
int mtp_copy_segments_f32(const float* const* src, float* const* dst, const int64_t* count, int n, mtp_stream_t stream);
/* dst (rows, C) ACT = scale[row / rows_per_sample] * src (rows, C) f32   (scale may be NULL = plain cast) */
int mtp_scale_rows_cast(const float* src, void* dst, int dst_dtype, const float* scale, int64_t rows_per_sample, int64_t rows, int64_t C, mtp_stream_t stream);

/* ---- InternImage layers around the DCNv3 core (csrc/conv.hip) ------------------------------------- */
/* Conv2d(kernel 3, stride s, padding 1) as im2col + mtp_gemm_nt (StemLayer intern_image.py:239-276, DownsampleLayer :279-300):
 * cols (N*Ho*Wo, Kp) ACT with column (kh*3 + kw)*Cin + c, zero for padding taps and for columns 9*Cin .. Kp (Kp: the caller's
 * multiple of 8).  The source is addressed by ELEMENT strides (sN, sH, sW, sC): NCHW images and channels-last maps both fit. */
int mtp_im2col3x3(const void* x, int x_dtype, int64_t sN, int64_t sH, int64_t sW, int64_t sC, void* cols, int cols_dtype,
                  int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t stride, int64_t Kp, mtp_stream_t stream);
/* dx (f32, element strides) = / += the transposed gather of dcols (N*Ho*Wo, Kp) */
int mtp_col2im3x3(const void* dcols, int cols_dtype, float* dx, int64_t sN, int64_t sH, int64_t sW, int64_t sC,
                  int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t stride, int64_t Kp, int accumulate, mtp_stream_t stream);
/* (Cout, Cin, 3, 3) f32 weight -> w2 (Cout, Kp) and / or w2t (Kp, Cout) in the im2col column order; and the gradient back */
int mtp_conv3x3_pack(const float* w, void* w2, void* w2t, int dtype, int64_t Cout, int64_t Cin, int64_t Kp, mtp_stream_t stream);
int mtp_conv3x3_unpack_grad(const float* dw2, float* dw, int64_t Cout, int64_t Cin, int64_t Kp, mtp_stream_t stream);
/* Linear weight (R, C) f32 -> wp (Rp, C) and / or wpt (C, Rp), zero rows / columns R .. Rp (mask head: 9 * groups rows) */
int mtp_pack_rows_padded(const float* w, void* wp, void* wpt, int dtype, int64_t R, int64_t C, int64_t Rp, mtp_stream_t stream);
/* dst (rows, ld) ACT: dst[r][c] = c < n ? src[r][c] : 0 -- an f32 gradient as the zero-padded contraction operand of a GEMM */
int mtp_cast_pad_rows(const float* src, int64_t n, void* dst, int dst_dtype, int64_t ld, int64_t rows, mtp_stream_t stream);
/* dst[r][0..n) = src[r][0..n) with separate row pitches (elements): compacts a GEMM output written with a padded pitch */
int mtp_copy_rows(const void* src, int64_t src_ld, void* dst, int64_t dst_ld, int dtype, int64_t n, int64_t rows, mtp_stream_t stream);
/* depth-wise Conv2d(C, C, 3, 1, 1, groups=C) of DCNv3 (ops_dcnv3/modules/dcnv3.py:262-272), channels-last (N,H,W,C) ACT;
 * w (C,1,3,3) f32, bias (C) f32.  bwd_dx: f32 output (= / +=).  bwd_dw: per-block partials (partial_rows, 10 C) f32 of
 * [dweight (C, 9) | dbias (C)], to be summed by mtp_reduce_rows_f32. */
int mtp_dwconv3x3_fwd(const void* x, const float* w, const float* bias, void* y, int dtype, int64_t N, int64_t H, int64_t W, int64_t C, mtp_stream_t stream);
int mtp_dwconv3x3_bwd_dx(const void* dy, int dtype, const float* w, float* dx, int accumulate, int64_t N, int64_t H, int64_t W, int64_t C, mtp_stream_t stream);
int64_t mtp_dwconv3x3_bwd_dw_partial_rows(int64_t N, int64_t H, int64_t W);
int mtp_dwconv3x3_bwd_dw(const void* dy, const void* x, int dtype, float* part, int64_t N, int64_t H, int64_t W, int64_t C, mtp_stream_t stream);
/* The same for any odd k (InternImage-H/G's dw_kernel_size, ops_dcnv3/modules/dcnv3.py:124, 146-151): plain per-(pixel, 4 channels) kernels; w (C, 1, k, k) f32.
 * mtp_dwconv_bwd_dw ACCUMULATES into dw / db with f32 atomics (clear them first); db may be NULL.  (k = 3: the mtp_dwconv3x3_* entries above are the fast path.) */
int mtp_dwconv_fwd(const void* x, const float* w, const float* bias, void* y, int dtype, int64_t N, int64_t H, int64_t W, int64_t C, int k, mtp_stream_t stream);
int mtp_dwconv_bwd_dx(const void* dy, int dtype, const float* w, float* dx, int accumulate, int64_t N, int64_t H, int64_t W, int64_t C, int k, mtp_stream_t stream);
int mtp_dwconv_bwd_dw(const void* dy, const void* x, int dtype, float* dw, float* db, int64_t N, int64_t H, int64_t W, int64_t C, int k, mtp_stream_t stream);
/* center_feature_scale (InternImage-H/G; ops_dcnv3/modules/dcnv3.py:80-88, 209-215): out (rows, G * GC) = y (1 - s) + xp s with s = sigmoid(logits[row][group]);
 * logits: rows of ld >= G elements (the G-output Linear on the depth-wise branch).  bwd: dy (`dtype`) = dout (1 - s); dxp (f32) = dout s; dlogits (rows of ld,
 * pad columns zeroed) = s (1 - s) x the sum over the group's channels of dout (xp - y). */
int mtp_center_feature_scale_fwd(const void* y, const void* xp, const void* logits, int64_t ld, void* out, int dtype, int64_t rows, int64_t G, int64_t GC, mtp_stream_t stream);
int mtp_center_feature_scale_bwd(const void* dout, const void* y, const void* xp, const void* logits, int64_t ld, void* dy, float* dxp, void* dlogits, int dtype, int64_t rows,
                                 int64_t G, int64_t GC, mtp_stream_t stream);
/* softmax over the P <= 32 sampling points of each of G groups (dcnv3.py:341-342): logits (rows, ld >= G*P) -> prob (rows, G*P).
 * bwd: dprob (rows, G*P) f32 -> dlogits (rows, ld) ACT, columns G*P .. ld zeroed. */
int mtp_softmax_groups_fwd(const void* logits, int64_t ld, void* prob, int dtype, int64_t rows, int64_t G, int64_t P, mtp_stream_t stream);
int mtp_softmax_groups_bwd(const void* prob, const float* dprob, void* dlogits, int64_t ld, int dtype, int64_t rows, int64_t G, int64_t P, mtp_stream_t stream);
/* layer scale + drop path + residual (intern_image.py:424-426): out (rows, C) f32 = x + sample_scale[row / rows_per_sample] *
 * gamma * z (z ACT; sample_scale may be NULL), out_act = the same in ACT (may be NULL).
 * bwd: dz (ACT) = sample_scale * gamma * dout; part (partial_rows, C) f32 = per-block partials of dgamma. */
int mtp_scale_residual_fwd(const float* x, const void* z, int dtype, const float* gamma, const float* sample_scale, int64_t rows_per_sample,
                           float* out, void* out_act, int64_t rows, int64_t C, mtp_stream_t stream);
int64_t mtp_scale_residual_bwd_partial_rows(int64_t rows);
int mtp_scale_residual_bwd(const float* dout, const void* z, int dtype, const float* gamma, const float* sample_scale, int64_t rows_per_sample,
                           void* dz, float* part, int64_t rows, int64_t C, mtp_stream_t stream);

/* ---- attention --------------------------------------------------------------------------------------------- */
/* Attention.forward core (VIT:97-108 + calc_rel_pos_spatial VIT:142-193): qkv (T,3C) ACT [q|k|v][head][hd] ->
 * o (T,C) ACT; lse (B, heads, N) f32.  logits = s*q.k + s*q.Rh[hq-hk+Hp-1] + s*q.Rw[wq-wk+Wp-1]. */
int mtp_full_attn_fwd(const void* qkv, void* o, float* lse, int dtype, const float* rel_h, const float* rel_w,
                      int64_t B, int64_t Hp, int64_t Wp, int64_t heads, int64_t hd, float scale, mtp_stream_t stream);
/* drel partials: (B*heads, 2*Hp-1 + 2*Wp-1, hd) f32, reduced by mtp_reduce_rows_f32.
 * Token grids of more than 256 tokens (448^2 pretraining inputs: 28 x 28) run a three-pass f32-math backward that keeps
 * per-query quantities in `workspace` (f32, mtp_full_attn_bwd_workspace_floats(...) elements; NULL / 0 for N <= 256). */
int64_t mtp_full_attn_bwd_workspace_floats(int64_t B, int64_t Hp, int64_t Wp, int64_t heads);
int mtp_full_attn_bwd(const void* qkv, const void* o, const void* dout, const float* lse, void* dqkv, int dtype,
                      const float* rel_h, const float* rel_w, float* drel_part, float* workspace,
                      int64_t B, int64_t Hp, int64_t Wp, int64_t heads, int64_t hd, float scale, mtp_stream_t stream);

/* RVSA sampling heads, stage 1 (VIT:347 zero pad, AvgPool2d(7,7), LeakyReLU): x (T,C) ACT -> avg, pooled (B*nh*nw, C) f32 */
int mtp_rvsa_pool_fwd(const void* x, int dtype, float* avg, float* pooled, int64_t B, int64_t Hp, int64_t Wp, int64_t C, mtp_stream_t stream);
/* dx (T,C) ACT += / = dpooled * leaky'(avg) / 49 broadcast over the window */
int mtp_rvsa_pool_bwd(const float* dpooled, const float* avg, void* dx, int dtype, int accumulate, int64_t B, int64_t Hp, int64_t Wp, int64_t C, mtp_stream_t stream);
/* The same two stages fused per window, one launch each way: avg, pooled (B*nh*nw, C) f32 and samp (B*nh*nw, N) f32 =
 * LeakyReLU(AvgPool(x)) . w^T + bias with w (N, C) f32 = the three heads stacked (N = 5 * heads);
 * backward: dx (T, C) ACT += (dsamp . w) * leaky'(avg) / 49 over each window's tokens (the weight / bias gradients stay with
 * mtp_small_linear_bwd, dx = NULL). */
int mtp_rvsa_sampling_fwd(const void* x, int dtype, const float* w, const float* bias, float* avg, float* pooled, float* samp,
                          int64_t B, int64_t Hp, int64_t Wp, int64_t C, int64_t N, mtp_stream_t stream);
int mtp_rvsa_sampling_bwd(const float* dsamp, const float* w, const float* avg, void* dx, int dtype,
                          int64_t B, int64_t Hp, int64_t Wp, int64_t C, int64_t N, mtp_stream_t stream);
/* the per-window factor of that update alone: g (windows, C) f32 = (dsamp . w) * leaky'(avg) / 49, for mtp_layernorm_bwd_win */
int mtp_rvsa_sampling_bwd_win(const float* dsamp, const float* w, const float* avg, float* g, int64_t windows, int64_t C, int64_t N, mtp_stream_t stream);
/* small f32 linear for the three 1x1 conv heads (VIT:231,236,242): y (R,N) = x (R,K) W(N,K)^T + b; and its backward */
int mtp_small_linear_fwd(const float* x, const float* w, const float* b, float* y, int64_t R, int64_t N, int64_t K, mtp_stream_t stream);
int mtp_small_linear_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int64_t R, int64_t N, int64_t K, mtp_stream_t stream);
/* weight / bias gradients of nseg <= 4 layers stacked along N (the three RVSA heads), ACCUMULATED into their own buffers:
 * dw[j] (rows_j, K) += dy[:, r0_j : r0_j + rows_j]^T x, db[j] (rows_j) += column sums.  seg_rows, dw, db: host arrays; db may be NULL. */
int mtp_small_linear_dw_segments(const float* x, const float* dy, int64_t R, int64_t N, int64_t K, int nseg, const int64_t* seg_rows,
                                 float* const* dw, float* const* db, mtp_stream_t stream);
/* the same for count <= 8 problems of one shape in ONE launch (the stacked heads of a burst of RVSA blocks, round 4): xs / dys are host arrays
 * of device pointers, dw / db host arrays of count * nseg device pointers, problem-major */
int mtp_small_linear_dw_segments_batched(const float* const* xs, const float* const* dys, int count, int64_t R, int64_t N, int64_t K, int nseg,
                                         const int64_t* seg_rows, float* const* dw, float* const* db, mtp_stream_t stream);
/* RotatedVariedSizeWindowAttention core (VIT:312-428): samp (B*nh*nw, 5*heads) f32 = [off(h,2)|scale(h,2)|angle(h)];
 * lse (B, heads, nh*nw, 49) f32 */
int mtp_rvsa_attn_fwd(const void* qkv, const float* samp, void* o, float* lse, int dtype,
                      const float* rel_h, const float* rel_w, const float* bias_table,
                      int64_t B, int64_t Hp, int64_t Wp, int64_t heads, int64_t hd, float scale, mtp_stream_t stream);
/* dqkv (T,3C) ACT: q part written directly; k/v parts are scattered through the bilinear weights with f32 atomics into
 * dkv_f32 (T, 2C) (zeroed by the callee) and then converted into dqkv by the callee.  dsamp (B*nh*nw, 5*heads) f32.
 * rel_part (B*nh*nw*heads, 26*hd) f32 = per-workgroup partials of [drel_h (13,hd) | drel_w (13,hd)];
 * tab_part (B*nh*nw, heads, 169) f32 = per-(window, head) partials of the bias-table gradient (the parameter is (169, heads):
 * the caller sums over the windows and transposes). */
int mtp_rvsa_attn_bwd(const void* qkv, const float* samp, const void* o, const void* dout, const float* lse,
                      void* dqkv, float* dkv_f32, float* dsamp, float* rel_part, float* tab_part, int dtype,
                      const float* rel_h, const float* rel_w, const float* bias_table,
                      int64_t B, int64_t Hp, int64_t Wp, int64_t heads, int64_t hd, float scale, mtp_stream_t stream);

/* ---- optimizer (the step recipe around the path: MAIN:424-457,783-788) ---------------------------------------- */
/* sum of squares of g[0:n] accumulated into *out (f32, zeroed by caller) -- for clip_grad_norm_ */
/* base[start[i] .. start[i] + count[i]) = 0, i < n; start / count: DEVICE arrays (the accumulating segments of a flat gradient buffer;
 * one workgroup per entry: split long runs) */
int mtp_zero_segments_f32(float* base, const int64_t* start, const int64_t* count, int n, mtp_stream_t stream);
int mtp_sqnorm_f32(const float* g, float* out, int64_t n, mtp_stream_t stream);
/* out += sum of squares over n runs base[start[i] .. start[i] + count[i]) (device tables, as mtp_zero_segments_f32): the share of the gradient norm that is not a
 * by-product of mtp_gemm_tn_grouped (mtp_gemm_args.workspace) */
int mtp_sqnorm_segments_f32(const float* base, const int64_t* start, const int64_t* count, int n, float* out, mtp_stream_t stream);
/* AdamW over a flat f32 buffer; per-segment weight decay via sorted seg_start[nseg] (element offsets) and seg_wd[nseg];
 * hyper (device, f32[6]) = {lr, beta1, beta2, eps, bias_corr1, bias_corr2}; clip_coef = min(1, max_norm / (sqrt(*sqnorm)+1e-6)) if sqnorm */
int mtp_adamw_flat(float* p, const float* g, float* m, float* v, int64_t n, const int64_t* seg_start, const float* seg_wd, int nseg,
                   const float* hyper, const float* sqnorm, float max_norm, float grad_scale, mtp_stream_t stream);

/* ---- DCNv3 core (InternImage) ------------------------------------------------------------------------
 * The reference's own native extension: `dcnv3_forward(input, offset, mask, kernel_h, kernel_w, stride_h, stride_w, pad_h,
 * pad_w, dilation_h, dilation_w, group, group_channels, offset_scale, im2col_step, remove_center) -> output` and
 * `dcnv3_backward(..., grad_output, im2col_step, remove_center) -> [grad_input, grad_offset, grad_mask]`
 * (Multi-Task_Pretrain/backbone/ops_dcnv3/src/vision.cpp:14-17, dcnv3.h:20-59; kernels src/cuda/dcnv3_im2col_cuda.cuh).
 * Same argument list, carried in one struct.  Tensors are the reference's: input (N, H, W, group*group_channels)
 * channels-last; offset (N, Ho, Wo, group*P*2) as (group, point, (x, y)); mask (N, Ho, Wo, group*P); output
 * (N, Ho, Wo, group*group_channels); P = kernel_h*kernel_w - remove_center, points ordered i*kernel_h + j with i over
 * kernel_w.  Differences, all on the caller-owns-buffers side of the boundary: outputs are passed in instead of being
 * allocated by the callee (dcnv3_cuda.cu:55-57, 131-133); dtypes are f32 and bf16 (the reference dispatches float / double /
 * half, :61, :150); gradients are always f32 (the reference promotes half to float the same way, :125-128).
 * im2col_step is validated as the reference does (batch % min(batch, im2col_step) == 0, dcnv3_cuda.cu:46-49) but does not
 * chunk the launch.  Errors: the reference's AT_ASSERTM exceptions become MTP_ERR_ARG. */
typedef struct {
    int64_t N, H, W;
    int kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w;
    int group, group_channels;
    float offset_scale;
    int im2col_step;
    int remove_center;
    int variant;   /* 0 = default; bit0 = plain workgroup order instead of one contiguous pixel range per XCD (A/B);
                    * bit1 = backward always as the per-corner f32-atomic scatter (default: the gather form where the geometry allows it);
                    * bit2 = the gather backward in its 3 x 3 form (dilation 1, offset_scale 1 or 2): 81 instead of 441 candidate samples per input
                    *        pixel, faster while the offsets stay below one pixel, slower beyond (its reach is narrower than the window form's);
                    * bit3 = the generic forward kernel also for 3 x 3 kernels (default there: the unrolled 9-point forward, round 6) */
} mtp_dcnv3_geom;
/* Ho = (H + 2 pad_h - (dilation_h (kernel_h - 1) + 1)) / stride_h + 1, Wo likewise (dcnv3_cuda.cu:40-45) */
int mtp_dcnv3_out_size(const mtp_dcnv3_geom* geom, int64_t* Ho, int64_t* Wo);
int mtp_dcnv3_fwd(const void* input, const void* offset, const void* mask, void* output, int dtype, const mtp_dcnv3_geom* geom, mtp_stream_t stream);
/* grad_input (N, H, W, C), grad_offset, grad_mask: f32, shaped like input / offset / mask; zero-filled by the callee where
 * the kernel accumulates into them.  grad_input: stride 1, "same" padding, 16-channel groups, <= 9 points (every InternImage
 * level) -> summed per input pixel from the output pixels around it, plain stores, atomics only for samples displaced by more
 * than a pixel beyond the kernel's reach; any other geometry -> bilinear scatter with f32 atomics, like the reference. */
int mtp_dcnv3_bwd(const void* input, const void* offset, const void* mask, const void* grad_output, int dtype, float* grad_input, float* grad_offset,
                  float* grad_mask, const mtp_dcnv3_geom* geom, mtp_stream_t stream);
/* dtype MTP_F64 (round 6; the reference dispatches float / double / half, dcnv3_cuda.cu:69, and its own test-suite checks gradients numerically in double,
 * ops_dcnv3/test.py): every operand, the output and -- through the same three pointers -- the three gradients are double; plain per-(pixel, group, channel)
 * kernels with double arithmetic and f64 atomics, any geometry.  A validation path, not a fast one. */
/* The same, and grad_offset once more in the input dtype as rows of act_ld >= group * P * 2 elements (pad columns zero): the operand the offset
 * head's dgrad / wgrad GEMMs read, written by the kernel that computes it instead of a cast-and-pad pass (round 4).  Only in the gather form of
 * the backward (every InternImage level); MTP_ERR_UNSUPPORTED otherwise, with nothing launched. */
int mtp_dcnv3_bwd_act(const void* input, const void* offset, const void* mask, const void* grad_output, int dtype, float* grad_input, float* grad_offset,
                      float* grad_mask, void* grad_offset_act, int64_t act_ld, const mtp_dcnv3_geom* geom, mtp_stream_t stream);

const char* mtp_version(void);

/* A non-blocking stream of the lowest priority the device offers, for launches that are off the critical path and should only take the CUs the
 * main stream leaves idle (the reference has no counterpart: its weight gradients are ATen calls on the one autograd stream).  The caller owns the
 * handle and frees it with mtp_stream_destroy. */
int mtp_stream_create_low_priority(mtp_stream_t* stream);
/* A stream restricted to the CUs of a bit mask (`words` 32-bit words; gfx950, SPX: bit i = XCC i % 8, CU i / 8 of that XCC).  The two half-batch
 * schedule runs each half of the batch on its own 128 CUs (16 of every XCC); the reference has no counterpart (one autograd stream,
 * main_pretrain.py:508-518).  MTP_ERR_ARG for a mask that leaves an XCC without a CU.  Freed with mtp_stream_destroy. */
int mtp_stream_create_cu_mask(const uint32_t* mask, int words, mtp_stream_t* stream);
/* Diagnostic: one record {XCC id, HW_ID register (cu_id [11:8], sh_id [12], se_id [15:13])} per workgroup of a `blocks`-workgroup launch on
 * `stream`, each workgroup resident for `spin_clocks` shader clocks; out = (blocks, 2) int32 in device memory. */
int mtp_probe_placement(int32_t* out, int blocks, int64_t spin_clocks, mtp_stream_t stream);
int mtp_stream_destroy(mtp_stream_t stream);

/* ---- gradient all-reduce over RCCL (reference: DistributedDataParallel, main_pretrain.py:508-518) ------ */
/* One communicator per process / GPU.  Rank 0 draws a 128-byte id (mtp_comm_unique_id) and hands it to every rank out of band;
 * mtp_comm_init is collective.  mtp_comm_allreduce_bucket: in-place SUM of `count` f32 values on `stream` (asynchronous; the
 * caller orders it against the kernels that produce / consume the bucket).  RCCL is resolved at run time: MTP_ERR_UNSUPPORTED when
 * no librccl can be loaded; RCCL's own error codes come back as 10000 + ncclResult_t. */
int mtp_comm_unique_id(void* id128);
int mtp_comm_init(const void* id128, int rank, int world, void** comm);
int mtp_comm_allreduce_bucket(void* comm, float* bucket, int64_t count, mtp_stream_t stream);
/* the same for a bucket of `dtype` (MTP_F32 / MTP_BF16: gradient buckets exchanged as bf16 move half the xGMI bytes) */
int mtp_comm_allreduce_bucket_dt(void* comm, void* bucket, int64_t count, int dtype, mtp_stream_t stream);
/* Direct exchange, in place (xGMI is point-to-point: every GPU owns 1 / world of the bucket): after mtp_comm_reduce_scatter_bucket
 * rank r holds the SUM of bucket[r * count_per_rank, (r + 1) * count_per_rank) (the other shards are unspecified);
 * mtp_comm_allgather_bucket spreads every rank's shard r back into all buckets.  Together = mtp_comm_allreduce_bucket on
 * world * count_per_rank elements (ncclReduceScatter + ncclAllGather; DistributedDataParallel's reducer, main_pretrain.py:508-518,
 * has no such mode). */
int mtp_comm_reduce_scatter_bucket(void* comm, void* bucket, int64_t count_per_rank, int rank, int dtype, mtp_stream_t stream);
int mtp_comm_allgather_bucket(void* comm, void* bucket, int64_t count_per_rank, int rank, int dtype, mtp_stream_t stream);
/* info4 = {ranks in the communicator, this rank, its device ordinal, RCCL's version code}; -1 where RCCL has no answer.  (The reference reads the
 * same from torch.distributed, main_pretrain.py:132-140.) */
int mtp_comm_info(void* comm, int* info4);
int mtp_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* MTP_HIP_H_ */
