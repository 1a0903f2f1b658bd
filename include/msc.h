/* msc.h — C-ABI of libmsc_hip.so: the MI355X (gfx950) kernels behind the multimodal
 * semantic-communication forward/backward path named by BASELINE.json `north_star`.
 *
 * Reference interface each entry replaces: NONE EXISTS. /root/reference/README.md:1-2 is the
 * whole upstream snapshot (SURVEY.md "finding zero", §8b), so there is no plugin/FFI surface to
 * mirror.  The subsystem names (ImageEncoder / BEV encoder / FusionTransformer / SemanticEncoder /
 * SemanticDecoder / Channel / policy-value head / DP all-reduce) come from BASELINE.json; each
 * group below says which of them it serves.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions (every entry point):
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers unless a comment
 *     says "host"; the caller owns every buffer; nothing here allocates, frees or synchronises.
 *   - `dtype` selects the storage/MFMA input type of activations and compute copies of weights:
 *       MSC_F32  = fp32 storage, exact-fp32 MFMA (v_mfma_f32_32x32x2_f32)   — parity mode
 *       MSC_BF16 = bf16 storage, bf16 MFMA with fp32 accumulation           — throughput mode
 *     statistics, losses, gradients of parameters and optimizer state are always fp32.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Work is enqueued only.
 *   - return value: MSC_OK (0) or a negative MSC_ERR_*; never throws, never prints.
 *   - activations are channels-last: images/feature maps NHWC, token tensors [B,T,D].
 *   - conv weights are OHWI = [Cout][KH][KW][Cin]; linear weights are [out][in].
 */
#ifndef MSC_H_
#define MSC_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { MSC_F32 = 0, MSC_BF16 = 1 };
enum { MSC_OK = 0, MSC_ERR_BAD_ARG = -1, MSC_ERR_LAUNCH = -2, MSC_ERR_UNSUPPORTED = -3 };
enum { MSC_ACT_NONE = 0, MSC_ACT_RELU = 1, MSC_ACT_GELU = 2, MSC_ACT_TANH = 3, MSC_ACT_SIGMOID = 4,
       /* msc_linear_fwd only: y = gelu(u) and the side output `pre` receives gelu'(u) instead of u — the derivative shares the
        * exponential of the erf evaluation, and the backward epilogue then multiplies instead of evaluating erf + exp per element */
       MSC_ACT_GELU_DSAVE = 5,
       /* msc_linear_dgrad's `dact` only: dact_ref holds the derivative itself (what MSC_ACT_GELU_DSAVE stored): dx *= dact_ref */
       MSC_ACT_MUL_REF = 6 };
enum { MSC_CH_AWGN = 0, MSC_CH_RAYLEIGH = 1 };

/* ABI version of this header: bumped whenever an entry point's parameter list changes (a C caller built against another value
 * must refuse to run: msc_abi_version() != MSC_ABI_VERSION).  3: msc_conv2d_dgrad gained relu_ref, msc_linear_dgrad gained res
 * (round 3, both in the middle of their lists); msc_set_knob / msc_get_knob / msc_debug_last_gemm added (round 4). */
#define MSC_ABI_VERSION 3

const char* msc_strerror(int code);
int msc_version(void);
int msc_abi_version(void);
/* Tuning / A-B knobs of the launch planner (csrc/gemm_plan.h `Knobs`: gemm_tile = bm*1000+bn, gemm_t128, conv_splitk, wgrad_blocks,
 * wgrad_small, wgrad_small_max, wgrad_xcd_split, dgrad_cls_interleave, gemm_pp, gemm_pp_mask, att_bwd_waves, att_dq_waves).  Initial values come
 * from the environment (MSC_<NAME>) ONCE, when the library is loaded; no launch reads the environment.  `name` is a host string.
 * msc_set_knob: MSC_ERR_BAD_ARG for an unknown name; msc_get_knob: INT64_MIN for an unknown name. */
int msc_set_knob(const char* name, int64_t value);
int64_t msc_get_knob(const char* name);
/* host_info8 (HOST pointer, 8 ints): the most recent contraction launch of this process — kind (1 one tile per workgroup, 2 persistent
 * 256-row tiles, 3 persistent ping-pong), tile rows, tile columns, threads per workgroup, K slices, xcd_split, cls_interleave,
 * extended-epilogue id.  Lets a test assert WHICH kernel variant a shape ran on. */
int msc_debug_last_gemm(int* host_info8);
/* sha256 (hex, first 16 digits) of the kernel sources + msc.h this library was built from (csrc/Makefile embeds it), so
   a test record can show that the loaded .so matches the checked-out sources */
const char* msc_build_id(void);
/* non-blocking HIP stream at a relative priority (-1 lowest, 0 default, +1 highest of the device's range); returns the
   hipStream_t as an integer, 0 on failure.  Used for the training step's side stream (weight gradients). */
int64_t msc_stream_create(int level);
int msc_stream_destroy(int64_t stream);
/* reusable hipEvent_t (timing disabled) as an integer, 0 on failure; msc_stream_wait_stream: `waiter` waits for all work
   enqueued so far on `signaler` (hipEventRecord + hipStreamWaitEvent on that event) */
int64_t msc_event_create(void);
int msc_stream_wait_stream(void* waiter, void* signaler, int64_t event);
/* the same in two halves, for a dependency signalled now and waited for later (staged input batches) */
int msc_event_record(int64_t event, void* stream);
int msc_stream_wait_event(void* waiter, int64_t event);
/* HOST-side blocking wait for the work recorded with `event` (frame stager's filler thread only; never part of a step) */
int msc_event_synchronize(int64_t event);
/* bytes of fp32 workspace msc_conv2d_wgrad / msc_linear_wgrad need for their split-K slabs */
int64_t msc_wgrad_workspace_bytes(int64_t rows_out, int64_t cols_out, int64_t reduce_len);

/* ---------------------------------------------------------------------------------------------
 * ImageEncoder / BEV conv stack: implicit-GEMM convolution (no im2col buffer).
 * x:[B,H,W,Cin] w:[Cout,KH,KW,Cin] y:[B,Ho,Wo,Cout], Ho=(H+2p-KH)/s+1.  Cin must be a power of two
 * >= 16/sizeof(T) (the RGB image is padded to 8 channels by msc_image_to_nhwc).
 * bn_partial (may be NULL): fp32 [msc_conv2d_fwd_stat_rows(B*Ho*Wo, Cout, dtype)][2][Cout] per-row-block column
 * sums / sums of squares of the fp32 accumulators, reduced in fixed order by msc_bn_finalize (deterministic). */
int64_t msc_conv2d_fwd_stat_rows(int64_t rows, int cout, int dtype);
/* ws / ws_bytes (may be NULL / 0): fp32 scratch of msc_conv2d_splitk_workspace_bytes(rows_out, cols_out, k_len, dtype)
 * bytes.  Launches that cannot fill the chip (deep layers: few output rows, long K) then run as split-K over 128x128
 * tiles + a fixed-order finish kernel that also writes bn_partial; without the workspace the un-split kernel runs.
 * forward: rows_out = B*Ho*Wo, cols_out = Cout, k_len = KH*KW*Cin; dgrad: rows_out = B*H*W, cols_out = Cin,
 * k_len = KH*KW*Cout (stride-1 dgrad only). */
int64_t msc_conv2d_splitk_workspace_bytes(int64_t rows_out, int cols_out, int64_t k_len, int dtype);
int msc_conv2d_fwd(const void* x, const void* w, void* y, float* bn_partial, int B, int H, int W,
                   int Cin, int Cout, int KH, int KW, int stride, int pad, float* ws, int64_t ws_bytes, int dtype,
                   void* stream);
/* ImageEncoder stem (7x7 / stride 2 / pad 3 over RGB) in space-to-depth form: the image re-blocked as xs [B][H/2][W/2][16]
 * (msc_image_to_s2d: channel (dy*2+dx)*3+c = pixel (2i+dy, 2j+dx), channels 12..15 zero) turns it into a 4x4 / stride-1
 * convolution with K = 256 instead of 392 (RGB padded to 8): 57 % instead of 37 % of the MFMA work is real.  Same sums, other
 * order.  w_s2d [Cout][4][4][16] from msc_stem_weight_s2d(conv1 fp32 OHWI [Cout][7][7][8]); the weight gradient dw_s2d fp32
 * [Cout][4][4][16] goes back into the OHWI layout with msc_stem_wgrad_unpack.  y: [B][H/2][W/2][Cout]; bn_partial and ws as for
 * msc_conv2d_fwd (rows = B*H/2*W/2) / msc_conv2d_wgrad (msc_wgrad_workspace_bytes(Cout, 256, rows)). */
int msc_image_to_s2d(const float* img_nchw, void* out, int B, int H, int W, int dtype, void* stream);
int msc_stem_weight_s2d(const float* w7, void* w_s2d, int Cout, int dtype, void* stream);
int msc_stem_wgrad_unpack(const float* dw_s2d, float* dw7, int Cout, int accumulate, void* stream);
int msc_stem_s2d_conv_fwd(const void* xs, const void* w_s2d, void* y, float* bn_partial, int B, int Hs, int Ws, int Cout,
                          int dtype, void* stream);
int msc_stem_s2d_conv_wgrad(const void* xs, const void* dy, float* dw_s2d, float* ws, int64_t ws_bytes, int B, int Hs, int Ws,
                            int Cout, int accumulate, int dtype, void* stream);
/* dx:[B,H,W,Cin] = ( conv-transpose of dy:[B,Ho,Wo,Cout] with wt:[Cin,KH,KW,Cout] (msc_permute3_cast)
 *                    + res ) * [relu_ref > 0]
 * res (optional T [B,H,W,Cin]): the gradient arriving over a residual/identity branch.
 * relu_ref (optional T [B,H,W,Cin]): this layer's INPUT when that input is the output of a ReLU — the ReLU's backward is applied
 * here, in the epilogue (or the split-K finish), so the producer of that input receives an already masked gradient and neither
 * re-reads its activation for the mask nor writes a separate masked copy for its identity branch. */
int msc_conv2d_dgrad(const void* dy, const void* wt, const void* res, const void* relu_ref, void* dx, int B, int H, int W,
                     int Cin, int Cout, int KH, int KW, int stride, int pad, float* ws, int64_t ws_bytes, int dtype, void* stream);
/* msc_conv2d_dgrad whose output dx is the gradient w.r.t. the output of a BatchNorm + ReLU (no residual): the epilogue also does the
 * STATISTICS PASS of that BatchNorm's backward.  bn_x: that BatchNorm's input (the previous convolution's output, [B,H,W,Cin]);
 * bn_stats: fp32 [4][Cin] mean, rstd, scale, shift (msc_bn_finalize); bn_partial: fp32 [msc_conv2d_dgrad_stat_rows(B*H*W, Cin, dtype)][2][Cin]
 * per-row-block sums of dz and dz*xhat, dz = dx * [bn_x*scale+shift > 0] (dx as stored, i.e. rounded to T).  dx itself is stored unmasked;
 * msc_bn_bwd_from_partials finishes the job.  Stride-1 (row-major tiled) launches only: MSC_ERR_UNSUPPORTED otherwise. */
int64_t msc_conv2d_dgrad_stat_rows(int64_t rows, int cin, int dtype);
int msc_conv2d_dgrad_bnstat(const void* dy, const void* wt, const void* res, void* dx, const void* bn_x, const float* bn_stats,
                            float* bn_partial, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                            float* ws, int64_t ws_bytes, int dtype, void* stream);
/* dw (fp32, OHWI) = sum over pixels dy^T x.  accumulate!=0 adds to dw.  ws: see msc_wgrad_workspace_bytes. */
int msc_conv2d_wgrad(const void* x, const void* dy, float* dw, float* ws, int64_t ws_bytes, int B, int H,
                     int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int accumulate,
                     int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * FusionTransformer / SemanticEncoder / SemanticDecoder / heads: dense layers.
 * y[M,N] = act(x[M,K] w[N,K]^T + bias[N]) + res[M,N];  pre (may be NULL) receives the
 * pre-activation (needed by GELU/tanh backward; act = MSC_ACT_GELU_DSAVE: the derivative gelu'(pre-activation) instead).
 * bias,res,pre optional.  bias is fp32. */
int msc_linear_fwd(const void* x, const void* w, const float* bias, const void* res, void* y, void* pre,
                   int M, int N, int K, int act, int dtype, void* stream);
/* dx[M,K] = (dy[M,N] wt[K,N]^T + res[M,K]) * act'(dact_ref[M,K])   (wt = w transposed, [K][N] row-major).
 * res (optional, T): a gradient w.r.t. this layer's input arriving over another branch (a tensor with two consumers), summed here
 * in fp32 before the one rounding of dx.
 * dact_ref/dact (optional): fused backward of the activation that FED this layer — dact_ref is that activation's
 * output (relu/tanh/sigmoid), pre-activation (gelu) or stored derivative (MSC_ACT_MUL_REF); dx is then the gradient w.r.t.
 * the previous layer's pre-activation.  colsum_partial (optional): fp32 [msc_linear_dgrad_colsum_rows(M,K,dtype)][K] per-row-block column sums
 * of dx (= the previous layer's bias gradient after msc_colsum_finalize). */
int64_t msc_linear_dgrad_colsum_rows(int64_t M, int K, int dtype);
int msc_linear_dgrad(const void* dy, const void* wt, const void* res, const void* dact_ref, int dact, float* colsum_partial, void* dx,
                     int M, int N, int K, int dtype, void* stream);
int msc_colsum_finalize(const float* partial, int nblk, int64_t cols, float* out, int accumulate, void* stream);
/* dw[N,K] (fp32) = dy[M,N]^T x[M,K];  db (optional, fp32 [N], N % 4 == 0) = column sums of dy = the bias gradient, produced by
 * the same launch (one extra MFMA per dy fragment against a ones fragment in the first column tile) — no separate pass over dy */
int msc_linear_wgrad(const void* x, const void* dy, float* dw, float* db, float* ws, int64_t ws_bytes, int M, int N,
                     int K, int accumulate, int dtype, void* stream);

/* General strided-batched GEMM used by the unfused attention path:
 *   C[z][m][n] = alpha * sum_k A(z,m,k) B(z,n,k)
 * a_kmajor/b_kmajor = 1: operand stored [row][k] (k contiguous); 0: stored [k][row] (row contiguous).
 * Leading dimensions must be multiples of 16 bytes; a row may end inside a 16-byte chunk provided the
 * padded tail up to ld is allocated and holds zeros (the softmax kernels write such pads).
 * batch index z = zo*batch_inner + zi; operand offset = zo*s?_o + zi*s?_i (in elements).
 * c_f32 != 0 writes fp32 output, else T. */
int msc_gemm_batched(const void* A, const void* Bm, void* C, int M, int N, int K, int64_t lda, int64_t ldb,
                     int64_t ldc, int a_kmajor, int b_kmajor, int batch, int batch_inner, int64_t sa_o,
                     int64_t sa_i, int64_t sb_o, int64_t sb_i, int64_t sc_o, int64_t sc_i, float alpha,
                     int c_f32, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * FusionTransformer: fused (flash-style) multi-head attention, head_dim 64, bf16 (MSC_ERR_UNSUPPORTED otherwise —
 * the exact-fp32 parity mode uses msc_gemm_batched + msc_softmax_*).  qkv: [B*T][3*H*64] (q | k | v, heads inside),
 * o: [B*T][H*64], lse: fp32 [B*H][msc_attention_tpad(T)] (log2-domain log-sum-exp, saved for the backward),
 * delta: fp32 scratch [2][B*H][msc_attention_tpad(T)] (the dQ pass leaves the row constants of the dK/dV pass there: -rowsum(dO*O)
 * and -lse/c, the initial values of that pass's accumulators), dqkv: [B*T][3*H*64].  No score matrix is written to HBM. */
int64_t msc_attention_tpad(int T);
int msc_attention_fwd(const void* qkv, void* o, float* lse, int B, int T, int H, int hd, float scale, int dtype,
                      void* stream);
int msc_attention_bwd(const void* qkv, const void* o, const void* dout, const float* lse, float* delta, void* dqkv,
                      int B, int T, int H, int hd, float scale, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * BatchNorm (training-mode batch statistics, per-GPU under DP) and LayerNorm.
 * msc_bn_finalize: mean/rstd [C] from the conv epilogue partials (fp64 combine, fixed order) and the folded
 *                  per-channel affine scale = gamma*rstd, shift = beta - mean*scale used by msc_bn_apply.
 * msc_bn_apply   : y = relu?(x*scale + shift + res?)                         x,y,res: T [rows][C]
 * msc_bn_bwd     : dz = dy*(y>0 if relu); dx (T); dres = dz (optional, T); dgamma/dbeta fp32
 *                  scale/shift (both or neither; from msc_bn_finalize): when given, the ReLU mask is rebuilt as
 *                  x*scale+shift > 0 and y is not read (may be NULL) — only valid when the forward had no residual input.
 *                  ws >= msc_bn_bwd_workspace_bytes(C). */
int msc_bn_finalize(const float* partial, int nblk, int C, int64_t count, float eps, const float* gamma,
                    const float* beta, float* mean, float* rstd, float* scale, float* shift, void* stream);
int msc_bn_apply(const void* x, const float* scale, const float* shift, const void* res, void* y, int64_t rows,
                 int C, int relu, int dtype, void* stream);
int64_t msc_bn_bwd_workspace_bytes(int C);
int msc_bn_bwd(const void* dy, const void* x, const void* y, const float* mean, const float* rstd,
               const float* gamma, const float* scale, const float* shift, void* dx, void* dres, float* dgamma,
               float* dbeta, float* ws, int64_t rows, int C, int relu, int accumulate, int dtype, void* stream);
/* msc_bn_bwd (relu = 1, mask from x*scale+shift, no dres) with the statistics pass already done by msc_conv2d_dgrad_bnstat:
 * partial = its bn_partial, nblk = its row count.  One finalize + one apply launch. */
int msc_bn_bwd_from_partials(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma, const float* scale,
                             const float* shift, const float* partial, int nblk, void* dx, float* dgamma, float* dbeta, float* ws,
                             int64_t rows, int C, int accumulate, int dtype, void* stream);
/* ImageEncoder stem, BN + ReLU + max-pool (KxK / S, pad P) fused: the post-activation map is never materialised.
 * fwd: x = conv output [B,H,W,C], scale/shift from msc_bn_finalize -> pooled y [B,Ho,Wo,C], arg-max tap ids idx (same values
 *      and tie-breaks as msc_bn_apply + msc_maxpool_fwd) and xsel [B,Ho,Wo,C] = the conv output at each window's arg-max.
 * bwd: dpool = gradient w.r.t. y.  Statistics pass on the pooled size (dpool, xsel; ReLU mask = xsel*scale+shift > 0), then one
 *      pass that gathers dpool through idx -> dx [B,H,W,C] (gradient w.r.t. the conv output), dgamma, dbeta.
 *      ws >= msc_bn_bwd_workspace_bytes(C). */
int msc_bn_relu_pool_fwd(const void* x, const float* scale, const float* shift, void* y, uint8_t* idx, void* xsel, int B, int H,
                         int W, int C, int K, int S, int P, int dtype, void* stream);
int msc_bn_relu_pool_bwd(const void* dpool, const uint8_t* idx, const void* xsel, const void* x, const float* mean,
                         const float* rstd, const float* gamma, const float* scale, const float* shift, void* dx, float* dgamma,
                         float* dbeta, float* ws, int B, int H, int W, int C, int K, int S, int P, int accumulate, int dtype,
                         void* stream);
/* out[i] = (accumulate ? out[i] : 0) + (a[i] + b[i]), fp32: sums the LayerNorm parameter gradients of the two half-batch chains */
int msc_add2_f32(const float* a, const float* b, float* out, int64_t n, int accumulate, void* stream);
int msc_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                      int64_t rows, int D, float eps, int dtype, void* stream);
int64_t msc_layernorm_bwd_workspace_bytes(int64_t rows, int D);
/* dx = LN-backward(dy) + dadd (dadd optional: gradient arriving over the residual branch).  ws receives the per-row-block partials
 * of dgamma / dbeta; with dgamma == dbeta == NULL they are NOT reduced here: msc_layernorm_bwd_params finishes them (fixed order,
 * the partials of ws_a then of the optional ws_b = the other half-batch chain) on whatever stream the caller likes — the
 * parameter gradients are needed by the optimizer only, so the training step keeps that launch off its critical chains. */
int msc_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                      const void* dadd, void* dx, float* dgamma, float* dbeta, float* ws, int64_t rows, int D,
                      int accumulate, int dtype, void* stream);
int msc_layernorm_bwd_params(const float* ws_a, int64_t rows_a, const float* ws_b, int64_t rows_b, int D, float* dgamma,
                             float* dbeta, int accumulate, void* stream);
/* out[c] (+)= sum_r x[r][c] — bias and positional-embedding gradients.  ws >= msc_colsum_workspace_bytes(rows, cols) */
int64_t msc_colsum_workspace_bytes(int64_t rows, int64_t cols);
int msc_colsum(const void* x, float* out, float* ws, int64_t rows, int64_t cols, int accumulate, int dtype,
               void* stream);

/* ---------------------------------------------------------------------------------------------
 * Layout, pooling, activation-backward, weight preparation, optimizer (all HBM-bound). */
int msc_image_to_nhwc(const float* img_nchw, void* out, int B, int C, int H, int W, int Cpad, int dtype,
                      void* stream);
/* Camera frames as stored / received: uint8 RGB pixel-interleaved [B][H][W][3] (device) -> fp32 planes [B][3][H][W] in [0,1]
 * (IEEE division by 255 — bit-equal to float32(u8) / float32(255)); H*W % 4 == 0.  The fp32 planes are what msc_image_to_nhwc and
 * msc_recon_loss read.  msc_memcpy_h2d_async: hipMemcpyAsync of `bytes` from pinned HOST memory (src_host) on `stream` — the
 * staging copy of the next batch (multimodal_sc_amd/frames.py). */
int msc_frames_u8_to_nchw(const uint8_t* frames_nhwc, float* img_nchw, int B, int H, int W, void* stream);
int msc_memcpy_h2d_async(void* dst, const void* src_host, int64_t bytes, void* stream);
int msc_maxpool_fwd(const void* x, void* y, uint8_t* idx, int B, int H, int W, int C, int K, int S, int P,
                    int dtype, void* stream);
int msc_maxpool_bwd(const void* dy, const uint8_t* idx, void* dx, int B, int H, int W, int C, int K, int S, int P,
                    int dtype, void* stream);
/* tokens[b] = concat(s0[b] (n0 rows), s1[b] (n1), s2[b] (n2)) + pos[T][D] */
int msc_tokens_assemble(const void* s0, const void* s1, const void* s2, const float* pos, void* out, int B, int n0,
                        int n1, int n2, int D, int dtype, void* stream);
int msc_tokens_split(const void* dout, void* d0, void* d1, void* d2, int B, int n0, int n1, int n2, int D,
                     int dtype, void* stream);
int msc_meanpool_fwd(const void* x, void* y, int B, int T, int D, int dtype, void* stream);
int msc_meanpool_bwd(const void* dy, void* dx, int B, int T, int D, int dtype, void* stream);
/* ref = output for relu/tanh/sigmoid, pre-activation for gelu, the derivative itself for MSC_ACT_MUL_REF */
int msc_act_bwd(const void* dy, const void* ref, void* dx, int64_t n, int act, int dtype, void* stream);
int msc_cast(const float* src, void* dst, int64_t n, int dtype, void* stream);

/* ---- OCP e4m3 fp8 path of the semantic encoder / decoder GEMMs (configs[4]) ----
 * Per-tensor quantisation by ONE workgroup (codec-sized tensors, n % 4 == 0, n <= 2^24): scale_out[0] = amax/448 (1 when
 * amax == 0); q[i] = rne_e4m3(x[i] * (448/amax)) as bytes; deq (optional) = q * scale in deq_dtype — the straight-through
 * operand the backward GEMMs use. */
int msc_fp8_quantize(const void* x, void* q, void* deq, float* scale_out, int64_t n, int in_dtype, int deq_dtype,
                     void* stream);
/* y[M][N] = act(sx[0]*sw[0] * (xq[M][K] . wq[N][K]^T) + bias) on v_mfma_f32_32x32x16_fp8_fp8; K % 32 == 0 */
int msc_linear_fp8_fwd(const void* xq, const void* wq, const float* sx, const float* sw, const float* bias, void* y,
                       int M, int N, int K, int act, int dtype, void* stream);
/* dst[k][j][i] = src[i][j][k]; OHWI -> [Cin][KH*KW][Cout] with (d0,d1,d2)=(Cout,KH*KW,Cin); linear W^T with d1=1 */
int msc_permute3_cast(const float* src, void* dst, int d0, int d1, int d2, int dtype, void* stream);
/* every weight transpose of a step in one launch.  table (device, int64 [n][6]) = {src_off, dst_off, d0, d1, d2,
 * first_block}: offsets in elements into the flat fp32 parameter buffer P / the flat transposed buffer TR;
 * first_block = running sum of ceil(d2/32)*ceil(d0/32)*d1; total_blocks = that sum over all entries. */
int msc_permute3_cast_batched(const float* P, void* TR, const int64_t* table, int n_entries, int total_blocks,
                              int dtype, void* stream);
/* torch.optim.Adam semantics; shadow (T, may be NULL) receives the updated low-precision copy */
/* step_dev (optional, device int64): step counter read by the kernel instead of `step` (survives hipGraph replay) */
int msc_adam_step(float* p, const float* g, float* m, float* v, void* shadow, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int step, const int64_t* step_dev, float grad_scale,
                  int dtype, void* stream);
int msc_counter_add(int64_t* ctr, int64_t inc, void* stream);

/* ---------------------------------------------------------------------------------------------
 * SemanticEncoder power normalisation + Channel (AWGN / Rayleigh).  x,y: T [B][k]; noise/fading:
 * fp32 [B][k] or NULL (then Philox4x32-10 with `seed`); snr_db: fp32 [B]; scale: fp32 [B] (saved). */
/* In-kernel noise: Philox4x32-10 with key = seed (lo, hi words; the host side puts the DP rank in the high word) and
 * counter = (block lo, block hi, stream_id + 2*draw_hi, draw_lo) with draw = `draw` (host-side count of forwards, eager
 * launches) + draw_dev[0] (optional device int64 the caller bumps with msc_counter_add after every forward — used where the
 * launch sequence is replayed from a hipGraph and a host value would be frozen): fresh noise per forward either way; seed, rank and draw never share a word, so (seed 1, draw 0) and (seed 0, draw 1) are different streams. */
int msc_channel_fwd(const void* x, const float* noise, const float* fading, const float* snr_db, void* y,
                    float* scale, int B, int k, int mode, uint64_t seed, uint64_t draw, const int64_t* draw_dev,
                    int dtype, void* stream);
int msc_channel_bwd(const void* dy, const void* x, const float* scale, void* dx, int B, int k, int dtype,
                    void* stream);
int msc_philox_normal(float* out, int64_t nquads, uint64_t seed, uint32_t stream_id, uint64_t draw, void* stream);

/* ---------------------------------------------------------------------------------------------
 * LiDAR / BEV encoder front end: point MLP (4->64, ReLU) + max over the points of each BEV cell -> [B][G][G][64].
 * Gather formulation: the forward bins each frame's points by cell (sorted_pts fp32 [B*Np][4], cell_start int32
 * [B][G*G+1] = msc_pillar_index_bytes, rank int32 [B*Np] scratch) and writes the grid in the storage type; the backward walks
 * the same bins (points outside the grid are dropped by the binning).  G*G <= 32768. */
int64_t msc_pillar_index_bytes(int B, int G);
int msc_pillar_fwd(const float* points, const float* w, const float* bias, float* sorted_pts, int32_t* cell_start, int32_t* rank,
                   void* grid_out, int B, int Np, int G, int C, float range_lo, float cell, int dtype, void* stream);
int64_t msc_pillar_bwd_workspace_bytes(void);
int msc_pillar_bwd(const float* sorted_pts, const int32_t* cell_start, const float* w, const float* bias, const void* dgrid,
                   float* dw, float* db, float* ws, int B, int G, int C, int accumulate, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Policy / value head, losses, attention softmax. */
/* Action selection (integer outputs are bit-exact): order = value descending, index ascending on exact ties (0.0 == -0.0
 * ties), NaN after everything else.  msc_topk_rows: out_idx int64 [rows][k], out_val fp32 [rows][k] (may be NULL), 1 <= k <= 8,
 * k <= n <= 32; k = 1 gives msc_argmax_rows.  oracle: oracle/rl.py topk_rows (stable lexicographic sort). */
int msc_argmax_rows(const void* x, int64_t* out, int64_t rows, int n, int64_t ld, int dtype, void* stream);
int msc_topk_rows(const void* x, int64_t* out_idx, float* out_val, int64_t rows, int n, int k, int64_t ld, int dtype,
                  void* stream);
/* head: T [B][ld], cols 0..A-1 logits, col A value.  terms fp32[4] = total, policy, value, entropy. */
int msc_ppo_loss(const void* head, const int64_t* actions, const float* old_logp, const float* adv,
                 const float* ret, float* terms, void* dhead, int B, int A, int ld, float clip_eps, float c_v,
                 float c_e, int dtype, void* stream);
int64_t msc_recon_loss_workspace_bytes(int B);
/* recon: T [B][G*G][P*P*3]; img fp32 NCHW; sse fp32 [B]; drecon = gscale*(recon-img) (may be NULL) */
int msc_recon_loss(const void* recon, const float* img_nchw, float* sse, void* drecon, float* ws, int B, int G,
                   int P, float gscale, int dtype, void* stream);
int msc_unpatchify(const void* recon, float* img_nchw, int B, int G, int P, int dtype, void* stream);
int msc_softmax_fwd(const float* S, void* P, int64_t rows, int n, int ld, int dtype, void* stream);
int msc_softmax_bwd(const float* dP, const void* P, void* dS, int64_t rows, int n, int ld, int dtype,
                    void* stream);

/* ---------------------------------------------------------------------------------------------
 * PPO loop around the path — device-side pieces; oracle: oracle/rl.py
 * msc_gae: rewards/dones [T][N], values [T+1][N] (last row = bootstrap) -> adv, ret [T][N] (fp32, unfused arithmetic:
 *   bit-exact against the oracle's recurrence).  msc_normalize: x <- (x-mean)/(std+eps), unbiased std, n <= 2^22.
 * msc_sample_actions: inverse-CDF categorical sample per row of the policy head ([B][ld], logits 0..A-1, value at A)
 *   with u = Philox4x32-10(counter = (row, step, 2), key = seed + seed_dev[0]); emits action, log pi(action), value. */
int msc_gae(const float* rewards, const float* values, const float* dones, float* adv, float* ret, int T, int N,
            float gamma, float lam, void* stream);
int msc_normalize(float* x, int64_t n, float eps, void* stream);
int msc_sample_actions(const void* head, int64_t* action, float* logp, float* value, int B, int A, int ld,
                       uint64_t seed, const int64_t* seed_dev, uint64_t step, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Wire format of the semantic symbols — oracle: oracle/wire.py
 * per frame: fp32 scale = amax/127 (1 if all zero) + k signed bytes I0,Q0,I1,Q1,..: q = clamp(rne(x/scale), -127, 127) */
int msc_symbols_pack(const void* x, void* q, float* scale, int B, int k, int dtype, void* stream);
int msc_symbols_unpack(const void* q, const float* scale, void* y, int B, int k, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Launch replay (host side of the training step; csrc/replay.hip, multimodal_sc_amd/replay.py).  A program is the recorded list
 * of int-returning entry-point calls of one eager step (function id + argument values, stream handles and event waits
 * included), cut into segments where the host must act between launches (the DP all-reduces).  msc_replay_run re-issues a
 * segment through the same entry points: one call from the host per segment instead of one per launch.  The caller keeps every
 * recorded buffer alive and unchanged in address (per-step bump arena); step / noise-draw numbers must be device-side counters.
 * iargs[j] carries integer and pointer arguments (as int64), fargs[j] float arguments (as double). */
#define MSC_REPLAY_MAX_ARGS 24
int msc_replay_fn_id(const char* name);              /* -1 when `name` is not a replayable entry point */
int msc_replay_fn_nargs(int fn_id);
int64_t msc_replay_create(void);
int msc_replay_destroy(int64_t prog);
int msc_replay_add(int64_t prog, int fn_id, const int64_t* iargs, const double* fargs, int nargs);
int msc_replay_cut(int64_t prog);                    /* closes the current segment; returns the index of the next one */
int64_t msc_replay_count(int64_t prog, int segment); /* records in a segment (all when segment < 0) */
int msc_replay_run(int64_t prog, int segment);       /* segment < 0: every segment */
int64_t msc_replay_last_failed(int64_t prog);

#ifdef __cplusplus
}
#endif
#endif /* MSC_H_ */
