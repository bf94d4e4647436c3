/*
 * favit.h -- C ABI of libfavit.so: the MI355X (gfx950 / CDNA4) kernels behind the
 * focused-attention ViT encoder forward/backward hot path.
 *
 * The reference (zser092/Focused-Attention-ViT) has NO native/FFI layer: its hot path
 * is the Python nn.Module surface of models/{vit,mhla,vit_mhla,sppp,sppp_mhla,attention}.py
 * executing aten ops.  This header is the new boundary UNDER that surface: each entry
 * point replaces the aten op sequence of the cited reference lines.  The host-side
 * mirror of the reference classes (the models package of focused-attention-vit_amd) binds these
 * with ctypes (focused-attention-vit_amd/_abi.py); INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - plain pointers + sizes only; every pointer is a DEVICE pointer owned by the caller
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is
 *     enqueued asynchronously on it, nothing synchronises, nothing allocates
 *   - return 0 on success, a negative FAVIT_ERR_* otherwise; never throws
 *   - dtype codes: FAVIT_F32 (exact fp32 path, f32 MFMA) / FAVIT_BF16 (bf16 operands,
 *     fp32 accumulation, bf16 MFMA) / FAVIT_FP8 (GEMM operands only: OCP e4m3 / e5m2 bytes
 *     with per-tensor scales, fp32 accumulation, fp8 MFMA)
 *   - kernels are stateless and thread-compatible
 */
#ifndef FAVIT_H_
#define FAVIT_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FAVIT_ABI_VERSION 8
#define FAVIT_FP8_AMAX_SLOTS 256   /* partial maxima per tensor with delayed fp8 scaling (favit_fp8_quantize) */

enum { FAVIT_F32 = 0, FAVIT_BF16 = 1, FAVIT_FP8 = 2 };
/* OCP 8-bit float formats of gfx950 (NOT the MI300X fnuz encodings) */
enum { FAVIT_E4M3 = 0, FAVIT_E5M2 = 1 };

enum {
  FAVIT_OK = 0,
  FAVIT_ERR_INVALID = -1,     /* bad argument (null pointer, non-positive size, bad enum) */
  FAVIT_ERR_UNSUPPORTED = -2, /* shape/dtype combination this build has no kernel for */
  FAVIT_ERR_ALIGN = -3,       /* pointer / leading dimension alignment requirement violated */
  FAVIT_ERR_LAUNCH = -4       /* hipLaunchKernel reported an error */
};

enum {
  FAVIT_ACT_NONE = 0,
  FAVIT_ACT_GELU = 1,           /* out = GELU(v); aux_out (optional) receives the pre-activation v                   */
  FAVIT_ACT_DGELU = 2,          /* out = v * GELU'(aux_in)            (aux_in = the saved pre-activation)            */
  FAVIT_ACT_GELU_SAVEGRAD = 3,  /* out = GELU(v); aux_out (required) receives GELU'(v) instead of v                  */
  FAVIT_ACT_MULAUX = 4          /* out = v * aux_in                   (aux_in = the saved GELU' of mode 3)           */
};
enum { FAVIT_POOL_MEAN = 0, FAVIT_POOL_MAX = 1, FAVIT_POOL_ATTENTION = 2 };

int favit_abi_version(void);
const char* favit_strerror(int code);

/* Dropout epoch: `device_word` (8-byte aligned device pointer, or NULL to switch the feature off) is read by every
 * kernel that draws a dropout mask -- effective seed = seed + *device_word * 0x9E3779B97F4A7C15 -- at EXECUTION time.
 * A training step captured once in a HIP graph has its dropout seeds frozen into the kernel arguments; with the
 * graph's first node incrementing the word, every replay still draws fresh masks, and the forward and backward
 * kernels of one replay (which recompute the same masks) agree.  Process-wide; the caller owns the word. */
int favit_set_dropout_epoch(const uint64_t* device_word);

/* Health word (v7): `device_words` = four zero-initialised uint32 on the device (16-byte aligned), or NULL to switch
 * it off.  favit_cross_entropy and favit_adamw -- which read every logit row / every gradient and write every
 * parameter anyway -- note the FIRST non-finite value they meet, at no memory traffic and without atomics in a clean
 * run:  [0] flags: 1 = a loss row of an in-range label was non-finite (non-finite logits), 2 = a gradient handed to
 * AdamW was non-finite, 4 = an updated parameter is non-finite;  [1] = (AdamW launches so far) + 1 when bit 1 was
 * first set, [2] = the same for bits 2 | 4, [3] = AdamW launches so far.  bench.py registers one and refuses to print
 * a metric line for a poisoned run; a harness can poll it every N steps without synchronising per step.  Process-wide;
 * the caller owns the words (and keeps them alive while captured HIP graphs hold their address). */
int favit_set_health_word(uint32_t* device_words);

/* ------------------------------------------------------------------------------------
 * GEMM with fused epilogue: every nn.Linear on the path and the dense QK^T / attn.V
 * contractions.   C[m,n] = epilogue( alpha * sum_k A[m,k] * B[n,k] )
 *   replaces: aten::addmm / mm / bmm behind models/vit.py:40,72,74,95,100,119,121,252,
 *             models/mhla.py:37,38, models/attention.py:30-33,63,75,131,144 and their
 *             autograd backward (dX = dY.W, dW = dY^T.X, db = colsum dY).
 * Operand layouts ("kmajor" = the contraction index is contiguous in memory):
 *   a_kmajor=1: A[m*lda + k]   a_kmajor=0: A[k*lda + m]
 *   b_kmajor=1: B[n*ldb + k]   b_kmajor=0: B[k*ldb + n]
 *   forward  Y = X.W^T      : A=X (kmajor), B=W (kmajor)
 *   dX = dY.W               : A=dY (kmajor), B=W with b_kmajor=0
 *   dW = dY^T.X             : A=dY with a_kmajor=0, B=X with b_kmajor=0, K = tokens
 * Epilogue order: v=alpha*acc; v+=bias[n]; aux_out=v; act; dropout; v+=residual; C (=|+=) v.
 *   act=GELU: v=gelu_erf(v) (models/vit.py:135); act=DGELU: v*=gelu'(aux_in[m,n]).
 *   dropout_p>0: v = keep(seed, m*N+n) ? v/(1-p) : 0  (nn.Dropout sites models/vit.py:102,
 *   136,138, models/mhla.py:159; the same (seed,index) draw is reused by the backward GEMMs).
 *   keep(seed, i): one 32-bit counter-based draw per element PAIR (i >> 1); element i takes its low (i even) or high
 *   16 bits and is kept when they are >= floor(p * 65536) -- p is realised to 1 / 65536, in every kernel of the library.
 * accumulate=1 (or split_k>1) adds into C with fp32 atomics; C must then be FAVIT_F32.
 * a_rowsum (a_kmajor=0 only): a_rowsum[m] += sum_k A[m,k]  (bias gradient, fused).
 * Batched: z in [0,batch): ptr += (z / batch_inner) * s?o + (z % batch_inner) * s?i.
 * in_dtype = FAVIT_FP8: A and B hold fp8 bytes (a_fp8_fmt / b_fp8_fmt = FAVIT_E4M3 | FAVIT_E5M2; B must
 *   be E4M3), both k-major, K a multiple of 64 and 16-byte aligned rows, batch = 1; the true operands
 *   are A*scale_a[0] and B*scale_b[0] (device scalars written by favit_fp8_quantize, NULL = 1), i.e.
 *   v = alpha*scale_a*scale_b*acc.  aux_in (DGELU) is bf16 in this mode.
 * ---------------------------------------------------------------------------------- */
typedef struct favit_gemm_t {
  const void* A;
  const void* B;
  void* C;
  const float* bias;     /* [N] fp32 or NULL */
  const void* aux_in;    /* DGELU: pre-activation [M,N], dtype = in_dtype; else NULL */
  void* aux_out;         /* optional copy of the pre-activation [M,N], dtype = out_dtype */
  const float* residual; /* [M,N] fp32 or NULL */
  float* a_rowsum;       /* [M] fp32 or NULL */
  int64_t M, N, K;
  int64_t lda, ldb, ldc, ld_aux_in, ld_aux_out, ld_res;
  int64_t sAo, sAi, sBo, sBi, sCo, sCi; /* batch strides in elements */
  int32_t batch, batch_inner;
  int32_t a_kmajor, b_kmajor;
  int32_t in_dtype, out_dtype;
  int32_t act;
  int32_t accumulate;
  int32_t split_k; /* 0 = library decides */
  float alpha;
  float dropout_p;
  int32_t fp8_fmt;        /* FAVIT_FP8 only: bit 0 = A is E5M2, bit 1 = B is E5M2 (unsupported) */
  uint64_t dropout_seed;
  const float* scale_a;   /* FAVIT_FP8 only: device scalars (dequantisation factors) or NULL */
  const float* scale_b;
} favit_gemm_t;

int favit_gemm(const favit_gemm_t* g, void* stream);
/* Diagnostic: the kernel family ("p4" 256x128 tiles, "p7" 256x256, "pp" ping-pong, "s64" 64-row, "t128"
 * 128x128 / exact-fp32) the calling host thread's last favit_gemm dispatched to.  Static string, never NULL. */
const char* favit_gemm_last_kernel(void);

/* LayerNorm fused into a small-M forward GEMM (v7):  C = epilogue( LN(x; gamma, beta, eps) . B^T )  in ONE launch, for
 * the short-token configurations where a block's forward is a chain of 5-15 us launches.  `g` describes the GEMM as for
 * favit_gemm with A ignored: bf16 k-major weights B [N, K], K = the LayerNorm width (a multiple of 64, <= 512, not 320 /
 * 448), every epilogue of favit_gemm (bias, GELU variants, dropout, residual), no split-K / batch / accumulate.
 * x: fp32 rows of ldx.  Also written: xn [M, K] bf16 (the normalised rows, the A operand of the weight-gradient GEMM),
 * mean [M], rstd [M] (fp32) for favit_layernorm_bwd.  Same arithmetic as favit_layernorm_fwd followed by favit_gemm
 * (two-pass variance, bf16 rounding of xn before the product); FAVIT_ERR_UNSUPPORTED for other shapes / dtypes (the
 * caller then issues the two launches). */
int favit_ln_gemm(const favit_gemm_t* g, const float* x, int64_t ldx, const float* gamma, const float* beta, float eps,
                  void* xn, float* mean, float* rstd, void* stream);

/* Grouped weight-gradient GEMMs: `count` (<= 48) problems dW_i = dY_i^T . X_i that share the token
 * dimension K (the four nn.Linear layers of one transformer block, or of several consecutive blocks) as
 * ONE launch.  Every problem must be bf16 in / fp32 out with a_kmajor = b_kmajor = 0, no epilogue other
 * than a_rowsum (bias gradient, always ADDED to its destination) and accumulate; returns
 * FAVIT_ERR_UNSUPPORTED otherwise (the caller then issues them one by one).  The number of K-splits is
 * chosen by a cost model (v7): with enough tiles in the launch it is 1, and then no workspace is used. */
int favit_gemm_grouped_tn(const favit_gemm_t* gs, int32_t count, void* stream);
/* The same launch with a caller-provided device workspace of favit_gemm_grouped_tn_workspace(gs, count) bytes:
 * every K-split writes its partial results to a slab of the workspace with plain stores and a second kernel adds
 * the slabs in a fixed order (no fp32 atomics: faster -- atomics run at ~1.3 TB/s chip-wide -- and bitwise
 * reproducible).  A NULL / too small workspace falls back to the atomic path. */
int64_t favit_gemm_grouped_tn_workspace(const favit_gemm_t* gs, int32_t count);
int favit_gemm_grouped_tn_ws(const favit_gemm_t* gs, int32_t count, void* workspace, int64_t workspace_bytes,
                             void* stream);
/* Diagnostic: K-splits of the calling host thread's last grouped launch (1 = tiles wrote dW directly). */
int favit_gemm_grouped_last_splits(void);

/* ------------------------------------------------------------------------------------
 * FP8 operand preparation (BASELINE.json configs[3] "fp8 MFMA path"; no reference counterpart:
 * the reference is fp32 only).  Per-tensor scaling, computed on the device:
 *   favit_fp8_amax:     amax[0] = max(amax[0], max |src[i]|)   (caller zeroes amax first)
 *   favit_fp8_quantize: q = sat(src * fmax / amax[0]) in OCP e4m3 (fmax 448) or e5m2 (fmax 57344),
 *                       round-to-nearest-even; scale_inv[0] = amax / fmax (the GEMM's scale_a/b).
 *     dst   [rows, ld_dst]   same orientation as src (NULL to skip)
 *     dst_t [cols, ld_t]     transposed copy (NULL to skip); columns rows..ld_t-1 are zero-filled,
 *                            so ld_t (a multiple of 64) can serve as a padded GEMM K
 *     colsum [cols] fp32     optional: colsum[c] += sum_r src[r,c] (bias gradient of an fp8 Linear)
 *     amax_next, amax_clear  optional, both or neither (delayed scaling).  Then amax, amax_next and amax_clear are
 *                            three DIFFERENT arrays of FAVIT_FP8_AMAX_SLOTS floats (rotating roles): the scale comes
 *                            from max(amax[0..]) -- what this site's previous call measured --, max |src| is taken in
 *                            the same pass into amax_next (atomics spread over the slots), amax_clear is zeroed
 *                            for the call after next.  No separate favit_fp8_amax pass.
 * src dtype is FAVIT_F32 or FAVIT_BF16, row stride ld_src (elements).
 * ---------------------------------------------------------------------------------- */
int favit_fp8_amax(const void* src, int src_dtype, int64_t rows, int64_t cols, int64_t ld_src, float* amax,
                   void* stream);
int favit_fp8_quantize(const void* src, int src_dtype, int64_t rows, int64_t cols, int64_t ld_src, void* dst,
                       int64_t ld_dst, void* dst_t, int64_t ld_t, int fmt, const float* amax, float* scale_inv,
                       float* colsum, float* amax_next, float* amax_clear, void* stream);

/* dst[i] = (dst_dtype) src[i] */
int favit_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------
 * LayerNorm over the last dim (nn.LayerNorm, eps 1e-5, biased variance):
 *   models/vit.py:155,157,251; models/vit_mhla.py:45,64,88,107,188,241.
 * x is the fp32 residual stream with row stride ldx (lets the head normalise x[:,0]
 * only); y has dtype y_dtype; mean/rstd [rows] are saved for backward.
 * Backward: dx = LN'(dy) (+ dres if given); optional low-precision copy dx_lp, optionally with the dropout mask
 * (lp_dropout_p, lp_dropout_seed; element index row*D + col, as favit_dropout) of the branch it feeds; the
 * affine gradients are produced as `nparts` partial sums in a [2][nparts][D] workspace
 * (dbeta_part = dgamma_part + nparts*D) and folded deterministically (no atomics) into dgamma[D]
 * and dbeta[D] (accumulate=1 adds to them); dgamma = NULL skips the fold.
 * ---------------------------------------------------------------------------------- */
int favit_layernorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, void* y, int y_dtype,
                        float* mean, float* rstd, int64_t rows, int32_t D, float eps, void* stream);
int favit_layernorm_bwd(const void* dy, int dy_dtype, const float* x, int64_t ldx, const float* gamma,
                        const float* mean, const float* rstd, const float* dres, float* dx, int64_t lddx,
                        void* dx_lp, int lp_dtype, float* dgamma_part, float* dbeta_part, int32_t nparts,
                        float* dgamma, float* dbeta, int32_t accumulate, int64_t rows, int32_t D,
                        float lp_dropout_p, uint64_t lp_dropout_seed, void* stream);
/* fp8 mode (ABI 8): the same two passes, the bf16 tensor they write (y / dx_lp) ALSO leaving as fp8 in q [rows, D]
 * (fmt = FAVIT_E4M3 | FAVIT_E5M2) with delayed scaling -- amax / scale_inv / amax_next / amax_clear exactly as in
 * favit_fp8_quantize (three different arrays of FAVIT_FP8_AMAX_SLOTS floats; all required).  Bytes, scale and
 * history are bit-identical to favit_layernorm_* followed by favit_fp8_quantize on the bf16 tensor; the stand-alone
 * pass (one read of the tensor + one write) is what is saved.  y and dy / dx_lp are bf16 here.  No reference
 * counterpart (the reference is fp32 only); consumers: the fp8 GEMMs of BASELINE.json configs[3]. */
int favit_layernorm_fwd_q8(const float* x, int64_t ldx, const float* gamma, const float* beta, void* y, float* mean,
                           float* rstd, int64_t rows, int32_t D, float eps, void* q, int fmt, const float* amax,
                           float* scale_inv, float* amax_next, float* amax_clear, void* stream);
int favit_layernorm_bwd_q8(const void* dy, const float* x, int64_t ldx, const float* gamma, const float* mean,
                           const float* rstd, const float* dres, float* dx, int64_t lddx, void* dx_lp,
                           float* dgamma_part, float* dbeta_part, int32_t nparts, float* dgamma, float* dbeta,
                           int32_t accumulate, int64_t rows, int32_t D, float lp_dropout_p, uint64_t lp_dropout_seed,
                           void* q, int fmt, const float* amax, float* scale_inv, float* amax_next, float* amax_clear,
                           void* stream);
/* The classification head as exact-fp32 dot products (ABI 8): y[M,N] = x[M,K] (row stride ldx) . w[N,K]^T + bias, and its
 * backward dx[M,K] (NULL: skipped) = dy . w, dw[N,K] (+)= dy^T . x, db[N] (NULL: skipped) (+)= column sums of dy
 * (accumulate = 1 adds to dw / db).  N <= 64 (FAVIT_ERR_UNSUPPORTED beyond: those heads are GEMMs); no atomics.
 * Reference: nn.Linear(embed_dim, num_classes) on the normalised CLS row, models/vit.py:259,306,
 * models/vit_mhla.py:156,249, models/sppp_mhla.py:240,318. */
int favit_small_linear_fwd(const float* x, int64_t ldx, const float* w, const float* bias, float* y, int32_t M, int32_t N,
                           int32_t K, void* stream);
int favit_small_linear_bwd(const float* dy, const float* x, int64_t ldx, const float* w, float* dx, float* dw, float* db,
                           int32_t accumulate, int32_t M, int32_t N, int32_t K, void* stream);
/* out[c] (+)= sum_r in[r*ld + c] */
int favit_reduce_rows(const float* in, int64_t ld, float* out, int64_t rows, int32_t cols, int32_t accumulate,
                      void* stream);
/* n (<= 32) stacked-pair reductions in ONE launch: entry e adds the column sums of in[e] ([2][rows][cols], contiguous)
 * to out0[e] / out1[e] ([cols] each).  Used for the dgamma / dbeta partials of several favit_layernorm_bwd calls
 * (each called with dgamma = NULL): the arguments are HOST arrays of n device pointers. */
int favit_reduce_rows_multi(int32_t n, const float* const* in, float* const* out0, float* const* out1, int64_t rows,
                            int32_t cols, void* stream);

/* ------------------------------------------------------------------------------------
 * MHLA (models/mhla.py).  latent_proj (mhla.py:41,105-106) is one Linear(hd,hd) shared
 * by K, V and all heads; it is folded algebraically into the qkv projection:
 *   Weff[s,h] = Wl . Wqkv[s,h] , beff[s,h] = Wl . bqkv[s,h] + bl   for s in {k,v}
 * so that one GEMM produces q, k~, v~.  fold_bwd maps the gradients of (Weff, beff)
 * back to the real parameters qkv.{weight,bias} and latent_proj.{weight,bias}.
 * ---------------------------------------------------------------------------------- */
int favit_mhla_fold_fwd(const float* wqkv, const float* bqkv, const float* wl, const float* bl, void* weff,
                        int weff_dtype, float* weff_f32 /* optional fp32 copy */, float* beff, int32_t D,
                        int32_t H, void* stream);
/* accumulate=1: the four outputs are added to (gradient buffers), else overwritten */
/* The forward fold of n (<= 32) independent blocks -- e.g. every layer of an encoder -- in ONE launch.
 * The arguments are HOST arrays of n device pointers; weff[i] / beff[i] as in favit_mhla_fold_fwd. */
int favit_mhla_fold_fwd_multi(int32_t n, const float* const* wqkv, const float* const* bqkv, const float* const* wl,
                              const float* const* bl, void* const* weff, int weff_dtype, float* const* beff,
                              int32_t D, int32_t H, void* stream);

int favit_mhla_fold_bwd(const float* dweff, const float* dbeff, const float* wqkv, const float* bqkv,
                        const float* wl, float* dwqkv, float* dbqkv, float* dwl, float* dbl, int32_t D, int32_t H,
                        int32_t accumulate, void* stream);
/* The backward fold of n (<= 16) layers in ONE launch, always ACCUMULATING into the gradient buffers (HOST arrays of n
 * device pointers, as favit_mhla_fold_fwd_multi).  dwqkv[i] = dbqkv[i] = NULL for every layer: frozen qkv projection
 * (the fine-tuning setup of experiments/sppp_mhla_pretrained.py:236-247), only dwl / dbl are produced. */
int favit_mhla_fold_bwd_multi(int32_t n, const float* const* dweff, const float* const* dbeff, const float* const* wqkv,
                              const float* const* bqkv, const float* const* wl, float* const* dwqkv,
                              float* const* dbqkv, float* const* dwl, float* const* dbl, int32_t D, int32_t H,
                              void* stream);

/* Windowed attention core: window index rule (mhla.py:46-83, closed form in-kernel, the
 * duplicated pad indices take part in the softmax), gather (117-126, never materialised),
 * scores / sqrt(hd) (130-133), optional mask[B,L,L]!=0 keep (136-143), softmax + dropout
 * (146-147), attn.V (151-154), head merge (157).  qkv is the [B*L, 3D] output of the folded
 * projection (column = s*D + h*hd + d); out is [B*L, D] (column = h*hd + d).
 * Backward recomputes the probabilities and writes dqkv [B*L, 3D]. */
int favit_mhla_attn_fwd(const void* qkv, void* out, const uint8_t* mask, int32_t B, int32_t L, int32_t H,
                        int32_t hd, int32_t W, int dtype, float dropout_p, uint64_t seed, void* stream);
int favit_mhla_attn_bwd(const void* qkv, const void* dout, void* dqkv, const uint8_t* mask, int32_t B, int32_t L,
                        int32_t H, int32_t hd, int32_t W, int dtype, float dropout_p, uint64_t seed,
                        void* stream);
/* The same attention core with the forward's softmax statistics handed to backward (ABI 6): `lse` fp32 [B, H, L] =
 * log sum_w exp(s_w) of every row's window (pad copies counted, mask applied), `o` = the forward's output.  With them
 * backward recomputes no softmax for rows outside a workgroup's block (P = exp(s - lse); delta = dO . O for the halo
 * rows: o is in bf16, so delta of those rows carries its rounding, ~2^-9 relative) -- the cfg2 launch takes 66 us
 * against 79.  favit_mhla_attn_lse_supported: 1 where both entry points run (bf16, hd = 64, odd W <= 7, or <= 11 with
 * L > 16; L >= W + 1), else the callers use the pair above. */
int favit_mhla_attn_lse_supported(int32_t L, int32_t hd, int32_t W, int dtype);
int favit_mhla_attn_fwd_lse(const void* qkv, void* out, float* lse, const uint8_t* mask, int32_t B, int32_t L, int32_t H,
                            int32_t hd, int32_t W, int dtype, float dropout_p, uint64_t seed, void* stream);
int favit_mhla_attn_bwd_lse(const void* qkv, const void* dout, const void* o, const float* lse, void* dqkv,
                            const uint8_t* mask, int32_t B, int32_t L, int32_t H, int32_t hd, int32_t W, int dtype,
                            float dropout_p, uint64_t seed, void* stream);

/* ------------------------------------------------------------------------------------
 * Row softmax for the dense attention variants (models/vit.py:96, attention.py:71,140,
 * nn.MultiheadAttention): S fp32 [Z, Lq, Lk] (already scaled by the GEMM alpha).
 * mask (uint8, 0 = -inf) is addressed mask[(z / H) * m_sb + q * m_sq + k].
 * P (dtype p_dtype) gets softmax(S); if dropout_p>0, Pd gets the dropped+rescaled P.
 * Backward: dS = P * (dPd*keep/(1-p) - sum_k(...)) in ds_dtype.
 * ---------------------------------------------------------------------------------- */
int favit_softmax_fwd(const float* S, void* P, void* Pd, int p_dtype, const uint8_t* mask, int64_t m_sb,
                      int64_t m_sq, int32_t H, int64_t Z, int32_t Lq, int32_t Lk, float dropout_p, uint64_t seed,
                      void* stream);
int favit_softmax_bwd(const void* P, int p_dtype, const float* dPd, void* dS, int ds_dtype, int64_t Z, int32_t Lq,
                      int32_t Lk, float dropout_p, uint64_t seed, void* stream);

/* ------------------------------------------------------------------------------------
 * Device-side input transforms (reference utils/data_utils.py:21-81): crop (+ zero padding) ->
 * optional flip -> Pillow-exact bilinear resize (8-bit two-pass resampling, 22-bit fixed-point coefficients:
 * bit-identical to PIL.Image.resize(BILINEAR), which is what torchvision's Resize / RandomResizedCrop run on PIL
 * images) -> output window -> ToTensor (/255) -> Normalize, for a batch of raw uint8 HWC images on the device.
 *   src [B,Hs,Ws,C] uint8; params int32 [B,12] (device) = crop top, left, h, w (in the padded source), pad,
 *   resized h, w, window origin y, x, flip_src, flip_out, 0; tmp uint8 [B,ch_max,S,C] workspace (ch_max >= every
 *   crop height); out fp32 [B,C,S,S]; out_u8 (optional) the resized bytes [B,S,S,C]; mean / std: HOST arrays [C].
 * ---------------------------------------------------------------------------------- */
int favit_image_transform(const uint8_t* src, uint8_t* tmp, float* out, uint8_t* out_u8, const int32_t* params, int32_t B,
                          int32_t Hs, int32_t Ws, int32_t C, int32_t ch_max, int32_t S, const float* mean,
                          const float* std, void* stream);

/* ------------------------------------------------------------------------------------
 * SLIC superpixels on the device: replaces the per-image D2H -> skimage.segmentation.slic -> H2D
 * hop of SuperpixelSegmentation.segment (reference models/sppp.py:44-74).  scikit-image is an unpinned third-party
 * dependency absent from the image: parity with skimage is UNPINNED; the algorithm (csrc/slic.hip header) is the
 * published SLIC as skimage parametrises it, integer-exact after the colour conversion, and is checked against
 * the CPU restatement oracle/slic_oracle.py.
 *   features: img fp32 [B,3,H,W] -> feat int16 [B,H*W,4] = round(16 * CIELAB(gaussian_sigma(rescale(img)))) clamped to +-8191, lane 3 = 0;
 *             rescale != 0: every image is first rescaled to [0, 1] by its own minimum and maximum over all channels,
 *             as scikit-image >= 0.19 does before smoothing (the reference passes mean/std-normalised tensors,
 *             models/sppp_mhla.py:278); 0 = no rescale (scikit-image < 0.19).  ws: device workspace of
 *             favit_slic_features_workspace(B, H, W) bytes (min / max per image and the horizontal pass of the
 *             separable gaussian)
 *   cluster : (feat values must lie in [-8191, 8191], as `features` produces them: the distance arithmetic relies on it)
 *             k-means, K <= 64 centres seeded at init_yx [K,2] (int32 y, x), window +-2*step, distance
 *             256*spatial^2 + coef*dq^2 (int64), `iters` rounds -> labels uint8 [B,H*W]; ws: 8-byte aligned device
 *             workspace of favit_slic_cluster_workspace(K, B) bytes (per-centre sums and coordinates)
 *   connect : 4-connected components in raster order, components < min_size merged into a labelled neighbour,
 *             consecutive labels from 0 -> out int64 [B,H*W]; n_regions int32 [B] (-1: more than 2048 components,
 *             out = the cluster map); ws_comp / ws_aux: int32 [B,H*W] workspaces.
 * ---------------------------------------------------------------------------------- */
int64_t favit_slic_features_workspace(int32_t B, int32_t H, int32_t W);
int favit_slic_features(const float* img, int16_t* feat, int32_t B, int32_t H, int32_t W, float sigma, int32_t rescale,
                        float* ws, void* stream);
int64_t favit_slic_cluster_workspace(int32_t K, int32_t B);
int favit_slic_cluster(const int16_t* feat, uint8_t* labels, const int32_t* init_yx, int32_t K, int32_t B, int32_t H,
                       int32_t W, int32_t step, int64_t coef, int32_t iters, void* ws, void* stream);
int favit_slic_connect(const uint8_t* labels, int32_t* ws_comp, int32_t* ws_aux, int64_t* out, int32_t* n_regions,
                       int32_t B, int32_t H, int32_t W, int32_t min_size, void* stream);

/* ------------------------------------------------------------------------------------
 * Fused scaled-dot-product attention for the dense variants (replaces, on one kernel family, the reference's
 *   scores = q @ k^T * scale -> masked_fill(mask == 0, -inf) -> softmax -> dropout -> @ v
 * of models/vit.py:95-100 (MultiHeadAttention), models/attention.py:63-75 (CrossAttention, one head,
 * scale 1/sqrt(embed_dim)) and 131-144 (MultiHeadCrossAttention), and of the nn.MultiheadAttention branch
 * models/vit_mhla.py:57-62).  No [B*H, Lq, Lk] tensor is written to memory.
 * Element (b, h, l, d) of a matrix lives at ptr + b*str[1] + h*str[2] + l*str[0] + d (strides in elements,
 * str = {row stride, batch stride, head stride}; multiples of 16 bytes), so q / k / v can be column blocks of
 * one fused projection output and o / dq / dk / dv are written in place, head-merged.
 * mask (uint8, 0 = -inf, NULL = none) is addressed mask[b*m_sb + q*m_sq + k]  (key-keep [B,Lk]: m_sq = 0).
 * fwd writes o and lse [B*H, Lq] (row log-sum-exp of the scaled, masked scores); bwd needs q, k, v, o, dout,
 * lse and a workspace delta [B*H, Lq], and writes dq, dk, dv (probabilities are recomputed; deterministic).
 * dropout: inverted, counter-based on (seed, ((b*H + h)*Lq + q)*Lk + k), the same draw in fwd and bwd.
 * dtype FAVIT_BF16 (bf16 MFMA, fp32 accumulate / softmax) or FAVIT_F32 (exact-fp32 MFMA); hd % 16 == 0.
 * ---------------------------------------------------------------------------------- */
typedef struct favit_sdpa {
  const void* q;
  const void* k;
  const void* v;
  void* o;               /* fwd: output; bwd: input */
  float* lse;            /* fwd: output; bwd: input */
  const void* dout;      /* bwd only from here */
  void* dq;
  void* dk;
  void* dv;
  float* delta;
  const uint8_t* mask;
  int64_t m_sb, m_sq;
  int64_t q_str[3], k_str[3], v_str[3], o_str[3], do_str[3], dq_str[3], dk_str[3], dv_str[3];
  int32_t B, H, Lq, Lk, hd, dtype;
  float scale, dropout_p;
  uint64_t seed;
} favit_sdpa_t;
int favit_sdpa_fwd(const favit_sdpa_t* s, void* stream);
int favit_sdpa_bwd(const favit_sdpa_t* s, void* stream);

/* ------------------------------------------------------------------------------------
 * Patch embedding front end.
 * patchify: einops 'b c (h p1) (w p2) -> b (h w) (p1 p2 c)' (models/vit.py:38-39),
 *   img fp32 NCHW -> rows [B*N, P*P*C] of out_dtype (channel fastest).
 * embed_prologue: cat(cls, tokens) + pos_embed (models/vit.py:292-296,
 *   models/vit_mhla.py:229-233); pos may be NULL (SPPP path adds its own encoding).
 * ---------------------------------------------------------------------------------- */
int favit_patchify_fwd(const float* img, void* out, int out_dtype, int32_t B, int32_t C, int32_t HW, int32_t P,
                       void* stream);
int favit_patchify_bwd(const float* dpatch, float* dimg, int32_t B, int32_t C, int32_t HW, int32_t P, void* stream);
int favit_embed_prologue_fwd(const float* tok, const float* cls, const float* pos, float* x, int32_t B, int32_t N,
                             int32_t D, void* stream);
/* dtok (dtype dtok_dtype, [B*N, D], feeds the patch-embedding weight-gradient GEMM), dcls, dpos may be NULL */
int favit_embed_prologue_bwd(const float* dx, void* dtok, int dtok_dtype, float* dcls, float* dpos, int32_t B,
                             int32_t N, int32_t D, void* stream);

/* Inverted dropout with a counter-based RNG (nn.Dropout sites, models/vit.py:136,138,
 * 102, mhla.py:159); the mask is recomputed from (seed, index) in backward. */
int favit_dropout(const void* x, void* y, int dtype, int64_t n, float p, uint64_t seed, void* stream);

/* ------------------------------------------------------------------------------------
 * SPPP (models/sppp.py, models/sppp_mhla.py); the label map [B,HW,HW] int64 is an input.
 * map_patches (sppp.py:91-128): dominant label per patch (max count, ties -> smallest
 *   label), token rank = first-appearance order of the dominant labels in raster scan.
 *   Outputs: patch_rank[B,N] int32, n_tokens[B] int32, perm[B,N] int32 (patches grouped
 *   by rank, raster order inside), offs[B,N+1] int32 (group offsets into perm).
 * pool (sppp.py:192-223): mean / max / 'attention' pooling of patch embeddings
 *   emb[B,N,D] fp32 into tokens [B,R,D] fp32 (R = the common token count).
 * centroids (sppp_mhla.py:226-262): per LABEL s<S mean (x/w, y/h), empty -> 0.5.
 * posenc (sppp.py:267-300): x + [sin(cx*f), cos(cy*f)], CLS centroid (0.5,0.5)
 *   prepended when n_cent < L; cent == NULL selects the index-sinusoid branch (sppp.py:257-266).
 * ---------------------------------------------------------------------------------- */
/* dom_ws: caller-provided workspace [B,N] int64, receives the dominant label of every patch */
int favit_sppp_map_patches(const int64_t* seg, int32_t* patch_rank, int32_t* n_tokens, int32_t* perm,
                           int32_t* offs, int64_t* dom_ws, int32_t B, int32_t HW, int32_t P, void* stream);
int favit_sppp_pool_fwd(const float* emb, const int32_t* perm, const int32_t* offs, float* out, int32_t* argmax,
                        int32_t kind, int32_t B, int32_t N, int32_t R, int32_t D, void* stream);
int favit_sppp_pool_bwd(const float* dout, const float* emb, const int32_t* patch_rank, const int32_t* perm,
                        const int32_t* offs, const int32_t* argmax, float* demb, int32_t kind, int32_t B,
                        int32_t N, int32_t R, int32_t D, void* stream);
int favit_sppp_centroids(const int64_t* seg, float* cent, int32_t B, int32_t HW, int32_t S, void* stream);
int favit_sppp_posenc_fwd(const float* x, const float* cent, float* y, int32_t B, int32_t L, int32_t D,
                          int32_t n_cent, void* stream);

/* ------------------------------------------------------------------------------------
 * Harness-side pieces of the training step (experiments/mhla_pretrained.py:363-367):
 * mean cross-entropy with its gradient, and a fused multi-tensor AdamW step.
 * ---------------------------------------------------------------------------------- */
int favit_cross_entropy(const float* logits, const int64_t* labels, float* loss_rows, float* dlogits, int32_t B,
                        int32_t C, float grad_scale, void* stream);
/* p_bf16 (optional, [n] bf16): refreshed compute-dtype copy of the updated parameters */
int favit_adamw(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr, float beta1,
                float beta2, float eps, float weight_decay, float bias_c1, float bias_c2, float grad_scale,
                void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FAVIT_H_ */
