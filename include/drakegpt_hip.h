/*
 * drakegpt_hip.h -- C ABI of libdrakegpt_hip.so, the MI355X (gfx950) kernels behind
 * DrakeGPT's transformer training hot path.
 *
 * The reference (ChrisTho23/DrakeGPT) has no native code and no FFI: every number is produced
 * by stock ATen ops called from src/model.py and src/model_component.py (SURVEY.md section 2b
 * lists the call sites K1..K19).  Each entry point below therefore names the reference call
 * site(s) whose arithmetic it replaces; the Python layer in drakegpt_amd/ binds these symbols
 * with ctypes and presents the reference's nn.Module surface on top.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless the name ends in _host; the library allocates
 *    nothing and owns nothing: workspaces are passed in by the caller;
 *  - `stream` is a hipStream_t; calls only enqueue work and return immediately, so they may be
 *    captured into a hipGraph;
 *  - all matrices are row-major with an explicit leading dimension in ELEMENTS;
 *  - dtype codes: DG_F32 = 0 (float), DG_BF16 = 1 (bfloat16, round-to-nearest-even);
 *  - return value: 0 on success, a negative DG_ERR_* for a rejected argument, or a positive
 *    hipError_t from the launch; dg_error_string() explains either;
 *  - no global mutable state: safe to call from the autograd worker thread and the main
 *    thread concurrently (forward runs on one, backward on the other).
 *
 * Dropout stream: a keep decision is a stateless hash of (seed, step, site, element index)
 * (csrc/common.h: dg_keep): elements 2i and 2i + 1 read the low / high 16 bits of one 32-bit
 * word hash(key, i) and are kept iff that field >= floor(p * 65536).
 * `rng_state` points at 4 device uint32: {seed_lo, seed_hi, step, 0 (scratch of dg_adamw_step's advance)}; the step word is
 * advanced on the device (dg_state_advance, or inside dg_adamw_step) so a captured graph replays with fresh masks.  A NULL rng_state or p == 0 disables dropout (eval mode).
 */
#ifndef DRAKEGPT_HIP_H
#define DRAKEGPT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DG_F32 0
#define DG_BF16 1
#define DG_FP8_E4M3 2      /* OCP e4m3fn (gfx950's fp8; NOT MI300's fnuz encoding), max 448 */
#define DG_FP8_E5M2 3      /* OCP e5m2, max 57344 */

#define DG_OK 0
#define DG_ERR_ARG (-1)       /* bad size / null pointer / unsupported combination */
#define DG_ERR_ALIGN (-2)     /* pointer or leading dimension not 16-byte aligned */
#define DG_ERR_DTYPE (-3)

#define DG_ABI_VERSION 20   /* bump whenever a signature or struct of this header changes: the Python binding refuses a stale library */

int dg_version(void);
const char* dg_error_string(int code);

/* {seed_lo, seed_hi, step, 0}: step += 1.  One launch of one thread. */
int dg_state_advance(uint32_t* rng_state, void* stream);

/* ---------------------------------------------------------------------------------------
 * get_batch -- ref: src/preprocessing.py:43-45.  The corpus stays resident in HBM; the B window
 * offsets are still drawn by the host CPU generator (torch.randint) for parity and passed in.
 * x[b, t] = corpus[off[b] + t], y[b, t] = corpus[off[b] + t + 1].  All int64. */
int dg_batch_gather(const int64_t* corpus, int64_t n_corpus, const int64_t* offsets,
                    int64_t* x, int64_t* y, int B, int T, void* stream);

/* ---------------------------------------------------------------------------------------
 * Token + position embedding -- ref: src/model.py:595-597 (K1-K3).
 * x[b,t,:] = tok[idx[b,t],:] + pos[t,:]   (pos == NULL: BigramLM-style plain gather, :96).
 * T <= rows of pos is the caller's job to check.  idx values are clamped to [0, V) on the device so that a bad id can
 * never fault the GPU; torch raises IndexError instead, and so does the Python layer, at the points where ids enter
 * (drakegpt_amd.ops.check_ids: module forward / generate, TrainEngine.set_corpus / set_batch) -- never inside the step. */
int dg_embed_fwd(const int64_t* idx, const float* tok, const float* pos, float* x,
                 int B, int T, int C, int V, void* onehot, int64_t ld_onehot, void* stream);
/* onehot (nullable): bf16 [B*T, ld_onehot >= V, multiple of 8], row m = e_{idx[m]}.  With it the token-table gradient
 * is dg_gemm_tn_grouped problem (A = onehot, B = bf16 dx): deterministic, no atomics. */
/* get_batch and the embedding in ONE launch (ref: src/preprocessing.py:43-45 + src/model.py:595-597): the ids are gathered
 * from the resident corpus at this step's window offsets and x_ids / y_ids [B, T] (the batch and its targets) are written
 * beside x.  `offsets` is a staged block [n_rows, B] of host-drawn offsets; the row is step_state[2] - ctl[0], clamped to
 * ctl[1] rows (ctl: 2 device uint32 {step word at staging time, n_rows}; step_state as in the header comment), so a captured
 * step walks through the block without any per-step host copy.  ctl == step_state == NULL: row 0. */
int dg_batch_embed_fwd(const int64_t* corpus, int64_t n_corpus, const int64_t* offsets, const uint32_t* step_state,
                       const uint32_t* ctl, int64_t* x_ids, int64_t* y_ids, const float* tok, const float* pos, float* x,
                       int B, int T, int C, int V, void* onehot, int64_t ld_onehot, void* stream);
/* Backward of dg_embed_fwd (embedding_dense_backward).  dtok [V,C] (nullable) is zero-filled here and then
 * accumulated with fp32 atomics; dpos [T,C] (nullable) is overwritten with sum over b. */
int dg_embed_bwd(const int64_t* idx, const void* dx, int dx_dtype, float* dtok, float* dpos,
                 int B, int T, int C, int V, void* stream);
/* dx_dtype: DG_F32, or DG_BF16 (the bf16 gradient stream of the engine's bf16 / fp8 modes) */

/* ---------------------------------------------------------------------------------------
 * LayerNorm over the last dim, eps inside the sqrt, biased variance -- ref: nn.LayerNorm at
 * src/model_component.py:488-489 applied at :505-506 (K4).  x is the fp32 residual stream;
 * y is written in y_dtype (the next GEMM's operand type). mean/rstd [M] are saved for bwd. */
int dg_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y, int y_dtype,
                     float* mean, float* rstd, int M, int C, float eps, void* stream);
/* precision fp8 (round 3): the same LayerNorm with its output as OCP e4m3 (q8 [M][C] bytes: the operand of the fp8 GEMM that
 * follows and of the fp8 dW launch) with delayed per-tensor scaling -- the value is first rounded to bf16, as the separate cast
 * launch saw it -- one partial maximum per workgroup: parts2 [2][n_parts] floats with n_parts = dg_layernorm_fwd_fp8_parts(M)
 * (slot step_state[2] & 1 written, the other read), *scale_inv the factor the consumer multiplies back.  y_bf16 nullable (nobody
 * reads the bf16 form in the engine's fp8 step).  C % 4 == 0, C <= 1024. */
int dg_layernorm_fwd_fp8_parts(int M);
int dg_layernorm_fwd_fp8(const float* x, const float* gamma, const float* beta, void* y_bf16, void* q8, float* mean, float* rstd,
                         int M, int C, float eps, float* parts2, int n_parts, const uint32_t* step_state, float* scale_inv,
                         void* stream);
/* dx = dresid (nullable, the residual branch's gradient) + LN'(dy).  dgamma/dbeta are emitted
 * as n_partials row-chunk partial sums: partial g at dgamma_part + g*part_stride (same for
 * dbeta_part); finish with dg_reduce_partials.  dy_dtype: DG_F32, or DG_BF16 (C % 4 == 0 only) when dy comes
 * straight out of a bf16-output dX GEMM. */
int dg_layernorm_bwd(const void* dy, int dy_dtype, const float* x, const float* gamma, const float* mean,
                     const float* rstd, const float* dresid, float* dx,
                     float* dgamma_part, float* dbeta_part, int64_t part_stride, int n_partials,
                     int M, int C, void* stream);

/* The same, and additionally g[m,c] = (g_dtype)(dx[m,c] * keep/(1-p)) with its column-sum partials in
 * gbias_part (same stride / count as dgamma_part): the operand and the bias gradient that the sub-layer
 * which runs next in backward would otherwise get from dg_dropout_bwd_cast(dx, site) -- one 38 MB pass and
 * one launch less per sub-layer (gbias_part may be NULL).  Returns DG_ERR_ARG for shapes the fused kernel does not cover
 * (C % 4 != 0 or C > 1024): call dg_layernorm_bwd + dg_dropout_bwd_cast then.
 * resid_dtype: type of dresid and dx, DG_F32 or (bf16 dy and bf16 g only) DG_BF16: the engine's bf16 / fp8 modes keep the
 * residual-branch gradient stream in bf16 (25 MB less per launch at the scaled configuration). */
int dg_layernorm_bwd_fused(const void* dy, int dy_dtype, const float* x, const float* gamma, const float* mean,
                           const float* rstd, const void* dresid, void* dx, int resid_dtype,
                           float* dgamma_part, float* dbeta_part, int64_t part_stride, int n_partials,
                           int M, int C,
                           void* g, int g_dtype, float dropout_p, const uint32_t* rng_state, uint32_t site,
                           float* gbias_part, void* stream);
/* fp8 mode: the same with g in bf16 AND a second time as OCP e5m2 (g8 [M][C] bytes: the gradient operand of the dX GEMM that
 * runs next) with delayed per-tensor scaling as in dg_fp8_quantize_delayed (g8_parts2 [2][DG_FP8_AMAX_PARTS], step_state,
 * *g8_scale_inv).  n_partials must be DG_FP8_AMAX_PARTS: every workgroup leaves one partial maximum. */
int dg_layernorm_bwd_fused_fp8(const void* dy, int dy_dtype, const float* x, const float* gamma, const float* mean,
                               const float* rstd, const void* dresid, void* dx, int resid_dtype,
                               float* dgamma_part, float* dbeta_part, int64_t part_stride, int n_partials,
                               int M, int C,
                               void* g, float dropout_p, const uint32_t* rng_state, uint32_t site, float* gbias_part,
                               void* g8, float* g8_parts2, const uint32_t* step_state, float* g8_scale_inv, int g8_only, void* stream);
/* g8_only != 0: the bf16 form of g is not written (its consumers read the e5m2 copy); g must still be a valid pointer. */

/* ---------------------------------------------------------------------------------------
 * GEMM "NT": C[M,N] = epilogue(A[M,K] . B[N,K]^T), MFMA with fp32 accumulation.
 * Replaces every nn.Linear forward (y = x W^T + b; W is [out,in]) -- ref:
 * src/model_component.py:392-393,404 (packed q/k/v), :454 (proj), :321-323 (FFN),
 * src/model.py:599 (lm_head) -- and, with B = W^T, every Linear's dX = dY . W.
 * Epilogue, in this order (each optional):
 *    v = acc + bias[n];  v = max(v,0) if relu;  v = 0 where relu_mask[m,n] <= 0 (or its sign_bits bit is clear);
 *    v = dropout(v; p, site);  v += residual[m,n];  C[m,n] = (out_dtype) v
 * in_dtype: type of A, B and relu_mask.  K and lda/ldb must be multiples of 16 bytes' worth of
 * elements (8 bf16 / 4 f32) and A, B 16-byte aligned.
 * sign_bits_out / sign_bits: the ReLU mask as one BIT per element (C[m,n] > 0).  The forward Linear+ReLU of
 * FeedForward (ref: src/model_component.py:321-322) emits it next to C; the dX GEMM of the second Linear consumes it
 * instead of re-reading the 16x larger activation.  The buffer (dg_gemm_nt_sign_bits_bytes(M, N) bytes) is OPAQUE: bits
 * are stored in the order the kernel's lanes own the output tile, so it is only meaningful between a producing and a
 * consuming dg_gemm_nt call with the same M and N.  Supported when dg_gemm_nt_sign_bits_supported(args) is non-zero
 * (bf16 operands, N % 8 == 0, K % 64 == 0, K >= 128); DG_ERR_ARG otherwise. */
typedef struct dg_gemm_nt_args {
    const void* A; int64_t lda;
    const void* B; int64_t ldb;
    void* C; int64_t ldc;
    int32_t M, N, K;
    int32_t in_dtype, out_dtype;
    const float* bias;
    int32_t relu;
    const void* relu_mask; int64_t ldmask;
    const float* residual; int64_t ldr;
    float dropout_p;
    const uint32_t* rng_state;
    uint32_t site;
    uint8_t* sign_bits_out;        /* nullable */
    const uint8_t* sign_bits;      /* nullable; excludes relu_mask */
    int64_t sign_bits_bytes;       /* size of either buffer, >= dg_gemm_nt_sign_bits_bytes(M, N) */
    float* colsum_part;            /* nullable: [colsum_rows][colsum_ld] fp32, see below */
    int64_t colsum_ld;
    int32_t colsum_rows;           /* rows of colsum_part, >= dg_gemm_nt_colsum_rows(args) */
    /* fp8 operands (in_dtype = DG_FP8_E4M3 for forward activations or DG_FP8_E5M2 for gradients; b_dtype = DG_FP8_E4M3, the
     * weight operand): one byte per element, K % 128 == 0, K >= 256, block-scaled MFMA (v_mfma_f32_16x16x128_f8f6f4) with unit
     * block scales; the per-tensor dequantisation factors *scale_a and *scale_b (device scalars written by dg_fp8_quantize)
     * multiply the fp32 accumulator before the epilogue.  b_dtype 0 = same as in_dtype (every non-fp8 call). */
    int32_t b_dtype;
    const float* scale_a;
    const float* scale_b;
    /* fp8_out (nullable): the output is ALSO written as OCP e4m3, [M][ld_fp8_out] bytes -- the operand of the next fp8 GEMM,
     * without a cast launch that re-reads C -- with delayed per-tensor scaling exactly as dg_fp8_quantize_delayed does it:
     * fp8_out_parts2 = [2][DG_FP8_AMAX_PARTS] partial maxima of the call site (slot rng_state-style step & 1 written, the other
     * read; fp8_out_step = the device step word array), *fp8_out_scale_inv = the factor the consumer multiplies back.  The cast
     * sees the fp32 value (before the rounding to out_dtype).  Only where dg_gemm_nt_fp8_out_supported(args) is non-zero
     * (fp8 e4m3 operands, bf16 out, bias + ReLU + sign_bits_out epilogue, whole tiles, exactly DG_FP8_AMAX_PARTS workgroups:
     * the first FFN Linear of the engine's step); DG_ERR_ARG otherwise. */
    void* fp8_out; int64_t ld_fp8_out;
    float* fp8_out_parts2;
    const uint32_t* fp8_out_step;
    float* fp8_out_scale_inv;
    /* fp8_out_only != 0 (with fp8_out): C is NOT written -- the fp8 copy (and the sign bits / column sums) is all the callers of
     * this output read (the engine's fp8 step: the FFN hidden layer and its gradient feed an fp8 GEMM and the fp8 dW launch only);
     * C must still be a valid pointer (shape / alignment checks). */
    int32_t fp8_out_only;
} dg_gemm_nt_args;
int dg_gemm_nt(const dg_gemm_nt_args* args, void* stream);
int dg_gemm_nt_sign_bits_supported(const dg_gemm_nt_args* args);
int dg_gemm_nt_fp8_out_supported(const dg_gemm_nt_args* args);
/* colsum_part: the epilogue also leaves the column sums of C (fp32, of the values before rounding to out_dtype) as
 * dg_gemm_nt_colsum_rows(args) partial rows: sum them with dg_reduce_partials.  Which rows of C a partial row covers is
 * the kernel's business (one per 32 rows of C, or one per workgroup and wave row when a workgroup's tiles share their
 * columns); every partial row is written by every call.  With C = d(pre-activation) of FeedForward's first Linear this is
 * that Linear's bias gradient (ref: autograd of src/model_component.py:321), for which the step used to re-read all of C
 * (dg_colsum).  Only for the sign_bits-consuming form with whole tiles: dg_gemm_nt_colsum_supported(args) non-zero,
 * DG_ERR_ARG otherwise. */
int dg_gemm_nt_colsum_supported(const dg_gemm_nt_args* args);
int dg_gemm_nt_colsum_rows(const dg_gemm_nt_args* args);
int64_t dg_gemm_nt_sign_bits_bytes(int M, int N);

/* fp8 operand preparation (per-tensor scaling, just in time) for dg_gemm_nt's fp8 form -- precision = "fp8".
 * dg_fp8_amax: DG_FP8_AMAX_PARTS partial maxima of |x| per segment of the flat buffer x (n elements) into amax_parts
 * [n_seg * DG_FP8_AMAX_PARTS] (no atomics, nothing to zero).  seg = NULL: one segment; else a device table of n_seg x 2 int64
 * {first element, element count}, both multiples of 8 (all weight matrices of the step in one launch).
 * dg_fp8_quantize: q[i] = fp8(clamp(x[i] * FMAX / amax[s])) in format fmt (DG_FP8_E4M3 / DG_FP8_E5M2), one byte per element,
 * with amax[s] = the maximum of segment s's partial maxima, and scale_inv[s] = amax[s] / FMAX (1 when amax is 0), the factor
 * dg_gemm_nt multiplies back.  n % 8 == 0. */
#define DG_FP8_AMAX_PARTS 256
int dg_fp8_amax(const void* x, int dtype, int64_t n, const int64_t* seg, int n_seg, float* amax_parts, void* stream);
int dg_fp8_quantize(const void* x, int dtype, void* q, int fmt, int64_t n, const int64_t* seg, int n_seg,
                    const float* amax_parts, float* scale_inv, void* stream);
/* Delayed scaling in ONE pass (the training engine's activations and gradients: the amax pass is what made just-in-time
 * scaling slower than the bf16 GEMMs it replaced): q = fp8(clamp(x * FMAX / amax_prev)), amax_prev = what this call site
 * recorded one step ago, and this tensor's own partial maxima are recorded for the next step.  parts2: [2][DG_FP8_AMAX_PARTS]
 * floats owned by the call site; slot (rng_state[2] & 1) is written, the other one read (rng_state[2] is the step word that
 * dg_state_advance increments).  Seed both slots with dg_fp8_amax before the first use.  scale_inv[0] = amax_prev / FMAX. */
int dg_fp8_quantize_delayed(const void* x, int dtype, void* q, int fmt, int64_t n, float* parts2,
                            const uint32_t* rng_state, float* scale_inv, void* stream);

/* GEMM "TN": weight gradients, dW[P,Q] = sum_r A[r,P] * B[r,Q]  (A = dY [R,P], B = X [R,Q]).
 * The contraction over the R = B*T rows is split n_splits ways across workgroups; split s
 * writes its fp32 partial to out + s*split_stride (finish with dg_reduce_partials).
 * P and Q need no alignment; lda/ldb must be multiples of 8 bf16 / 4 f32 and every 16-byte
 * chunk that starts left of P (resp. Q) must be readable. */
int dg_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb,
               float* out, int64_t ldo, int64_t split_stride, int n_splits,
               int R, int P, int Q, int dtype, void* stream);

/* Grouped form of dg_gemm_tn: n independent weight-gradient problems in ONE launch, each over its whole
 * contraction (no split, no partial slabs, no reduction pass): out_i[P_i,Q_i] = sum_r A_i[r,P_i] * B_i[r,Q_i].
 * The 128x128 output tiles of all problems are dealt to persistent workgroups, so the step's ~25 small dW
 * GEMMs (ref: autograd of every nn.Linear, src/model_component.py:321-323,392-393,454, src/model.py:599) fill
 * the chip together at the end of backward.  dtype DG_BF16: bf16 operands, R_i a multiple of 64, alignment as for dg_gemm_tn.
 * dtype DG_FP8_E5M2 (precision "fp8"): A = dY as OCP e5m2, B = X as OCP e4m3 -- the very copies the fp8 dX / forward GEMMs
 * consumed -- one byte per element, R_i a multiple of 128, lda / ldb multiples of 16, scale_a / scale_b the dequantisation
 * factors dg_fp8_quantize* left for them; fp32 accumulation, v_mfma_f32_16x16x128_f8f6f4.
 * The operands must stay alive until this call (the caller keeps dY of every Linear). */
typedef struct dg_tn_problem {
    const void* A; int64_t lda;     /* dY [R,P] */
    const void* B; int64_t ldb;     /* X  [R,Q] */
    float* out; int64_t ldo;        /* dW [P,Q] fp32, overwritten */
    int R, P, Q, reserved;
    const float* scale_a;           /* fp8 operands only: device scalars, dW = scale_a[0] * scale_b[0] * sum_r qA qB */
    const float* scale_b;
} dg_tn_problem;
int dg_gemm_tn_grouped(const dg_tn_problem* problems, int n, int dtype, void* workspace, int64_t workspace_bytes, void* stream);
/* Optional workspace (device memory of dg_gemm_tn_grouped_workspace_bytes(problems, n) bytes, zero-filled ONCE by the
 * caller and then left to the library between calls on one stream): lets the kernel cut every tile's contraction into
 * two halves run by different workgroups when that shortens the schedule (381 tiles on 256 CUs: 3 rounds of half
 * tiles instead of 2 rounds of whole ones).  The halves are summed first + second, a fixed order.  NULL = no split.
 * A second half waits (spins on a flag) for the first half of its tile.  The launch rules make the producer an earlier item
 * of a LOWER-numbered workgroup that never waits itself: every tile is cut only when each workgroup gets one item (2 * tiles
 * <= #CUs: workgroup tiles + t waits for workgroup t); otherwise only the leftover tiles of the last round are cut, their
 * first halves run FIRST on even XCDs and their second halves LAST on the next-numbered workgroup.  Under in-order workgroup
 * dispatch the wait therefore resolves whatever part of the grid is resident.  It is bounded all the same (~1 s): a wave
 * that runs out sets the sticky error word in the LAST 16 BYTES of the workspace (uint32, non-zero = some result of some
 * launch since the workspace was zeroed is invalid) instead of hanging the GPU; the host may read it at any sync point. */
int64_t dg_gemm_tn_grouped_workspace_bytes(const dg_tn_problem* problems, int n);

/* out[i] = sum_{g < n_partials} partials[g*stride + i], i < n.  Deterministic order. */
int dg_reduce_partials(const float* partials, int64_t stride, int n_partials,
                       float* out, int64_t n, void* stream);

/* Column sums (bias gradients): partial g (< n_partials) of sum_m A[m, n], fp32. */
int dg_colsum(const void* A, int64_t lda, int dtype, float* part, int64_t part_stride,
              int n_partials, int M, int N, void* stream);

/* g = (dtype)(dy * keep/(1-p)) -- backward of the nn.Dropout after proj / FFN
 * (ref: src/model_component.py:454,324); with p == 0 a plain cast.  relu_mask (nullable, fp32
 * [M,N]): additionally g = 0 where relu_mask <= 0 (backward of the ReLU that ends FeedForward,
 * ref: src/model_component.py:118-121).  Optionally also emits the column-sum partials of g (the
 * bias gradient of the Linear in front). g may be NULL when only the column sums are wanted. */
int dg_dropout_bwd_cast(const void* dy, int dy_dtype /* DG_F32; DG_BF16 with a bf16 g and no relu_mask */, int64_t lddy, void* g, int64_t ldg, int dtype,
                        int M, int N, float p, const uint32_t* rng_state, uint32_t site,
                        const float* relu_mask, int64_t ldmask,
                        float* colsum_part, int64_t part_stride, int n_partials, void* stream);

/* out = (out_dtype) in, n elements: f32 -> bf16 (shadow copies), bf16 -> f32, f32 -> f32. */
int dg_cast(const void* in, int in_dtype, void* out, int out_dtype, int64_t n, void* stream);
/* out[c, r] = (dtype) in[r, c] for r < R, c < Cc; out has leading dim ldo >= R and columns
 * R..ldo-1 are zero-filled (W^T operands for the dX GEMMs, K padded to the MFMA granule). */
int dg_transpose_cast(const float* in, int64_t ldi, void* out, int64_t ldo, int dtype,
                      int R, int Cc, void* stream);

/* The same for n_desc matrices in ONE launch.  desc: device array of n_desc x 8 int64
 * {in ptr, out ptr, ldi, ldo, R, Cc, first_tile, tiles_x} where a matrix owns tiles_x * ceil(ldo/64)
 * consecutive 64x64 tiles starting at first_tile (tiles_x = ceil(Cc/64)); total_tiles = their sum.
 * in_dtype: DG_F32 (the fp32 masters) or, for bf16 output, DG_BF16 (the bf16 shadow copy: same values, half the read). */
int dg_transpose_cast_batched(const int64_t* desc, int n_desc, int total_tiles, int in_dtype, int out_dtype, void* stream);
/* The same for one-byte elements (precision fp8: the e4m3 copy of every W^T from the e4m3 copy of W, scales unchanged), tiles of
 * 128 x 128: first_tile / tiles_x of the desc rows count 128-wide tiles (tiles_x = ceil(Cc / 128), a matrix owns tiles_x *
 * ceil(ldo / 128) of them).  Rows of `out` beyond R (ldo > R: padding) are written as zeros. */
int dg_transpose_u8_batched(const int64_t* desc, int n_desc, int total_tiles, void* stream);

/* ---------------------------------------------------------------------------------------
 * Causal multi-head attention -- ref: Head2.forward src/model_component.py:392-405 for every
 * head of MultiHeadAttention3 (:453), i.e. K5's output to K11:
 *   S = (Q K^T) * scale; mask j > i to -inf; P = softmax(S); P = dropout(P) (no renorm); O = P V.
 * qkv: [B*T, 3*NH*H] with column blocks [Q heads | K heads | V heads]; out: [B*T, NH*H], head h
 * at columns h*H.. (the torch.cat of :453 disappears).  lse [B,NH,T] = log-sum-exp of the scaled
 * masked scores, saved for backward.  Dropout element index = ((b*NH+h)*T+i)*T+j. */
/* keep_bits (nullable; round 3): dg_attn_keep_bits_bytes(...) bytes in which the forward pass leaves its dropout keep decisions
 * (16 wave masks per unmasked 32 x 32 tile, opaque layout); handed to dg_attn_bwd the dQ pass selects with them instead of
 * hashing every score again (same decisions either way: the masks are the hash's compare results).  0 bytes = the shape takes
 * the generic kernels, which have no such path: pass NULL. */
int64_t dg_attn_keep_bits_bytes(int B, int T, int NH, int H, int dtype);
int dg_attn_fwd(const void* qkv, void* out, float* lse, int B, int T, int NH, int H,
                float scale, float dropout_p, const uint32_t* rng_state, uint32_t site,
                int dtype, void* keep_bits, int64_t keep_bits_bytes, void* stream);
/* dqkv [B*T, 3*NH*H] from dout.  workspace: at least B*NH*T floats (delta = rowsum(dO * O)); with
 * dg_attn_bwd_workspace_bytes(...) bytes the dQ pass also leaves the dropped-out probabilities and dS of every
 * unmasked 32 x 32 tile behind it and the dK/dV pass consumes them instead of recomputing scores, exp and the
 * dropout hash (bf16, head size 64; 55 MB at the scaled config). */
int64_t dg_attn_bwd_workspace_bytes(int B, int T, int NH, int H, int dtype);
int dg_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse,
                void* dqkv, void* workspace, int64_t workspace_bytes, int B, int T, int NH, int H,
                float scale, float dropout_p, const uint32_t* rng_state, uint32_t site,
                int dtype, const void* keep_bits, int64_t keep_bits_bytes, void* stream);

/* Precision "fp8" (round 3): the attention kernels leave their outputs a second time as fp8, so that the cast launches in front of
 * the projection (o as e4m3: the x operand of `proj` and of its weight gradient -- ref: MultiHeadAttention3.proj,
 * src/model_component.py:454) and in front of the QKV dX GEMM (dqkv as e5m2: its gradient operand and the dY of the QKV weight
 * gradient -- ref: autograd through the packed query / key / value Linears, :392-393) disappear.  Delayed per-tensor scaling as in
 * dg_fp8_quantize_delayed, with a history format of the kernels' own: hist3 = DG_ATTN_FP8_HIST floats per call site (16-byte
 * aligned; three slots of 64 partial maxima, each partial on a 128-byte line of its own -- partial i of slot s is word
 * (s * 64 + i) * 32) -- slot step % 3 (step = step_state[2]) collects this step's maxima by atomic max, slot (step + 2) % 3 holds
 * last step's (the scale of this launch, *scale_inv = its inverse), slot (step + 1) % 3 is cleared for the next step; seed every
 * partial with a just-in-time maximum.  q8: [B*T, NH*H] (forward) / [B*T, 3*NH*H] (backward), one byte per element, 16-byte aligned.  only8 (backward only): the
 * bf16 dqkv is not written.  dg_attn_fp8_out_supported: 1 when the shape takes the MFMA kernels in their default forms (else cast). */
#define DG_ATTN_FP8_HIST (3 * 64 * 32)
typedef struct dg_attn_fp8_out {
    void* q8; float* hist3; const uint32_t* step_state; float* scale_inv; int only8;
} dg_attn_fp8_out;
int dg_attn_fp8_out_supported(int B, int T, int NH, int H, int dtype);
int dg_attn_fwd_fp8(const void* qkv, void* out, float* lse, int B, int T, int NH, int H,
                    float scale, float dropout_p, const uint32_t* rng_state, uint32_t site,
                    int dtype, void* keep_bits, int64_t keep_bits_bytes, const dg_attn_fp8_out* fp8, void* stream);
int dg_attn_bwd_fp8(const void* qkv, const void* out, const void* dout, const float* lse,
                    void* dqkv, void* workspace, int64_t workspace_bytes, int B, int T, int NH, int H,
                    float scale, float dropout_p, const uint32_t* rng_state, uint32_t site,
                    int dtype, const void* keep_bits, int64_t keep_bits_bytes, const dg_attn_fp8_out* fp8, void* stream);

/* Single-query attention against a K/V cache for generate() -- ref: src/model.py:625-635 re-runs the
 * whole forward per new token; with the cache only the new position is computed.  qkv_cache:
 * [B, Tcap, 3*NH*H] in the training layout (row = position; the QKV GEMM of the new token writes row t);
 * out [B, NH*H] = attention output of query row t over keys 0..t.  Same arithmetic and order as the
 * generic forward kernel, so fp32 decoding is bit-identical to the uncached forward. */
int dg_attn_decode(const void* qkv_cache, void* out, int B, int Tcap, int t, int NH, int H,
                   float scale, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------
 * Cross entropy, mean over rows -- ref: F.cross_entropy at src/model.py:604-607 (K16).
 * loss_rows[m] = logsumexp(logits[m,:]) - logits[m,target[m]].  If dlogits != NULL also writes
 * dlogits[m,n] = (softmax(logits[m,:])[n] - [n == target]) * grad_scale (* *grad_scale_dev) in
 * `dtype`, with columns
 * V..ldd-1 zero-filled.  dg_reduce_mean then gives the scalar loss.
 * logits_dtype DG_F32, or DG_BF16 (rows up to 53248 wide, bf16 gradient): the engine's lm_head at the GPT-2 vocabulary writes
 * bf16 logits (0.82 GB instead of 1.65 GB at M = 8192) and lets this kernel overwrite them IN PLACE with their gradient
 * (dlogits == logits, ldd == ldl). */
int dg_cross_entropy(const void* logits, int logits_dtype, int64_t ldl, const int64_t* targets, float* loss_rows,
                     void* dlogits, int64_t ldd, int dtype, float grad_scale,
                     const float* grad_scale_dev /* nullable: multiplies grad_scale */,
                     int M, int V, void* stream);
/* dg_cross_entropy on bf16 logits (large vocabularies, gradient written in place or elsewhere in bf16) that ALSO leaves the
 * gradient as OCP e5m2 for precision "fp8": dlogits_fp8 [M, ld8] = e5m2(dlogits * 57344 / grad_scale) -- |softmax - onehot| <= 1
 * bounds |dlogits| by grad_scale, so the scale is known a priori (no amax pass, no history) and the consumer's dequantisation
 * factor is the constant grad_scale / 57344.  ld8 % 16 == 0, V <= ld8 <= ldd; columns V .. ld8 - 1 are written as zeros (the dX
 * contraction of lm_head runs over the padded width).  ref: F.cross_entropy at src/model.py:604-607. */
int dg_cross_entropy_fp8(const void* logits, int64_t ldl, const int64_t* targets, float* loss_rows, void* dlogits, int64_t ldd,
                         float grad_scale, int M, int V, void* dlogits_fp8, int64_t ld8, void* stream);
/* The loss head of a captured step at a small vocabulary (ldd <= 128, fp32 logits): dg_cross_entropy + the column sums of
 * the gradient (lm_head bias gradient: partial row b of colsum_part = rows [b * ceil(M / n_partials), ...), fp32, nullable) +
 * the scalar loss (loss_out[0] = loss_scale * sum of loss_rows, added in a fixed order by the workgroup that finishes last;
 * loss_part: n_partials floats of scratch, loss_counter: one device word, zero before the first launch and again after
 * every launch; all three nullable together) in ONE launch of n_partials (<= 2048) workgroups. */
int dg_cross_entropy_fused(const float* logits, int64_t ldl, const int64_t* targets, float* loss_rows, void* dlogits, int64_t ldd,
                           int dtype, float grad_scale, int M, int V, float* colsum_part, int64_t part_stride, int n_partials,
                           float* loss_part, uint32_t* loss_counter, float* loss_out, float loss_scale, void* stream);
/* out[0] = scale * sum_i x[i] (single workgroup, fixed order). */
int dg_reduce_sum(const float* x, int64_t n, float scale, float* out, void* stream);

/* probs = softmax(logits) row-wise, fp32 -- ref: F.softmax at src/model.py:631 (generate). */
int dg_softmax_rows(const float* logits, int64_t ldl, float* probs, int64_t ldp, int M, int V,
                    void* stream);

/* ---------------------------------------------------------------------------------------
 * AdamW over one flat fp32 buffer -- ref: torch.optim.AdamW(model.parameters(), lr, betas)
 * at src/train.py:121,151 (eps 1e-8, weight_decay 1e-2 unless overridden):
 *   p *= 1 - lr*wd; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
 *   p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps),   t = rng_state.step + 1
 * hyper (device): {lr, beta1, beta2, eps, weight_decay}.  grad_scale multiplies g first
 * (1/world_size after a sum all-reduce).  Optionally refreshes a bf16 shadow copy of p.
 * advance_step != 0: the launch also does dg_state_advance -- the workgroup that finishes last writes step + 1 (word 3 of
 * rng_state is its arrival counter: zero before and after every launch). */
int dg_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper,
                  uint32_t* rng_state, float grad_scale, void* shadow_bf16, int advance_step, void* stream);

/* ---------------------------------------------------------------------------------------
 * The row-local chain of one residual block in ONE launch (bf16 operands, C = 384, M % 64 == 0) -- ref:
 * src/model_component.py:454 (proj) + :505 (x + ...), :506 + :488-489 (LayerNorm 2), :320-325 (FeedForward3), the second
 * residual add, and the NEXT block's :505 LayerNorm 1 + :392-393,404 (its 3 * NH per-head Linears as one packed operand):
 *     x1 = x + dropout(o wproj^T + bproj; site_proj)      h2 = LN(x1; ln2w, ln2b)      f = relu(h2 w1^T + b1)
 *     x2 = x1 + dropout(f w2^T + b2; site_ffn)            h1 = LN(x2; ln1w, ln1b)      qkv = h1 wqkv^T
 * It replaces four dg_gemm_nt launches and two dg_layernorm_fwd launches per block and writes exactly what they wrote (the
 * tensors the backward pass reads): x1, mean2 / rstd2, h2, f, the ReLU sign bits in dg_gemm_nt's opaque layout for an
 * [M, 4C] output (dg_gemm_nt_sign_bits_bytes(M, 4 * C) bytes), x2, mean1 / rstd1, h1, qkv.  Dropout masks are those of
 * dg_gemm_nt's epilogue (same site keys and element indices).  A workgroup owns 64 rows and walks the whole chain for them.
 * The four weight operands (wproj [C, C], w1 [4C, C], w2 [C, 4C], wqkv [3C, C]; bf16, [out, in]) are passed PACKED: in the
 * order the kernel streams them, as written by dg_pack_chain_weights (same byte count as the matrix; refresh after every
 * optimizer step, like the W^T operands of the dX GEMMs).
 * mode 0: everything above.  mode 1 (last block): stops behind the second residual add and writes x2 as bf16 (x2_bf16, the
 * operand of lm_head; x2 / ln1* / h1 / wqkv / qkv unused).  mode 2 (head: the first block's LayerNorm 1 on the embedding
 * output): h1 = LN(x), qkv = h1 wqkv^T only. */
typedef struct dg_block_chain_args {
    int32_t mode, M, C;
    float eps;
    const void* o; const float* x;
    const void* wproj; const float* bproj; float* x1;
    const float* ln2w; const float* ln2b; float* mean2; float* rstd2; void* h2;
    const void* w1; const float* b1; void* f; uint8_t* sign_bits; int64_t sign_bits_bytes;
    const void* w2; const float* b2; float* x2; void* x2_bf16;
    const float* ln1w; const float* ln1b; float* mean1; float* rstd1; void* h1;
    const void* wqkv; void* qkv;
    float dropout_p; const uint32_t* rng_state; uint32_t site_proj, site_ffn;
} dg_block_chain_args;
int dg_block_chain_supported(int M, int C);
int dg_block_chain_fwd(const dg_block_chain_args* args, void* stream);
/* w [N, K] bf16 row-major with leading dimension ld (N % 384 == 0, K % 32 == 0) -> packed (N * K elements).  Batched form: a
 * device table of n_desc rows {src, dst, N, K, first stage of the matrix (prefix sum of N / 384 * K / 32), ld} (int64 each):
 * every weight matrix of a model in one launch of total_stages workgroups. */
/* touch every 128-byte line of [p, p + bytes) once from every XCD (p 128-byte aligned): an L2 warm-up for a weight stream that the
 * next launch reads from all workgroups in lockstep */
int dg_l2_warm(const void* p, int64_t bytes, void* stream);
int dg_pack_chain_weights(const void* w, int64_t ld, void* packed, int N, int K, void* stream);
int dg_pack_chain_weights_batched(const int64_t* desc, int n_desc, int total_stages, void* stream);

/* ---------------------------------------------------------------------------------------
 * The row-local chain of the BACKWARD pass between two attention-backward calls in ONE launch (bf16 operands and gradient
 * stream, C = 384, M % 64 == 0, M / 64 <= #CUs) -- ref: autograd through src/model_component.py:392-393,404 (dX of the packed
 * q / k / v Linears), :505 (LayerNorm 1 + residual branch), :324 (the Dropout of the block below), :322-323 (dX of the
 * second FFN Linear through the ReLU), :321 (dX of the first), :506 (LayerNorm 2 + residual branch), :454 (proj's Dropout, dX):
 *   [block l]     dh  = dqkv wqkvT        dx1 = LN'(dh; x, mean1, rstd1, ln1w) + dresid1       g1 = dropout_bwd(dx1; site_ffn_below)
 *   [block l - 1] df  = (g1 w2T) masked by sign_bits      dh2 = df w1T
 *                 dx2 = LN'(dh2; x1, mean2, rstd2, ln2w) + dx1      g2 = dropout_bwd(dx2; site_proj)      dout = g2 wprojT
 * It replaces four dg_gemm_nt launches and two dg_layernorm_bwd_fused launches and leaves what they left: df, g1, g2 (the dY
 * operands of the weight gradients), dx1 / dx2 (the bf16 gradient stream), dout (the attention backward's input) and the
 * partial rows of b1 (column sums of df), b2 / bproj (column sums of g1 / g2) and both LayerNorms' dgamma / dbeta: TWO rows
 * per 64-row block, row 2 * block + {0, 1} at `part_stride` floats (2 * M / 64 rows for dg_reduce_partials).  The LayerNorm
 * backward consumes the dX GEMM's fp32 accumulators (the separate launches round them to bf16 in between).  The four weight
 * operands are the W^T shadows ([in, out] bf16) PACKED by dg_pack_chain_weights (wqkvT: N = C, K = 3C; w2T: N = 4C, K = C;
 * w1T: N = C, K = 4C; wprojT: N = C, K = C).  sign_bits: dg_gemm_nt's layout for an [M, 4C] output.
 * mode 0: everything above (dresid2 is dx1: pass the same pointer).  mode 1 (top of the stack): the second half only, g1
 * arrives as g_in.  mode 2 (block 0): the first half only; gbias1_part NULL = no dropout and no bias behind LayerNorm 1. */
typedef struct dg_block_chain_bwd_args {
    int32_t mode, M, C, reserved;
    const void* dqkv; const void* wqkvT; const float* x; const float* mean1; const float* rstd1; const float* ln1w;
    const void* dresid1; void* dx1; void* g1;
    float* dln1w_part; float* dln1b_part; float* gbias1_part;
    const void* g_in;
    const void* w2T; const uint8_t* sign_bits; int64_t sign_bits_bytes; void* df; float* db1_part;
    const void* w1T; const float* x1; const float* mean2; const float* rstd2; const float* ln2w;
    const void* dresid2; void* dx2; void* g2;
    float* dln2w_part; float* dln2b_part; float* gbias2_part;
    const void* wprojT; void* dout;
    int64_t part_stride;
    float dropout_p; const uint32_t* rng_state; uint32_t site_ffn_below, site_proj;
} dg_block_chain_bwd_args;
int dg_block_chain_bwd_supported(int M, int C);
int dg_block_chain_bwd(const dg_block_chain_bwd_args* args, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DRAKEGPT_HIP_H */
