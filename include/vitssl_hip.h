/*
 * vitssl_hip.h -- C ABI of libvitssl_hip.so, the MI355X (gfx950) compute library
 * behind the vit_core hot path of kristi700/ViT-SSL.
 *
 * The reference has no FFI layer: its "operator interface" for this path is the
 * set of torch ops that vit_core lowers to.  Each
 * entry point below names the reference lines it replaces (paths relative to the
 * reference repo root).  The Python host mirror (vit-ssl_amd/) binds these with
 * ctypes; see INTEGRATION.md for the binding a reference maintainer would add.
 *
 * Conventions
 *   - every entry returns 0 on success, <0 on error; vitssl_last_error() gives text
 *   - no allocation, no ownership transfer: all pointers are device pointers owned
 *     by the caller (torch-allocated); shapes are passed explicitly
 *   - kernels are enqueued on `stream` (a hipStream_t passed as void*); no entry
 *     synchronises the device
 *   - bf16 tensors are passed as void* (raw uint16 storage), fp32 as float*
 *   - all matrices are dense row-major; "ld" is always the logical column count
 *   - callable from any host thread (autograd's backward thread included): the only process-wide
 *     state are idempotent "attribute set" / "knob read" flags and the last-error string, which is
 *     thread-local; one process drives one GPU (launches go to the CURRENT device's stream)
 */
#ifndef VITSSL_HIP_H
#define VITSSL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VITSSL_OK 0
#define VITSSL_ERR_ARG (-1)     /* bad shape / unsupported configuration */
#define VITSSL_ERR_LAUNCH (-2)  /* HIP launch error */

const char* vitssl_last_error(void);
int vitssl_version(void);

/* ---- CUs left to other kernels (data parallelism; no reference counterpart: the reference is single-device,
 *      utils/train_utils.py:12-16) -------------------------------------------------------------------------
 * The forward / input-gradient / weight-gradient GEMMs, the persistent attention forward and the LayerNorm backward run ONE persistent workgroup per
 * CU that owns the CU's whole register file and 129 of its 160 KiB of LDS, so the collective library's all-reduce
 * kernels cannot share a CU with them.  vitssl_set_reserved_cus(n) makes every persistent grid launched AFTER the
 * call leave n CUs unoccupied (clamped to [0, CUs - 8]; process-wide; read on every launch, so the order of
 * "first forward" and "reducer built" does not matter).  The initial value is VITSSL_RESERVE_CUS (default 0). */
int vitssl_set_reserved_cus(int n);
int vitssl_get_reserved_cus(void);          /* returns the count in force (not an error code) */
int vitssl_debug_last_nt_grid(void);        /* workgroups of the last persistent NT GEMM launch (tests) */
int vitssl_debug_last_attn_fwd_grid(void);  /* workgroups of the last attention-forward launch (tests) */

/* ---- dropout stream -------------------------------------------------------
 * Counter-based: keep(e) = [bits16(seed, site, e) >= round(p * 65536)] for element index e = row * cols + col of the
 * tensor the dropout acts on (rows * cols < 2^34; four consecutive elements share one draw of 64 bits).  The effective
 * drop probability is round(p * 65536) / 65536 and survivors are scaled by its exact reciprocal.  The same triple
 * regenerates the mask in backward.  Replaces nn.Dropout at vit_core/encoder_block.py:29-30,45,51 and
 * vit_core/feed_forward.py:16,27.  The mixer (a 32-bit multiply / xor-shift hash, csrc/common.h) is restated in NumPy
 * in tests/test_dropout_stream.py, which holds its statistical checks; vitssl_dropout_mask exports the mask. */
typedef struct {
  float p;        /* drop probability; 0 disables */
  uint32_t site;  /* per (block, site) stream id */
  uint64_t seed;  /* per-step seed */
} vitssl_dropout_t;

/* Materialise the 0/1 keep mask (uint8 [rows*cols]); test/debug helper. */
int vitssl_dropout_mask(uint8_t* keep, int64_t rows, int64_t cols, vitssl_dropout_t d, void* stream);

/* ---- LayerNorm (nn.LayerNorm eps=1e-5 affine; vit_core/encoder_block.py:26-27,41,49;
 *      vit_core/mlp_head.py:9,13) ------------------------------------------- */
/* y_bf16[rows,cols] = LN(x) ; saves mean/rstd [rows] (fp32 statistics). */
int vitssl_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y_bf16,
                         float* mean, float* rstd, int64_t rows, int cols, float eps, void* stream);

/* g_out = g_res + LN'(dy); optionally also emits the dropout-masked bf16 copy that
 * the next (reverse-order) GEMM consumes, and its column sum (bias gradient).
 *   dy_bf16      [rows,cols] gradient wrt the LN output (bf16)
 *   g_res        [rows,cols] fp32 residual-branch gradient, may be NULL
 *   g_out        [rows,cols] fp32, may alias g_res
 *   gm_bf16      [rows,cols] bf16 = keepmask(drop) * g_out / (1-p), may be NULL
 *   dgamma/dbeta [cols] fp32, ATOMICALLY ACCUMULATED (caller zeroes)
 *   gm_colsum    [cols] fp32 accumulated column sums of gm_bf16, may be NULL */
int vitssl_layernorm_bwd(const void* dy_bf16, const float* x, const float* mean, const float* rstd,
                         const float* gamma, const float* g_res, float* g_out, void* gm_bf16,
                         float* dgamma, float* dbeta, float* gm_colsum, vitssl_dropout_t drop,
                         int64_t rows, int cols, void* stream);

/* Standalone "mask + cast + column-sum" of an fp32 gradient (top of the backward chain). */
int vitssl_grad_mask_cast(const float* g, void* gm_bf16, float* gm_colsum, vitssl_dropout_t drop,
                          int64_t rows, int cols, void* stream);

/* ---- bf16 MFMA GEMM, C[M,N] = A[M,K] . B[N,K]^T, fp32 accumulate ----------------
 * (nn.Linear at vit_core/attention.py:54-58,82-84,105; feed_forward.py:14-15,26-28;
 *  ssl/simmim/model.py:28-30,35-37,45,57; Conv2d k=s=P at patch_embedding.py:22,79-84;
 *  mlp_head.py:10,14; ssl/dino/head.py:10-17,20,22 -- and their autograd dgrads,
 *  which are the same contraction against the transposed weight copy). */
enum {
  VITSSL_EPI_BF16 = 0,         /* out0(bf16) = acc + bias                                         */
  VITSSL_EPI_F32 = 1,          /* out0(f32)  = acc + bias                                         */
  VITSSL_EPI_GELU = 2,         /* u = bf16(acc+bias); out1(bf16) = keep*s*gelu_erf(u); out0(bf16) = g' = keep*s*gelu_erf'(u) */
  VITSSL_EPI_RESID = 3,        /* out0(f32)  = aux(f32 [M,N]) + drop(acc + bias)                  */
  VITSSL_EPI_DGELU = 4,        /* out0(bf16) = acc * aux(bf16 g' [M,N] saved by EPI_GELU)         */
  VITSSL_EPI_EMBED = 5         /* patch-embedding epilogue, see vitssl_embed_t                    */
};

/* Extra arguments of VITSSL_EPI_EMBED: token = mask ? mask_token : (acc + bias);
 * token += pos[row_in_img + tok_offset]; stored (fp32) at
 * out0[(img * out_tokens + tok_offset + row_in_img), :].
 * (vit_core/ssl/simmim/model.py:45-49; patch_embedding.py:58-63,92-95,125-127) */
typedef struct {
  const uint8_t* mask;      /* [M] 1 = masked token, NULL = none */
  const float* mask_token;  /* [N] */
  const float* pos;         /* [out_tokens, N] (already interpolated if needed) */
  int tokens;               /* patches per image (rows of A per image) */
  int out_tokens;           /* tokens per image in the output (tokens + tok_offset) */
  int tok_offset;           /* 1 when a CLS slot precedes the patches, else 0 */
} vitssl_embed_t;

typedef struct {
  const void* A;            /* bf16 [M,K] */
  const void* B;            /* bf16 [N,K] */
  int64_t M;
  int N, K;                 /* K % 64 == 0, N % 4 == 0 (fp8 operands: K % 128 == 0, N % 8 == 0) */
  int epilogue;
  const float* bias;        /* [N] or NULL */
  const void* aux;          /* epilogue-specific input or NULL */
  void* out0;
  void* out1;
  float* colsum;            /* [N] accumulated column sums of out0 (bias grad), or NULL */
  vitssl_dropout_t drop;
  vitssl_embed_t embed;
} vitssl_gemm_t;

int vitssl_gemm_bf16_nt(const vitssl_gemm_t* g, void* stream);

/* ---- fp8 (OCP e4m3fn) operand path of the FORWARD Linear layers: BASELINE.json configs[4],
 * "ViT-L/16 SimMIM, fp8 MFMA weight path" (the same nn.Linear call sites as above:
 * vit_core/attention.py:54-58,105; feed_forward.py:14-15,26-28).  Same contraction, epilogues and
 * output types as vitssl_gemm_bf16_nt, but A [M,K] and B [N,K] are e4m3 bytes, K % 128 == 0, and the
 * MFMA is v_mfma_f32_16x16x128_f8f6f4 (fp32 accumulate).  Epilogues BF16 / F32 / GELU / RESID / DGELU: the
 * forward products and the input-gradient products dX = dY . W (dY quantised with a per-tensor scale that
 * follows the previous step's max |dY|); the weight gradients have their own entry, vitssl_gemm_fp8_tn. */
typedef struct {
  const float* alpha;  /* device scalar: acc is multiplied by *alpha before the epilogue (dequantisation scale of the
                          weight operand; vitssl_fp8_quantize_weights writes it), or NULL = 1 */
  const float* alpha2; /* second device scalar multiplied in (1 / scale of a scaled gradient operand), or NULL = 1 */
  void* out_fp8;       /* EPI_GELU: e4m3 [M,N] image of out1; EPI_DGELU: of out0 (the A operand of the next fp8 GEMM); or NULL.
                          When it is given, the bf16 image it mirrors (out1 / out0) may be NULL and is then not written */
  const float* out_scale; /* device scalar the values are multiplied by before quantisation into out_fp8, or NULL = 1 */
  float* out_amax;     /* device slot: atomic max of |value| written to out_fp8 (before scaling; caller zeroes), or NULL */
} vitssl_fp8_gemm_t;
int vitssl_gemm_fp8_nt(const vitssl_gemm_t* g, const vitssl_fp8_gemm_t* q, void* stream);

/* vitssl_attn_fwd that also writes the e4m3 image of `out` (made from the fp32 values before the bf16 store) */
int vitssl_attn_fwd_fp8(const void* qkv, void* out, void* out_fp8, float* lse, float* probs, int B, int N, int H, int dh,
                        void* stream);
/* Weight gradient on e4m3 operands: C[N1,N2] (fp32) += alpha * alpha2 * A8[M,N1]^T . B8[M,N2]  (dY scaled image x the
 * unit-scale activation image; transposed 1-byte LDS reads, v_mfma_f32_16x16x128_f8f6f4).  N1 % 16 == 0, N2 % 16 == 0;
 * workspace as for vitssl_gemm_bf16_tn (its size from vitssl_gemm_fp8_tn_workspace_floats). */
int64_t vitssl_gemm_fp8_tn_workspace_floats(int64_t M, int N1, int N2);
int vitssl_gemm_fp8_tn(const void* A8, const void* B8, float* C, int64_t M, int N1, int N2, const float* alpha,
                       const float* alpha2, float* workspace, int64_t workspace_floats, void* stream);
/* vitssl_attn_bwd (one-launch form) that also writes dqkv_fp8 = e4m3(dqkv * *qscale) and records max|dqkv| in *qamax;
 * dqkv (bf16) may be NULL (not written) */
int vitssl_attn_bwd_fp8(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, void* dqkv_fp8,
                        const float* qscale, float* qamax, int B, int N, int H, int dh, void* stream);
/* y_fp8[n] = e4m3(clamp(x, -448, 448)), round to nearest even (activations are quantised at unit scale) */
int vitssl_quantize_fp8(const void* x_bf16, void* y_fp8, int64_t n, void* stream);
/* y_fp8[n] = e4m3(x * *qscale) (qscale NULL = 1); *qamax = max(*qamax, max|x|) (NULL = not recorded; caller zeroes) */
int vitssl_quantize_fp8_scaled(const void* x_bf16, void* y_fp8, int64_t n, const float* qscale, float* qamax, void* stream);
/* LayerNorm forward that also emits the e4m3 image of its output (operand of the next fp8 GEMM);
 * y_bf16 may be NULL (not written): with vitssl_gemm_fp8_tn nothing reads the bf16 image. */
int vitssl_layernorm_fwd_fp8(const float* x, const float* gamma, const float* beta, void* y_bf16, void* y_fp8,
                             float* mean, float* rstd, int64_t rows, int cols, float eps, void* stream);
/* vitssl_layernorm_bwd / vitssl_grad_mask_cast that also write gm_fp8 = e4m3(gm * *qscale) and record max|gm| in
 * *qamax: the scaled e4m3 operand of the fp8 GEMMs that consume gm; gm_bf16 may be NULL (not written) */
int vitssl_layernorm_bwd_fp8(const void* dy_bf16, const float* x, const float* mean, const float* rstd, const float* gamma,
                             const float* g_res, float* g_out, void* gm_bf16, void* gm_fp8, const float* qscale, float* qamax,
                             float* dgamma, float* dbeta, float* gm_colsum, vitssl_dropout_t drop, int64_t rows, int cols,
                             void* stream);
int vitssl_grad_mask_cast_fp8(const float* g, void* gm_bf16, void* gm_fp8, const float* qscale, float* qamax, float* gm_colsum,
                              vitssl_dropout_t drop, int64_t rows, int cols, void* stream);
/* Per-step fp8 refresh of a table of weights in one call (three launches): per tensor j,
 * amax_j = max|src|, k_j = floor(log2(448 / amax_j)) (0 when amax_j = 0; a power-of-two scale is exact),
 * dst_fp8 [R,C] = e4m3(src * 2^k_j) (forward operand), dst_t_fp8 [C,R] = its transpose (input-gradient operand),
 * alpha[j] = 2^-k_j.  `jobs`, `tile_start`, `amax_ws` [njobs] and `alpha` [njobs] live in DEVICE memory;
 * tile_start[njobs + 1] is the exclusive prefix sum of ceil(R/64)*ceil(C/64). */
typedef struct {
  const float* src; /* f32 [R, C] */
  void* dst_fp8;    /* e4m3 [R, C] or NULL */
  void* dst_t_fp8;  /* e4m3 [C, R] or NULL */
  int R, C;
} vitssl_fp8_weight_job_t;
int vitssl_fp8_quantize_weights(const vitssl_fp8_weight_job_t* jobs, const int* tile_start, int njobs, int total_tiles,
                                float* amax_ws, float* alpha, void* stream);

/* Weight gradient: C[N1,N2] (fp32) += A[M,N1]^T . B[M,N2]  (contraction over rows).
 * Split over M across workgroups.  With a workspace of at least
 * vitssl_gemm_tn_workspace_floats(M,N1,N2) floats the partial tiles are combined through
 * per-split slabs (deterministic, plain stores + one reduce kernel); with workspace=NULL
 * they are combined with fp32 atomics.  N1 % 8 == 0, N2 % 8 == 0. */
int64_t vitssl_gemm_tn_workspace_floats(int64_t M, int N1, int N2);
int vitssl_gemm_bf16_tn(const void* A, const void* B, float* C, int64_t M, int N1, int N2, float* workspace,
                        int64_t workspace_floats, void* stream);

/* Several weight gradients over the SAME M rows in one launch: C_j[N1_j, N2_j] (fp32) += A_j[M, N1_j]^T . B_j[M, N2_j], j < njobs <= 8
 * (the four weight gradients of a transformer block: reference vit_core/encoder_block.py:40-53, feed_forward.py:26-28,
 * attention.py:30-47 backward).  The tiles of all jobs share one split count, one launch and one reduce pass, which cuts the
 * partial-tile traffic of vitssl_gemm_bf16_tn (one round of the CUs per launch, whatever the shape).  Workspace: at least
 * vitssl_gemm_tn_batch_workspace_floats(jobs, njobs, M) floats (0 when every tile has a single owner).  N1, N2 % 8 == 0. */
typedef struct {
  const void* A; /* bf16 [M, N1] */
  const void* B; /* bf16 [M, N2] */
  float* C;      /* f32 [N1, N2], accumulated into */
  int N1, N2;
} vitssl_tn_job_t;
int64_t vitssl_gemm_tn_batch_workspace_floats(const vitssl_tn_job_t* jobs, int njobs, int64_t M);
int vitssl_gemm_bf16_tn_batch(const vitssl_tn_job_t* jobs, int njobs, int64_t M, float* workspace, int64_t workspace_floats,
                              void* stream);

/* The same for e4m3 operands (vitssl_gemm_fp8_tn per job): C_j += alpha_j * alpha2_j * A8_j^T . B8_j.  N1, N2 % 16 == 0. */
typedef struct {
  const void* A8;      /* e4m3 [M, N1] */
  const void* B8;      /* e4m3 [M, N2] */
  float* C;            /* f32 [N1, N2], accumulated into */
  int N1, N2;
  const float* alpha;  /* device scalars multiplied into the product, or NULL */
  const float* alpha2;
} vitssl_fp8_tn_job_t;
int64_t vitssl_gemm_fp8_tn_batch_workspace_floats(const vitssl_fp8_tn_job_t* jobs, int njobs, int64_t M);
int vitssl_gemm_fp8_tn_batch(const vitssl_fp8_tn_job_t* jobs, int njobs, int64_t M, float* workspace, int64_t workspace_floats,
                             void* stream);

/* ---- fused multi-head self-attention (vit_core/attention.py:20-23,86-103) ----------
 * qkv  bf16 [B, N, 3, H, dh]  (the fused projection output: q | k | v per token)
 * out  bf16 [B, N, H*dh]      (heads merged, ready for final_linear)
 * lse  f32  [B, H, N]         (log-sum-exp of the scaled scores, saved for backward)
 * probs f32 [B, H, N, N] or NULL (return_attn=True path)
 * dh = 64 (every ViT family of BASELINE.json), N <= 256; other geometries return VITSSL_ERR_ARG;
 * scale = 1/sqrt(dh). */
int vitssl_attn_fwd(const void* qkv, void* out, float* lse, float* probs, int B, int N, int H, int dh,
                    void* stream);
/* delta_ws: f32 [B,H,N] scratch for rowsum(dO*O) (used by the two-launch variant only; the default
 * one-launch backward keeps delta in LDS); dqkv bf16 [B,N,3,H,dh] fully overwritten. */
int vitssl_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                    float* delta_ws, int B, int N, int H, int dh, void* stream);

/* ---- patch handling (nn.Unfold + permute, ssl/simmim/model.py:27,43; masking.py:35) */
/* img f32 [B,C,H,W] -> patches bf16 [B*gh*gw, C*P*P], feature order (c,kh,kw). */
int vitssl_patchify_bf16(const float* img, void* patches, int B, int C, int H, int W, int P, void* stream);
/* targets f32 [n_idx, C*P*P] = patches[idx] gathered straight from the image (exact fp32). */
int vitssl_gather_patches_f32(const float* img, const int32_t* idx, float* out, int n_idx, int C, int H,
                              int W, int P, void* stream);
/* out bf16 [n_idx, cols] = x f32 [rows, cols][idx]   (x[bool_mask], ssl/simmim/model.py:56) */
int vitssl_gather_rows_bf16(const float* x, const int32_t* idx, void* out, int n_idx, int cols, void* stream);
/* g f32 [rows, cols] = 0 except g[idx[i]] = src_bf16[i]; inv[row] = i or -1 */
int vitssl_scatter_rows_f32(const void* src_bf16, const int32_t* inv, float* g, int64_t rows, int cols,
                            void* stream);
/* CLS-row helpers for ViT / DINO (x[:,0], vit_core/vit.py:39): f32 [B, T, D] row 0 of each image */
int vitssl_gather_cls_f32(const float* x, float* out, int B, int T, int D, void* stream);
int vitssl_scatter_cls_f32(const float* gcls, float* g, int B, int T, int D, void* stream);

/* Patch-embedding backward glue: from dtok f32 [B*T_out, D] produce the bf16 operand
 * of the projection wgrad (masked rows zeroed, CLS rows dropped) and accumulate
 * d(pos) [T_out, D], d(mask_token) [D], d(bias) [D], d(cls_token) [D]. */
int64_t vitssl_embed_bwd_workspace_floats(int B, int tokens, int tok_offset, int D);
/* workspace (may be NULL = atomics only): vitssl_embed_bwd_workspace_floats() floats of scratch
 * through which the single-row accumulators d(mask_token) / d(bias) are reduced without
 * thousands of colliding atomics. */
int vitssl_embed_bwd(const float* dtok, const uint8_t* mask, void* dproj_bf16, float* dpos, float* dmask_token,
                     float* dbias, float* dcls, int B, int tokens, int tok_offset, int D, float* workspace,
                     int64_t workspace_floats, void* stream);

/* ---- losses ------------------------------------------------------------------- */
/* nn.L1Loss(mean) (configs/simmim/training.yaml:2-5): loss_sum += sum|p-t| (caller zeroes,
 * divides by n); dpred_bf16 = sign(p-t) * gscale (gscale = upstream_grad / n) or NULL. */
int vitssl_l1_loss(const float* pred, const float* target, float* loss_sum, void* dpred_bf16, float gscale,
                   int64_t n, void* stream);
/* nn.CrossEntropyLoss(mean) on f32 logits [B,C]; dlogits_bf16 = (softmax - onehot) * gscale */
int vitssl_cross_entropy(const float* logits, const int64_t* labels, float* loss_sum, void* dlogits_bf16,
                         float gscale, int B, int C, void* stream);

/* out[cols] (fp32, ACCUMULATED) += column sums of x bf16 [rows, cols] (bias gradients). */
int vitssl_colsum_bf16(const void* x_bf16, float* out, int64_t rows, int cols, void* stream);

/* ---- parameter plumbing ----------------------------------------------------------- */
int vitssl_cast_bf16(const float* src, void* dst, int64_t n, void* stream);
/* dst_t bf16 [C,R] = transpose(src f32 [R,C]) ; dst bf16 [R,C] = src (either may be NULL) */
int vitssl_cast_transpose_bf16(const float* src, void* dst, void* dst_t, int R, int C, void* stream);
/* The same for a whole table of weights in ONE launch (the per-step bf16 refresh of every
 * nn.Linear weight that torch.autocast performs implicitly in the reference's trainers,
 * utils/trainers/simmim_trainer.py:37, dino_trainer.py:86).  `jobs` and `tile_start` live in
 * DEVICE memory: tile_start[njobs + 1] is the exclusive prefix sum of ceil(R/64)*ceil(C/64). */
typedef struct {
  const float* src; /* f32 [R, C] */
  void* dst;        /* bf16 [R, C] or NULL */
  void* dst_t;      /* bf16 [C, R] or NULL */
  int R, C;
} vitssl_cast_job_t;
int vitssl_cast_transpose_batch(const vitssl_cast_job_t* jobs, const int* tile_start, int njobs, int total_tiles,
                                void* stream);
/* torch.optim.AdamW step over a flat fp32 buffer (utils/train_utils.py:25-29). step is 1-based;
 * g is multiplied by gscale first (1/world_size for DP averaging). */
int vitssl_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                 float eps, float wd, int step, float gscale, void* stream);
/* teacher = m*teacher + (1-m)*student over flat buffers (ssl/dino/model.py:126-139) */
int vitssl_ema(float* teacher, const float* student, int64_t n, float m, void* stream);

/* ---- DINO head / loss (vit_core/ssl/dino/head.py:19-23, loss.py:13-29, model.py:91-99) ---- */
/* F.normalize(dim=1, eps=1e-12): zn bf16 [rows,cols] = z / max(||z||, eps); inv_norm [rows] */
int vitssl_rownorm_fwd(const float* z, void* zn_bf16, float* inv_norm, int64_t rows, int cols, void* stream);
/* dz bf16 = inv_norm * (dzn - zn <zn, dzn>) */
int vitssl_rownorm_bwd(const float* dzn, const void* zn_bf16, const float* inv_norm, void* dz_bf16, int64_t rows, int cols,
                       void* stream);
/* weight_norm(dim=0): w_f32[k,:] = g[k] v[k,:] / ||v[k,:]||, inv_vnorm[k] = 1/||v[k,:]|| (cast with
 * vitssl_cast_transpose_bf16 for the GEMM operands) */
int vitssl_weightnorm_fold(const float* g, const float* v, float* w_f32, float* inv_vnorm, int K, int D, void* stream);
/* dg[k] += <dW[k,:], vhat_k>; dv[k,:] += g_k/||v_k|| (dW[k,:] - vhat_k <dW[k,:], vhat_k>) */
int vitssl_weightnorm_bwd(const float* dw, const float* g, const float* v, const float* inv_vnorm, float* dg, float* dv, int K,
                          int D, void* stream);
/* DINOLoss: teacher f32 [G,B,K], student f32 [V,B,K], center f32 [K];
 * loss_sum += -(1/(G B K)) sum_{b,k} (sum_g softmax((t-c)/tt))(sum_v log_softmax(s/ts));
 * dstudent bf16 [V,B,K] = gscale * dloss/dstudent (or NULL); t_ws = f32 scratch of t_ws_floats >=
 * vitssl_dino_loss_workspace_floats(G, B, K) floats ([B,K] teacher probabilities, then the per-slice softmax statistics of
 * the teacher rows: B*K + VITSSL_DINO_TWS_EXTRA(G,B)); a smaller buffer is refused (ABI version 2: version 1 took the bare
 * pointer and wrote the statistics behind [B,K] unchecked). */
#define VITSSL_DINO_TWS_EXTRA(G, B) (8 * (G) * (B))
int64_t vitssl_dino_loss_workspace_floats(int G, int B, int K);
int vitssl_dino_loss(const float* teacher, const float* student, const float* center, float* t_ws, int64_t t_ws_floats,
                     float* loss_sum, void* dstudent_bf16, int G, int V, int B, int K, float teacher_temp, float student_temp,
                     float gscale, void* stream);
/* out[cols] = column sums of x f32 [rows, cols] (overwrites) */
int vitssl_colsum_f32(const float* x, float* out, int64_t rows, int cols, void* stream);
/* center = m center + (1-m) colsum * inv_rows   (all-reduce colsum first under data parallelism) */
int vitssl_center_ema(float* center, const float* colsum, int K, float momentum, float inv_rows, void* stream);

/* Positional-table resize of DynamicPatchEmbedding.interpolate_pos_encoding
 * (vit_core/patch_embedding.py:26-48): F.interpolate(mode="bicubic", align_corners=False) of the
 * patch rows, channel-last.  src f32 [gh0*gw0, D] -> dst f32 [gh*gw, D]; the backward
 * ACCUMULATES into dsrc f32 [gh0*gw0, D]. */
int vitssl_bicubic_resize_fwd(const float* src, float* dst, int gh0, int gw0, int gh, int gw, int D, void* stream);
int vitssl_bicubic_resize_bwd(const float* ddst, float* dsrc, int gh0, int gw0, int gh, int gw, int D, void* stream);

/* ---- DINO multi-crop input pipeline -------------------------------
 * Replaces, per view, the torchvision transform list that data/datasets.py:80-123
 * (STL10DINODataset._get_dino_views) applies on the CPU with PIL images:
 * configs/dino/globals.yaml / locals.yaml = RandomResizedCrop, RandomHorizontalFlip,
 * ColorJitter, [RandomGrayscale], GaussianBlur(7), ToTensor (built by utils/train_utils.py:54-68).
 * The random parameters are drawn on the host (vit-ssl_amd/data/multicrop.py) and handed over
 * as device arrays:
 *   iparams int32 [B, 11] = top, left, h, w, flip, order[4] (0 brightness 1 contrast
 *                           2 saturation 3 hue), gray, hue_shift (uint8 added to H, wraps)
 *   fparams f32   [B, 10] = brightness, contrast, saturation factors, k1d[7] (normalised
 *                           Gaussian taps of this image's sigma)
 * Results are bit-identical to Pillow's uint8 arithmetic (oracle/augment_oracle.py). */
/* src u8 [B,H,W,3] -> dst u8 [B,S,S,3]: crop box, Pillow BILINEAR resize (horizontal then
 * vertical 8bpc pass), optional horizontal flip.  tmp: u8 scratch [B,H,S,3]. */
int vitssl_aug_resized_crop_u8(const uint8_t* src, const int32_t* iparams, uint8_t* tmp, uint8_t* dst, int B, int H,
                               int W, int S, void* stream);
/* in place on u8 [B,S,S,3]: the four ColorJitter ops in the sampled order, then grayscale if
 * iparams[9]; S*S*3 <= 150 KiB (S <= 224) */
int vitssl_aug_color_u8(uint8_t* img, const int32_t* iparams, const float* fparams, int B, int S, void* stream);
/* u8 [B,S,S,3] -> f32 [B,3,S,S]: ksize x ksize Gaussian blur (reflect padding, float32, rounded
 * back to uint8) followed by ToTensor (/255) */
int vitssl_aug_blur_to_tensor(const uint8_t* img, const float* fparams, float* out, int B, int S, int ksize,
                              void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VITSSL_HIP_H */
