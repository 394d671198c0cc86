/*
 * vitk.h -- C ABI of libvitk.so: hand-written gfx950 (MI355X) kernels for the ViT encoder
 * forward/backward hot path.
 *
 * The reference (TidalPaladin/vit) has no FFI: its "operators" are pure-Python functions
 * decorated with torch.compile.  Each entry point below therefore cites the reference
 * FUNCTION whose arithmetic it replaces (paths relative to the reference checkout).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; the library never
 *     allocates, frees or keeps a pointer after the call returns (workspaces come from the caller);
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*); no implicit sync;
 *   - return value 0 = launched; non-zero = rejected, with the reason in vitk_last_error()
 *     (thread-local, valid until the next failing call on the same thread);
 *   - dtype codes: VITK_F32 = 0, VITK_BF16 = 1;
 *   - functions are re-entrant (forward runs on the caller's thread, backward on the autograd engine's
 *     thread).  State kept between calls: the thread-local error string, the CU count of the first device
 *     used, and (vitk_colsum_static only) the per-device address of its built-in scratch array.  Nothing
 *     else: options travel in the argument structs.
 */
#ifndef VITK_H
#define VITK_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { VITK_F32 = 0, VITK_BF16 = 1 };

/* activation codes: vit/helpers.py:35-48 */
enum { VITK_ACT_RELU = 0, VITK_ACT_SILU = 1, VITK_ACT_GELU = 2, VITK_ACT_SRELU = 3, VITK_ACT_OPENSWIGLU = 4 };

int vitk_version(void);
const char* vitk_last_error(void);

/* ---- normalisation: vit/norm.py:40-50 (apply_norm -> F.rms_norm / F.layer_norm) ------------
 * y[r,:] = norm(x[row(r),:]) * w (+ b);  row(r) = row_idx ? row_idx[r] : r  (the gather of
 * vit/transformer.py:14-27 fused into the load).  rstd (and mean for LayerNorm) are saved per
 * output row for the backward pass.  Statistics are fp32. */
int vitk_norm_fwd(const void* x, int x_dtype, const void* w, const void* b, int w_dtype, void* y, int y_dtype,
                  float* rstd, float* mean, const int64_t* row_idx, int64_t rows, int64_t D, float eps,
                  int is_layernorm, void* stream);

/* Backward of the above.  dx[row(r),:] = (add ? add[row(r),:] : 0) + d norm / d x.
 * dw/db (dtype w_dtype) are reduced over rows through `workspace` (fp32, vitk_norm_bwd_workspace_bytes(D)).
 * `add` may alias `dx`. dw/db may be NULL (frozen parameters).  dx_bf16 (or NULL): a bf16 copy of the rows written to
 * dx, for the consumer that multiplies this gradient in bf16 next (saves it a pass over the fp32 rows). */
size_t vitk_norm_bwd_workspace_bytes(int64_t D);
int vitk_norm_bwd(const void* dy, int dy_dtype, const void* x, int x_dtype, const void* w, int w_dtype,
                  const float* rstd, const float* mean, const void* add, void* dx, int dx_dtype, void* dw, void* db,
                  const int64_t* row_idx, int64_t rows, int64_t D, int is_layernorm, void* workspace, void* dx_bf16, void* stream);

/* Backward of the per-head normalisation of q and k (vit/attention.py:25-41, applied at :95-105), in place on the packed
 * gradient dqkv [M, ld] whose first two sections of H * Dh columns hold the gradients of the normalised q and k heads:
 * each head row is replaced by the gradient of the un-normalised head.  pre_q / pre_k [M * H, Dh] (dtype of dqkv) are the
 * heads before normalisation, rstd / mean [M * H] their row statistics (mean NULL for RMSNorm), w_q / w_k [Dh] the weights.
 * dw / db [Dh] (dtype w_dtype; NULL for a frozen parameter) are summed through `workspace`
 * (vitk_qk_norm_bwd_workspace_bytes(Dh)).  head_dim: a multiple of 8, at most 128. */
size_t vitk_qk_norm_bwd_workspace_bytes(int64_t Dh);
int vitk_qk_norm_bwd(void* dqkv, int dtype, const void* pre_q, const void* pre_k, const float* rstd_q, const float* rstd_k,
                     const float* mean_q, const float* mean_k, const void* w_q, const void* w_k, int w_dtype, void* dw_q,
                     void* db_q, void* dw_k, void* db_k, int64_t M, int64_t H, int64_t Dh, int64_t ld, int is_layernorm,
                     void* workspace, void* stream);

/* ---- GEMM: F.linear at vit/attention.py:93,213, vit/fused.py:114,117,153,164, the conv of
 *      vit/patch_embed.py:54 (as im2col GEMM), and their autograd backward (dgrad / wgrad) -------
 * C[M,N] = op(A) x op(B), fp32 accumulate.
 *   trans_a = 0: A is [M,K] row-major (lda);   trans_a = 1: A is [K,M] row-major (lda)
 *   trans_b = 0: B is [N,K] row-major (ldb) -- nn.Linear weight layout;  trans_b = 1: B is [K,N] row-major
 * Epilogues (vit/transformer.py:30-38 merge, vit/layer_scale.py:8-13, F.dropout at attention.py:214 /
 * fused.py:118,165, `y + pos` at patch_embed.py:57, vit/vit.py:500-505 prefix offset):
 *   VITK_EPI_STORE    out[m,n] = acc + bias[n]
 *   VITK_EPI_RESIDUAL t = dropout(round(acc + bias[n])); aux[m,n] = t (optional);
 *                     out[row(m),n] = res[row(m),n] + alpha * gamma[n] * t      (gamma optional)
 *   VITK_EPI_PATCH    out[(m / tokens) * seq + prefix + m % tokens, n] = acc + bias[n] + pos[m % tokens, n]
 * algo: 0 = auto, 1 = portable scalar kernel (any dtype/shape), 2 = bf16 MFMA kernel (rejects what it cannot do),
 *       3 / 4 / 5 = MFMA kernel forced to the 256x256 tile with register staging / the 128x128 tile / the 256x256 tile
 *       with LDS-DMA staging (tuning and tests; 0 picks between the last two).
 * split-K (wgrad) partials live in `workspace` (vitk_gemm_workspace_bytes). */
enum {
  VITK_EPI_STORE = 0, VITK_EPI_RESIDUAL = 1, VITK_EPI_PATCH = 2,
  /* MLP activation fused into the producing GEMM (MFMA kernel only; vit/fused.py:114-116,153-161 and their backward):
   *   ACT  (fc1, plain)    aux[m,n] = acc + bias (pre-activation, [M,N]);  out[m,n] = dropout(act(aux))
   *   GLU  (fc1, gated)    N = 2F: aux[M,2F] = acc + bias;  out[m,f] = dropout(act(aux[m,F+f]) * aux[m,f]), out is [M,F]
   *   DACT (fc2 dgrad)     out[m,n] = dropmask * acc * act'(aux[m,n])
   *   DGLU (fc2 dgrad)     N = F, aux = pre-activations [M,2F]: out[m,f] = d lin, out[m,F+f] = d gate, out is [M,2F]
   * dropout index = m*F + f as in vitk_act_fwd/bwd, so fused and unfused paths draw the same mask. */
  VITK_EPI_ACT = 3, VITK_EPI_GLU = 4, VITK_EPI_DACT = 5, VITK_EPI_DGLU = 6,
  /* ROPE (the fused QKV projection of vit/attention.py:93 followed by the optional per-head normalisation and by
   * apply_rope on q and k, vit/rope.py:148-163, for head_dim 64; MFMA kernel only, bf16 operands and output,
   * trans_a = trans_b = 0):
   *   x = round(acc + bias);  rows whose index inside their sample (m % seq) is below `prefix`, and columns from
   *   2 * rope_width on (the v section), are stored as they are; elsewhere, with t = m % seq - prefix and d = n % 64,
   *   out[m,n] = x[m,n] cos[t,d] - x[m,n+32] sin[t,d]  (d < 32),   x[m,n] cos[t,d] + x[m,n-32] sin[t,d]  (d >= 32)
   *   -- bit-identical to VITK_EPI_STORE followed by vitk_rope_qk with the same table. */
  VITK_EPI_ROPE = 7
};

typedef struct VitkGemm {
  const void* a;
  const void* b;
  void* out;
  const void* bias;      /* [N] or NULL, dtype = in_dtype */
  const void* res;       /* RESIDUAL: [*, ldo] dtype out_dtype */
  const void* gamma;     /* RESIDUAL: [N] dtype out_dtype, or NULL */
  const int64_t* row_idx;/* RESIDUAL: [M] or NULL */
  void* aux;             /* RESIDUAL: [M, N] dtype in_dtype, or NULL */
  const void* pos;       /* PATCH: [tokens, N] dtype in_dtype (float32 if pos_f32), or NULL */
  void* workspace;
  int64_t workspace_bytes;
  int64_t M, N, K;
  int64_t lda, ldb, ldo;
  int64_t tokens, seq, prefix; /* PATCH; ROPE: seq = rows per sample, prefix = leading rows of a sample left unrotated */
  uint64_t seed;         /* dropout stream */
  float alpha;
  float drop_p;
  int32_t trans_a, trans_b;
  int32_t in_dtype;      /* dtype of a, b, bias, aux, pos */
  int32_t out_dtype;     /* dtype of out, res, gamma */
  int32_t epilogue;
  int32_t algo;
  int32_t act;           /* ACT/GLU/DACT/DGLU: activation code */
  int32_t use_limit;     /* GLU/DGLU: clamp as vitk_act_fwd */
  int32_t use_extra;
  float limit;
  float extra_bias;
  void* a_colsum;        /* STORE with trans_a = 1 only, or NULL: a_colsum[m] = sum_k A[k,m], dtype out_dtype -- the bias
                            gradient (autograd's dy.sum(0)) that accompanies a weight gradient dW = dy^T x, from the same pass */
  int32_t dynamic_tiles; /* 1: the workgroups of the large-tile kernels draw their tiles from counters (kept in the last
                            256 bytes of the workspace and zeroed on `stream` by the call) instead of taking fixed shares;
                            for callers whose collectives run on another stream beside the GEMMs and hold CUs meanwhile
                            (vit_amd/ddp.py).  vitk_gemm_workspace_bytes accounts for the counters. */
  int32_t pos_f32;       /* PATCH: 1 = pos holds float32 values whatever in_dtype is, and is added to the rounded tokens
                            without a further rounding; out_dtype must be float32.  This is a learned position table
                            under autocast: vit/pos_enc.py:79-90 returns the fp32 parameter, vit/patch_embed.py:57 adds
                            it to the bf16 convolution output, and the promoted sum is fp32. */
  const float* rope;     /* ROPE: [2, seq - prefix, 64] float32, sin then cos (one table for all samples) */
  int64_t rope_width;    /* ROPE: columns per section, a multiple of 64 (N = 3 * rope_width for the packed q | k | v) */
  int32_t rope_emulate_bf16; /* ROPE: as in vitk_rope_qk */
  /* ROPE epilogue, optional per-head normalisation of q and k BEFORE the rotation (_apply_qk_norm, vit/attention.py:25-41,
   * :95-105; head_dim 64): y = (x - mean) * rstd * w (+ b) over the 64 columns of a head, x = round(acc + bias), the result
   * rounded to bf16 -- bit-identical to vitk_norm_fwd on the heads copied out as [M * heads, 64].  rope may be NULL then
   * (normalisation only).  What the backward needs is written beside the output:
   *   aux     [2][M][rope_width] bf16: the un-normalised q, then k
   *   qk_rstd [2][M * heads] float32, qk_mean likewise (LayerNorm only). */
  const float* qk_norm_weight;  /* [2][64] float32: q_norm.weight then k_norm.weight, or NULL (no normalisation) */
  const float* qk_norm_bias;    /* [2][64] float32 or NULL */
  float* qk_rstd;
  float* qk_mean;
  float qk_eps;
  int32_t qk_layernorm;
  /* Tuning knobs of the MFMA kernels; 0 = the library's choice (what production callers pass).  They exist for A/B
   * measurements (tools/bench_policy.py); neither changes a result.
   *   store_policy  1: every epilogue output store is plain (its lines stay in the XCD's L2);
   *                 2: the 16-byte output stores of the activation-backward epilogues (DGLU, DACT) are written through
   *                    (sc1: the line is not kept in L2); the other epilogues' stores stay plain (write-through stores
   *                    measured slower on them);
   *                 3: as 2, and the register path of the epilogues asks for its read-once rows (saved pre-activations,
   *                    residual rows) with the streaming (nt) policy.
   *                 0 behaves as 1 unless the environment sets VITK_WT_STORES=1.
   *   panel_m       row tiles per panel of the tile walk (tiles are walked panel by panel, column tile by column tile
   *                 inside a panel, so that a panel's A rows stay in L2 across its column tiles).  0: 4 when the output
   *                 is at most four 256-column tiles wide, 8 otherwise. */
  int32_t store_policy;
  int32_t panel_m;
} VitkGemm;

size_t vitk_gemm_workspace_bytes(const VitkGemm* g);
int vitk_gemm(const VitkGemm* g, void* stream);

/* ---- attention: F.scaled_dot_product_attention at vit/attention.py:209-211 (non-causal, no mask,
 *      scale Dh^-0.5, dropout on the probabilities) with the head split/merge of :15-22 folded into
 *      the strides -------------------------------------------------------------------------------
 * q,k,v,o (and dq,dk,dv,do) are addressed as ptr[b*sb + h*sh + s*ss + d], d contiguous; this lets the
 * kernels read the packed [B,S,3,H,Dh] projection output and write [B,S,H*Dh] directly.
 * lse [B,H,S] fp32 = log-sum-exp of the scaled scores (saved for backward). delta [B,H,S] fp32 scratch.
 * Cross-attention (CrossAttention / AttentivePool, vit/attention.py:218-280, :594-667): Sk > 0 keys per sample
 * with their own strides kv_* (k, v) and dkv_* (dk, dv); S then counts the queries.  Sk = 0 is self-attention
 * (keys = queries, k/v/dk/dv use the q and dq strides).  Both kernel families take either form; the fused inverse
 * rotation (`rope`) is for self-attention only. */
typedef struct VitkAttn {
  const void* q; const void* k; const void* v; void* o;
  const void* dout; void* dq; void* dk; void* dv;   /* backward only */
  float* lse; float* delta;
  int64_t B, H, S, Dh;
  int64_t q_sb, q_sh, q_ss;     /* strides of q/k/v (elements) */
  int64_t o_sb, o_sh, o_ss;     /* strides of o / dout */
  int64_t dq_sb, dq_sh, dq_ss;  /* strides of dq/dk/dv */
  uint64_t seed;
  float scale;
  float drop_p;
  int32_t dtype;
  int32_t algo;                 /* 0 auto, 1 portable, 2 MFMA */
  int64_t Sk;                   /* 0 = self-attention */
  int64_t kv_sb, kv_sh, kv_ss;     /* strides of k/v when Sk > 0 */
  int64_t dkv_sb, dkv_sh, dkv_ss;  /* strides of dk/dv when Sk > 0 */
  /* vitk_attn_bwd only, MFMA kernels with head_dim 64: q and k were rotated (apply_rope, vit/rope.py:148-163) before the
   * attention, so dq and dk leave through the transposed rotation -- bit-identical to vitk_rope_qk(inverse = 1) on the
   * dq / dk this call would otherwise store.  rope = NULL: no rotation. */
  const float* rope;            /* [2, S - rope_prefix, 64] float32, sin then cos */
  int64_t rope_prefix;          /* leading rows of every sample that were not rotated */
  int32_t rope_emulate_bf16;    /* as in vitk_rope_qk */
} VitkAttn;

int vitk_attn_fwd(const VitkAttn* a, void* stream);
int vitk_attn_bwd(const VitkAttn* a, void* stream);
/* softmax(scale * q k^T) itself, [B, H, S, Sk or S] fp32: attention_weights_qkv_packed / _q_kv_packed,
 * vit/attention.py:281-373 (diagnostics; v, o, lse of the descriptor are not used). */
int vitk_attn_weights(const VitkAttn* a, float* weights, void* stream);

/* ---- RoPE: vit/rope.py:139-163 (apply_rope: half-split rotation, prefix rows untouched) -------------
 * In place on the q and k sections of the packed projection qkv[B*S, 3*H*Dh]. table = [2, S-prefix, Dh]
 * fp32 (sin then cos).  inverse = 1 applies the transpose (backward).  emulate_bf16 = 1 rounds each
 * product and the sum to bf16 like eager bf16 arithmetic does when the table itself is bf16.
 * per_sample = 1: table = [2, B, S-prefix, Dh], one table per sample -- the masked-token path, where every
 * sample keeps its own subset of positions (vit/vit.py:522-527, vit/rope.py:139-163 with a 5-D table). */
int vitk_rope_qk(void* qkv, int dtype, const float* table, int64_t B, int64_t S, int64_t H, int64_t Dh,
                 int64_t prefix, int inverse, int emulate_bf16, int per_sample, void* stream);
/* The same rotation on the first nsec sections (H*Dh wide each) of rows that are row_stride elements apart:
 * q alone (row_stride = H*Dh, nsec = 1) or the k half of a packed [k | v] projection (row_stride = 2*H*Dh,
 * nsec = 1) -- rope_q / rope_k of project_q_kv_packed, vit/attention.py:156-159. */
int vitk_rope_rows(void* x, int dtype, const float* table, int64_t B, int64_t S, int64_t H, int64_t Dh,
                   int64_t prefix, int64_t row_stride, int64_t nsec, int inverse, int emulate_bf16,
                   int per_sample, void* stream);

/* ---- MLP activation: vit/fused.py:115-116 (plain) and :154-161 (GLU), vit/helpers.py:9-48 ----------
 * plain: h = dropout(act(p));  GLU: lin = p[:, :F], glu = p[:, F:], h = dropout(act(min(glu,limit)) *
 * (clamp(lin,+-limit) + extra_bias)).  use_limit/use_extra select the optional terms. */
int vitk_act_fwd(const void* p, void* h, int dtype, int64_t rows, int64_t F, int is_glu, int act, int use_limit,
                 float limit, int use_extra, float extra_bias, float drop_p, uint64_t seed, void* stream);
int vitk_act_bwd(const void* dh, const void* p, void* dp, int dtype, int64_t rows, int64_t F, int is_glu, int act,
                 int use_limit, float limit, int use_extra, float extra_bias, float drop_p, uint64_t seed,
                 void* stream);

/* ---- patch unfolding for the conv-as-GEMM of vit/patch_embed.py:25-31,54 ---------------------------
 * cols[(b*gh + i)*gw + j, (c*ph + u)*pw + v] = x[b, c, i*ph + u, j*pw + v], zero padded to ldc columns.
 * vitk_col2im is the exact inverse scatter (gradient w.r.t. the image). */
int vitk_im2col(const void* x, int x_dtype, void* cols, int cols_dtype, int64_t B, int64_t C, int64_t H, int64_t W,
                int64_t ph, int64_t pw, int64_t ldc, void* stream);
int vitk_col2im(const void* cols, int cols_dtype, void* dx, int dx_dtype, int64_t B, int64_t C, int64_t H, int64_t W,
                int64_t ph, int64_t pw, int64_t ldc, void* stream);

/* ---- reductions over rows: bias gradients, layer-scale gradients, register/cls token gradients ----
 * out[n] = sum_r a[ra(r), n] * (b ? b[r, n] : 1) * scale;  ra(r) = (a_idx ? a_idx[r] : r) * 1.
 * `a` rows are addressed with stride lda, `b` with ldb.  workspace: vitk_colsum_workspace_bytes(N) fp32. */
size_t vitk_colsum_workspace_bytes(int64_t N);
int vitk_colsum(const void* a, int a_dtype, int64_t lda, const int64_t* a_idx, const void* b, int b_dtype, int64_t ldb,
                void* out, int out_dtype, int64_t rows, int64_t N, float scale, void* workspace, void* stream);

/* vitk_colsum without a workspace argument and without any allocation.  No counterpart in the reference: it is for
 * callers that cannot allocate device memory around the call (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  The per-chunk partial sums go to a device array built into the library (8 MiB per
 * GPU, held until the process ends), so N is limited to 8192 columns; larger N is rejected with an error.  All calls on
 * one GPU share that array: calls that overlap on one GPU (different streams) race -- serialise them.  The first call on
 * a device looks up the array's address (no allocation); make that call outside graph capture.  The array used is the
 * CURRENT device's: the current device must be the one that owns `stream` and the operands, and a call whose `out`
 * lives on another device is rejected. */
int vitk_colsum_static(const void* a, int a_dtype, int64_t lda, const int64_t* a_idx, const void* b, int b_dtype, int64_t ldb,
                       void* out, int out_dtype, int64_t rows, int64_t N, float scale, void* stream);

/* ---- gradient preparation for the residual branch: backward of merge + layer scale + dropout -------
 * (vit/transformer.py:30-38, vit/layer_scale.py:8-13, F.dropout):
 * out[r, n] = dropmask(r, n) * alpha * gamma[n] * g[row(r), n], cast to out_dtype. */
int vitk_scale_cast(const void* g, int g_dtype, int64_t ldg, const int64_t* row_idx, const void* gamma, void* out,
                    int out_dtype, int64_t rows, int64_t N, float alpha, float drop_p, uint64_t seed, void* stream);

/* The same, with the gradient of the layer scale (vit/layer_scale.py:8-13) from the same pass over g:
 * dgamma[n] = alpha * sum_r g[row(r), n] * aux[r, n], aux [rows, N] = the branch output the scale multiplied (what the
 * residual epilogue of vitk_gemm saves), dgamma in the dtype of g / gamma.  N a multiple of 4, at most 4096. */
size_t vitk_scale_cast_dgamma_workspace_bytes(int64_t N);
int vitk_scale_cast_dgamma(const void* g, int g_dtype, int64_t ldg, const int64_t* row_idx, const void* gamma, const void* aux,
                           int aux_dtype, void* out, int out_dtype, void* dgamma, int64_t rows, int64_t N, float alpha,
                           float drop_p, uint64_t seed, void* workspace, void* stream);

/* ---- prefix tokens (vit/vit.py:500-505: torch.cat([cls_tokens.expand(B), register_tokens.expand(B), x], dim=1)) ----
 * out is the [B, S, D] sequence whose rows ncls + nreg .. S - 1 the patch-embedding GEMM writes (VITK_EPI_PATCH with
 * seq = S, prefix = ncls + nreg); this call fills rows 0 .. ncls + nreg - 1 of every sample with the learned tokens
 * cls [ncls, D] and reg [nreg, D] (both of prefix_dtype), cast to out_dtype.  D % 4 == 0. */
int vitk_prefix_fill(const void* cls, int64_t ncls, const void* reg, int64_t nreg, int prefix_dtype, void* out, int out_dtype,
                     int64_t B, int64_t S, int64_t D, void* stream);

/* ---- Fourier position features (vit/pos_enc.py:137-149: `features = grid @ w_fourier`, `cat([sin, cos], -1) / sqrt(J)`) --
 * grid [N, Kc] fp32 coordinates (Kc = 1..3), w [Kc, J] fp32; out [N, 2 J] of out_dtype = sin | cos, scaled by 1 / sqrt(J).
 * The D x D projection that follows is a vitk_gemm call.  Backward: dw [Kc, J] fp32 from g = d out [N, 2 J] of g_dtype
 * (the features are recomputed, nothing is kept from the forward call). */
int vitk_fourier_features(const float* grid, const float* w, void* out, int out_dtype, int64_t N, int64_t Kc, int64_t J,
                          void* stream);
int vitk_fourier_features_bwd(const float* grid, const float* w, const void* g, int g_dtype, float* dw, int64_t N, int64_t Kc,
                              int64_t J, void* stream);

/* ---- drop-path pass-through (vit/transformer.py:41-46: `x.flatten(1).index_add(0, keep_indices, ...)` leaves the samples
 * outside the subset unchanged) ---------------------------------------------------------------------------------------
 * dst[idx[i]] = src[idx[i]] for i < n, a sample being `sample_bytes` contiguous bytes (a multiple of 16; src and dst 16-byte
 * aligned).  Called with the dropped samples before the residual epilogue (or the norm backward) writes the kept ones. */
int vitk_copy_samples(const void* src, void* dst, const int64_t* idx, int64_t n, int64_t sample_bytes, void* stream);

/* Diagnostics (no counterpart in the reference): `blocks` idle workgroups that hold their CUs for about `cycles` shader
 * clocks, to measure what a launch loses when another stream's kernel occupies part of the chip. */
int vitk_debug_hold_cus(int blocks, long long cycles, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VITK_H */
