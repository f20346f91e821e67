/* lt_amd.h -- C ABI of the MI355X-native DINOv2 training-step kernels (liblt_amd.so).
 *
 * The reference (lightly-ai/lightly-train v0.17.0) has no FFI for this path: every op below
 * replaces a PyTorch ATen call sequence inside the reference's Python `Method.training_step_impl`
 * / `ModelWrapper.forward_features`.  Each entry point cites the reference
 * lines (relative to src/lightly_train/, "LT/") whose arithmetic it implements.
 *
 * Conventions
 *   - all pointers are DEVICE pointers borrowed from the caller (torch tensors); nothing is
 *     allocated or freed inside the library; every launch goes to the caller's `stream`
 *     (a hipStream_t passed as void*; NULL = default stream);
 *   - bf16 tensors are raw uint16 bfloat16 bits; "f32" = IEEE float; row-major, last dim contiguous;
 *   - return 0 (LT_OK) or a negative code; lt_last_error() returns a thread-local message;
 *   - re-entrant per stream; the library allocates nothing.  The only process-global state is the scratch of the deterministic
 *     grad-norm reduction (lt_sumsq_f32: 16 slots of per-block partials in device memory, handed out round-robin, i.e. at most
 *     16 lt_sumsq_f32 launches may be in flight at once) and the one-time hipFuncSetAttribute of the 128-KiB-LDS GEMM kernels.
 */
#ifndef LT_AMD_H
#define LT_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LT_OK 0
#define LT_ERR_INVALID (-22)
#define LT_ERR_HIP (-5)

const char* lt_last_error(void);
int lt_abi_version(void);
/* device name / CU count of the current device (diagnostics for bench.py) */
int lt_device_info(char* name, int name_len, int* compute_units, int* clock_khz);

/* ------------------------------------------------------------------------------------------
 * GEMM  (LT/_models/dinov2_vit/dinov2_vit_src/layers/attention.py:44,51-55,64  qkv / proj Linear;
 *        layers/mlp.py:36-42 fc1/GELU/fc2; layers/patch_embed.py:77-79,108-110 Conv2d as GEMM;
 *        layers/layer_scale.py:27-28 + layers/block.py:90-115 gamma*branch + residual;
 *        LT/_methods/dinov2/dinov2_head.py:66-71,85-95 head MLP and prototype layer; and the
 *        autograd backward of all of them)
 *   C[M,N] = opA(A) . opB(B), bf16 operands, fp32 accumulate (v_mfma_f32_32x32x16_bf16).
 *   trans_a = 0: A is [M][K] (lda = row stride, K contiguous); 1: A is [K][M] (M contiguous)
 *   trans_b = 0: B is [N][K] (nn.Linear weight layout);          1: B is [K][N]
 * ------------------------------------------------------------------------------------------ */
enum {
  LT_EPI_BF16 = 0,          /* C(bf16) = alpha*acc + bias                                        */
  LT_EPI_BF16_GELU = 1,     /* C2(bf16, optional) = pre = alpha*acc + bias; C(bf16) = gelu(pre)   */
  LT_EPI_RESID = 2,         /* y = alpha*acc + bias; C2(bf16, optional) = y; C(f32) = resid + branch_scale*rowscale[m]*gamma*y */
  LT_EPI_F32 = 3,           /* C(f32) = alpha*acc + bias                                         */
  LT_EPI_BF16_GELUGRAD = 4, /* C(bf16) = alpha*acc * gelu'(aux(bf16))                            */
  LT_EPI_F32_ACCUM = 5      /* C(f32) += alpha*acc   (split_k > 1 -> atomic adds)                */
};

typedef struct lt_gemm_desc {
  const void* A; const void* B;   /* bf16 */
  int M, N, K;
  int lda, ldb;
  int trans_a, trans_b;
  int epilogue;
  void* C; int ldc;               /* primary output (dtype by epilogue) */
  void* C2; int ldc2;             /* optional secondary bf16 output */
  const float* bias;              /* [N] or NULL */
  const float* gamma;             /* [N] LayerScale (LT_EPI_RESID) or NULL (=1) */
  const float* resid; int ldr;    /* [M][N] f32 residual (LT_EPI_RESID) or NULL (=0) */
  const void* aux; int ldaux;     /* [M][N] bf16 pre-activation (LT_EPI_BF16_GELUGRAD) */
  float alpha;
  int split_k;                    /* >1 only honoured for LT_EPI_F32_ACCUM */
  int force_kernel;               /* 0 = auto, 1 = 128x128 register-staged kernel, 2 = 256x256 LDS-DMA kernel with the 2-stage K-loop,
                                     8 = 256x256 LDS-DMA kernel with the four-phase ping-pong K-loop, 11 = the same tile and phases with the static-address
                                     K-loop (round 6; the default for large shapes with K % 64 == 0; falls back to 8 where it is not eligible) */
  const float* rowscale;          /* [M] per-row multiplier of the LayerScale branch (LT_EPI_RESID; per-sample DropPath) or NULL */
  float branch_scale;             /* scalar multiplier of the branch (LT_EPI_RESID; batch-subset stochastic depth b/s); 0 = 1 */
  void* workspace; size_t workspace_bytes; /* optional f32 scratch for deterministic slab split-K (LT_EPI_F32_ACCUM) */
  int batch;                      /* > 1: `batch` independent problems of this shape, operands `stride_*` elements apart (plain
                                     epilogues only; 0 / 1 = single problem) */
  int64_t stride_a, stride_b, stride_c;
  float* colsum;                  /* trans_a only (weight gradients dW = dY^T X, A = dY stored [K][M]): colsum[m] += sum_k A[k][m], m < M -- the bias
                                     gradient of the same Linear (the column sums of dY), formed from the A fragments the kernel holds anyway
                                     instead of a second pass over dY.  Fused into the four-phase slab kernel when the reduction ledger
                                     (lt_reduce_begin) is open; otherwise a separate column-sum launch precedes the GEMM.  NULL = none */
  /* LT_EPI_RESID only (round 6): the NEXT LayerNorm of the residual stream handed to the GEMM call -- x + ls(attn(norm1(x))) followed by norm2,
   * x + ls(mlp(norm2(x))) followed by the next block's norm1 (LT/.../layers/block.py:90-115, :60,74):
   *   ln_out bf16 [M][N] (row stride N) = LayerNorm(C[m][0..N), ln_eps) * ln_weight + ln_bias,  ln_mean / ln_rstd f32 [M] (may be NULL).
   * The library issues lt_layernorm_fwd on the stream behind the GEMM (one call across this ABI instead of two; a row-owning kernel that
   * normalised inside the GEMM was measured slower and removed, profiles/r06_rowln_probe.md).  ln_out NULL = none. */
  const float* ln_weight; const float* ln_bias; void* ln_out; float* ln_mean; float* ln_rstd; float ln_eps;
} lt_gemm_desc;

int lt_gemm_bf16(const lt_gemm_desc* d, void* stream);
/* one-thread-per-output fp32-accumulate GEMM on the same bf16 operands (cross-check only) */
int lt_gemm_bf16_naive(const void* A, const void* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                       int trans_a, int trans_b, void* stream);
/* tiny fp32 matmul C[M,N] (+)= op(A)[M,K] . B[K,N]  (pos-embed bicubic map, vision_transformer.py:251-305) */
int lt_matmul_f32(const float* A, const float* B, float* C, int M, int N, int K, int trans_a, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------
 * Token path (vision_transformer.py:307-329 prepare_tokens_with_masks; patch_embed.py:86-114)
 * ------------------------------------------------------------------------------------------ */
/* separable 4-tap image resize (the bicubic pad-resize to the next multiple of the patch size, patch_embed.py:90-99):
 * out[p,y,x] = sum_{a,b<4} wy[y,a]*wx[x,b]*in[p, iy[y,a], ix[x,b]]; taps precomputed on the host from F.interpolate.
 * The four tap tables ([Ho,4] / [Wo,4]) must be 16-byte aligned (LT_ERR_INVALID otherwise). */
int lt_resize_4tap(const float* in, float* out, const int32_t* iy, const float* wy, const int32_t* ix, const float* wx,
                   int planes, int H, int W, int Ho, int Wo, void* stream);
/* img f32 [B,C,H,W] -> cols bf16 [B*gh*gw, kpad], k = (c*p + py)*p + px, zero padded to kpad */
int lt_im2col_bf16(const float* img, void* cols, int B, int C, int H, int W, int p, int kpad, void* stream);
/* Rotary position embedding of DINOv3 (reference _models/dinov3/dinov3_src/layers/attention.py:23-34,79-103): in place on
 * the packed qkv [B,N,3,H,dh] bf16, q and k of tokens >= prefix become x*cos + rotate_half(x)*sin; sin/cos f32 [N-prefix, dh]
 * (layers/rope_position_encoding.py:62-117, computed by the caller); inverse = 1 applies the transposed rotation (backward) */
int lt_rope_apply(void* qkv_bf16, const float* sin_t, const float* cos_t, int B, int N, int H, int dh, int prefix, int inverse,
                  void* stream);

/* SwiGLU FFN gate (reference layers/swiglu_ffn.py:31-35): x12 bf16 [rows, 2H] = [x1 | x2], out bf16 [rows, H] =
 * silu(x1) * x2.  Backward: d12 bf16 [rows, 2H] = [dh * x2 * silu'(x1) | dh * silu(x1)].  H % 8 == 0. */
int lt_swiglu_fwd(const void* x12_bf16, void* out_bf16, int64_t rows, int H, void* stream);
int lt_swiglu_bwd(const void* x12_bf16, const void* dh_bf16, void* d12_bf16, int64_t rows, int H, void* stream);

/* nn.GELU() (erf form) on n bf16 elements, and its backward dx = dy * gelu'(x) with x the saved pre-activation: the activation of the
 * BatchNorm projection heads (reference dinov2_head.py:86-92: Linear, BatchNorm1d, GELU), where it cannot ride a GEMM epilogue.
 * n % 8 == 0, tensors 16-byte aligned; in-place (y == x, dx == dy) is allowed. */
int lt_gelu_fwd_bf16(const void* x, void* y, int64_t n, void* stream);
int lt_gelu_bwd_bf16(const void* dy, const void* x, void* dx, int64_t n, void* stream);

/* out f32 [B, n_out, D] = sparse linear map of in f32 [B, n_in, D]: out[b,o,:] = sum_{a<taps} w[o,a] * in[b, idx[o,a], :]
 * (bilinear resize of the student's spatial features onto the teacher grid, distillationv3.py:338-345; backward = the
 * transposed table).  Tables int32 / f32 [n_out, taps], built by the caller from F.interpolate. */
int lt_resample_tokens(const float* in, const int32_t* idx, const float* w, float* out, int B, int n_in, int n_out, int D, int taps,
                       void* stream);
/* tokens [cls | n_reg registers | n_p patches]: x[b,0]=cls+pos[0]; x[b,1+r]=reg[r] (no pos-embed);
 * x[b,1+n_reg+i]=(mask[b,i]?mask_token:patch[b*n_p+i])+pos[1+i]; masks / reg may be NULL */
int lt_assemble_tokens(const float* patch, const float* cls, const float* pos, const float* mask_token,
                       const uint8_t* masks, const float* reg, float* x, int B, int n_p, int n_reg, int D, void* stream);
/* backward: dpatch bf16 [B*n_p,D] (0 where masked); dcls[D], dpos[(1+n_p),D], dmask_token[D], dreg[n_reg,D] accumulate */
int lt_assemble_tokens_bwd(const float* dx, const uint8_t* masks, void* dpatch_bf16, float* dcls, float* dpos,
                           float* dmask_token, float* dreg, int B, int n_p, int n_reg, int D, void* stream);

/* ------------------------------------------------------------------------------------------
 * LayerNorm eps=1e-6 (vision_transformer.py:138; block.py:60,74)  x f32 [rows,D]
 * ------------------------------------------------------------------------------------------ */
int lt_layernorm_fwd(const float* x, const float* w, const float* b, void* y_bf16, float* y_f32, float* mean,
                     float* rstd, int rows, int D, float eps, void* stream);
/* dx = (dres ? dres : 0) + LN'(dy); dw/db accumulate. dy is bf16 unless dy_is_f32.
 * ws (optional, ws_floats >= 128*D): per-block partial sums + deterministic reduce instead of atomics. */
int lt_layernorm_bwd(const float* x, const float* w, const float* mean, const float* rstd, const void* dy,
                     int dy_is_f32, const float* dres, float* dx, float* dw, float* db, float* ws, int64_t ws_floats,
                     int rows, int D, void* stream);
/* same, plus the fused producer of the NEXT branch's upstream gradient (block.py:90-115 backward, the branch whose output
 * was added to this residual stream): dnext bf16 [rows,D] = dx * gamma_next * scale_next * rowscale_next[row] and
 * dbias_next[D] += column sums of dnext (bias gradient of the Linear that feeds that LayerScale).  dnext NULL = plain. */
int lt_layernorm_bwd_fused(const float* x, const float* w, const float* mean, const float* rstd, const void* dy, int dy_is_f32,
                           const float* dres, float* dx, float* dw, float* db, float* ws, int64_t ws_floats, void* dnext_bf16,
                           const float* gamma_next, const float* rowscale_next, float scale_next, float* dbias_next, int rows,
                           int D, void* stream);
/* the LayerNorm backward of a branch that ran on a row subset (batch-subset stochastic depth, block.py:118-141): row r of (x, mean, rstd, dy)
 * belongs to row ridx[r] (int64, device, no repeats) of the gradient stream: dx[ridx[r]] = dres[ridx[r]] + LN'(dy[r]); dres may be dx (in
 * place).  bf16 dy, D % 4 == 0, D <= 1024 */
int lt_layernorm_bwd_rows(const float* x, const float* w, const float* mean, const float* rstd, const void* dy, int dy_is_f32,
                          const float* dres, float* dx, const int64_t* ridx, float* dw, float* db, int rows, int D, void* stream);

/* LayerScale (+ stochastic depth) backward (layer_scale.py:27-28, block.py:118-141, drop_path.py:16-28):
 *   m_r = scale * (rowscale ? rowscale[r] : 1);  dy(bf16) = dout*gamma*m_r;  dgamma += sum_r dout*y*m_r;
 *   dbias (optional) += sum_r dy  (bias gradient of the Linear in front of LayerScale, fused).
 * gamma == NULL: dy = bf16(dout*m_r). */
int lt_layerscale_bwd(const float* dout, const void* y_bf16, const float* gamma, void* dy_bf16, float* dgamma,
                      float* dbias, const float* rowscale, float scale, int rows, int D, void* stream);
/* same, with row r of the branch reading its upstream gradient at row ridx[r] of `dout` (int64 on the device; NULL = row r): the rows of a
 * batch-subset stochastic-depth branch (block.py:118-141) or of the last block's loss rows inside the full gradient stream, without a gathered
 * copy in between.  y / dy / rowscale stay indexed by r.  D % 4 == 0 */
int lt_layerscale_bwd_rows(const float* dout, const int64_t* ridx, const void* y_bf16, const float* gamma, void* dy_bf16, float* dgamma,
                           float* dbias, const float* rowscale, float scale, int rows, int D, void* stream);
/* LayerScale gradient from the weight gradient instead of the saved branch output (layer_scale.py:27-28 backward):
 * dgamma[c] += (sum_k W[c,k] dW[c,k] + bias[c] dbias[c]) / gamma[c], W bf16 [N,K] = the Linear feeding the LayerScale,
 * dW/dbias = its accumulated gradients (computed from dD = dx*gamma).  Call once per step after all weight gradients.
 * The identity divides by gamma: a channel with gamma == 0 exactly (|gamma| <= 1e-30) has no recoverable gradient and is left
 * unchanged -- callers must not train a LayerScale initialised at 0 on this path (ViTEngine refuses init_values == 0). */
int lt_layerscale_dgamma(const void* W_bf16, const float* dW, const float* bias, const float* dbias, const float* gamma,
                         float* dgamma, int N, int K, void* stream);
/* the same for `batch` identically shaped layers in ONE launch: layer i's six tensors start `stride` elements behind layer i - 1's (the
 * blocks of a ViT in the flat parameter / gradient / bf16-shadow storages, which share their offsets) */
int lt_layerscale_dgamma_batched(const void* W_bf16, const float* dW, const float* bias, const float* dbias, const float* gamma,
                                 float* dgamma, int N, int K, int batch, int64_t stride, void* stream);
/* out[N] += column sums of a bf16 [rows,N] matrix (bias gradients) */
int lt_colsum_bf16(const void* x, float* out, int rows, int N, void* stream);
/* out[N] (+)= column sums of an f32 [rows,N] matrix (teacher center, dinov2_loss.py:139-145,274-282) */
int lt_colsum_f32(const float* x, float* out, int rows, int N, int accumulate, void* stream);
/* row gather: out[m,:] = src[idx[m],:]  (index_select, dinov2.py:427-431,496-500; cls rows incl. the
 * teacher half-swap dinov2.py:414-420).  Either output may be NULL. */
int lt_gather_rows(const float* src, int ld_src, const int64_t* idx, void* out_bf16, float* out_f32, int M, int D,
                   void* stream);
/* dst[idx[m],:] += src[m,:] (unique idx) */
int lt_scatter_add_rows(const float* src, const int64_t* idx, float* dst, int ld_dst, int M, int D, void* stream);
int lt_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream);
/* dst bf16 [R,Cpad] = zero-padded cast of src f32 [R,C]; dst f32 [R,C] += src f32 [R,Cpad][:, :C]  (patch-embed weights whose
 * 3*p*p is not a multiple of 8, e.g. patch 14: K = 588 -> 592) */
int lt_cast_pad_rows(const float* src, void* dst_bf16, int R, int C, int Cpad, void* stream);
int lt_unpad_accumulate(const float* src, float* dst, int R, int C, int Cpad, void* stream);
int lt_fill_f32(float* dst, float value, int64_t n, void* stream);
int lt_scale_f32(float* dst, float alpha, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------
 * Attention (attention.py:49-66): softmax(q*scale k^T) v on the packed qkv bf16 [B,N,3,H,dh].
 * head_dim 64 runs the MFMA flash kernels; other head dims run the generic kernels.
 * ------------------------------------------------------------------------------------------ */
int lt_attention_fwd(const void* qkv, void* out_bf16, float* lse, int B, int N, int H, int dh, float scale, void* stream);
/* dqkv bf16 [B,N,3,H,dh]; ws = f32 workspace of lt_attention_bwd_ws_floats(B,N,H,dh) floats */
int64_t lt_attention_bwd_ws_floats(int B, int N, int H, int dh);
int lt_attention_bwd(const void* qkv, const void* out_bf16, const void* dout_bf16, const float* lse, float* ws,
                     void* dqkv, int B, int N, int H, int dh, float scale, void* stream);

/* ------------------------------------------------------------------------------------------
 * Projection head pieces (dinov2_head.py:54-58,66-71)
 * ------------------------------------------------------------------------------------------ */
/* y = x / max(||x||, eps) per row; outputs bf16 y and the f32 inverse norm */
int lt_l2norm_fwd(const float* x, void* y_bf16, float* inv_norm, int rows, int D, float eps, void* stream);
/* dx(bf16) = (dy - y*(y.dy)) * inv_norm with y = x*inv_norm */
int lt_l2norm_bwd(const float* dy, const float* x, const float* inv_norm, void* dx_bf16, int rows, int D, void* stream);
/* weight_norm(dim=0): w[k,:] = v[k,:]*g[k]/||v[k,:]|| -> bf16 */
int lt_weightnorm_fwd(const float* v, const float* g, void* w_bf16, int K, int D, void* stream);
/* dv += g/||v|| * (dw - v*(v.dw)/||v||^2);  dg += (v.dw)/||v|| */
int lt_weightnorm_bwd(const float* dw, const float* v, const float* g, float* dv, float* dg, int K, int D, void* stream);

/* ------------------------------------------------------------------------------------------
 * Losses (LT/_methods/dinov2/dinov2_loss.py)
 * ------------------------------------------------------------------------------------------ */
/* probs = softmax((logits - center) * inv_temp) per row (:76-82, :178-186); center may be NULL */
int lt_softmax_center(const float* logits, const float* center, float* probs, int rows, int K, float inv_temp, void* stream);
/* the same centering without the probability matrix: stats f32 [rows][2] = (max_k z, 1 / sum_k exp(z - max)) of z = (logits - center) *
 * inv_temp per row, and colsum f32 [K] = column sums of the raw logits (overwritten; the center update of :139-145 / :274-282) -- one
 * pass over the teacher logits instead of three (probabilities written, column sums, cross-entropy).  lt_ce_fwd_bwd_logits rebuilds the
 * probabilities from (logits, stats, center).  scratch: caller-owned, >= K floats (256 * K for one workgroup per CU): per-workgroup
 * column sums, added in workgroup order (the library allocates nothing) */
int lt_softmax_stats_colsum(const float* logits, const float* center, float* stats, float* colsum, int rows, int K, float inv_temp,
                            float* scratch, int64_t scratch_floats, void* stream);
/* center = center*momentum + colsum*scale*(1-momentum) (:147-160, :284-297) */
int lt_center_ema(float* center, const float* colsum, float scale, float momentum, int K, void* stream);
/* student CE against 1 or 2 teacher rows (:117-133, :246-268):
 *   lsm = log_softmax(s*inv_temp);  l_r = -sum_k (t_a + t_b) * lsm
 *   loss[slot ? slot[r] : 0] += coef_r * l_r ;  dlogits(bf16)[r,:] = coef_r * inv_temp * (softmax(s*inv_temp)*sum(t) - (t_a+t_b))
 *   coef_r = scale * (row_weight ? row_weight[r] : 1);  t_a = teacher[ta[r]], t_b = (tb && tb[r] >= 0) ? teacher[tb[r]] : 0 */
int lt_ce_fwd_bwd(const float* s, const float* teacher, const int32_t* ta, const int32_t* tb, const float* row_weight,
                  const int32_t* slot, float scale, float inv_temp, float* loss, void* dlogits_bf16, int rows, int K,
                  void* stream);
/* lt_ce_fwd_bwd with t_x = softmax((t_logits[x] - center(x)) * inv_temp_t) rebuilt per element from the teacher logits, their row
 * statistics t_stats [rows_t][2] (lt_softmax_stats_colsum) and the center the statistics were formed with: center_a for teacher rows
 * < split_row (the DINO cls rows), center_b for the others (the iBOT patch rows); either may be NULL (= 0) */
int lt_ce_fwd_bwd_logits(const float* s, const float* t_logits, const float* t_stats, const float* center_a, const float* center_b,
                         int split_row, const int32_t* ta, const int32_t* tb, const float* row_weight, const int32_t* slot, float scale,
                         float inv_temp, float inv_temp_t, float* loss, void* dlogits_bf16, int rows, int K, void* stream);
/* The same two entry points on bf16 logit rows (round 6, the `bf16_logits` option of the DINOv2 step): what the reference's own bf16-mixed
 * path holds -- the prototype Linear runs under autocast (LT/_methods/dinov2/dinov2_head.py:66-71) and the losses cast back with .float()
 * (dinov2_loss.py:37-38,88).  Arithmetic in fp32, statistics / column sums / loss terms fp32; the generic path needs K % 8 == 0. */
int lt_softmax_stats_colsum_bf16(const void* logits_bf16, const float* center, float* stats, float* colsum, int rows, int K, float inv_temp,
                                 float* scratch, int64_t scratch_floats, void* stream);
int lt_ce_fwd_bwd_logits_bf16(const void* s_bf16, const void* t_logits_bf16, const float* t_stats, const float* center_a, const float* center_b,
                              int split_row, const int32_t* ta, const int32_t* tb, const float* row_weight, const int32_t* slot, float scale,
                              float inv_temp, float inv_temp_t, float* loss, void* dlogits_bf16, int rows, int K, void* stream);
/* Distillation v3 (reference _methods/distillationv3/distillationv3_loss.py:60-115): per row KL(softmax(t/T) || softmax(s/T));
 * loss[0] += coef * KL, dlogits bf16 = coef/T * (softmax(s/T) - softmax(t/T)); rows `ld` (dlogits: `ldd`) elements apart */
int lt_kl_fwd_bwd(const float* s_logits, const float* t_logits, int ld, float inv_temp, float coef, float* loss,
                  void* dlogits_bf16, int ldd, int rows, int K, void* stream);
/* g[b] = d[b] + d[b]^T for `batch` square bf16 matrices (n x n, row stride ld): upstream gradient of X X^T */
int lt_symmetrize_bf16(const void* d, void* g, int batch, int n, int ld, void* stream);
/* DistillationV3._mixup_data (distillationv3.py:356-368): out[b] = lam * x[b] + (1 - lam) * x[index[b]], index int64 [B] */
int lt_mixup(const float* x, const int64_t* index, float lam, float* out, int B, int64_t per_image, void* stream);

/* Sinkhorn-Knopp pieces (:84-115, :188-224); Q f32 [rows,K] holds exp(logits*inv_temp) */
int lt_sk_exp(const float* logits, float* Q, int64_t n, float inv_temp, void* stream);
/* Q[r,k] *= 1/(colsum[k]*K); then row-normalise: Q[r,:] /= (rowsum(r) * n_total); final: Q *= final_mul */
int lt_sk_iter(float* Q, const float* colsum, int rows, int K, float n_total, float final_mul, void* stream);
/* DistillationV2Loss (nn.MSELoss, LT/_methods/distillationv2/distillationv2_loss.py:14-44): *loss += scale * sum (s - t)^2 and
 * ds = 2 * scale * (s - t) (ds may be null); scale = 1 / numel for the mean.  *loss is ACCUMULATED into: zero it before the first call of a
 * step.  Deterministic two-level sum through library-owned scratch that rotates per call (launches on several streams may overlap). */
int lt_mse_fwd_bwd(const float* s, const float* t, float* ds, int64_t n, float scale, float* loss, void* stream);
/* KoLeo (lightly.loss.KoLeoLoss, call site dinov2.py:377-380): *loss += weight*L(x), dx += weight*dL/dx.
 * weight == 0: value only -- *loss += L(x) and dx is not touched (the reference logs the term at weight 0 too, dinov2.py:377-396).
 * ws: f32 workspace of 2*n*D + 2*n floats, nn: int32 workspace [n] */
int lt_koleo_fwd_bwd(const float* x, int ld, float* loss, float* dx, int ld_dx, int n, int D, float eps, float weight,
                     float* ws, int32_t* nn, void* stream);

/* ------------------------------------------------------------------------------------------
 * Optimizer / EMA on flat parameter storage (LT/_methods/dinov2/utils.py:191-250 AdamW groups,
 * dinov2.py:588-660 clip / WD schedule / EMA; LT/_torch_helpers.py:75-96)
 * The flat f32 buffer is split into segments; seg_of_chunk[i] gives the segment of 1024-element chunk i.
 * ------------------------------------------------------------------------------------------ */
/* out[0] += sum(g^2); deterministic (fixed summation order: identical gradients give identical bits on every rank).
 * At most 16 calls may be in flight on different streams at once. */
int lt_sumsq_f32(const float* g, float* out, int64_t n, void* stream);
/* AdamW (torch.optim.AdamW semantics, decoupled wd). clip_coef = min(1, max_norm/(||g||+1e-6)) is computed on
 * device from *sumsq.  lr = seg_lr[seg]*lr_factor (0 if seg_frozen[seg] & freeze: both are bit masks, bit 0 = last-layer freeze
 * (dinov2.py:627-635), bit 1 = backbone freeze (dinov2.py:619-625)); wd = seg_wd_on[seg] ? wd : 0.  beta1 / beta2 are doubles so that
 * 1 - beta and the bias corrections 1 - beta^step are formed in double like torch.optim.AdamW does on Python floats.
 * Also writes the bf16 shadow copy of the updated parameters. */
int lt_adamw_flat(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, const int32_t* seg_of_chunk,
                  const float* seg_lr, const uint8_t* seg_wd_on, const uint8_t* seg_frozen, int freeze,
                  float lr_factor, float wd, double beta1, double beta2, float eps, int step, const float* sumsq,
                  float max_norm, void* stream);
/* LARS as LT/_optim/lars_args.py:12-37 builds it from lightly.utils.lars (the "auto" optimizer of Distillation / DistillationV2,
 * distillation.py:140-147,294; an option of DistillationV3, distillationv3.py:147-157): per parameter tensor (= segment), and only where
 * the weight decay applies and both norms are non-zero, d = (g + wd p) * trust ||p|| / (||g|| + wd ||p|| + eps); then torch SGD's
 * momentum rule on d (first_step: the buffer becomes a copy of d) and p -= lr d, lr = seg_lr[seg] * lr_factor.  g is taken as clipped by
 * min(1, max_norm / (sqrt(*sumsq) + 1e-6)) as in lt_adamw_flat.  lt_lars_norms writes seg_norms [nseg][2] = (||p||, ||g|| unclipped):
 * per-chunk partials in ws (2 n / 1024 floats), added per segment in chunk order (deterministic); seg_chunk_begin [nseg + 1]. */
int lt_lars_norms(const float* p, const float* g, int64_t n, const int32_t* seg_chunk_begin, int nseg, float* ws, float* seg_norms, void* stream);
int lt_lars_flat(float* p, const float* g, float* buf, void* p_bf16, int64_t n, const int32_t* seg_of_chunk, const float* seg_lr,
                 const uint8_t* seg_wd_on, const float* seg_norms, float lr_factor, float wd, float momentum, float dampening, int nesterov,
                 float trust, float eps, int first_step, const float* sumsq, float max_norm, void* stream);
/* torch.optim.SGD as LT/_optim/sgd_args.py:19-31 builds it (the "auto" optimizer of DINO, LT/_methods/dino/dino.py:213-216,343-352):
 * d = g + wd p on every tensor of a decayed segment (coupled weight decay, no trust ratio), then the momentum rule and the clipping of
 * lt_lars_flat.  A segment with seg_lr 0 keeps its parameters but still moves its momentum buffer, as torch does for a group at lr 0
 * (DINO's frozen last layer, dino.py:470-473). */
int lt_sgd_flat(float* p, const float* g, float* buf, void* p_bf16, int64_t n, const int32_t* seg_of_chunk, const float* seg_lr,
                const uint8_t* seg_wd_on, float lr_factor, float wd, float momentum, float dampening, int nesterov, int first_step,
                const float* sumsq, float max_norm, void* stream);
/* ------------------------------------------------------------------------------------------
 * PaKA pieces of the DINOv31 method (LT/_methods/dinov31/dinov31.py:258-437; the loss itself and the RoI resampling are LightlySSL
 * code the reference tree does not vendor: oracle/dinov31_oracle.py restates them, parity unpinned)
 * ------------------------------------------------------------------------------------------ */
/* RoI resampling of token maps with per-image 4-tap bilinear tables (dinov31.py:338-437 `_align_cross_view_pair` / `_roi_align_view`; the
 * tables come from the crop geometry on the host, flips folded into the indices): out[(b * n_out + j), :] = sum_a w[b, j, a] *
 * in[src_image[b] * img_stride + idx[b, j, a] * D + :].  `in` = first patch token of image 0, images img_stride floats apart;
 * src_image NULL = identity; out_bf16 and / or out_f32 [B * n_out, D]. */
int lt_roi_resample_tokens(const float* in, const int32_t* src_image, const int32_t* idx, const float* w, void* out_bf16, float* out_f32,
                           int B, int64_t img_stride, int n_out, int D, void* stream);
/* its backward in gather form (no atomics): din[b * img_stride + i * D + :] = sum_{(j, a): idx[b, j, a] == i} w[b, j, a] * dout[b, j, :]
 * for every input cell i < n_in (overwrites those rows). */
int lt_roi_resample_tokens_bwd(const float* dout, const int32_t* idx, const float* w, float* din, int B, int64_t img_stride, int n_in,
                               int n_out, int D, void* stream);
/* out[b, j, :] = z[b, j, :] - mean_j z[b, :, :] (centring the token kernel K = Z Z^T is centring Z over the image's n tokens) */
int lt_center_tokens(const float* z, void* out_bf16, float* out_f32, int B, int n, int C, void* stream);
/* per image b: cka = <Ks, Kt> / (||Ks|| ||Kt|| + eps) over the n x n centred Gram matrices (rows ld floats apart);
 * loss[0] += sum_b coef[b] * (1 - cka_b) in image order;  G(bf16)[b] = d(coef[b] * (1 - cka_b)) / dKs[b], pad columns zero */
int lt_cka_fwd_bwd(const float* Ks, const float* Kt, const float* coef, float* loss, void* G_bf16, int B, int n, int ld, float eps,
                   void* stream);

/* Order-fixed reductions (bitwise reproducible steps).  Between lt_reduce_begin and lt_reduce_end the kernels that end in a sum over
 * workgroups -- lt_layernorm_bwd(_fused) (dw, db, dbias_next), lt_layerscale_bwd (dgamma, dbias), lt_colsum_bf16, lt_assemble_tokens_bwd
 * (mask-token gradient) -- store per-workgroup partial rows into `scratch` instead of issuing fp32 atomics; their destinations are
 * complete only after the next lt_reduce_flush / lt_reduce_end, which adds each destination's partial rows in call order with a fixed
 * summation tree (one launch for everything recorded).  Calls are recorded at enqueue time: flush on a stream that is ordered after
 * every stream the producers ran on.  When `scratch` runs out the kernels fall back to atomics; lt_reduce_overflows counts that.
 * (The loss scalars of lt_ce_fwd_bwd / lt_kl_fwd_bwd / lt_koleo_fwd_bwd, the KoLeo gradient and lt_colsum_f32 are order-fixed always.) */
int lt_reduce_begin(float* scratch, int64_t floats);
/* Same, with the cached flush tables of this region numbered from `first_slot`: a step that opens several regions (an eagerly launched
 * part, a part replayed from a HIP graph, an eager tail) gives each its own slot range, so that a captured flush kernel keeps reading
 * the table it was captured with.  A flush may be captured into a graph once the same region has run eagerly (its table buffers exist). */
int lt_reduce_begin_at(float* scratch, int64_t floats, int first_slot);
int lt_reduce_flush(void* stream);
int lt_reduce_end(void* stream);
int64_t lt_reduce_overflows(void);
/* RCCL communicator handle of this process -- the library's only process-wide state besides scratch.  Rank 0 draws
 * a 128-byte unique id (lt_comm_unique_id) and hands it to the other ranks by any out-of-band means; every rank calls lt_comm_init once.
 * lt_comm_allreduce_f32 sums buf over the ranks in place on the communicator's own stream, ordered after what `after_stream` holds at the
 * call; lt_comm_wait makes `stream` wait for every collective enqueued so far (no host synchronisation in either).  The reference does
 * the same through DDP's bucket hooks and torch.distributed (LT/_commands/train.py: strategy "ddp"); the Python driver here can use either
 * (parallel.GradSync, LT_GRAD_COMM=abi).  librccl.so is loaded at lt_comm_init, not at library load. */
int lt_comm_unique_id(void* id_out, int bytes);
int lt_comm_init(int rank, int world, const void* unique_id, int bytes);
int lt_comm_allreduce_f32(float* buf, int64_t n, void* after_stream);
int lt_comm_wait(void* stream);
int lt_comm_size(void);
int lt_comm_destroy(void);
/* teacher = m*teacher + (1-m)*student ; also refresh the teacher's bf16 shadow.  m is a double: 1 - m (~1e-6 at the end of the
 * cosine momentum schedule) is formed in double before the cast, as update_momentum does (_torch_helpers.py:75-96). */
int lt_ema_flat(float* teacher, const float* student, void* teacher_bf16, int64_t n, double m, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Convolutional student (torchvision ResNet-50 of BASELINE configs[3]; reference wrapper LT/_models/torchvision/resnet.py:21-47,
 * driven from LT/_methods/distillationv3/distillationv3.py:324-354).  Activations are NHWC bf16 = [B*H*W, C] matrices, so a
 * convolution is lt_gemm_bf16 on an im2col matrix; weights are held [Cout][kh][kw][Cin].  C must be a multiple of 8.
 * ------------------------------------------------------------------------------------------------------------------ */
/* cols[(b,oy,ox)][(ky*KW+kx)*C + c] = x[b][oy*stride-pad+ky][ox*stride-pad+kx][c] (zero outside); ld = row stride of cols (>= KH*KW*C).
 * KH = KW = 1 with stride 2 is the row gather of a strided 1x1 convolution (ResNet downsample path). */
int lt_im2col_nhwc_bf16(const void* x, void* cols, int B, int H, int W, int C, int KH, int KW, int stride, int pad, int ld, void* stream);
/* transpose of the above (dgrad of the gather), deterministic gather form: dx[pixel] = sum of the dcols entries that read it (+ add[pixel]) */
int lt_col2im_nhwc_bf16(const void* dcols, const void* add, void* dx, int B, int H, int W, int C, int KH, int KW, int stride, int pad, int ld,
                        void* stream);
/* stem: NCHW f32 image -> cols[(b,oy,ox)][c*KH*KW + ky*KW + kx] bf16 (the order of conv.weight.flatten(1)), zero-padded to ld columns */
int lt_im2col_nchw_f32(const float* x, void* cols, int B, int Cin, int H, int W, int KH, int KW, int stride, int pad, int ld, void* stream);
/* training-mode BatchNorm (nn.BatchNorm2d / BatchNorm1d in train(): batch statistics over the rows of x bf16 [rows, C], fp32 math):
 * y = act(gamma * (x - mean) * rstd + beta (+ resid)), act = ReLU if relu else identity; saves mean / rstd [C] for backward and
 * updates the running estimates (momentum, unbiased variance) when given.  ws: lt_batchnorm_ws_floats(C) floats of scratch. */
#define LT_BN_MAX_CHUNKS 512
int64_t lt_batchnorm_ws_floats(int C);
int lt_batchnorm_fwd(const void* x, const float* gamma, const float* beta, const void* resid, void* y, float* mean, float* rstd,
                     float* running_mean, float* running_var, int64_t rows, int C, float eps, float momentum, int relu, float* ws, void* stream);
/* eval-mode BatchNorm: the same affine map with caller-provided statistics (running_mean, 1/sqrt(running_var + eps)) */
int lt_batchnorm_apply(const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta, const void* resid, void* y,
                       int64_t rows, int C, int relu, void* stream);
/* backward of the training-mode forward: dz = dy * (y > 0) when y (the activated output) is given (stored to dz, which the residual path re-uses),
 * else dz = dy; dgamma += sum dz*xhat, dbeta += sum dz, dx = gamma * rstd * (dz - mean(dz) - xhat * mean(dz*xhat)). */
int lt_batchnorm_bwd(const void* dy, const void* y, const void* x, const float* gamma, const float* mean, const float* rstd, void* dz, void* dx,
                     float* dgamma, float* dbeta, int64_t rows, int C, float* ws, void* stream);
/* The same two passes split around a collective: torch.nn.SyncBatchNorm, which Lightning's sync_batchnorm=True (the reference sets it whenever
 * the accelerator is a GPU: LT/_commands/train_helpers.py:223,335-342) puts in place of every BatchNorm layer.  sums: doubles [2C + 1]
 * on the device, (sum x, sum x^2, rows) for the forward and (sum dz, sum dz*xhat, rows) for the backward; the caller adds them over the
 * ranks (one all-reduce) between the two calls.  Statistics, running estimates (unbiased over the GLOBAL row count) and the input-gradient
 * means are formed from the reduced sums; dgamma / dbeta accumulate the LOCAL sums (the data-parallel gradient mean handles them). */
int lt_batchnorm_stats(const void* x, int64_t rows, int C, float* ws, double* sums, void* stream);
int lt_batchnorm_fwd_from_sums(const void* x, const double* sums, const float* gamma, const float* beta, const void* resid, void* y, float* mean,
                               float* rstd, float* running_mean, float* running_var, int64_t rows, int C, float eps, float momentum, int relu,
                               void* stream);
int lt_batchnorm_bwd_sums(const void* dy, const void* y, const void* x, const float* mean, const float* rstd, void* dz, float* dgamma, float* dbeta,
                          int64_t rows, int C, float* ws, double* sums, void* stream);
int lt_batchnorm_bwd_from_sums(const void* dz, const void* x, const float* gamma, const float* mean, const float* rstd, const double* sums, void* dx,
                               int64_t rows, int C, float* ws, void* stream);
/* nn.MaxPool2d(3, stride 2, padding 1) on NHWC bf16 with the arg-max tap saved per element (uint8, first maximum in scan order wins) */
int lt_maxpool3x3s2_fwd(const void* x, void* y, uint8_t* idx, int B, int H, int W, int C, void* stream);
int lt_maxpool3x3s2_bwd(const void* dy, const uint8_t* idx, void* dx, int B, int H, int W, int C, void* stream);
/* AdaptiveAvgPool2d(1) over the n positions of every image: x bf16 [B, n, C] -> out bf16 [B, C] */
int lt_token_mean_bf16(const void* x, void* out, int B, int n, int C, void* stream);
/* gradient reaching the feature map: out[b,p,:] = d_tok[b,p,:] + d_pool[b,:] / n (either input may be null), bf16 out */
int lt_pool_bwd_add(const float* d_tok, const float* d_pool, void* out_bf16, int B, int n, int C, void* stream);
int lt_add_bf16(const void* a, const void* b, void* out, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Host-side (CPU) helper: the iBOT block masks of one step (MaskingGenerator + create_collated_masks, LT/_methods/dinov2/utils.py:41-152)
 * sampled in C++ from CPython's own Mersenne-Twister stream.  mt_state[624] / *mt_pos = the words of random.getstate()[1]; both are
 * advanced in place so that random.setstate() continues the stream exactly where the reference's Python loop would have left it.
 * ratio_edges[n_masked_crops + 1] = numpy.linspace(mask_ratio_min, mask_ratio_max, n_masked_crops + 1); masks: uint8 [n_crops, H*W],
 * returned in the final (shuffled) order.
 * ------------------------------------------------------------------------------------------------------------------ */
int lt_sample_block_masks(uint32_t* mt_state, int* mt_pos, const double* ratio_edges, int n_masked_crops, int n_crops, int H, int W,
                          int max_num_patches, int min_num_patches, double log_aspect_min, double log_aspect_max, uint8_t* masks);

/* ------------------------------------------------------------------------------------------------------------------
 * GPU multi-crop augmentation: the DINO view pipeline of LT/_transforms/view_transform.py:133-215 /
 * LT/_methods/dino/dino_transform.py:129-202 (RandomResizedCrop(INTER_AREA) -> HorizontalFlip -> ColorJitter -> ToGray ->
 * GaussianBlur -> Solarize -> Normalize) from decoded uint8 HWC images resident in HBM.  The random parameters are drawn on the
 * host and handed over as per-view records (device arrays); one call processes all n views of one output size S.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct lt_aug_crop_item {
  int64_t src_off;          /* byte offset of the image inside the packed uint8 [H, W, 3] source buffer */
  int32_t H, W;             /* source image size */
  float x0, y0, cw, ch;     /* crop box in source pixels (RandomResizedCrop) */
  int32_t flip;             /* HorizontalFlip fired */
} lt_aug_crop_item;
typedef struct lt_aug_color_item {
  int32_t apply;            /* ColorJitter fired */
  int32_t order;            /* the four ops in application order, 2 bits each: 0 brightness, 1 contrast, 2 saturation, 3 hue */
  float fb, fc, fs, fh;     /* brightness / contrast / saturation factors, hue shift in turns */
  int32_t gray;             /* ToGray fired */
} lt_aug_color_item;
typedef struct lt_aug_finish_item {
  float sigma;              /* GaussianBlur sigma, 0 = not fired; kernel radius ceil(3 sigma) <= LT_AUG_MAX_RADIUS */
  int32_t solarize;         /* Solarize fired: x >= threshold -> 1 - x */
  float threshold;
} lt_aug_finish_item;
#define LT_AUG_MAX_RADIUS 6
/* A: views f32 [n, 3, S, S] in [0,1] = area-resampled ("pixel area relation", cv2.INTER_AREA) crop of each item's box, flipped if asked */
int lt_aug_crop_resize(const uint8_t* src, const lt_aug_crop_item* items, float* views, int n, int S, void* stream);
/* B: in place: ColorJitter (torchvision semantics: brightness / contrast (blend with the view-wide mean luminance) / saturation / hue in
 * the item's order, each clamped to [0,1]) then ToGray (0.299 R + 0.587 G + 0.114 B on all three channels) */
int lt_aug_color(float* views, const lt_aug_color_item* items, int n, int S, void* stream);
/* C: out f32 [n, 3, S, S] = Normalize(Solarize(GaussianBlur(views))); separable Gaussian, reflect-101 border.  mean3 / std3: HOST float[3] */
int lt_aug_finish(const float* views, const lt_aug_finish_item* items, float* out, int n, int S, const float* mean3, const float* std3,
                  void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LT_AMD_H */
