/* cxrk — C-ABI of the MI355X (gfx950) kernels behind the joint image+text contrastive training step.
 *
 * The reference (marcomistretta/incremental_multimodal_medical_learning_II) is pure Python on PyTorch: it has no
 * FFI layer of its own, so each entry point cites the Python function whose arithmetic it replaces
 * (paths relative to the reference root).  The host side (incremental_multimodal_medical_learning_ii_amd/) binds this
 * library with ctypes and keeps the reference's class / method surface; INTEGRATION.md shows the binding stub.
 *
 * Conventions (all entry points):
 *   - plain device pointers + sizes; fp32 data; int64 ids/masks; no hidden allocation, no hidden synchronisation;
 *   - work is enqueued on `stream` (the caller's current HIP stream) and the call returns immediately;
 *   - scratch memory is passed in (`ws`, `ws_bytes`); the matching `*_ws_bytes` function sizes it;
 *   - return 0 on success, <0 on error: -1 bad argument/alignment, -2 workspace too small, -3 launch failure,
 *     -4 unsupported shape.  The Python wrappers raise RuntimeError/ValueError for these, mirroring the
 *     reference's exception behaviour;
 *   - row-major tensors; activations of the image encoder are NHWC, filters are [Ko][R][S][C] (= a torch OIHW
 *     tensor in channels_last memory format).
 */
#ifndef CXRK_H
#define CXRK_H

#include <stddef.h>
#include <hip/hip_runtime_api.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------------------
 * gemm_bias_act — every nn.Linear of the path: BERT query/key/value/dense/intermediate/output
 * (HF BertLayer under health_multimodal/text/model/modelling_cxrbert.py:87-95), BertProjectionHead
 * (modelling_cxrbert.py:43-49), models.myMLP / myLinearModel (models.py:7-26), the projector's 1x1 conv with bias
 * (health_multimodal/image/model/modules.py:32-33) and the logits block I_hat @ T_hat^T.
 *   C[M,N] = act( alpha * op(A)[M,K] @ op(B)[K,N] + bias[N] + R ) (* gelu'(aux) | * (aux>0))
 *   transA=0: A[m*lda+k]   transA=1: A[k*lda+m]      transB=0: B[k*ldb+n]   transB=1: B[n*ldb+k]
 *   act: 0 none, 1 relu, 2 gelu(erf).  auxmode: 0 none, 1 multiply by (aux>0), 2 multiply by gelu'(aux).
 *   C2 (optional) receives the pre-activation value.  accumulate: C += result (R must be NULL).
 *   splitk>1: deterministic split-K through `ws` (plain epilogue only) — used for weight gradients.
 * fwd:  Y = X W^T + b           -> transA=0, transB=1
 * bwd:  dX = dY W               -> transA=0, transB=0 ;  dW = dY^T X -> transA=1, transB=0 (split-K)
 */
size_t cxrk_gemm_splitk_ws_bytes(int M, int N, int splitk);
/* Split-K factor the library recommends for a weight-gradient shaped GEMM (M x N output, K = rows reduced over): enough
 * slabs to fill the chip with the tile the launch will take in the current precision mode. */
int cxrk_gemm_wgrad_splitk(int M, int N, int K, int planes);
/* 1 when a launch of this GEMM shape takes the 256x256 tile (reporting only).  kind: 0 or 3 dense layer / weight gradient,
 * 1 convolution forward, 2 convolution data gradient. */
int cxrk_gemm_wide_tile(int M, int N, long K, int splitk, int kind);
int cxrk_gemm_f32(int transA, int transB, int M, int N, int K, const float* A, long lda, const float* B, long ldb,
                  float* C, long ldc, const float* bias, const float* R, long ldr, const float* aux, long ldaux,
                  int auxmode, float* C2, long ldc2, int act, float alpha, int accumulate, int splitk, float* ws,
                  size_t ws_bytes, hipStream_t stream);

/* The same contraction on planes operands (see "Storage formats" below): A and B are planes; the output is fp32 (C) or planes
 * (Cp), the residual fp32 (R) or planes (Rp).  auxmode 2 = multiply by gelu'(aux) (aux fp32), 3 = multiply by the ReLU decision
 * bits `maskin`; `maskout` (with act 1) receives the decision bits of this launch's ReLU.  transA && transB is not provided. */
int cxrk_gemm_pl(int transA, int transB, int M, int N, int K, const void* A, long lda, long aplane, const void* B, long ldb,
                 long bplane, float* C, void* Cp, long ldc, long cplane, const float* bias, const float* R, const void* Rp,
                 long ldr, long rplane, const float* aux, long ldaux, int auxmode, const unsigned char* maskin, long ldmaskin,
                 unsigned char* maskout, long ldmaskout, float* C2, long ldc2, int act, float alpha, int accumulate, int splitk,
                 float* colsum, int colsum_accumulate, float* ws, size_t ws_bytes, hipStream_t stream);
size_t cxrk_gemm_pl_colsum_ws_bytes(int M, int N);   /* ws for the fused column sums (`colsum` != NULL; needs splitk == 1) */
/* fp32 tensor -> planes (weights once per step, inputs of the path) and back (host-side consumers, tests); n % 8 == 0. */
int cxrk_split_planes(const float* x, long n, void* out, long plane, hipStream_t stream);
int cxrk_merge_planes(const void* x, long plane, long n, float* out, hipStream_t stream);

/* out[cols] (+)= alpha * sum_rows X[rows, cols] — bias gradients, BN beta gradients, position/type embedding grads. */
size_t cxrk_colsum_ws_bytes(long rows, int cols);
int cxrk_colsum(const float* X, long ldx, long rows, int cols, float* out, float alpha, int accumulate, float* ws,
                size_t ws_bytes, hipStream_t stream);
int cxrk_colsum_pl(const void* X, long ldx, long plane, long rows, int cols, float* out, float alpha, int accumulate, float* ws,
                   size_t ws_bytes, hipStream_t stream);
/* out[cols] = alpha * sum_rows (X[r][c] - mean[c])^2, X fp32 (plane == 0) or planes (plane > 0); ws as for cxrk_colsum.  The second
 * pass of the batch variance `torch.nn.BatchNorm2d` stores in train mode (momentum 1): `ImageModel.calibrate_batchnorm_`, which
 * gives synthetic weights statistics that match their activations (the reference only loads trained weights, model.py:117-118). */
int cxrk_colvar(const void* X, long ldx, long plane, long rows, int cols, const float* mean, float* out, float alpha, float* ws,
                size_t ws_bytes, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * conv_bn_act — torchvision Bottleneck conv + eval-mode BatchNorm + ReLU (+ residual) as used by
 * health_multimodal/image/model/resnet.py:34-47 and the projector's conv+BN+ReLU (modules.py:43-46).
 * cxrk_bn_fold: w_scaled = w * gamma*rsqrt(var+eps) (per output channel, channel-padded to Cpad),
 *               scale = gamma*rstd, shift = beta - mean*scale, rstd = rsqrt(var+eps).
 * fwd:        y = relu?( conv(x, w_scaled) + shift + residual? )
 * bwd_data:   dx = (relu_src>0)? * ( conv^T(dy, w_scaled) + residual? )        (dy already masked by its own ReLU)
 * bwd_params: dW = scale * wgrad(x, dy);  dbeta = sumdy = sum dy;
 *             dgamma = sum dy * xhat = rstd * (<w, wgrad(x, dy)> - mean * sumdy): taken from the raw weight gradient the
 *             call forms anyway (no activation is re-read, gamma is never divided by).
 *
 * Storage formats.  The fp32 entry points take fp32 tensors.  The `_pl` entry points take "planes" tensors: x stored as two
 * bf16 planes hi = bf16(x), lo = bf16(x - hi) of the same shape, the lo plane `*plane` ELEMENTS behind the hi plane (4 bytes
 * per element like fp32; pointers are `void*`).  They are what the encoders use in split-bf16 mode: the MFMA mainloops then
 * load operands without any conversion work.  ReLU decisions travel as bit masks: byte [pixel][channel / 8], bit channel % 8.
 */
int cxrk_bn_fold(const float* w, const float* gamma, const float* beta, const float* rmean, const float* rvar,
                 float eps, int Ko, int taps, int C, int Cpad, float* w_scaled, float* scale, float* shift,
                 float* rstd, hipStream_t stream);
int cxrk_bn_fold_pl(const float* w, const float* gamma, const float* beta, const float* rmean, const float* rvar,
                    float eps, int Ko, int taps, int C, int Cpad, void* w_scaled, long wplane, float* scale, float* shift,
                    float* rstd, hipStream_t stream);
int cxrk_conv_bn_act_fwd(const float* x, const float* w_scaled, const float* shift, const float* residual, float* y,
                         int N, int H, int W, int C, int Ko, int R, int S, int stride, int pad, int relu,
                         hipStream_t stream);
/* y (and the optional residual) are planes; in_planes = 1: x and w_scaled are planes too, 0: they are fp32 (the stem).
 * maskout (optional, needs relu) receives the ReLU decision bits of y. */
int cxrk_conv_bn_act_fwd_pl(const void* x, long xplane, const void* w_scaled, long wplane, int in_planes, const float* shift,
                            const void* residual, long rplane, void* y, long yplane, unsigned char* maskout, int N, int H,
                            int W, int C, int Ko, int R, int S, int stride, int pad, int relu, hipStream_t stream);
/* sums (optional, [C]) = column sums of dx: the BatchNorm beta gradient of the unit that produced relu_src / maskin (dx is
 * that unit's masked output gradient), reduced in the data-gradient epilogue; needs ws of
 * cxrk_conv_bwd_data_colsum_ws_bytes().  Not for 1x1 stride-2 filters. */
size_t cxrk_conv_bwd_data_colsum_ws_bytes(int N, int H, int W, int C, int stride);
int cxrk_conv_bn_act_bwd_data(const float* dy, const float* w_scaled, const float* residual, const float* relu_src,
                              float* dx, int N, int H, int W, int C, int Ko, int R, int S, int stride, int pad,
                              float* sums, float* ws, size_t ws_bytes, hipStream_t stream);
int cxrk_conv_bn_act_bwd_data_pl(const void* dy, long dyplane, const void* w_scaled, long wplane, const void* residual,
                                 long rplane, const unsigned char* maskin, void* dx, long dxplane, int N, int H, int W, int C,
                                 int Ko, int R, int S, int stride, int pad, float* sums, float* ws, size_t ws_bytes,
                                 hipStream_t stream);
/* The same with a COMPACT residual: planes [N, (H + 1) / 2, (W + 1) / 2, C] = the identity-branch gradient at the pixels with even
 * (h, w) only (the data gradient of a 1x1 / stride-2 projection shortcut, torchvision Bottleneck.downsample, resnet.py:36-47);
 * stride must be 1; CXRK_ERR_UNSUPPORTED when the compact tensor exceeds 2 GiB per plane (use the dense form then). */
int cxrk_conv_bn_act_bwd_data_pl_s2res(const void* dy, long dyplane, const void* w_scaled, long wplane, const void* residual_s2,
                                 long rplane, const unsigned char* maskin, void* dx, long dxplane, int N, int H, int W, int C,
                                 int Ko, int R, int S, int stride, int pad, float* sums, float* ws, size_t ws_bytes,
                                 hipStream_t stream);
size_t cxrk_conv_wgrad_ws_bytes(int N, int H, int W, int Cpad, int Ko, int R, int S, int stride, int pad);
int cxrk_conv_bn_act_bwd_params(const float* x, const float* dy, const float* w, const float* scale, const float* rstd,
                                const float* rmean, const float* sumdy, float* dw, float* dgamma, float* dbeta,
                                int accumulate, int N, int H, int W, int C, int Cpad, int Ko, int R, int S, int stride,
                                int pad, float* ws, size_t ws_bytes, hipStream_t stream);
int cxrk_conv_bn_act_bwd_params_pl(const void* x, long xplane, const void* dy, long dyplane, const float* w, const float* scale,
                                   const float* rstd, const float* rmean, const float* sumdy, float* dw, float* dgamma,
                                   float* dbeta, int accumulate, int N, int H, int W, int C, int Ko, int R, int S, int stride,
                                   int pad, float* ws, size_t ws_bytes, hipStream_t stream);

/* Boundary layout transforms: torch NCHW input (model.py:141) <-> NHWC working layout. */
int cxrk_nchw_to_nhwc(const float* x, float* y, int N, int C, int H, int W, int Cpad, hipStream_t stream);
int cxrk_nhwc_to_nchw(const float* x, float* y, int N, int C, int H, int W, hipStream_t stream);

/* maxpool — nn.MaxPool2d(3, stride 2, pad 1) of the ResNet stem (resnet.py:37). idx = winning tap per output. */
int cxrk_maxpool_fwd(const float* x, float* y, unsigned char* idx, int N, int H, int W, int C, hipStream_t stream);
int cxrk_maxpool_bwd(const float* dy, const unsigned char* idx, const float* x, float* dx, int N, int H, int W, int C,
                     int relu_mask, hipStream_t stream);
/* planes variants: x / y / dy / pooled are planes.  The backward masks with the stem ReLU through the sign of `pooled` (the
 * pooled value IS the winning input) and writes dx in fp32 (it feeds the stem's exact-fp32 weight gradient). */
int cxrk_maxpool_fwd_pl(const void* x, long xplane, void* y, long yplane, unsigned char* idx, int N, int H, int W, int C,
                        hipStream_t stream);
int cxrk_maxpool_bwd_pl(const void* dy, long dyplane, const unsigned char* idx, const void* pooled, float* dx, int N, int H,
                        int W, int C, hipStream_t stream);

/* spatial_mean — torch.mean(projected_patch_embeddings, dim=(2,3)) (model.py:145). x[N][P][C] -> y[N][C]. */
int cxrk_spatial_mean_fwd(const float* x, float* y, int N, int P, int C, hipStream_t stream);
int cxrk_spatial_mean_bwd(const float* dy, float* dx, int N, int P, int C, hipStream_t stream);
/* dx (planes [N][P][C]) = dy[n][c] / P + add[n][p][c] (add optional, fp32) */
int cxrk_spatial_mean_bwd_pl(const float* dy, const float* add, void* dx, long dxplane, int N, int P, int C, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Train-mode BatchNorm2d around the convolutions (csrc/bn_train.hip): the mode the reference's `ImageModel` constructor leaves the
 * model in (`self.train()`, health_multimodal/image/model/model.py:119) — batch statistics in the forward, running statistics updated
 * with momentum, the batch-statistics terms in the backward (torch.nn.BatchNorm2d, training=True).  Tensors `void* p, long plane`:
 * fp32 when plane == 0, split-bf16 planes otherwise; [rows = pixels][C], C % 8 == 0.
 *   forward:   z = conv(x, w) (cxrk_conv_bn_act_fwd* with an identity fold); cxrk_colstats -> batch mean and biased variance;
 *              cxrk_bn_train_fwd_coeffs -> scale = gamma rstd, shift = beta - mean scale, rstd (+ running statistics update when
 *              rmean / rvar are given: r = (1 - momentum) r + momentum stat, variance unbiased);  cxrk_bn_apply: y = relu?(z scale +
 *              shift + residual?), ReLU decision bits (byte [row][c / 8]) when mask != null.
 *   backward:  dot = cxrk_coldot(dy, z, bshift = mean) = sum_rows dy (z - mean);  cxrk_bn_train_bwd_coeffs -> dbeta (+)= sum dy,
 *              dgamma (+)= rstd dot, and A, B, Cc with dz = A dy + B + Cc z = gamma rstd (dy - mean(dy) - xhat mean(dy xhat));  cxrk_bn_train_dz;
 *              dz then takes the place of dy in cxrk_conv_bn_act_bwd_data* / _bwd_params* (identity fold).
 */
/* mean[c], var[c] = var_scale * sum_rows (x - mean)^2 in ONE pass over x (per-block shifted sums merged with Chan's formula);
 * ws: 2 * cxrk_coldot_ws_bytes(rows, C) */
int cxrk_colstats(const void* x, long plane, long rows, int C, float* mean, float* var, float var_scale, float* ws, size_t ws_bytes,
                  hipStream_t stream);
int cxrk_bn_train_fwd_coeffs(const float* mean, const float* var, const float* gamma, const float* beta, float eps, long n, float momentum,
                             float* scale, float* shift, float* rstd, float* rmean, float* rvar, int C, hipStream_t stream);
int cxrk_bn_apply(const void* z, long zplane, const float* scale, const float* shift, const void* res, long rplane, void* y, long yplane,
                  unsigned char* mask, long rows, int C, int relu, hipStream_t stream);
size_t cxrk_coldot_ws_bytes(long rows, int C);
/* out[c] = sum_rows a[r][c] * (b[r][c] - bshift[c]); bshift may be null (= 0) */
int cxrk_coldot(const void* a, long aplane, const void* b, long bplane, const float* bshift, long rows, int C, float* out, float* ws,
                size_t ws_bytes, hipStream_t stream);
int cxrk_bn_train_bwd_coeffs(const float* gamma, const float* mean, const float* rstd, const float* sumdy, const float* dot, long n, float* A,
                             float* B, float* Cc, float* dgamma, float* dbeta, int accumulate, int C, hipStream_t stream);
int cxrk_bn_train_dz(const void* dy, long dyplane, const void* z, long zplane, const float* A, const float* B, const float* Cc, void* dz,
                     long dzplane, long rows, int C, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * CXR-BERT pieces (HF BertForMaskedLM under modelling_cxrbert.py:87-99; config configuration_cxrbert.py:11-22).
 * embed_ln:    y = LayerNorm(word[ids] + pos[t % L] + type[0])                    (BertEmbeddings)
 * residual_ln: y = LayerNorm(x + res)                                             (BertSelfOutput / BertOutput)
 *              xhat, rstd saved for bwd; bwd: dx = LN'(dy) + dx_add, dgamma/dbeta reduced deterministically.
 * attn:        ctx = softmax(Q K^T / sqrt(d) + keymask) V per (sequence, head); qkv is the fused [T][3*nH*d]
 *              projection output; probs [B][nH][L][L] saved for bwd.  d <= 64 (multiple of 4); L <= 64: one workgroup per
 *              (sequence, head), everything in LDS; 64 < L <= 512 (the reference's max_position_embeddings,
 *              text/inference_engine.py:30,44-46): tiled over 32 queries / 64 keys, bwd needs the dS workspace.
 * embed_bwd:   dword[ids[t]] += dx[t], summed in token order per row (sort by (id, position) + segmented sums: deterministic, no atomics)
 * Outputs declared `void* y, long yplane`: yplane = 0 -> fp32 tensor; yplane > 0 -> planes (bf16 hi at y, lo at y + yplane
 * elements), the format the following GEMM consumes in split-bf16 mode.
 */
int cxrk_embed_ln_fwd(const long* ids, const float* word, const float* pos, const float* type, const float* gamma,
                      const float* beta, float eps, long T, int L, int H, void* y, long yplane, float* xhat, float* rstd,
                      hipStream_t stream);
int cxrk_residual_ln_fwd(const float* x, const float* res, const float* gamma, const float* beta, float eps, long rows,
                         int H, void* y, long yplane, float* xhat, float* rstd, hipStream_t stream);
size_t cxrk_residual_ln_bwd_ws_bytes(long rows, int H);
/* dxsum (optional, [H]): column sums of the gradient this call writes (dx, dx_add included) = the bias gradient of the dense layer
 * whose output fed this LayerNorm (BertSelfOutput / BertOutput dense), reduced in the same pass. */
int cxrk_residual_ln_bwd(const float* dy, const float* xhat, const float* rstd, const float* gamma, long rows, int H,
                         const float* dx_add, void* dx, long dxplane, float* dgamma, float* dbeta, int accumulate, float* dxsum,
                         int dxsum_accumulate, float* ws, size_t ws_bytes, hipStream_t stream);
/* dst[r*ld + c] += src[r][c] for a planes tensor src [rows][cols] (the CLS-row gradient added into a [N, L, H] fp32 gradient) */
int cxrk_planes_add_rows(const void* src, long plane, long rows, int cols, float* dst, long ld, hipStream_t stream);
int cxrk_attn_fwd(const float* qkv, const long* mask, int B, int L, int nH, int dH, void* ctx, long ctxplane, float* probs,
                  hipStream_t stream);
size_t cxrk_attn_bwd_ws_bytes(int B, int L, int nH, int dH);
int cxrk_attn_bwd(const float* qkv, const float* probs, const float* dctx, int B, int L, int nH, int dH, void* dqkv,
                  long dqkvplane, float* ws, size_t ws_bytes, hipStream_t stream);
size_t cxrk_embed_bwd_ws_bytes(long T, int H);
int cxrk_embed_bwd(const long* ids, const float* dx, long T, int H, float* dword, float* ws, size_t ws_bytes, hipStream_t stream);
/* dx = dy * gelu'(pre): the erf-GELU between dense_to_hidden and LayerNorm of BertProjectionHead (modelling_cxrbert.py:45-46). */
int cxrk_gelu_bwd(const float* dy, const float* pre, long n, float* dx, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Similarity / loss heads.
 * l2norm:      F.normalize(x, dim=1) (modelling_cxrbert.py:138-139; vlp/inference_engine.py:51): xhat = x / max(|x|, eps).
 *              xhat rows are ldxhat floats apart (the data-parallel step writes image and text halves of one [B][2D] send buffer).
 * infonce:     north-star head (not in the reference): on a logits block S[rows][cols] = X_hat_local @ Y_hat_all^T / tau,
 *              row_lse gives lse + diagonal (+ accumulates sum(lse-diag)*scale into loss_out);
 *              grad_inplace turns S into exp(S-lse_row[i]) + exp(S-lse_col[j]) - 2*[j==diag_off+i].
 * pairwise_cosine: torchmetrics pairwise_cosine_similarity as called by Trainer.myCosineSimilarity (Trainer.py:1682-1704).
 * pairwise_cosine_max: the MAX_EMB branch of the same function (Trainer.py:1691-1693): y holds G groups of Pg prompt
 *              vectors (group g = rows g*Pg .. g*Pg+Pg-1); besides cos [B][G*Pg] it returns, per image and group, the maximum
 *              over the group's prompts (first winner on ties, as torch.max), the mean (logged at Trainer.py:1697-1703) and
 *              the winner's index, which max_bwd uses to route dmax [B][G] to one prompt per group.
 * patch_similarity: sim[r] = <patches[r,:], text> — `projected_patch_embeddings.view(-1, D) @ text.t()`
 *              (health_multimodal/vlp/inference_engine.py:104); smoothing and resizing stay on the host.
 * bce_posneg:  logits = cos_pos - cos_neg (Trainer.py:575) + nn.BCEWithLogitsLoss() mean (ZERO_JOINT_BOUNDS.py:36);
 *              cos is [B][2C] with column 2c = positive prompt of class c, 2c+1 = negative; writes dloss/dcos.
 * eval_score:  Trainer.val/test scoring (Trainer.py:825-836).
 * group_mean:  prompt-embedding mean over the prompts of a class (Trainer.py:1665-1666).
 */
int cxrk_l2norm_fwd(const float* x, long rows, int D, float eps, float* xhat, long ldxhat, float* norm, hipStream_t stream);
int cxrk_l2norm_bwd(const float* dxhat, const float* xhat, long ldxhat, const float* norm, long rows, int D, float* dx,
                    hipStream_t stream);
int cxrk_infonce_row_lse(const float* S, long ld, int rows, int cols, int diag_off, float* lse, float* diag,
                         float* loss_out, float loss_scale, int loss_accumulate, hipStream_t stream);
int cxrk_infonce_grad_inplace(float* S, long ld, int rows, int cols, int diag_off, const float* lse_row,
                              const float* lse_col, hipStream_t stream);
int cxrk_pairwise_cosine_fwd(const float* x, const float* y, long B, int P, int D, float* cosv, float* xnorm,
                             float* ynorm, hipStream_t stream);
size_t cxrk_pairwise_cosine_bwd_ws_bytes(long B, int P, int D);
int cxrk_pairwise_cosine_bwd(const float* x, const float* y, const float* cosv, const float* dcos, const float* xnorm,
                             const float* ynorm, long B, int P, int D, float* dx, float* dy, int accumulate_dy,
                             float* ws, size_t ws_bytes, hipStream_t stream);
int cxrk_pairwise_cosine_max_fwd(const float* x, const float* y, long B, int G, int Pg, int D, float* cosv, float* xnorm,
                                 float* ynorm, float* maxv, float* meanv, int* argmax, hipStream_t stream);
int cxrk_pairwise_cosine_max_bwd(const float* x, const float* y, const float* cosv, const float* dmax, const int* argmax,
                                 const float* xnorm, const float* ynorm, long B, int G, int Pg, int D, float* dx, float* dy,
                                 int accumulate_dy, float* ws, size_t ws_bytes, hipStream_t stream);
int cxrk_patch_similarity(const float* patches, const float* text, long R, int D, float* sim, hipStream_t stream);
size_t cxrk_bce_posneg_ws_bytes(void);
int cxrk_bce_posneg_fwd_bwd(const float* cosv, const float* labels, long B, int C, int ldlab, int diff, float* logits,
                            float* dcos, float* loss, float* ws, size_t ws_bytes, hipStream_t stream);
int cxrk_eval_score(const float* cosv, long B, int C, int pred_diff, float* score, float* pred, hipStream_t stream);
int cxrk_group_mean_fwd(const float* in, int G, int n, int D, float* out, hipStream_t stream);
int cxrk_group_mean_bwd(const float* dout, int G, int n, int D, float* din, hipStream_t stream);
/* out = alpha * (alpha_dev ? *alpha_dev : 1) * x * (mask_src > 0): nn.ReLU backward (models.py:10) and chain-rule scaling
 * by an upstream scalar gradient that lives on the device. */
int cxrk_scale_mask(const float* x, const float* mask_src, const float* alpha_dev, float alpha, long n, float* out,
                    hipStream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Optimiser / continual learning.
 * adam_fused:   torch.optim.Adam defaults (Trainer.py:172-175), one launch over a flat parameter buffer.
 * sgd:          optim.SGD (Trainer.py:176-178).
 * weight_reset: Trainer.myIncremental per tensor (Trainer.py:1562-1572); counters[0] += #restored.
 */
int cxrk_adam_fused(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                    float eps, float weight_decay, int step, float grad_scale, hipStream_t stream);
int cxrk_sgd(float* p, const float* g, long n, float lr, float weight_decay, float grad_scale, hipStream_t stream);
size_t cxrk_weight_reset_ws_bytes(void);
int cxrk_weight_reset(float* pnew, const float* pold, long n, float threshold, unsigned long long* counters, float* ws,
                      size_t ws_bytes, hipStream_t stream);

/* Contraction precision of every GEMM / implicit-GEMM entry point above (process-wide):
 *   0 (default) exact fp32: v_mfma_f32_32x32x2_f32, a k-ordered fmaf chain;
 *   1 split-bf16: operands split into bf16 hi+lo while staged, a*b = hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16
 *     with fp32 accumulation (~2^-16 relative per product; inputs, outputs and all other kernels stay fp32). */
int cxrk_set_precision(int mode);
int cxrk_get_precision(void);
/* Policy of the 256x256 LDS-DMA kernel (planes operands): 0 never, 1 where it pays (default; CXRK_WIDE), 2 every planes launch
 * (test coverage of small / ragged shapes).  Returns the previous mode. */
int cxrk_set_wide_mode(int mode);

/* Library identification: returns a static string "cxrk <version> gfx950". */
const char* cxrk_version(void);

#ifdef __cplusplus
}
#endif
#endif /* CXRK_H */
