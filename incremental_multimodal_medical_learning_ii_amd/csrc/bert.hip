// CXR-BERT text-encoder kernels that are not GEMMs: embedding gather + LayerNorm, residual LayerNorm (fwd/bwd),
// short-sequence multi-head attention (fwd/bwd), embedding scatter-add.
//
// Reference semantics: HuggingFace BertForMaskedLM as configured by
// health_multimodal/text/model/configuration_cxrbert.py:11-22 and used at modelling_cxrbert.py:87-99
// (post-LN encoder, LayerNorm eps 1e-12, additive key mask, softmax(QK^T/sqrt(d)) V, dropout inactive).
#include "cxrk.h"
#include "cxrk_common.h"

using namespace cxrk;

namespace {

// One wave per row.  xhat/rstd are saved for the backward.  H <= 64*MAXV.
constexpr int LN_MAXV = 16;  // H <= 1024

template <bool EMBED>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                     const long* __restrict__ ids, const float* __restrict__ word,
                                                     const float* __restrict__ pos, const float* __restrict__ type,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float eps, long rows, int H, int L, float* __restrict__ y,
                                                     float* __restrict__ xhat, float* __restrict__ rstd_out) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  float v[LN_MAXV];
  float s = 0.f;
  const float* xr = EMBED ? word + ids[row] * H : x + row * H;
  const float* rr = EMBED ? pos + (row % L) * H : (res ? res + row * H : nullptr);
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = lane + i * 64;
    float t = 0.f;
    if (c < H) {
      t = xr[c];
      if (rr) t += rr[c];
      if (EMBED) t += type[c];
    }
    v[i] = t; s += t;
  }
  const float mean = wave_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = lane + i * 64;
    const float d = c < H ? v[i] - mean : 0.f;
    v[i] = d; q += d * d;
  }
  const float rs = 1.0f / sqrtf(wave_sum(q) / (float)H + eps);
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < H) {
      const float xh = v[i] * rs;
      if (xhat) xhat[row * H + c] = xh;
      y[row * H + c] = xh * gamma[c] + beta[c];
    }
  }
  if (rstd_out && lane == 0) rstd_out[row] = rs;
}

// dx = rstd * (g - mean(g) - xhat*mean(g*xhat)), g = dy*gamma; block partial sums of dgamma/dbeta.
// Each block handles `rows_per` rows; 4 waves take rows round-robin; partials [nblk][2][H].
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ xhat,
                                                     const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                     long rows, int H, int rows_per, float* __restrict__ dx,
                                                     const float* __restrict__ dx_add, float* __restrict__ part) {
  __shared__ float sh[2][4][64 * LN_MAXV];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long r0 = (long)blockIdx.x * rows_per;
  const long r1 = min(rows, r0 + rows_per);
  float dg[LN_MAXV], db[LN_MAXV], gm[LN_MAXV];
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) { dg[i] = 0.f; db[i] = 0.f; const int c = lane + i * 64; gm[i] = c < H ? gamma[c] : 0.f; }
  for (long row = r0 + w; row < r1; row += 4) {
    float g[LN_MAXV], xh[LN_MAXV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      const int c = lane + i * 64;
      float d = 0.f, x_ = 0.f;
      if (c < H) { d = dy[row * H + c]; x_ = xhat[row * H + c]; }
      xh[i] = x_; g[i] = d * gm[i];
      dg[i] += d * x_; db[i] += d;
      s1 += g[i]; s2 += g[i] * x_;
    }
    s1 = wave_sum(s1) / (float)H; s2 = wave_sum(s2) / (float)H;
    const float rs = rstd[row];
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      const int c = lane + i * 64;
      if (c < H) {
        float o = rs * (g[i] - s1 - xh[i] * s2);
        if (dx_add) o += dx_add[row * H + c];
        dx[row * H + c] = o;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) { sh[0][w][lane + i * 64] = dg[i]; sh[1][w][lane + i * 64] = db[i]; }
  __syncthreads();
  for (int c = threadIdx.x; c < H; c += 256) {
    part[((long)blockIdx.x * 2 + 0) * H + c] = (sh[0][0][c] + sh[0][1][c]) + (sh[0][2][c] + sh[0][3][c]);
    part[((long)blockIdx.x * 2 + 1) * H + c] = (sh[1][0][c] + sh[1][1][c]) + (sh[1][2][c] + sh[1][3][c]);
  }
}
__global__ void ln_bwd_final_kernel(const float* __restrict__ part, int nblk, int H, float* __restrict__ dgamma,
                                    float* __restrict__ dbeta, int accumulate) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= H) return;
  float a = 0.f, b = 0.f;
  for (int p = 0; p < nblk; ++p) { a += part[((long)p * 2) * H + c]; b += part[((long)p * 2 + 1) * H + c]; }
  dgamma[c] = accumulate ? dgamma[c] + a : a;
  dbeta[c] = accumulate ? dbeta[c] + b : b;
}

// ---------------------------------------------------------------------------------------------------------------
// Attention for short sequences (L <= 64, head dim 64): one 256-thread workgroup per (sequence, head).
// Q,K,V live in one fused [T][3*nH*64] buffer (the QKV GEMM output).  Everything is staged in LDS; the products are
// small (2*L*L*64 FLOP each) and run on the fp32 VALU.  probs [B][nH][L][L] are saved for the backward.
// ---------------------------------------------------------------------------------------------------------------
constexpr int AD = 64;       // max head dim
constexpr int AL = 64;       // max sequence length handled here

__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ qkv, const long* __restrict__ mask, int L,
                                                       int nH, int dH, float scale, float* __restrict__ ctx,
                                                       float* __restrict__ probs) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ALD = dH + 1;
  float* Qs = smem; float* Ks = Qs + L * ALD; float* Vs = Ks + L * ALD; float* Ps = Vs + L * ALD;  // Ps[L][L+1]
  const int b = blockIdx.x / nH, hd = blockIdx.x % nH;
  const int ld = 3 * nH * dH;
  const int d4 = dH / 4;
  const float* base = qkv + (long)b * L * ld + hd * dH;
  for (int i = threadIdx.x; i < L * d4; i += 256) {
    const int row = i / d4, c4 = i % d4;
    const float4 q = *reinterpret_cast<const float4*>(base + (long)row * ld + c4 * 4);
    const float4 k = *reinterpret_cast<const float4*>(base + (long)row * ld + nH * dH + c4 * 4);
    const float4 v = *reinterpret_cast<const float4*>(base + (long)row * ld + 2 * nH * dH + c4 * 4);
    float* qd = Qs + row * ALD + c4 * 4; qd[0] = q.x; qd[1] = q.y; qd[2] = q.z; qd[3] = q.w;
    float* kd = Ks + row * ALD + c4 * 4; kd[0] = k.x; kd[1] = k.y; kd[2] = k.z; kd[3] = k.w;
    float* vd = Vs + row * ALD + c4 * 4; vd[0] = v.x; vd[1] = v.y; vd[2] = v.z; vd[3] = v.w;
  }
  __syncthreads();
  const int LP = L + 1;
  for (int e = threadIdx.x; e < L * L; e += 256) {
    const int i = e / L, j = e % L;
    float s = 0.f;
#pragma unroll 16
    for (int d = 0; d < dH; ++d) s = fmaf(Qs[i * ALD + d], Ks[j * ALD + d], s);
    s *= scale;
    if (mask && mask[(long)b * L + j] == 0) s = -INFINITY;
    Ps[i * LP + j] = s;
  }
  __syncthreads();
  // softmax: 4 lanes per row
  {
    const int row = threadIdx.x >> 2, sub = threadIdx.x & 3;
    if (row < L) {
      float m = -INFINITY;
      for (int j = sub; j < L; j += 4) m = fmaxf(m, Ps[row * LP + j]);
      m = fmaxf(m, __shfl_xor(m, 1, 64)); m = fmaxf(m, __shfl_xor(m, 2, 64));
      float sum = 0.f;
      for (int j = sub; j < L; j += 4) { const float p = expf(Ps[row * LP + j] - m); Ps[row * LP + j] = p; sum += p; }
      sum += __shfl_xor(sum, 1, 64); sum += __shfl_xor(sum, 2, 64);
      const float inv = 1.0f / sum;
      for (int j = sub; j < L; j += 4) {
        const float p = Ps[row * LP + j] * inv;
        Ps[row * LP + j] = p;
        if (probs) probs[(((long)b * nH + hd) * L + row) * L + j] = p;
      }
    }
  }
  __syncthreads();
  float* out = ctx + (long)b * L * (nH * dH) + hd * dH;
  for (int e = threadIdx.x; e < L * dH; e += 256) {
    const int i = e / dH, d = e % dH;
    float s = 0.f;
    for (int j = 0; j < L; ++j) s = fmaf(Ps[i * LP + j], Vs[j * ALD + d], s);
    out[(long)i * (nH * dH) + d] = s;
  }
}

// dV = P^T dO ; dP = dO V^T ; dS = P o (dP - rowsum(dP o P)) ; dQ = scale * dS K ; dK = scale * dS^T Q
__global__ __launch_bounds__(256) void attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ probs,
                                                       const float* __restrict__ dctx, int L, int nH, int dH, float scale,
                                                       float* __restrict__ dqkv) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ALD = dH + 1;
  float* Qs = smem; float* Ks = Qs + L * ALD; float* Vs = Ks + L * ALD; float* Os = Vs + L * ALD;
  float* Ps = Os + L * ALD; float* Ds = Ps + L * (L + 1);  // Ps = P, Ds = dS
  const int b = blockIdx.x / nH, hd = blockIdx.x % nH;
  const int ld = 3 * nH * dH;
  const int d4 = dH / 4;
  const float* base = qkv + (long)b * L * ld + hd * dH;
  const float* dob = dctx + (long)b * L * (nH * dH) + hd * dH;
  for (int i = threadIdx.x; i < L * d4; i += 256) {
    const int row = i / d4, c4 = i % d4;
    const float4 q = *reinterpret_cast<const float4*>(base + (long)row * ld + c4 * 4);
    const float4 k = *reinterpret_cast<const float4*>(base + (long)row * ld + nH * dH + c4 * 4);
    const float4 v = *reinterpret_cast<const float4*>(base + (long)row * ld + 2 * nH * dH + c4 * 4);
    const float4 o = *reinterpret_cast<const float4*>(dob + (long)row * (nH * dH) + c4 * 4);
    float* qd = Qs + row * ALD + c4 * 4; qd[0] = q.x; qd[1] = q.y; qd[2] = q.z; qd[3] = q.w;
    float* kd = Ks + row * ALD + c4 * 4; kd[0] = k.x; kd[1] = k.y; kd[2] = k.z; kd[3] = k.w;
    float* vd = Vs + row * ALD + c4 * 4; vd[0] = v.x; vd[1] = v.y; vd[2] = v.z; vd[3] = v.w;
    float* od = Os + row * ALD + c4 * 4; od[0] = o.x; od[1] = o.y; od[2] = o.z; od[3] = o.w;
  }
  const int LP = L + 1;
  const float* pb = probs + ((long)b * nH + hd) * L * L;
  for (int e = threadIdx.x; e < L * L; e += 256) Ps[(e / L) * LP + (e % L)] = pb[e];
  __syncthreads();
  // dP -> Ds
  for (int e = threadIdx.x; e < L * L; e += 256) {
    const int i = e / L, j = e % L;
    float s = 0.f;
#pragma unroll 16
    for (int d = 0; d < dH; ++d) s = fmaf(Os[i * ALD + d], Vs[j * ALD + d], s);
    Ds[i * LP + j] = s;
  }
  __syncthreads();
  {
    const int row = threadIdx.x >> 2, sub = threadIdx.x & 3;
    if (row < L) {
      float t = 0.f;
      for (int j = sub; j < L; j += 4) t += Ds[row * LP + j] * Ps[row * LP + j];
      t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64);
      for (int j = sub; j < L; j += 4) Ds[row * LP + j] = Ps[row * LP + j] * (Ds[row * LP + j] - t) * scale;
    }
  }
  __syncthreads();
  float* dq = dqkv + (long)b * L * ld + hd * dH;
  for (int e = threadIdx.x; e < L * dH; e += 256) {
    const int i = e / dH, d = e % dH;
    float sq = 0.f, sk = 0.f, sv = 0.f;
    for (int j = 0; j < L; ++j) {
      sq = fmaf(Ds[i * LP + j], Ks[j * ALD + d], sq);   // dQ[i] = sum_j dS[i][j] K[j]
      sk = fmaf(Ds[j * LP + i], Qs[j * ALD + d], sk);   // dK[i] = sum_j dS[j][i] Q[j]
      sv = fmaf(Ps[j * LP + i], Os[j * ALD + d], sv);   // dV[i] = sum_j P[j][i] dO[j]
    }
    dq[(long)i * ld + d] = sq;
    dq[(long)i * ld + nH * dH + d] = sk;
    dq[(long)i * ld + 2 * nH * dH + d] = sv;
  }
}

// dword[ids[t]] += dx[t]  (fp32 atomics: 256 contiguous bytes per wave-instruction; MI355X_MICROARCH "Global float atomics")
__global__ void embed_bwd_kernel(const long* __restrict__ ids, const float* __restrict__ dx, long T, int H,
                                 float* __restrict__ dword) {
  const long t = blockIdx.x;
  float* dst = dword + ids[t] * H;
  for (int c = threadIdx.x; c < H; c += blockDim.x) atomicAdd(dst + c, dx[t * H + c]);
}

// dx = dy * gelu'(pre)
__global__ void gelu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ pre, long n, float* __restrict__ dx) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dx[i] = dy[i] * gelu_erf_grad(pre[i]);
}

}  // namespace

extern "C" int cxrk_gelu_bwd(const float* dy, const float* pre, long n, float* dx, hipStream_t stream) {
  CXRK_CHECK_ARG(dy && pre && dx && n > 0);
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dy, pre, n, dx);
  CXRK_LAUNCH_CHECK();
  return CXRK_OK;
}

extern "C" int cxrk_embed_ln_fwd(const long* ids, const float* word, const float* pos, const float* type,
                                 const float* gamma, const float* beta, float eps, long T, int L, int H, float* y,
                                 float* xhat, float* rstd, hipStream_t stream) {
  CXRK_CHECK_ARG(ids && word && pos && type && gamma && beta && y && T > 0 && L > 0 && H > 0 && H <= 64 * LN_MAXV);
  hipLaunchKernelGGL((ln_fwd_kernel<true>), dim3((unsigned)((T + 3) / 4)), dim3(256), 0, stream, nullptr, nullptr, ids, word,
                     pos, type, gamma, beta, eps, T, H, L, y, xhat, rstd);
  CXRK_LAUNCH_CHECK();
  return CXRK_OK;
}

extern "C" int cxrk_residual_ln_fwd(const float* x, const float* res, const float* gamma, const float* beta, float eps,
                                    long rows, int H, float* y, float* xhat, float* rstd, hipStream_t stream) {
  CXRK_CHECK_ARG(x && gamma && beta && y && rows > 0 && H > 0 && H <= 64 * LN_MAXV);
  hipLaunchKernelGGL((ln_fwd_kernel<false>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, x, res, nullptr, nullptr,
                     nullptr, nullptr, gamma, beta, eps, rows, H, 1, y, xhat, rstd);
  CXRK_LAUNCH_CHECK();
  return CXRK_OK;
}

static int ln_bwd_blocks(long rows) {
  long nb = (rows + 63) / 64;
  if (nb > 256) nb = 256;   // one block per CU; the final reduction walks nb partials per column
  if (nb < 1) nb = 1;
  return (int)nb;
}
extern "C" size_t cxrk_residual_ln_bwd_ws_bytes(long rows, int H) { return (size_t)ln_bwd_blocks(rows) * 2 * H * sizeof(float); }

extern "C" int cxrk_residual_ln_bwd(const float* dy, const float* xhat, const float* rstd, const float* gamma, long rows,
                                    int H, const float* dx_add, float* dx, float* dgamma, float* dbeta, int accumulate,
                                    float* ws, size_t ws_bytes, hipStream_t stream) {
  CXRK_CHECK_ARG(dy && xhat && rstd && gamma && dx && dgamma && dbeta && rows > 0 && H > 0 && H <= 64 * LN_MAXV);
  int nb = ln_bwd_blocks(rows);
  if (ws == nullptr || ws_bytes < (size_t)nb * 2 * H * sizeof(float)) return CXRK_ERR_WS;
  const int rows_per = (int)((rows + nb - 1) / nb);
  nb = (int)((rows + rows_per - 1) / rows_per);
  hipLaunchKernelGGL(ln_bwd_kernel, dim3(nb), dim3(256), 0, stream, dy, xhat, rstd, gamma, rows, H, rows_per, dx, dx_add, ws);
  CXRK_LAUNCH_CHECK();
  hipLaunchKernelGGL(ln_bwd_final_kernel, dim3(ceil_div(H, 256)), dim3(256), 0, stream, ws, nb, H, dgamma, dbeta, accumulate);
  CXRK_LAUNCH_CHECK();
  return CXRK_OK;
}

extern "C" int cxrk_attn_fwd(const float* qkv, const long* mask, int B, int L, int nH, int dH, float* ctx, float* probs,
                             hipStream_t stream) {
  CXRK_CHECK_ARG(qkv && ctx && B > 0 && nH > 0 && aligned16(qkv));
  if (dH > AD || dH < 4 || (dH % 4) != 0 || L > AL || L < 1) return CXRK_ERR_UNSUPPORTED;
  const int ALD = dH + 1;
  const size_t sh = (size_t)(3 * L * ALD + L * (L + 1)) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)((3 * AL * (AD + 1) + AL * (AL + 1)) * sizeof(float)));
    attr_set = true;
  }
  hipLaunchKernelGGL(attn_fwd_kernel, dim3((unsigned)(B * nH)), dim3(256), sh, stream, qkv, mask, L, nH, dH,
                     1.0f / sqrtf((float)dH), ctx, probs);
  CXRK_LAUNCH_CHECK();
  return CXRK_OK;
}

extern "C" int cxrk_attn_bwd(const float* qkv, const float* probs, const float* dctx, int B, int L, int nH, int dH,
                             float* dqkv, hipStream_t stream) {
  CXRK_CHECK_ARG(qkv && probs && dctx && dqkv && B > 0 && nH > 0 && aligned16(qkv) && aligned16(dctx));
  if (dH > AD || dH < 4 || (dH % 4) != 0 || L > AL || L < 1) return CXRK_ERR_UNSUPPORTED;
  const int ALD = dH + 1;
  const size_t sh = (size_t)(4 * L * ALD + 2 * L * (L + 1)) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)((4 * AL * (AD + 1) + 2 * AL * (AL + 1)) * sizeof(float)));
    attr_set = true;
  }
  hipLaunchKernelGGL(attn_bwd_kernel, dim3((unsigned)(B * nH)), dim3(256), sh, stream, qkv, probs, dctx, L, nH, dH,
                     1.0f / sqrtf((float)dH), dqkv);
  CXRK_LAUNCH_CHECK();
  return CXRK_OK;
}

extern "C" int cxrk_embed_bwd(const long* ids, const float* dx, long T, int H, float* dword, hipStream_t stream) {
  CXRK_CHECK_ARG(ids && dx && dword && T > 0 && H > 0);
  hipLaunchKernelGGL(embed_bwd_kernel, dim3((unsigned)T), dim3(256), 0, stream, ids, dx, T, H, dword);
  CXRK_LAUNCH_CHECK();
  return CXRK_OK;
}
