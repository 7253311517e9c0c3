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

// H % 8 == 0: a lane owns 8 consecutive columns (lane + 64*i)*8.. -> 16-byte accesses; y is written as fp32 or as split-bf16
// planes (yp: hi plane, lo plane yplane elements behind), the format the next GEMM consumes in split-bf16 mode.
constexpr int LN_MAXV8 = LN_MAXV / 8;
template <bool EMBED>
__global__ __launch_bounds__(256) void ln_fwd_vec8_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                          const long* __restrict__ ids, const float* __restrict__ word,
                                                          const float* __restrict__ pos, const float* __restrict__ type,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float eps, long rows, int H, int L, float* __restrict__ y,
                                                          unsigned short* __restrict__ yp, long yplane,
                                                          float* __restrict__ xhat, float* __restrict__ rstd_out) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const int H8 = H / 8;
  float v[LN_MAXV8][8];
  float s = 0.f;
  const float* xr = EMBED ? word + ids[row] * H : x + row * H;
  const float* rr = EMBED ? pos + (row % L) * H : (res ? res + row * H : nullptr);
  auto ld8 = [](const float* p, float (&o)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
  };
#pragma unroll
  for (int i = 0; i < LN_MAXV8; ++i) {
    const int c8 = lane + i * 64;
#pragma unroll
    for (int q = 0; q < 8; ++q) v[i][q] = 0.f;
    if (c8 < H8) {
      ld8(xr + c8 * 8, v[i]);
      if (rr) { float t[8]; ld8(rr + c8 * 8, t);
#pragma unroll
        for (int q = 0; q < 8; ++q) v[i][q] += t[q]; }
      if (EMBED) { float t[8]; ld8(type + c8 * 8, t);
#pragma unroll
        for (int q = 0; q < 8; ++q) v[i][q] += t[q]; }
#pragma unroll
      for (int q = 0; q < 8; ++q) s += v[i][q];
    }
  }
  const float mean = wave_sum(s) / (float)H;
  float qq = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV8; ++i) {
    const int c8 = lane + i * 64;
#pragma unroll
    for (int q = 0; q < 8; ++q) { const float d = c8 < H8 ? v[i][q] - mean : 0.f; v[i][q] = d; qq += d * d; }
  }
  const float rs = 1.0f / sqrtf(wave_sum(qq) / (float)H + eps);
#pragma unroll
  for (int i = 0; i < LN_MAXV8; ++i) {
    const int c8 = lane + i * 64;
    if (c8 < H8) {
      float g[8], b[8], o[8];
      ld8(gamma + c8 * 8, g); ld8(beta + c8 * 8, b);
#pragma unroll
      for (int q = 0; q < 8; ++q) { v[i][q] *= rs; o[q] = v[i][q] * g[q] + b[q]; }
      if (xhat) {
        *reinterpret_cast<float4*>(xhat + row * H + c8 * 8) = make_float4(v[i][0], v[i][1], v[i][2], v[i][3]);
        *reinterpret_cast<float4*>(xhat + row * H + c8 * 8 + 4) = make_float4(v[i][4], v[i][5], v[i][6], v[i][7]);
      }
      if (yp) planes_store8(yp, yplane, row * H + c8 * 8, o);
      else {
        *reinterpret_cast<float4*>(y + row * H + c8 * 8) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4*>(y + row * H + c8 * 8 + 4) = make_float4(o[4], o[5], o[6], o[7]);
      }
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
// H % 4 == 0: a lane owns float4 columns (lane + 64*i)*4..+3 -> 16-byte loads / stores, a quarter of the instructions.
constexpr int LN_MAXV4 = LN_MAXV / 4;
__global__ __launch_bounds__(256) void ln_bwd_vec_kernel(const float* __restrict__ dy, const float* __restrict__ xhat,
                                                         const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                         long rows, int H, int rows_per, float* __restrict__ dx,
                                                         unsigned short* __restrict__ dxp, long dxplane,
                                                         const float* __restrict__ dx_add, float* __restrict__ part, int np) {
  // np = 2: partials of dgamma, dbeta; np = 3: also the column sums of the OUTPUT (the bias gradient of the dense layer that
  // produced this LayerNorm's input: its dy is exactly what this kernel writes, so the separate column-sum pass over it goes away)
  __shared__ float sh[3][4][64 * LN_MAXV];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long r0 = (long)blockIdx.x * rows_per;
  const long r1 = min(rows, r0 + rows_per);
  const int H4 = H / 4;
  float4 dg[LN_MAXV4], db[LN_MAXV4], gm[LN_MAXV4], ds[LN_MAXV4];
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < LN_MAXV4; ++i) {
    dg[i] = z4; db[i] = z4; ds[i] = z4;
    const int c4 = lane + i * 64;
    gm[i] = c4 < H4 ? *reinterpret_cast<const float4*>(gamma + c4 * 4) : z4;
  }
  for (long row = r0 + w; row < r1; row += 4) {
    float4 g[LN_MAXV4], xh[LN_MAXV4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV4; ++i) {
      const int c4 = lane + i * 64;
      float4 d = z4, x_ = z4;
      if (c4 < H4) { d = *reinterpret_cast<const float4*>(dy + row * H + c4 * 4); x_ = *reinterpret_cast<const float4*>(xhat + row * H + c4 * 4); }
      xh[i] = x_;
      g[i] = make_float4(d.x * gm[i].x, d.y * gm[i].y, d.z * gm[i].z, d.w * gm[i].w);
      dg[i].x += d.x * x_.x; dg[i].y += d.y * x_.y; dg[i].z += d.z * x_.z; dg[i].w += d.w * x_.w;
      db[i].x += d.x; db[i].y += d.y; db[i].z += d.z; db[i].w += d.w;
      s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
      s2 += (g[i].x * x_.x + g[i].y * x_.y) + (g[i].z * x_.z + g[i].w * x_.w);
    }
    s1 = wave_sum(s1) / (float)H; s2 = wave_sum(s2) / (float)H;
    const float rs = rstd[row];
#pragma unroll
    for (int i = 0; i < LN_MAXV4; ++i) {
      const int c4 = lane + i * 64;
      if (c4 < H4) {
        float4 o = make_float4(rs * (g[i].x - s1 - xh[i].x * s2), rs * (g[i].y - s1 - xh[i].y * s2),
                               rs * (g[i].z - s1 - xh[i].z * s2), rs * (g[i].w - s1 - xh[i].w * s2));
        if (dx_add) { const float4 a = *reinterpret_cast<const float4*>(dx_add + row * H + c4 * 4); o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w; }
        ds[i].x += o.x; ds[i].y += o.y; ds[i].z += o.z; ds[i].w += o.w;
        if (dxp) { const float ov[4] = {o.x, o.y, o.z, o.w}; planes_store4(dxp, dxplane, row * H + c4 * 4, ov); }
        else *reinterpret_cast<float4*>(dx + row * H + c4 * 4) = o;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < LN_MAXV4; ++i) {
    const int c = (lane + i * 64) * 4;
    if (c < 64 * LN_MAXV) {
      *reinterpret_cast<float4*>(&sh[0][w][c]) = dg[i];
      *reinterpret_cast<float4*>(&sh[1][w][c]) = db[i];
      *reinterpret_cast<float4*>(&sh[2][w][c]) = ds[i];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < H; c += 256)
    for (int k = 0; k < np; ++k) part[((long)blockIdx.x * np + k) * H + c] = (sh[k][0][c] + sh[k][1][c]) + (sh[k][2][c] + sh[k][3][c]);
}
// Column sums of the block partials: 64 columns x 16 partial groups per block, groups added in order through LDS.
__global__ __launch_bounds__(1024) void ln_bwd_final_kernel(const float* __restrict__ part, int nblk, int np, int H, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int accumulate, float* __restrict__ dxsum,
                                                            int dxsum_accumulate) {
  __shared__ float sh[16][64];
  const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl, k = blockIdx.y;
  float a = 0.f;
  if (c < H)
    for (int p = pl; p < nblk; p += 16) a += part[((long)p * np + k) * H + c];
  sh[pl][cl] = a;
  __syncthreads();
  if (pl == 0 && c < H) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += sh[i][cl];
    float* out = k == 0 ? dgamma : (k == 1 ? dbeta : dxsum);
    out[c] = (k == 2 ? dxsum_accumulate : accumulate) ? out[c] + t : t;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Attention for short sequences (L <= 64, head dim 64): one 256-thread workgroup per (sequence, head).
// Q,K,V live in one fused [T][3*nH*64] buffer (the QKV GEMM output).  Everything is staged in LDS; the products are
// small (2*L*L*64 FLOP each) and run on the fp32 VALU.  probs [B][nH][L][L] are saved for the backward.
// ---------------------------------------------------------------------------------------------------------------
constexpr int AD = 64;       // max head dim
constexpr int AL = 64;       // max sequence length handled here

// Register tiling: a thread owns a 2x2 block of an L x L product (operands read as float4 along the head dimension:
// 4 ds_read_b128 per 16 FMAs) or 2 rows x 4 columns of an L x dH product (2-3 LDS reads per 8 FMAs).  With one output
// element per thread and two ds_read_b32 per FMA the kernels were bound by LDS bandwidth (0.22 ms / 0.46 ms per layer).
__device__ __forceinline__ float dot4(const float4& a, const float4& b, float s) {
  return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, fmaf(a.w, b.w, s))));
}
__device__ __forceinline__ void fma4(float4& acc, float p, const float4& v) {
  acc.x = fmaf(p, v.x, acc.x); acc.y = fmaf(p, v.y, acc.y); acc.z = fmaf(p, v.z, acc.z); acc.w = fmaf(p, v.w, acc.w);
}
// Out[i][j] = sum_d X[i][d] * Y[j][d] for a 2x2 block (rows clamped to L-1; the caller discards the clamped duplicates)
__device__ __forceinline__ void dot2x2(const float* X, const float* Y, int ALD, int d4n, int i0, int i1, int j0, int j1,
                                       float (&o)[2][2]) {
  o[0][0] = o[0][1] = o[1][0] = o[1][1] = 0.f;
  const float4* x0 = reinterpret_cast<const float4*>(X + i0 * ALD); const float4* x1 = reinterpret_cast<const float4*>(X + i1 * ALD);
  const float4* y0 = reinterpret_cast<const float4*>(Y + j0 * ALD); const float4* y1 = reinterpret_cast<const float4*>(Y + j1 * ALD);
#pragma unroll 4
  for (int d = 0; d < d4n; ++d) {
    const float4 a0 = x0[d], a1 = x1[d], b0 = y0[d], b1 = y1[d];
    o[0][0] = dot4(a0, b0, o[0][0]); o[0][1] = dot4(a0, b1, o[0][1]);
    o[1][0] = dot4(a1, b0, o[1][0]); o[1][1] = dot4(a1, b1, o[1][1]);
  }
}

__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ qkv, const long* __restrict__ mask, int L,
                                                       int nH, int dH, float scale, float* __restrict__ ctx,
                                                       unsigned short* __restrict__ ctxp, long ctxplane,
                                                       float* __restrict__ probs) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ALD = dH + 4;   // 16-byte aligned rows
  float* Qs = smem; float* Ks = Qs + L * ALD; float* Vs = Ks + L * ALD; float* Ps = Vs + L * ALD;  // Ps[L][L+1]
  const int b = blockIdx.x / nH, hd = blockIdx.x % nH;
  const int ld = 3 * nH * dH;
  const int d4 = dH / 4;
  const float* base = qkv + (long)b * L * ld + hd * dH;
  for (int i = threadIdx.x; i < L * d4; i += 256) {
    const int row = i / d4, c4 = i % d4;
    *reinterpret_cast<float4*>(Qs + row * ALD + c4 * 4) = *reinterpret_cast<const float4*>(base + (long)row * ld + c4 * 4);
    *reinterpret_cast<float4*>(Ks + row * ALD + c4 * 4) = *reinterpret_cast<const float4*>(base + (long)row * ld + nH * dH + c4 * 4);
    *reinterpret_cast<float4*>(Vs + row * ALD + c4 * 4) = *reinterpret_cast<const float4*>(base + (long)row * ld + 2 * nH * dH + c4 * 4);
  }
  __syncthreads();
  const int LP = L + 1;
  const int nb = (L + 1) / 2;
  for (int blk = threadIdx.x; blk < nb * nb; blk += 256) {
    const int i0 = 2 * (blk / nb), j0 = 2 * (blk % nb);
    const int i1 = min(i0 + 1, L - 1), j1 = min(j0 + 1, L - 1);
    float o[2][2];
    dot2x2(Qs, Ks, ALD, d4, i0, i1, j0, j1, o);
    const bool m0 = mask && mask[(long)b * L + j0] == 0, m1 = mask && mask[(long)b * L + j1] == 0;
    // HuggingFace adds finfo(float32).min to the masked scores (score + min == min in fp32): a finite value, so that a
    // sequence whose mask is all zeros gets a uniform attention row, as in the reference, instead of exp(-inf - -inf) = NaN
    constexpr float MASKED = -3.4028234663852886e38f;
    Ps[i0 * LP + j0] = m0 ? MASKED : o[0][0] * scale;
    Ps[i0 * LP + j1] = m1 ? MASKED : o[0][1] * scale;
    Ps[i1 * LP + j0] = m0 ? MASKED : o[1][0] * scale;
    Ps[i1 * LP + j1] = m1 ? MASKED : o[1][1] * scale;
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
  const long obase = (long)b * L * (nH * dH) + hd * dH;
  float* out = ctx + obase;
  for (int blk = threadIdx.x; blk < nb * d4; blk += 256) {
    const int i0 = 2 * (blk / d4), dq = blk % d4;
    const int i1 = min(i0 + 1, L - 1);
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    for (int j = 0; j < L; ++j) {
      const float4 v = *reinterpret_cast<const float4*>(Vs + j * ALD + dq * 4);
      fma4(a0, Ps[i0 * LP + j], v); fma4(a1, Ps[i1 * LP + j], v);
    }
    if (ctxp) {
      const float v0[4] = {a0.x, a0.y, a0.z, a0.w}, v1[4] = {a1.x, a1.y, a1.z, a1.w};
      planes_store4(ctxp, ctxplane, obase + (long)i0 * (nH * dH) + dq * 4, v0);
      if (i0 + 1 < L) planes_store4(ctxp, ctxplane, obase + (long)i1 * (nH * dH) + dq * 4, v1);
    } else {
      *reinterpret_cast<float4*>(out + (long)i0 * (nH * dH) + dq * 4) = a0;
      if (i0 + 1 < L) *reinterpret_cast<float4*>(out + (long)i1 * (nH * dH) + dq * 4) = a1;
    }
  }
}

// dV = P^T dO ; dP = dO V^T ; dS = P o (dP - rowsum(dP o P)) ; dQ = scale * dS K ; dK = scale * dS^T Q
__global__ __launch_bounds__(256) void attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ probs,
                                                       const float* __restrict__ dctx, int L, int nH, int dH, float scale,
                                                       float* __restrict__ dqkv, unsigned short* __restrict__ dqkvp, long dqkvplane) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ALD = dH + 4;
  float* Qs = smem; float* Ks = Qs + L * ALD; float* Vs = Ks + L * ALD; float* Os = Vs + L * ALD;
  float* Ps = Os + L * ALD; float* Ds = Ps + L * (L + 1);  // Ps = P, Ds = dS
  const int b = blockIdx.x / nH, hd = blockIdx.x % nH;
  const int ld = 3 * nH * dH;
  const int d4 = dH / 4;
  const float* base = qkv + (long)b * L * ld + hd * dH;
  const float* dob = dctx + (long)b * L * (nH * dH) + hd * dH;
  for (int i = threadIdx.x; i < L * d4; i += 256) {
    const int row = i / d4, c4 = i % d4;
    *reinterpret_cast<float4*>(Qs + row * ALD + c4 * 4) = *reinterpret_cast<const float4*>(base + (long)row * ld + c4 * 4);
    *reinterpret_cast<float4*>(Ks + row * ALD + c4 * 4) = *reinterpret_cast<const float4*>(base + (long)row * ld + nH * dH + c4 * 4);
    *reinterpret_cast<float4*>(Vs + row * ALD + c4 * 4) = *reinterpret_cast<const float4*>(base + (long)row * ld + 2 * nH * dH + c4 * 4);
    *reinterpret_cast<float4*>(Os + row * ALD + c4 * 4) = *reinterpret_cast<const float4*>(dob + (long)row * (nH * dH) + c4 * 4);
  }
  const int LP = L + 1;
  const float* pb = probs + ((long)b * nH + hd) * L * L;
  for (int e = threadIdx.x; e < L * L; e += 256) Ps[(e / L) * LP + (e % L)] = pb[e];
  __syncthreads();
  // dP -> Ds
  const int nb = (L + 1) / 2;
  for (int blk = threadIdx.x; blk < nb * nb; blk += 256) {
    const int i0 = 2 * (blk / nb), j0 = 2 * (blk % nb);
    const int i1 = min(i0 + 1, L - 1), j1 = min(j0 + 1, L - 1);
    float o[2][2];
    dot2x2(Os, Vs, ALD, d4, i0, i1, j0, j1, o);
    Ds[i0 * LP + j0] = o[0][0]; Ds[i0 * LP + j1] = o[0][1]; Ds[i1 * LP + j0] = o[1][0]; Ds[i1 * LP + j1] = o[1][1];
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
  const long qbase = (long)b * L * ld + hd * dH;
  float* dq = dqkv + qbase;
  auto put = [&](long off, const float4& v) {
    if (dqkvp) { const float t[4] = {v.x, v.y, v.z, v.w}; planes_store4(dqkvp, dqkvplane, qbase + off, t); }
    else *reinterpret_cast<float4*>(dq + off) = v;
  };
  for (int blk = threadIdx.x; blk < nb * d4; blk += 256) {
    const int i0 = 2 * (blk / d4), c = blk % d4;
    const int i1 = min(i0 + 1, L - 1);
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 q0 = z, q1 = z, k0 = z, k1 = z, v0 = z, v1 = z;
    for (int j = 0; j < L; ++j) {
      const float4 kj = *reinterpret_cast<const float4*>(Ks + j * ALD + c * 4);
      const float4 qj = *reinterpret_cast<const float4*>(Qs + j * ALD + c * 4);
      const float4 oj = *reinterpret_cast<const float4*>(Os + j * ALD + c * 4);
      fma4(q0, Ds[i0 * LP + j], kj); fma4(q1, Ds[i1 * LP + j], kj);   // dQ[i] = sum_j dS[i][j] K[j]
      fma4(k0, Ds[j * LP + i0], qj); fma4(k1, Ds[j * LP + i1], qj);   // dK[i] = sum_j dS[j][i] Q[j]
      fma4(v0, Ps[j * LP + i0], oj); fma4(v1, Ps[j * LP + i1], oj);   // dV[i] = sum_j P[j][i] dO[j]
    }
    const long o0 = (long)i0 * ld + c * 4;
    put(o0, q0); put(o0 + nH * dH, k0); put(o0 + 2 * nH * dH, v0);
    if (i0 + 1 < L) {
      const long o1 = (long)i1 * ld + c * 4;
      put(o1, q1); put(o1 + nH * dH, k1); put(o1 + 2 * nH * dH, v1);
    }
  }
}

// ---- sequences of 64 < L <= ALONG tokens (the reference tokenizer accepts prompts up to max_position_embeddings = 512,
// text/inference_engine.py:30,44-46): a head no longer fits in LDS, so the work is tiled.  Forward: one workgroup per (sequence,
// head, tile of TQ = 32 queries) keeps its 32 full score rows in LDS ([32][L + 1] floats), streams the keys and then the values
// through a 64-row LDS tile, and applies the same full-row softmax (maximum subtracted, finite mask constant) as the short kernel.
// Backward, without atomics: a query-tile kernel forms t_i = sum_j P_ij dP_ij, dS = P o (dP - t) * scale (written to a workspace)
// and dQ; a key-tile kernel then sums dK = dS^T Q and dV = P^T dO over the query tiles in ascending order.
constexpr int ALONG = 512;
constexpr int TQ = 32, TK = 64;

// rows [r0, r0 + nrows) of one head's [L][dH] slice (row stride ld) -> dst[nrows][ALD]; rows >= L are zero
__device__ __forceinline__ void load_rows(float* dst, const float* src, long ld, int r0, int nrows, int L, int d4, int ALD) {
  for (int i = threadIdx.x; i < nrows * d4; i += 256) {
    const int row = i / d4, c4 = i % d4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + row < L) v = *reinterpret_cast<const float4*>(src + (long)(r0 + row) * ld + c4 * 4);
    *reinterpret_cast<float4*>(dst + row * ALD + c4 * 4) = v;
  }
}
// o[c] = <X[i], Y[jg + 8 c]> for c = 0..7 (X, Y rows of ALD floats in LDS)
__device__ __forceinline__ void dot1x8(const float* X, const float* Y, int ALD, int d4n, int i, int jg, float (&o)[8]) {
#pragma unroll
  for (int c = 0; c < 8; ++c) o[c] = 0.f;
  const float4* x = reinterpret_cast<const float4*>(X + i * ALD);
  for (int d = 0; d < d4n; ++d) {
    const float4 a = x[d];
#pragma unroll
    for (int c = 0; c < 8; ++c) o[c] = dot4(a, reinterpret_cast<const float4*>(Y + (jg + 8 * c) * ALD)[d], o[c]);
  }
}

__global__ __launch_bounds__(256) void attn_fwd_long_kernel(const float* __restrict__ qkv, const long* __restrict__ mask, int L, int nH,
                                                            int dH, float scale, float* __restrict__ ctx,
                                                            unsigned short* __restrict__ ctxp, long ctxplane, float* __restrict__ probs) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ALD = dH + 4, d4 = dH / 4, LP = L + 1;
  float* Qs = smem; float* Ts = Qs + TQ * ALD; float* Ps = Ts + TK * ALD;     // Ps[TQ][L + 1]
  const int nqt = (L + TQ - 1) / TQ;
  const int qt = blockIdx.x % nqt, bh = blockIdx.x / nqt;
  const int b = bh / nH, hd = bh % nH;
  const int ld = 3 * nH * dH;
  const float* base = qkv + (long)b * L * ld + hd * dH;
  const int q0 = qt * TQ;
  const int i = threadIdx.x >> 3, sub = threadIdx.x & 7;      // query row of the tile, lane of the row
  constexpr float MASKED = -3.4028234663852886e38f;
  load_rows(Qs, base, ld, q0, TQ, L, d4, ALD);
  for (int k0 = 0; k0 < L; k0 += TK) {
    __syncthreads();                                            // Qs ready / previous key tile consumed
    load_rows(Ts, base + nH * dH, ld, k0, TK, L, d4, ALD);
    __syncthreads();
    float o[8];
    dot1x8(Qs, Ts, ALD, d4, i, sub, o);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int j = k0 + sub + 8 * c;
      if (j < L) Ps[i * LP + j] = (mask && mask[(long)b * L + j] == 0) ? MASKED : o[c] * scale;
    }
  }
  __syncthreads();
  {   // softmax of row i: 8 lanes
    float m = -INFINITY;
    for (int j = sub; j < L; j += 8) m = fmaxf(m, Ps[i * LP + j]);
    m = fmaxf(m, __shfl_xor(m, 1, 64)); m = fmaxf(m, __shfl_xor(m, 2, 64)); m = fmaxf(m, __shfl_xor(m, 4, 64));
    float sum = 0.f;
    for (int j = sub; j < L; j += 8) { const float pv = expf(Ps[i * LP + j] - m); Ps[i * LP + j] = pv; sum += pv; }
    sum += __shfl_xor(sum, 1, 64); sum += __shfl_xor(sum, 2, 64); sum += __shfl_xor(sum, 4, 64);
    const float inv = 1.0f / sum;
    for (int j = sub; j < L; j += 8) {
      const float pv = Ps[i * LP + j] * inv;
      Ps[i * LP + j] = pv;
      if (probs && q0 + i < L) probs[(((long)b * nH + hd) * L + q0 + i) * L + j] = pv;
    }
  }
  float4 acc[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};   // columns 4*sub.. and 4*(sub + 8)..
  for (int k0 = 0; k0 < L; k0 += TK) {
    __syncthreads();                                            // softmax done / previous value tile consumed
    load_rows(Ts, base + 2 * nH * dH, ld, k0, TK, L, d4, ALD);
    __syncthreads();
    const int jn = min(TK, L - k0);
    for (int j = 0; j < jn; ++j) {
      const float pv = Ps[i * LP + k0 + j];
      if (sub < d4) fma4(acc[0], pv, *reinterpret_cast<const float4*>(Ts + j * ALD + sub * 4));
      if (sub + 8 < d4) fma4(acc[1], pv, *reinterpret_cast<const float4*>(Ts + j * ALD + (sub + 8) * 4));
    }
  }
  if (q0 + i < L) {
    const long obase = (long)b * L * (nH * dH) + hd * dH + (long)(q0 + i) * (nH * dH);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int dq = sub + 8 * h;
      if (dq >= d4) continue;
      if (ctxp) { const float v[4] = {acc[h].x, acc[h].y, acc[h].z, acc[h].w}; planes_store4(ctxp, ctxplane, obase + dq * 4, v); }
      else *reinterpret_cast<float4*>(ctx + obase + dq * 4) = acc[h];
    }
  }
}

// query-tile pass of the backward: dS rows -> ws [B][nH][L][L], dQ
__global__ __launch_bounds__(256) void attn_bwd_long_q_kernel(const float* __restrict__ qkv, const float* __restrict__ probs,
                                                              const float* __restrict__ dctx, int L, int nH, int dH, float scale,
                                                              float* __restrict__ dS, float* __restrict__ dqkv,
                                                              unsigned short* __restrict__ dqkvp, long dqkvplane) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ALD = dH + 4, d4 = dH / 4, DP = TK + 1;
  float* Os = smem; float* Ks = Os + TQ * ALD; float* Vs = Ks + TK * ALD; float* Ds = Vs + TK * ALD;   // Ds[TQ][TK + 1]
  const int nqt = (L + TQ - 1) / TQ;
  const int qt = blockIdx.x % nqt, bh = blockIdx.x / nqt;
  const int b = bh / nH, hd = bh % nH;
  const int ld = 3 * nH * dH;
  const float* base = qkv + (long)b * L * ld + hd * dH;
  const float* dob = dctx + (long)b * L * (nH * dH) + hd * dH;
  const float* pb = probs + ((long)b * nH + hd) * L * L;
  float* dsb = dS + ((long)b * nH + hd) * L * L;
  const int q0 = qt * TQ;
  const int i = threadIdx.x >> 3, sub = threadIdx.x & 7;
  const bool rowok = q0 + i < L;
  load_rows(Os, dob, nH * dH, q0, TQ, L, d4, ALD);
  // pass 1: t_i = sum_j P_ij * <dO_i, V_j>
  float t = 0.f;
  for (int k0 = 0; k0 < L; k0 += TK) {
    __syncthreads();
    load_rows(Vs, base + 2 * nH * dH, ld, k0, TK, L, d4, ALD);
    __syncthreads();
    float o[8];
    dot1x8(Os, Vs, ALD, d4, i, sub, o);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int j = k0 + sub + 8 * c;
      if (rowok && j < L) t = fmaf(pb[(long)(q0 + i) * L + j], o[c], t);
    }
  }
  t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 4, 64);
  // pass 2: dS tile by tile, dQ_i += dS_ij K_j
  float4 acc[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
  for (int k0 = 0; k0 < L; k0 += TK) {
    __syncthreads();
    load_rows(Ks, base + nH * dH, ld, k0, TK, L, d4, ALD);
    load_rows(Vs, base + 2 * nH * dH, ld, k0, TK, L, d4, ALD);
    __syncthreads();
    float o[8];
    dot1x8(Os, Vs, ALD, d4, i, sub, o);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int jl = sub + 8 * c, j = k0 + jl;
      float v = 0.f;
      if (rowok && j < L) {
        v = pb[(long)(q0 + i) * L + j] * (o[c] - t) * scale;
        dsb[(long)(q0 + i) * L + j] = v;
      }
      Ds[i * DP + jl] = v;
    }
    __syncthreads();
    for (int j = 0; j < TK; ++j) {
      const float dv = Ds[i * DP + j];
      if (sub < d4) fma4(acc[0], dv, *reinterpret_cast<const float4*>(Ks + j * ALD + sub * 4));
      if (sub + 8 < d4) fma4(acc[1], dv, *reinterpret_cast<const float4*>(Ks + j * ALD + (sub + 8) * 4));
    }
  }
  if (rowok) {
    const long qoff = (long)b * L * ld + hd * dH + (long)(q0 + i) * ld;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int dq = sub + 8 * h;
      if (dq >= d4) continue;
      if (dqkvp) { const float v[4] = {acc[h].x, acc[h].y, acc[h].z, acc[h].w}; planes_store4(dqkvp, dqkvplane, qoff + dq * 4, v); }
      else *reinterpret_cast<float4*>(dqkv + qoff + dq * 4) = acc[h];
    }
  }
}

// key-tile pass of the backward: dK_j = sum_i dS_ij Q_i, dV_j = sum_i P_ij dO_i (query tiles in ascending order)
__global__ __launch_bounds__(256) void attn_bwd_long_kv_kernel(const float* __restrict__ qkv, const float* __restrict__ probs,
                                                               const float* __restrict__ dS, const float* __restrict__ dctx, int L,
                                                               int nH, int dH, float* __restrict__ dqkv,
                                                               unsigned short* __restrict__ dqkvp, long dqkvplane) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ALD = dH + 4, d4 = dH / 4, DP = TK + 1;
  float* Qs = smem; float* Os = Qs + TQ * ALD; float* Pt = Os + TQ * ALD; float* Dt = Pt + TQ * DP;   // Pt, Dt: [TQ][TK + 1]
  const int nkt = (L + TK - 1) / TK;
  const int kt = blockIdx.x % nkt, bh = blockIdx.x / nkt;
  const int b = bh / nH, hd = bh % nH;
  const int ld = 3 * nH * dH;
  const float* base = qkv + (long)b * L * ld + hd * dH;
  const float* dob = dctx + (long)b * L * (nH * dH) + hd * dH;
  const float* pb = probs + ((long)b * nH + hd) * L * L;
  const float* dsb = dS + ((long)b * nH + hd) * L * L;
  const int k0 = kt * TK;
  const int jr = threadIdx.x >> 2, sub = threadIdx.x & 3;     // key row of the tile, lane of the row: columns 4*(sub + 4 h)..
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 dk[4] = {z, z, z, z}, dv[4] = {z, z, z, z};
  for (int q0 = 0; q0 < L; q0 += TQ) {
    __syncthreads();
    load_rows(Qs, base, ld, q0, TQ, L, d4, ALD);
    load_rows(Os, dob, nH * dH, q0, TQ, L, d4, ALD);
    for (int e = threadIdx.x; e < TQ * TK; e += 256) {
      const int ii = e / TK, jj = e % TK;
      const bool ok = q0 + ii < L && k0 + jj < L;
      const long o = (long)(q0 + ii) * L + k0 + jj;
      Pt[ii * DP + jj] = ok ? pb[o] : 0.f;
      Dt[ii * DP + jj] = ok ? dsb[o] : 0.f;
    }
    __syncthreads();
    for (int ii = 0; ii < TQ; ++ii) {
      const float pv = Pt[ii * DP + jr], dsv = Dt[ii * DP + jr];
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const int dq = sub + 4 * h;
        if (dq < d4) {
          fma4(dk[h], dsv, *reinterpret_cast<const float4*>(Qs + ii * ALD + dq * 4));
          fma4(dv[h], pv, *reinterpret_cast<const float4*>(Os + ii * ALD + dq * 4));
        }
      }
    }
  }
  if (k0 + jr < L) {
    const long koff = (long)b * L * ld + hd * dH + (long)(k0 + jr) * ld;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int dq = sub + 4 * h;
      if (dq >= d4) continue;
      if (dqkvp) {
        const float a[4] = {dk[h].x, dk[h].y, dk[h].z, dk[h].w}, c[4] = {dv[h].x, dv[h].y, dv[h].z, dv[h].w};
        planes_store4(dqkvp, dqkvplane, koff + nH * dH + dq * 4, a);
        planes_store4(dqkvp, dqkvplane, koff + 2 * nH * dH + dq * 4, c);
      } else {
        *reinterpret_cast<float4*>(dqkv + koff + nH * dH + dq * 4) = dk[h];
        *reinterpret_cast<float4*>(dqkv + koff + 2 * nH * dH + dq * 4) = dv[h];
      }
    }
  }
}

// dst[r*ld + c] += src[r][c]  (src planes [rows][cols], cols % 8 == 0): adds a small planes tensor into strided rows of an fp32 one
// (the CLS rows of a [N, L, H] gradient)
__global__ void planes_add_rows_kernel(const unsigned short* __restrict__ src, long plane, long rows, int cols, float* __restrict__ dst,
                                       long ld) {
  const int c8n = cols / 8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < rows * c8n; i += (long)gridDim.x * blockDim.x) {
    const long r = i / c8n; const int c = (int)(i - r * c8n) * 8;
    float v[8]; planes_load8(src, plane, r * cols + c, v);
    float* d = dst + r * ld + c;
    float4 a = *reinterpret_cast<float4*>(d), b = *reinterpret_cast<float4*>(d + 4);
    a.x += v[0]; a.y += v[1]; a.z += v[2]; a.w += v[3]; b.x += v[4]; b.y += v[5]; b.z += v[6]; b.w += v[7];
    *reinterpret_cast<float4*>(d) = a; *reinterpret_cast<float4*>(d + 4) = b;
  }
}

// dx = dy * gelu'(pre)
__global__ void gelu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ pre, long n, float* __restrict__ dx) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dx[i] = dy[i] * gelu_erf_grad(pre[i]);
}

}  // namespace

extern "C" int cxrk_planes_add_rows(const void* src, long plane, long rows, int cols, float* dst, long ld, hipStream_t stream) {
  CXRK_CHECK_ARG(src && dst && rows > 0 && cols > 0 && (cols % 8) == 0 && (plane % 8) == 0 && (ld % 4) == 0 && aligned16(src) && aligned16(dst));
  long nb = (rows * (cols / 8) + 255) / 256; if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(planes_add_rows_kernel, dim3((unsigned)nb), dim3(256), 0, stream, static_cast<const unsigned short*>(src), plane, rows,
                     cols, dst, ld);
  CXRK_LAUNCH_CHECK();
  return CXRK_OK;
}

extern "C" int cxrk_gelu_bwd(const float* dy, const float* pre, long n, float* dx, hipStream_t stream) {
  CXRK_CHECK_ARG(dy && pre && dx && n > 0);
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dy, pre, n, dx);
  CXRK_LAUNCH_CHECK();
  return CXRK_OK;
}

static bool ln_vec8_ok(int H, const void* a, const void* b, const void* c, const void* d, const void* e, const void* f) {
  return (H % 8) == 0 && aligned16(a) && aligned16(b) && aligned16(c) && aligned16(d) && aligned16(e) && aligned16(f);
}

extern "C" int cxrk_embed_ln_fwd(const long* ids, const float* word, const float* pos, const float* type,
                                 const float* gamma, const float* beta, float eps, long T, int L, int H, void* y, long yplane,
                                 float* xhat, float* rstd, hipStream_t stream) {
  CXRK_CHECK_ARG(ids && word && pos && type && gamma && beta && y && T > 0 && L > 0 && H > 0 && H <= 64 * LN_MAXV && yplane >= 0);
  const bool v8 = ln_vec8_ok(H, word, pos, type, gamma, beta, y) && aligned16(xhat) && (yplane % 8) == 0;
  if (yplane > 0 && !v8) return CXRK_ERR_ARG;
  if (v8)
    hipLaunchKernelGGL((ln_fwd_vec8_kernel<true>), dim3((unsigned)((T + 3) / 4)), dim3(256), 0, stream, nullptr, nullptr, ids, word, pos,
                       type, gamma, beta, eps, T, H, L, yplane ? nullptr : static_cast<float*>(y),
                       yplane ? static_cast<unsigned short*>(y) : nullptr, yplane, xhat, rstd);
  else
    hipLaunchKernelGGL((ln_fwd_kernel<true>), dim3((unsigned)((T + 3) / 4)), dim3(256), 0, stream, nullptr, nullptr, ids, word,
                       pos, type, gamma, beta, eps, T, H, L, static_cast<float*>(y), xhat, rstd);
  CXRK_LAUNCH_CHECK();
  return CXRK_OK;
}

extern "C" int cxrk_residual_ln_fwd(const float* x, const float* res, const float* gamma, const float* beta, float eps,
                                    long rows, int H, void* y, long yplane, float* xhat, float* rstd, hipStream_t stream) {
  CXRK_CHECK_ARG(x && gamma && beta && y && rows > 0 && H > 0 && H <= 64 * LN_MAXV && yplane >= 0);
  const bool v8 = ln_vec8_ok(H, x, res, gamma, beta, y, xhat) && (yplane % 8) == 0;
  if (yplane > 0 && !v8) return CXRK_ERR_ARG;
  if (v8)
    hipLaunchKernelGGL((ln_fwd_vec8_kernel<false>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, x, res, nullptr, nullptr,
                       nullptr, nullptr, gamma, beta, eps, rows, H, 1, yplane ? nullptr : static_cast<float*>(y),
                       yplane ? static_cast<unsigned short*>(y) : nullptr, yplane, xhat, rstd);
  else
    hipLaunchKernelGGL((ln_fwd_kernel<false>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, x, res, nullptr, nullptr,
                       nullptr, nullptr, gamma, beta, eps, rows, H, 1, static_cast<float*>(y), xhat, rstd);
  CXRK_LAUNCH_CHECK();
  return CXRK_OK;
}

static int ln_bwd_blocks(long rows) {
  long nb = (rows + 15) / 16;
  if (nb > 1024) nb = 1024;   // 4 blocks per CU keep enough rows in flight; the final reduction sums the partials in 16 groups
  if (nb < 1) nb = 1;
  return (int)nb;
}
extern "C" size_t cxrk_residual_ln_bwd_ws_bytes(long rows, int H) { return (size_t)ln_bwd_blocks(rows) * 3 * H * sizeof(float); }

extern "C" int cxrk_residual_ln_bwd(const float* dy, const float* xhat, const float* rstd, const float* gamma, long rows,
                                    int H, const float* dx_add, void* dxv, long dxplane, float* dgamma, float* dbeta, int accumulate,
                                    float* dxsum, int dxsum_accumulate, float* ws, size_t ws_bytes, hipStream_t stream) {
  CXRK_CHECK_ARG(dy && xhat && rstd && gamma && dxv && dgamma && dbeta && rows > 0 && H > 0 && H <= 64 * LN_MAXV && dxplane >= 0);
  float* dx = dxplane ? nullptr : static_cast<float*>(dxv);
  unsigned short* dxp = dxplane ? static_cast<unsigned short*>(dxv) : nullptr;
  int nb = ln_bwd_blocks(rows);
  if (ws == nullptr || ws_bytes < (size_t)nb * 3 * H * sizeof(float)) return CXRK_ERR_WS;
  const int np = dxsum ? 3 : 2;
  const int rows_per = (int)((rows + nb - 1) / nb);
  nb = (int)((rows + rows_per - 1) / rows_per);
  const bool vec = (H % 4 == 0) && aligned16(dy) && aligned16(xhat) && aligned16(gamma) && aligned16(dxv) && (!dx_add || aligned16(dx_add)) &&
                   (dxplane % 4) == 0;
  if ((dxp || dxsum) && !vec) return CXRK_ERR_ARG;
  if (vec)
    hipLaunchKernelGGL(ln_bwd_vec_kernel, dim3(nb), dim3(256), 0, stream, dy, xhat, rstd, gamma, rows, H, rows_per, dx, dxp, dxplane, dx_add, ws, np);
  else
    hipLaunchKernelGGL(ln_bwd_kernel, dim3(nb), dim3(256), 0, stream, dy, xhat, rstd, gamma, rows, H, rows_per, dx, dx_add, ws);
  CXRK_LAUNCH_CHECK();
  hipLaunchKernelGGL(ln_bwd_final_kernel, dim3(ceil_div(H, 64), np), dim3(1024), 0, stream, ws, nb, np, H, dgamma, dbeta, accumulate, dxsum,
                     dxsum_accumulate);
  CXRK_LAUNCH_CHECK();
  return CXRK_OK;
}

extern "C" int cxrk_attn_fwd(const float* qkv, const long* mask, int B, int L, int nH, int dH, void* ctxv, long ctxplane, float* probs,
                             hipStream_t stream) {
  CXRK_CHECK_ARG(qkv && ctxv && B > 0 && nH > 0 && aligned16(qkv) && aligned16(ctxv) && ctxplane >= 0 && (ctxplane % 4) == 0);
  float* ctx = ctxplane ? nullptr : static_cast<float*>(ctxv);
  unsigned short* ctxp = ctxplane ? static_cast<unsigned short*>(ctxv) : nullptr;
  if (dH > AD || dH < 4 || (dH % 4) != 0 || L > ALONG || L < 1) return CXRK_ERR_UNSUPPORTED;
  const int ALD = dH + 4;
  if (L > AL) {   // tiled form: one workgroup per (sequence, head, 32 queries)
    const size_t shl = (size_t)((TQ + TK) * ALD + TQ * (L + 1)) * sizeof(float);
    static bool attr_long = false;
    if (!attr_long) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_long_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(((TQ + TK) * (AD + 4) + TQ * (ALONG + 1)) * sizeof(float)));
      attr_long = true;
    }
    hipLaunchKernelGGL(attn_fwd_long_kernel, dim3((unsigned)(B * nH * ceil_div(L, TQ))), dim3(256), shl, stream, qkv, mask, L, nH, dH,
                       1.0f / sqrtf((float)dH), ctx, ctxp, ctxplane, probs);
    CXRK_LAUNCH_CHECK();
    return CXRK_OK;
  }
  const size_t sh = (size_t)(3 * L * ALD + L * (L + 1)) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)((3 * AL * (AD + 4) + AL * (AL + 1)) * sizeof(float)));
    attr_set = true;
  }
  hipLaunchKernelGGL(attn_fwd_kernel, dim3((unsigned)(B * nH)), dim3(256), sh, stream, qkv, mask, L, nH, dH,
                     1.0f / sqrtf((float)dH), ctx, ctxp, ctxplane, probs);
  CXRK_LAUNCH_CHECK();
  return CXRK_OK;
}

// workspace of cxrk_attn_bwd: the dS matrix of the tiled form (L > 64); nothing for short sequences
extern "C" size_t cxrk_attn_bwd_ws_bytes(int B, int L, int nH, int dH) {
  (void)dH;
  return L > AL ? (size_t)B * nH * L * L * sizeof(float) : 0;
}

extern "C" int cxrk_attn_bwd(const float* qkv, const float* probs, const float* dctx, int B, int L, int nH, int dH,
                             void* dqkvv, long dqkvplane, float* ws, size_t ws_bytes, hipStream_t stream) {
  CXRK_CHECK_ARG(qkv && probs && dctx && dqkvv && B > 0 && nH > 0 && aligned16(qkv) && aligned16(dctx) && aligned16(dqkvv) && dqkvplane >= 0 &&
                 (dqkvplane % 4) == 0);
  float* dqkv = dqkvplane ? nullptr : static_cast<float*>(dqkvv);
  unsigned short* dqkvp = dqkvplane ? static_cast<unsigned short*>(dqkvv) : nullptr;
  if (dH > AD || dH < 4 || (dH % 4) != 0 || L > ALONG || L < 1) return CXRK_ERR_UNSUPPORTED;
  const int ALD = dH + 4;
  if (L > AL) {
    if (ws == nullptr || ws_bytes < cxrk_attn_bwd_ws_bytes(B, L, nH, dH)) return CXRK_ERR_WS;
    const float scale = 1.0f / sqrtf((float)dH);
    const size_t shq = (size_t)((TQ + 2 * TK) * ALD + TQ * (TK + 1)) * sizeof(float);
    const size_t shk = (size_t)(2 * TQ * ALD + 2 * TQ * (TK + 1)) * sizeof(float);
    hipLaunchKernelGGL(attn_bwd_long_q_kernel, dim3((unsigned)(B * nH * ceil_div(L, TQ))), dim3(256), shq, stream, qkv, probs, dctx, L, nH,
                       dH, scale, ws, dqkv, dqkvp, dqkvplane);
    CXRK_LAUNCH_CHECK();
    hipLaunchKernelGGL(attn_bwd_long_kv_kernel, dim3((unsigned)(B * nH * ceil_div(L, TK))), dim3(256), shk, stream, qkv, probs, ws, dctx, L,
                       nH, dH, dqkv, dqkvp, dqkvplane);
    CXRK_LAUNCH_CHECK();
    return CXRK_OK;
  }
  const size_t sh = (size_t)(4 * L * ALD + 2 * L * (L + 1)) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)((4 * AL * (AD + 4) + 2 * AL * (AL + 1)) * sizeof(float)));
    attr_set = true;
  }
  hipLaunchKernelGGL(attn_bwd_kernel, dim3((unsigned)(B * nH)), dim3(256), sh, stream, qkv, probs, dctx, L, nH, dH,
                     1.0f / sqrtf((float)dH), dqkv, dqkvp, dqkvplane);
  CXRK_LAUNCH_CHECK();
  return CXRK_OK;
}
