// Train-mode BatchNorm2d around the convolution kernels: `ImageModel.train()`, the mode the reference's constructor leaves the image
// model in (health_multimodal/image/model/model.py:119; torchvision Bottleneck / resnet.py:34-47 / modules.py:43-46 BatchNorm2d with
// training=True).  Forward: z = conv(x, w) (raw, written by the GEMM kernels), batch statistics over all pixels in one pass
// (cxrk_colstats), y = relu(gamma (z - mean) rstd + beta + residual) here, running statistics updated with the unbiased variance.
// Backward: dz = gamma rstd (dy - mean(dy) - xhat mean(dy xhat)), dgamma = sum dy xhat, dbeta = sum dy; dz then goes through the
// same data- and weight-gradient GEMMs as in eval mode (with unscaled filters).  Tensors are fp32 (plane == 0) or split-bf16
// planes (plane > 0: hi at the pointer, lo `plane` elements behind), [rows = pixels][C] with C % 8 == 0; all reductions two-stage,
// fixed order (no atomics).
#include "cxrk.h"
#include "cxrk_common.h"

using namespace cxrk;

namespace {
__device__ __forceinline__ void ld8(const void* p, long plane, long off, float (&v)[8]) {
  if (plane) { planes_load8(static_cast<const unsigned short*>(p), plane, off, v); return; }
  const float* f = static_cast<const float*>(p) + off;
  const float4 a = *reinterpret_cast<const float4*>(f), b = *reinterpret_cast<const float4*>(f + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void st8(void* p, long plane, long off, const float (&v)[8]) {
  if (plane) { planes_store8(static_cast<unsigned short*>(p), plane, off, v); return; }
  float* f = static_cast<float*>(p) + off;
  *reinterpret_cast<float4*>(f) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(f + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

// forward coefficients from the batch statistics (var = biased variance) + running-statistics update (momentum; unbiased variance)
__global__ void bn_train_fwd_coeffs_kernel(const float* __restrict__ mean, const float* __restrict__ var, const float* __restrict__ gamma,
                                           const float* __restrict__ beta, float eps, float n, float momentum, float* __restrict__ scale,
                                           float* __restrict__ shift, float* __restrict__ rstd, float* __restrict__ rmean,
                                           float* __restrict__ rvar, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float rs = 1.0f / sqrtf(var[c] + eps);
  const float sc = gamma[c] * rs;
  scale[c] = sc; shift[c] = beta[c] - mean[c] * sc; rstd[c] = rs;
  if (rmean) {
    rmean[c] = (1.0f - momentum) * rmean[c] + momentum * mean[c];
    rvar[c] = (1.0f - momentum) * rvar[c] + momentum * var[c] * (n > 1.0f ? n / (n - 1.0f) : 1.0f);
  }
}

// y = relu?(z * scale + shift + residual?), ReLU decision bits (byte [row][c / 8], bit c % 8) when mask != null
__global__ __launch_bounds__(256) void bn_apply_kernel(const void* __restrict__ z, long zplane, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const void* __restrict__ res, long rplane,
                                                       void* __restrict__ y, long yplane, unsigned char* __restrict__ mask, long rows, int C,
                                                       int relu) {
  const int C8 = C / 8;
  const long total = rows * C8;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int c = (int)(t % C8) * 8;
    float v[8]; ld8(z, zplane, t * 8, v);
    const float4 s0 = *reinterpret_cast<const float4*>(scale + c), s1 = *reinterpret_cast<const float4*>(scale + c + 4);
    const float4 h0 = *reinterpret_cast<const float4*>(shift + c), h1 = *reinterpret_cast<const float4*>(shift + c + 4);
    const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = fmaf(v[q], sc[q], sh[q]);
    if (res) {
      float r[8]; ld8(res, rplane, t * 8, r);
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] += r[q];
    }
    unsigned bits = 0;
    if (relu) {
#pragma unroll
      for (int q = 0; q < 8; ++q) { bits |= (v[q] > 0.f ? 1u : 0u) << q; v[q] = fmaxf(v[q], 0.f); }
    }
    st8(y, yplane, t * 8, v);
    if (mask) mask[t] = (unsigned char)bits;
  }
}

// Column reductions over [rows][C] (C % 8 == 0).  Lane map of the partial kernels: a block covers 8 * cqn columns (cqn = 1 << cql
// column lanes of 8 columns each, cqn = min(32, C / 8 rounded up to a power of two)) and 256 / cqn row lanes, so that narrow
// tensors (C = 64: 8 column lanes x 32 row lanes) still use every thread; block (bx, by) takes rows [by * rows_per, ...).
struct ColMap {
  int cq, rl, rln, col;
  __device__ __forceinline__ ColMap(int cql, int C) {
    const int cqn = 1 << cql;
    cq = threadIdx.x & (cqn - 1); rl = threadIdx.x >> cql; rln = 256 >> cql;
    col = (blockIdx.x * cqn + cq) * 8;
  }
};

// Batch statistics in ONE pass over z: a thread accumulates d = z - s and d^2 against the shift s = z[first row of the block] (so
// the squares do not cancel against a large mean); part[by] = (block mean, block M2 = sum (z - block mean)^2).  The final kernel
// merges the blocks with Chan's formula: n, mean, M2 <- n + nb, mean + delta nb / (n + nb), M2 + M2b + delta^2 n nb / (n + nb).
__global__ __launch_bounds__(256) void colstats_partial_kernel(const void* __restrict__ x, long plane, long rows, int C, int rows_per, int cql,
                                                               float* __restrict__ part_mean, float* __restrict__ part_m2) {
  __shared__ float sh[2][256][8];
  const ColMap m(cql, C);
  const long r0 = (long)blockIdx.y * rows_per, r1 = min(rows, r0 + rows_per);
  float s1[8], s2[8], sft[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { s1[q] = 0.f; s2[q] = 0.f; sft[q] = 0.f; }
  if (m.col < C) {
    ld8(x, plane, r0 * C + m.col, sft);
    for (long r = r0 + m.rl; r < r1; r += m.rln) {
      float v[8]; ld8(x, plane, r * C + m.col, v);
#pragma unroll
      for (int q = 0; q < 8; ++q) { const float d = v[q] - sft[q]; s1[q] += d; s2[q] = fmaf(d, d, s2[q]); }
    }
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) { sh[0][threadIdx.x][q] = s1[q]; sh[1][threadIdx.x][q] = s2[q]; }
  __syncthreads();
  if (m.rl == 0 && m.col < C) {
    const float nb = (float)(r1 - r0);
    const int cqn = 1 << cql;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float a = s1[q], b = s2[q];
      for (int i = 1; i < m.rln; ++i) { a += sh[0][i * cqn + m.cq][q]; b += sh[1][i * cqn + m.cq][q]; }
      part_mean[(long)blockIdx.y * C + m.col + q] = sft[q] + a / nb;
      part_m2[(long)blockIdx.y * C + m.col + q] = b - a * a / nb;
    }
  }
}
// 32 columns x 8 part lanes per block: part lane j merges the blocks p = j, j + 8, ... in ascending order, lane 0 then merges the
// eight lane results in ascending order (fixed order: reproducible)
__global__ __launch_bounds__(256) void colstats_final_kernel(const float* __restrict__ part_mean, const float* __restrict__ part_m2, int nparts,
                                                             long rows, int rows_per, int C, float* __restrict__ mean, float* __restrict__ var,
                                                             float var_scale) {
  __shared__ float sh[3][8][32];
  const int cl = threadIdx.x & 31, j = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float n = 0.f, mu = 0.f, m2 = 0.f;
  if (c < C)
    for (int p = j; p < nparts; p += 8) {
      const float nb = (float)(min(rows, (long)(p + 1) * rows_per) - (long)p * rows_per);
      const float delta = part_mean[(long)p * C + c] - mu;
      const float tot = n + nb, w = nb / tot;
      mu = fmaf(delta, w, mu);
      m2 += part_m2[(long)p * C + c] + delta * delta * n * w;
      n = tot;
    }
  sh[0][j][cl] = n; sh[1][j][cl] = mu; sh[2][j][cl] = m2;
  __syncthreads();
  if (j == 0 && c < C) {
    for (int i = 1; i < 8; ++i) {
      const float nb = sh[0][i][cl];
      if (nb == 0.f) continue;
      const float delta = sh[1][i][cl] - mu;
      const float tot = n + nb, w = nb / tot;
      mu = fmaf(delta, w, mu);
      m2 += sh[2][i][cl] + delta * delta * n * w;
      n = tot;
    }
    mean[c] = mu;
    var[c] = m2 * var_scale;
  }
}

// part[blockIdx.y][c] = sum over the block's rows of a[r][c] * (b[r][c] - bshift[c]) (bshift == null: 0)
__global__ __launch_bounds__(256) void coldot_partial_kernel(const void* __restrict__ a, long aplane, const void* __restrict__ b, long bplane,
                                                             const float* __restrict__ bshift, long rows, int C, int rows_per, int cql,
                                                             float* __restrict__ part) {
  __shared__ float sh[256][8];
  const ColMap m(cql, C);
  const long r0 = (long)blockIdx.y * rows_per, r1 = min(rows, r0 + rows_per);
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, sft[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (m.col < C) {
    if (bshift) ld8(bshift, 0, m.col, sft);
    for (long r = r0 + m.rl; r < r1; r += m.rln) {
      float x[8], w[8];
      ld8(a, aplane, r * C + m.col, x); ld8(b, bplane, r * C + m.col, w);
#pragma unroll
      for (int q = 0; q < 8; ++q) s[q] = fmaf(x[q], w[q] - sft[q], s[q]);
    }
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) sh[threadIdx.x][q] = s[q];
  __syncthreads();
  if (m.rl == 0 && m.col < C) {
    const int cqn = 1 << cql;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float t = s[q];
      for (int i = 1; i < m.rln; ++i) t += sh[i * cqn + m.cq][q];
      part[(long)blockIdx.y * C + m.col + q] = t;
    }
  }
}
// 32 columns x 8 part lanes per block, fixed order
__global__ __launch_bounds__(256) void colpart_final_kernel(const float* __restrict__ part, int nparts, int C, float* __restrict__ out) {
  __shared__ float sh[8][32];
  const int cl = threadIdx.x & 31, j = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float s = 0.f;
  if (c < C)
    for (int p = j; p < nparts; p += 8) s += part[(long)p * C + c];
  sh[j][cl] = s;
  __syncthreads();
  if (j == 0 && c < C) {
    for (int i = 1; i < 8; ++i) s += sh[i][cl];
    out[c] = s;
  }
}

// backward coefficients: dz = A * dy + B + Cc * z, and the parameter gradients
//   dbeta = sum dy, dgamma = sum dy * xhat = rstd * dot with dot = sum dy (z - mean) (cxrk_coldot with bshift = mean: the products are
//   centred before they are summed, so a channel whose mean is large against its spread loses no digits)
//   dz = gamma rstd (dy - dbeta / n - (z - mean) rstd dgamma / n)
__global__ void bn_train_bwd_coeffs_kernel(const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ rstd,
                                           const float* __restrict__ sumdy, const float* __restrict__ dot, float n, float* __restrict__ A,
                                           float* __restrict__ B, float* __restrict__ Cc, float* __restrict__ dgamma,
                                           float* __restrict__ dbeta, int accumulate, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float db = sumdy[c];
  const float dg = rstd[c] * dot[c];
  const float a = gamma[c] * rstd[c];
  const float cc = -a * rstd[c] * dg / n;
  A[c] = a; Cc[c] = cc; B[c] = -a * db / n - cc * mean[c];
  dgamma[c] = accumulate ? dgamma[c] + dg : dg;
  dbeta[c] = accumulate ? dbeta[c] + db : db;
}
__global__ __launch_bounds__(256) void bn_train_dz_kernel(const void* __restrict__ dy, long dyplane, const void* __restrict__ z, long zplane,
                                                          const float* __restrict__ A, const float* __restrict__ B,
                                                          const float* __restrict__ Cc, void* __restrict__ dz, long dzplane, long rows, int C) {
  const int C8 = C / 8;
  const long total = rows * C8;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int c = (int)(t % C8) * 8;
    float g[8], x[8];
    ld8(dy, dyplane, t * 8, g); ld8(z, zplane, t * 8, x);
#pragma unroll
    for (int q = 0; q < 8; ++q) g[q] = fmaf(A[c + q], g[q], fmaf(Cc[c + q], x[q], B[c + q]));
    st8(dz, dzplane, t * 8, g);
  }
}

inline unsigned grid_for(long work) { long nb = (work + 255) / 256; if (nb > 16384) nb = 16384; if (nb < 1) nb = 1; return (unsigned)nb; }
inline bool fmt_ok(const void* p, long plane) { return p && aligned16(p) && plane >= 0 && (plane % 8) == 0; }
}  // namespace

extern "C" int cxrk_bn_train_fwd_coeffs(const float* mean, const float* var, const float* gamma, const float* beta, float eps, long n,
                                        float momentum, float* scale, float* shift, float* rstd, float* rmean, float* rvar, int C,
                                        hipStream_t stream) {
  CXRK_CHECK_ARG(mean && var && gamma && beta && scale && shift && rstd && C > 0 && n > 0 && ((rmean == nullptr) == (rvar == nullptr)));
  hipLaunchKernelGGL(bn_train_fwd_coeffs_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, stream, mean, var, gamma, beta, eps, (float)n, momentum,
                     scale, shift, rstd, rmean, rvar, C);
  CXRK_LAUNCH_CHECK();
  return CXRK_OK;
}

extern "C" int cxrk_bn_apply(const void* z, long zplane, const float* scale, const float* shift, const void* res, long rplane, void* y,
                             long yplane, unsigned char* mask, long rows, int C, int relu, hipStream_t stream) {
  CXRK_CHECK_ARG(fmt_ok(z, zplane) && fmt_ok(y, yplane) && scale && shift && rows > 0 && C > 0 && (C % 8) == 0 && aligned16(scale) && aligned16(shift));
  CXRK_CHECK_ARG((res == nullptr || fmt_ok(res, rplane)) && (mask == nullptr || relu));
  hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(rows * (C / 8))), dim3(256), 0, stream, z, zplane, scale, shift, res, rplane, y, yplane, mask,
                     rows, C, relu);
  CXRK_LAUNCH_CHECK();
  return CXRK_OK;
}

// mean[c] = mean over the rows of x[r][c]; var[c] = var_scale * sum_r (x[r][c] - mean[c])^2 (var_scale = 1 / rows: biased variance;
// 1 / (rows - 1): unbiased), one pass over x; ws: 2 * cxrk_coldot_ws_bytes(rows, C)
extern "C" int cxrk_colstats(const void* x, long plane, long rows, int C, float* mean, float* var, float var_scale, float* ws, size_t ws_bytes,
                             hipStream_t stream);
// column lanes (log2) of a block and the number of row blocks: ~2048 blocks over the chip, >= 512 rows per block
static int col_lanes_log2(int C) { int l = 0; while (l < 5 && (8 << l) < C) ++l; return l; }
static int coldot_parts(long rows, int C) {
  const int colblocks = ceil_div(C, 8 << col_lanes_log2(C));
  long cap = 2048 / colblocks; if (cap < 64) cap = 64;
  long np = (rows + 511) / 512; if (np > cap) np = cap; if (np < 1) np = 1;
  return (int)np;
}
extern "C" size_t cxrk_coldot_ws_bytes(long rows, int C) { return (size_t)coldot_parts(rows, C) * (size_t)C * sizeof(float); }
extern "C" int cxrk_coldot(const void* a, long aplane, const void* b, long bplane, const float* bshift, long rows, int C, float* out, float* ws,
                           size_t ws_bytes, hipStream_t stream) {
  CXRK_CHECK_ARG(fmt_ok(a, aplane) && fmt_ok(b, bplane) && out && rows > 0 && C > 0 && (C % 8) == 0);
  int np = coldot_parts(rows, C);
  const int cql = col_lanes_log2(C);
  if (ws == nullptr || ws_bytes < (size_t)np * C * sizeof(float)) return CXRK_ERR_WS;
  const int rows_per = (int)((rows + np - 1) / np);
  np = (int)((rows + rows_per - 1) / rows_per);
  hipLaunchKernelGGL(coldot_partial_kernel, dim3(ceil_div(C, 8 << cql), np), dim3(256), 0, stream, a, aplane, b, bplane, bshift, rows, C, rows_per, cql,
                     ws);
  CXRK_LAUNCH_CHECK();
  hipLaunchKernelGGL(colpart_final_kernel, dim3(ceil_div(C, 32)), dim3(256), 0, stream, ws, np, C, out);
  CXRK_LAUNCH_CHECK();
  return CXRK_OK;
}

extern "C" int cxrk_bn_train_bwd_coeffs(const float* gamma, const float* mean, const float* rstd, const float* sumdy, const float* dot, long n,
                                        float* A, float* B, float* Cc, float* dgamma, float* dbeta, int accumulate, int C, hipStream_t stream) {
  CXRK_CHECK_ARG(gamma && mean && rstd && sumdy && dot && A && B && Cc && dgamma && dbeta && C > 0 && n > 0);
  hipLaunchKernelGGL(bn_train_bwd_coeffs_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, stream, gamma, mean, rstd, sumdy, dot, (float)n, A, B, Cc,
                     dgamma, dbeta, accumulate, C);
  CXRK_LAUNCH_CHECK();
  return CXRK_OK;
}

extern "C" int cxrk_bn_train_dz(const void* dy, long dyplane, const void* z, long zplane, const float* A, const float* B, const float* Cc, void* dz,
                                long dzplane, long rows, int C, hipStream_t stream) {
  CXRK_CHECK_ARG(fmt_ok(dy, dyplane) && fmt_ok(z, zplane) && fmt_ok(dz, dzplane) && A && B && Cc && rows > 0 && C > 0 && (C % 8) == 0);
  hipLaunchKernelGGL(bn_train_dz_kernel, dim3(grid_for(rows * (C / 8))), dim3(256), 0, stream, dy, dyplane, z, zplane, A, B, Cc, dz, dzplane, rows, C);
  CXRK_LAUNCH_CHECK();
  return CXRK_OK;
}

extern "C" int cxrk_colstats(const void* x, long plane, long rows, int C, float* mean, float* var, float var_scale, float* ws, size_t ws_bytes,
                             hipStream_t stream) {
  CXRK_CHECK_ARG(fmt_ok(x, plane) && mean && var && rows > 0 && C > 0 && (C % 8) == 0);
  int np = coldot_parts(rows, C);
  const int cql = col_lanes_log2(C);
  if (ws == nullptr || ws_bytes < 2 * (size_t)np * C * sizeof(float)) return CXRK_ERR_WS;
  const int rows_per = (int)((rows + np - 1) / np);
  np = (int)((rows + rows_per - 1) / rows_per);
  float* pm = ws; float* p2 = ws + (size_t)np * C;
  hipLaunchKernelGGL(colstats_partial_kernel, dim3(ceil_div(C, 8 << cql), np), dim3(256), 0, stream, x, plane, rows, C, rows_per, cql, pm, p2);
  CXRK_LAUNCH_CHECK();
  hipLaunchKernelGGL(colstats_final_kernel, dim3(ceil_div(C, 32)), dim3(256), 0, stream, pm, p2, np, rows, rows_per, C, mean, var, var_scale);
  CXRK_LAUNCH_CHECK();
  return CXRK_OK;
}
