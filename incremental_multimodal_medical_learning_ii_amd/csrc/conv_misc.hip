// ResNet-50 image-encoder kernels that are not GEMMs: eval-mode BatchNorm folding, NCHW <-> NHWC boundary transforms, max-pool,
// spatial mean (fp32 and split-bf16 planes storage).  Reference: resnet.py:34-47 (stem, maxpool), model.py:141-154, modules.py:29-47.
#include "conv_common.h"

using namespace cxrk;

namespace {
// w_scaled[ko][tap][c<Cpad] = w[ko][tap][c] * gamma[ko]*rsqrt(var[ko]+eps)  (0 for c >= C); fp32 (ws) or split-bf16 planes (wp)
__global__ void bn_fold_kernel(const float* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ rmean, const float* __restrict__ rvar, float eps, int Ko, int taps,
                               int C, int Cpad, float* __restrict__ ws, unsigned short* __restrict__ wp, long wplane,
                               float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ rstd) {
  const int ko = blockIdx.x;
  const float rs = 1.0f / sqrtf(rvar[ko] + eps);
  const float sc = gamma[ko] * rs;
  if (threadIdx.x == 0) { scale[ko] = sc; shift[ko] = beta[ko] - rmean[ko] * sc; rstd[ko] = rs; }
  const int n = taps * Cpad;
  if (wp) {   // Cpad % 8 == 0 (checked by the caller): 8 elements of one tap per thread
    for (int i8 = threadIdx.x; i8 < n / 8; i8 += blockDim.x) {
      const int i = i8 * 8, tap = i / Cpad, c = i - tap * Cpad;
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = c + q < C ? w[((long)ko * taps + tap) * C + c + q] * sc : 0.f;
      planes_store8(wp, wplane, (long)ko * n + i, v);
    }
    return;
  }
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int tap = i / Cpad, c = i - tap * Cpad;
    ws[(long)ko * n + i] = c < C ? w[((long)ko * taps + tap) * C + c] * sc : 0.f;
  }
}

// x[N][C][H][W] -> y[N][H][W][Cpad] (zero channel padding).  One block per (n, h): a W x C tile through LDS.
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int H, int W, int Cpad) {
  const long nh = blockIdx.x;  // n*H + h
  const long n = nh / H; const int hh = (int)(nh - n * H);
  for (int i = threadIdx.x; i < W * Cpad; i += blockDim.x) {
    const int wv = i / Cpad, c = i - wv * Cpad;
    y[(nh * W + wv) * Cpad + c] = c < C ? x[((n * C + c) * H + hh) * W + wv] : 0.f;
  }
}
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int H, int W) {
  const long nh = blockIdx.x;
  const long n = nh / H; const int hh = (int)(nh - n * H);
  for (int i = threadIdx.x; i < W * C; i += blockDim.x) {
    const int c = i / W, wv = i - c * W;
    y[((n * C + c) * H + hh) * W + wv] = x[(nh * W + wv) * C + c];
  }
}

// 3x3 / stride 2 / pad 1 max-pool, NHWC, 4 channels per thread.  idx = winning tap (first maximum in scan order,
// as torch's max_pool2d backward routes the gradient).
__global__ void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ idx,
                                   int N, int H, int W, int C, int Ho, int Wo) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int C4 = C / 4;
  const long total = (long)N * Ho * Wo * C4;
  if (t >= total) return;
  const int c4 = (int)(t % C4); long p = t / C4;
  const int wo = (int)(p % Wo); p /= Wo; const int ho = (int)(p % Ho); const long n = p / Ho;
  float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
  uchar4 bi = make_uchar4(0, 0, 0, 0);
  bool first = true;
  for (int r = 0; r < 3; ++r) {
    const int hi = ho * 2 - 1 + r;
    if ((unsigned)hi >= (unsigned)H) continue;
    for (int s = 0; s < 3; ++s) {
      const int wi = wo * 2 - 1 + s;
      if ((unsigned)wi >= (unsigned)W) continue;
      const float4 v = *reinterpret_cast<const float4*>(x + ((n * H + hi) * W + wi) * C + c4 * 4);
      const unsigned char tap = (unsigned char)(r * 3 + s);
      if (first || v.x > best.x) { best.x = v.x; bi.x = tap; }
      if (first || v.y > best.y) { best.y = v.y; bi.y = tap; }
      if (first || v.z > best.z) { best.z = v.z; bi.z = tap; }
      if (first || v.w > best.w) { best.w = v.w; bi.w = tap; }
      first = false;
    }
  }
  *reinterpret_cast<float4*>(y + t * 4) = best;
  *reinterpret_cast<uchar4*>(idx + t * 4) = bi;
}

// dx[n][hi][wi][c] = (x > 0) * sum over the <=4 windows covering (hi,wi) whose winning tap is this pixel.
__global__ void maxpool_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ idx,
                                   const float* __restrict__ x, float* __restrict__ dx, int N, int H, int W, int C, int Ho,
                                   int Wo, int relu_mask) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int C4 = C / 4;
  const long total = (long)N * H * W * C4;
  if (t >= total) return;
  const int c4 = (int)(t % C4); long p = t / C4;
  const int wi = (int)(p % W); p /= W; const int hi = (int)(p % H); const long n = p / H;
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int r = 0; r < 3; ++r) {
    const int hn = hi + 1 - r;
    if (hn < 0 || (hn & 1)) continue;
    const int ho = hn >> 1;
    if (ho >= Ho) continue;
    for (int s = 0; s < 3; ++s) {
      const int wn = wi + 1 - s;
      if (wn < 0 || (wn & 1)) continue;
      const int wo = wn >> 1;
      if (wo >= Wo) continue;
      const long o = (((n * Ho + ho) * Wo + wo) * C4 + c4) * 4;
      const uchar4 bi = *reinterpret_cast<const uchar4*>(idx + o);
      const float4 d = *reinterpret_cast<const float4*>(dy + o);
      const unsigned char tap = (unsigned char)(r * 3 + s);
      if (bi.x == tap) g.x += d.x;
      if (bi.y == tap) g.y += d.y;
      if (bi.z == tap) g.z += d.z;
      if (bi.w == tap) g.w += d.w;
    }
  }
  if (relu_mask) {
    const float4 xv = *reinterpret_cast<const float4*>(x + t * 4);
    g.x = xv.x > 0.f ? g.x : 0.f; g.y = xv.y > 0.f ? g.y : 0.f; g.z = xv.z > 0.f ? g.z : 0.f; g.w = xv.w > 0.f ? g.w : 0.f;
  }
  *reinterpret_cast<float4*>(dx + t * 4) = g;
}

// y[n][c] = mean_p x[n][p][c]
__global__ void spatial_mean_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int P, int C) {
  const int n = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float s = 0.f;
    for (int p = 0; p < P; ++p) s += x[((long)n * P + p) * C + c];
    y[(long)n * C + c] = s / (float)P;
  }
}
__global__ void spatial_mean_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int P, int C) {
  const int n = blockIdx.x;
  const float inv = 1.0f / (float)P;
  for (int i = threadIdx.x; i < P * C; i += blockDim.x) dx[(long)n * P * C + i] = dy[(long)n * C + (i % C)] * inv;
}

// ---- planes variants (split-bf16 storage) of the pooling kernels: 8 channels per thread -------------------------------
__global__ void maxpool_fwd_pl_kernel(const unsigned short* __restrict__ x, long xplane, unsigned short* __restrict__ y, long yplane,
                                      unsigned char* __restrict__ idx, int N, int H, int W, int C, int Ho, int Wo) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int C8 = C / 8;
  const long total = (long)N * Ho * Wo * C8;
  if (t >= total) return;
  const int c8 = (int)(t % C8); long p = t / C8;
  const int wo = (int)(p % Wo); p /= Wo; const int ho = (int)(p % Ho); const long n = p / Ho;
  float best[8]; unsigned char bi[8];
  bool first = true;
  for (int r = 0; r < 3; ++r) {
    const int hi = ho * 2 - 1 + r;
    if ((unsigned)hi >= (unsigned)H) continue;
    for (int s = 0; s < 3; ++s) {
      const int wi = wo * 2 - 1 + s;
      if ((unsigned)wi >= (unsigned)W) continue;
      float v[8]; planes_load8(x, xplane, ((n * H + hi) * W + wi) * C + c8 * 8, v);
      const unsigned char tap = (unsigned char)(r * 3 + s);
#pragma unroll
      for (int q = 0; q < 8; ++q) if (first || v[q] > best[q]) { best[q] = v[q]; bi[q] = tap; }
      first = false;
    }
  }
  planes_store8(y, yplane, t * 8, best);   // the maximum of values of the form hi + lo is again exactly hi + lo
  *reinterpret_cast<uint2*>(idx + t * 8) = make_uint2(bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24),
                                                      bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24));
}
// dx (fp32: it feeds the stem's exact-fp32 weight gradient) = sum over the <= 4 windows whose winning tap is this pixel, masked
// by the stem ReLU.  The pooled value IS the winning input, so (pooled > 0) is that input's ReLU decision: no second tensor.
// A thread owns the 2 x 2 input pixels (2a + i, 2b + j) of 8 channels: they lie in the four windows (a | a + 1, b | b + 1) only
// (3 x 3, stride 2, pad 1), which are loaded once (a thread per pixel loaded 9 windows for the same four pixels); each pixel's
// sum runs over its windows in ascending tap order (r, s), as the per-pixel form did.
struct PoolWin { float d[8]; unsigned tap[8]; };   // gradient (0 where the stem ReLU was off) and winning tap per channel
__device__ __forceinline__ void pool_win_load(PoolWin& w, bool valid, const unsigned short* __restrict__ dy, long dyplane,
                                              const unsigned char* __restrict__ idx, const unsigned short* __restrict__ pooled, long o) {
  if (!valid) {
#pragma unroll
    for (int q = 0; q < 8; ++q) { w.d[q] = 0.f; w.tap[q] = 0xffu; }
    return;
  }
  const uint2 bw = *reinterpret_cast<const uint2*>(idx + o);
  const uint4 ph = *reinterpret_cast<const uint4*>(pooled + o);   // hi plane: sign(hi) = sign(value)
  planes_load8(dy, dyplane, o, w.d);
  const unsigned pw[4] = {ph.x, ph.y, ph.z, ph.w};
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    w.tap[q] = ((q < 4 ? bw.x : bw.y) >> (8 * (q & 3))) & 0xffu;
    const float pv = __builtin_bit_cast(float, (q & 1) ? (pw[q >> 1] & 0xffff0000u) : (pw[q >> 1] << 16));
    if (!(pv > 0.f)) w.tap[q] = 0xffu;
  }
}
__device__ __forceinline__ void pool_acc(float (&g)[8], const PoolWin& w, unsigned tap) {
#pragma unroll
  for (int q = 0; q < 8; ++q) if (w.tap[q] == tap) g[q] += w.d[q];
}
__device__ __forceinline__ void pool_store(float* __restrict__ dx, long e, const float (&g)[8]) {
  *reinterpret_cast<float4*>(dx + e) = make_float4(g[0], g[1], g[2], g[3]);
  *reinterpret_cast<float4*>(dx + e + 4) = make_float4(g[4], g[5], g[6], g[7]);
}
__global__ __launch_bounds__(256) void maxpool_bwd_pl_kernel(const unsigned short* __restrict__ dy, long dyplane, const unsigned char* __restrict__ idx,
                                                             const unsigned short* __restrict__ pooled, float* __restrict__ dx, int N, int H,
                                                             int W, int C, int Ho, int Wo) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int C8 = C / 8, H2 = (H + 1) / 2, W2 = (W + 1) / 2;
  const long total = (long)N * H2 * W2 * C8;
  if (t >= total) return;
  const int c8 = (int)(t % C8); long p = t / C8;
  const int b = (int)(p % W2); p /= W2; const int a = (int)(p % H2); const long n = p / H2;
  PoolWin w00, w01, w10, w11;     // windows (a, b), (a, b + 1), (a + 1, b), (a + 1, b + 1)
  const bool va = a < Ho, va1 = a + 1 < Ho, vb = b < Wo, vb1 = b + 1 < Wo;
  const long o00 = (((n * Ho + a) * Wo + b) * C8 + c8) * 8;
  pool_win_load(w00, va && vb, dy, dyplane, idx, pooled, o00);
  pool_win_load(w01, va && vb1, dy, dyplane, idx, pooled, o00 + (long)C);
  pool_win_load(w10, va1 && vb, dy, dyplane, idx, pooled, o00 + (long)Wo * C);
  pool_win_load(w11, va1 && vb1, dy, dyplane, idx, pooled, o00 + (long)(Wo + 1) * C);
  const int hi = 2 * a, wi = 2 * b;
  const long e = (((n * H + hi) * W + wi) * C8 + c8) * 8;
  {  // (even row, even column): tap (1,1) of window (a, b)
    float g[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    pool_acc(g, w00, 4u);
    pool_store(dx, e, g);
  }
  if (wi + 1 < W) {  // (even, odd): tap (1,0) of (a, b + 1), then tap (1,2) of (a, b)
    float g[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    pool_acc(g, w01, 3u); pool_acc(g, w00, 5u);
    pool_store(dx, e + C, g);
  }
  if (hi + 1 < H) {  // (odd, even): tap (0,1) of (a + 1, b), then tap (2,1) of (a, b)
    float g[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    pool_acc(g, w10, 1u); pool_acc(g, w00, 7u);
    pool_store(dx, e + (long)W * C, g);
  }
  if (hi + 1 < H && wi + 1 < W) {  // (odd, odd): taps (0,0) of (a+1, b+1), (0,2) of (a+1, b), (2,0) of (a, b+1), (2,2) of (a, b)
    float g[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    pool_acc(g, w11, 0u); pool_acc(g, w10, 2u); pool_acc(g, w01, 6u); pool_acc(g, w00, 8u);
    pool_store(dx, e + (long)(W + 1) * C, g);
  }
}
__global__ void spatial_mean_bwd_pl_kernel(const float* __restrict__ dy, const float* __restrict__ add, unsigned short* __restrict__ dx,
                                           long dxplane, int P, int C) {
  const int n = blockIdx.x;
  const float inv = 1.0f / (float)P;
  const int C8 = C / 8;
  for (int i = threadIdx.x; i < P * C8; i += blockDim.x) {
    const int c = (i % C8) * 8;
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = dy[(long)n * C + c + q] * inv;
    const long o = ((long)n * P + i / C8) * C + c;
    if (add) {
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] += add[o + q];
    }
    planes_store8(dx, dxplane, o, v);
  }
}

}  // namespace

extern "C" int cxrk_bn_fold(const float* w, const float* gamma, const float* beta, const float* rmean, const float* rvar,
                            float eps, int Ko, int taps, int C, int Cpad, float* w_scaled, float* scale, float* shift,
                            float* rstd, hipStream_t stream) {
  CXRK_CHECK_ARG(w && gamma && beta && rmean && rvar && w_scaled && scale && shift && rstd && Ko > 0 && taps > 0 && C > 0 && Cpad >= C);
  hipLaunchKernelGGL(bn_fold_kernel, dim3(Ko), dim3(256), 0, stream, w, gamma, beta, rmean, rvar, eps, Ko, taps, C, Cpad,
                     w_scaled, (unsigned short*)nullptr, 0L, scale, shift, rstd);
  CXRK_LAUNCH_CHECK();
  return CXRK_OK;
}
extern "C" int cxrk_bn_fold_pl(const float* w, const float* gamma, const float* beta, const float* rmean, const float* rvar,
                               float eps, int Ko, int taps, int C, int Cpad, void* w_scaled, long wplane, float* scale,
                               float* shift, float* rstd, hipStream_t stream) {
  CXRK_CHECK_ARG(w && gamma && beta && rmean && rvar && w_scaled && scale && shift && rstd && Ko > 0 && taps > 0 && C > 0 && Cpad >= C);
  CXRK_CHECK_ARG((Cpad % 8) == 0 && (wplane % 8) == 0 && aligned16(w_scaled));
  hipLaunchKernelGGL(bn_fold_kernel, dim3(Ko), dim3(256), 0, stream, w, gamma, beta, rmean, rvar, eps, Ko, taps, C, Cpad,
                     (float*)nullptr, static_cast<unsigned short*>(w_scaled), wplane, scale, shift, rstd);
  CXRK_LAUNCH_CHECK();
  return CXRK_OK;
}

// ---- layout / pooling ------------------------------------------------------------------------------------------------
extern "C" int cxrk_nchw_to_nhwc(const float* x, float* y, int N, int C, int H, int W, int Cpad, hipStream_t stream) {
  CXRK_CHECK_ARG(x && y && N > 0 && C > 0 && Cpad >= C);
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((unsigned)((long)N * H)), dim3(256), 0, stream, x, y, C, H, W, Cpad);
  CXRK_LAUNCH_CHECK();
  return CXRK_OK;
}
extern "C" int cxrk_nhwc_to_nchw(const float* x, float* y, int N, int C, int H, int W, hipStream_t stream) {
  CXRK_CHECK_ARG(x && y && N > 0 && C > 0);
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)((long)N * H)), dim3(256), 0, stream, x, y, C, H, W);
  CXRK_LAUNCH_CHECK();
  return CXRK_OK;
}

extern "C" int cxrk_maxpool_fwd(const float* x, float* y, unsigned char* idx, int N, int H, int W, int C, hipStream_t stream) {
  CXRK_CHECK_ARG(x && y && idx && C % 4 == 0 && aligned16(x) && aligned16(y));
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long total = (long)N * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, x, y, idx, N, H, W, C, Ho, Wo);
  CXRK_LAUNCH_CHECK();
  return CXRK_OK;
}
extern "C" int cxrk_maxpool_bwd(const float* dy, const unsigned char* idx, const float* x, float* dx, int N, int H, int W,
                                int C, int relu_mask, hipStream_t stream) {
  CXRK_CHECK_ARG(dy && idx && dx && C % 4 == 0 && (!relu_mask || x));
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long total = (long)N * H * W * (C / 4);
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, dy, idx, x, dx, N, H, W, C,
                     Ho, Wo, relu_mask);
  CXRK_LAUNCH_CHECK();
  return CXRK_OK;
}
extern "C" int cxrk_maxpool_fwd_pl(const void* x, long xplane, void* y, long yplane, unsigned char* idx, int N, int H, int W, int C,
                                   hipStream_t stream) {
  CXRK_CHECK_ARG(x && y && idx && C % 8 == 0 && aligned16(x) && aligned16(y) && (xplane % 8) == 0 && (yplane % 8) == 0 && (((uintptr_t)idx) & 7) == 0);
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long total = (long)N * Ho * Wo * (C / 8);
  hipLaunchKernelGGL(maxpool_fwd_pl_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, static_cast<const unsigned short*>(x),
                     xplane, static_cast<unsigned short*>(y), yplane, idx, N, H, W, C, Ho, Wo);
  CXRK_LAUNCH_CHECK();
  return CXRK_OK;
}
// dy, pooled: planes [N,Ho,Wo,C] (pooled = the forward's output: its sign is the stem ReLU decision); dx: fp32 [N,H,W,C]
extern "C" int cxrk_maxpool_bwd_pl(const void* dy, long dyplane, const unsigned char* idx, const void* pooled, float* dx, int N, int H,
                                   int W, int C, hipStream_t stream) {
  CXRK_CHECK_ARG(dy && idx && pooled && dx && C % 8 == 0 && aligned16(dy) && aligned16(pooled) && aligned16(dx) && (dyplane % 8) == 0);
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long total = (long)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 8);   // one thread per 2 x 2 input pixels x 8 channels
  hipLaunchKernelGGL(maxpool_bwd_pl_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, static_cast<const unsigned short*>(dy),
                     dyplane, idx, static_cast<const unsigned short*>(pooled), dx, N, H, W, C, Ho, Wo);
  CXRK_LAUNCH_CHECK();
  return CXRK_OK;
}

extern "C" int cxrk_spatial_mean_fwd(const float* x, float* y, int N, int P, int C, hipStream_t stream) {
  CXRK_CHECK_ARG(x && y && N > 0 && P > 0 && C > 0);
  hipLaunchKernelGGL(spatial_mean_fwd_kernel, dim3(N), dim3(128), 0, stream, x, y, P, C);
  CXRK_LAUNCH_CHECK();
  return CXRK_OK;
}
extern "C" int cxrk_spatial_mean_bwd(const float* dy, float* dx, int N, int P, int C, hipStream_t stream) {
  CXRK_CHECK_ARG(dy && dx && N > 0 && P > 0 && C > 0);
  hipLaunchKernelGGL(spatial_mean_bwd_kernel, dim3(N), dim3(256), 0, stream, dy, dx, P, C);
  CXRK_LAUNCH_CHECK();
  return CXRK_OK;
}
// dx (planes [N,P,C]) = dy[n][c] / P (+ add[n][p][c], an optional fp32 gradient flowing into the same tensor)
extern "C" int cxrk_spatial_mean_bwd_pl(const float* dy, const float* add, void* dx, long dxplane, int N, int P, int C, hipStream_t stream) {
  CXRK_CHECK_ARG(dy && dx && N > 0 && P > 0 && C > 0 && (C % 8) == 0 && aligned16(dx) && (dxplane % 8) == 0);
  hipLaunchKernelGGL(spatial_mean_bwd_pl_kernel, dim3(N), dim3(256), 0, stream, dy, add, static_cast<unsigned short*>(dx), dxplane, P, C);
  CXRK_LAUNCH_CHECK();
  return CXRK_OK;
}
