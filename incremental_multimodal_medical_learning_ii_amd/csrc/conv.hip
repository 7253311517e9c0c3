// ResNet-50 image-encoder kernels: NHWC implicit-GEMM convolution (fprop / dgrad / wgrad) on the shared fp32 MFMA
// mainloop, eval-mode BatchNorm folding, the wgrad slab reduction that also produces the BN parameter gradients,
// max-pool, spatial mean and the NCHW<->NHWC boundary transforms.
//
// Reference semantics: health_multimodal/image/model/resnet.py:25-47 (stem, maxpool, layer1..4),
// torchvision ResNet-50 v1.5 Bottleneck (stride on the 3x3), model.py:141-154 and modules.py:29-47 (projector).
#include "cxrk.h"
#include "gemm_core.h"

using namespace cxrk;

namespace {

ConvGeom make_geom(int N, int H, int W, int C, int Ko, int R, int S, int stride, int pad) {
  ConvGeom g;
  g.N = N; g.H = H; g.W = W; g.C = C; g.Ko = Ko; g.R = R; g.S = S; g.stride = stride; g.pad = pad;
  g.Ho = (H + 2 * pad - R) / stride + 1;
  g.Wo = (W + 2 * pad - S) / stride + 1;
  return g;
}

// w_scaled[ko][tap][c<Cpad] = w[ko][tap][c] * gamma[ko]*rsqrt(var[ko]+eps)  (0 for c >= C)
__global__ void bn_fold_kernel(const float* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ rmean, const float* __restrict__ rvar, float eps, int Ko, int taps,
                               int C, int Cpad, float* __restrict__ ws, float* __restrict__ scale, float* __restrict__ shift,
                               float* __restrict__ rstd) {
  const int ko = blockIdx.x;
  const float rs = 1.0f / sqrtf(rvar[ko] + eps);
  const float sc = gamma[ko] * rs;
  if (threadIdx.x == 0) { scale[ko] = sc; shift[ko] = beta[ko] - rmean[ko] * sc; rstd[ko] = rs; }
  const int n = taps * Cpad;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int tap = i / Cpad, c = i - tap * Cpad;
    ws[(long)ko * n + i] = c < C ? w[((long)ko * taps + tap) * C + c] * sc : 0.f;
  }
}

// dW[ko][tap][c] = scale[ko] * sum_z slab_z[ko][tap][c<Cpad]  (channel un-padding), one block per (ko, 1024-element
// part); dbeta[ko] = sumdy[ko];  dgamma[ko] = sum dy*xhat = sumdyy[ko] / gamma[ko]  with sumdyy = sum dy*(y_bn - beta).
// Only when gamma == 0 (or no y_bn sums were supplied) the algebraically equal but badly conditioned form
// rstd*(<w[ko], dWraw[ko]> - rmean*sumdy) is evaluated (by the part-0 block, serially: it is the rare path).
// Slab sum: a block covers QPB = 256 >> sg_log2 float4 outputs of filter ko; its threads are split into SG = 1 << sg_log2
// groups that walk the slabs SG apart (4 loads in flight each), and the SG partial sums are added in group order through
// LDS (deterministic).  With one thread per output and a serial loop over up to 2048 slabs the layer-1 shapes (64x64
// filters: 16 active threads per block) took 0.5 ms for 25 MB.
__global__ __launch_bounds__(256) void wgrad_reduce_bn_kernel(const float* __restrict__ slabs, int nslab, long slab_stride,
                                                              int taps, int C, int Cpad, const float* __restrict__ w,
                                                              const float* __restrict__ scale, const float* __restrict__ rstd,
                                                              const float* __restrict__ rmean, const float* __restrict__ sumdy,
                                                              const float* __restrict__ gamma, const float* __restrict__ sumdyy,
                                                              float* __restrict__ dw, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int accumulate, int sg_log2) {
  __shared__ float sh[16];
  __shared__ float4 red[256];
  const int ko = blockIdx.x;
  const int n = taps * Cpad;
  const float sc = scale ? scale[ko] : 1.f;
  const int SG = 1 << sg_log2, QPB = 256 >> sg_log2;
  const int ql = threadIdx.x & (QPB - 1), grp = threadIdx.x >> (8 - sg_log2);
  const int i4 = (blockIdx.y * QPB + ql) * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i4 < n) {  // Cpad % 4 == 0 -> the 4 elements share a tap
    const float* src = slabs + (long)ko * n + i4;
    int z = grp;
    for (; z + 3 * SG < nslab; z += 4 * SG) {
      const float4 t0 = *reinterpret_cast<const float4*>(src + (long)z * slab_stride);
      const float4 t1 = *reinterpret_cast<const float4*>(src + (long)(z + SG) * slab_stride);
      const float4 t2 = *reinterpret_cast<const float4*>(src + (long)(z + 2 * SG) * slab_stride);
      const float4 t3 = *reinterpret_cast<const float4*>(src + (long)(z + 3 * SG) * slab_stride);
      s.x += (t0.x + t1.x) + (t2.x + t3.x); s.y += (t0.y + t1.y) + (t2.y + t3.y);
      s.z += (t0.z + t1.z) + (t2.z + t3.z); s.w += (t0.w + t1.w) + (t2.w + t3.w);
    }
    for (; z < nslab; z += SG) {
      const float4 t = *reinterpret_cast<const float4*>(src + (long)z * slab_stride);
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
  }
  if (SG > 1) {
    red[threadIdx.x] = s;
    __syncthreads();
    if (grp == 0) {
      for (int g2 = 1; g2 < SG; ++g2) { const float4 t = red[g2 * QPB + ql]; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
    }
  }
  if (i4 < n && grp == 0) {
    const int tap = i4 / Cpad, c = i4 - tap * Cpad;
    const float sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (c + q < C) {
        const long o = ((long)ko * taps + tap) * C + c + q;
        dw[o] = accumulate ? dw[o] + sc * sv[q] : sc * sv[q];
      }
    }
  }
  if (dgamma && blockIdx.y == 0) {
    float g;
    const bool direct = gamma && sumdyy && gamma[ko] != 0.f;
    if (direct) {
      g = sumdyy[ko] / gamma[ko];
    } else {
      float dot = 0.f;
      for (int i = threadIdx.x; i < n; i += blockDim.x) {
        float s = 0.f;
        for (int z = 0; z < nslab; ++z) s += slabs[(long)z * slab_stride + (long)ko * n + i];
        const int tap = i / Cpad, c = i - tap * Cpad;
        if (c < C) dot += w[((long)ko * taps + tap) * C + c] * s;
      }
      dot = block_sum(dot, sh);
      g = rstd[ko] * (dot - rmean[ko] * sumdy[ko]);
    }
    if (threadIdx.x == 0) {
      dgamma[ko] = accumulate ? dgamma[ko] + g : g;
      dbeta[ko] = accumulate ? dbeta[ko] + sumdy[ko] : sumdy[ko];
    }
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

// Per-channel reductions of the BN backward, stage 1.  Block = 4 row-lanes x 64 columns; blockIdx.y = row chunk.
//   p0[c] = sum_r dy[r][c],   p1[c] = sum_r dy[r][c] * (y[r][c] - sub[r][c] - beta[c])
__global__ __launch_bounds__(256) void bn_bwd_reduce_partial_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                                    const float* __restrict__ sub, const float* __restrict__ beta,
                                                                    long rows, int C, int rows_per, float* __restrict__ part) {
  __shared__ float sh[2][4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const long r0 = (long)blockIdx.y * rows_per, r1 = min(rows, r0 + rows_per);
  float s0 = 0.f, s1 = 0.f;
  if (c < C) {
    const float b = beta[c];
    for (long r = r0 + rl; r < r1; r += 4) {
      const float d = dy[r * C + c];
      float v = y[r * C + c] - b;
      if (sub) v -= sub[r * C + c];
      s0 += d; s1 += d * v;
    }
  }
  sh[0][rl][cl] = s0; sh[1][rl][cl] = s1;
  __syncthreads();
  if (rl == 0 && c < C) {
    part[((long)blockIdx.y * 2 + 0) * C + c] = (sh[0][0][cl] + sh[0][1][cl]) + (sh[0][2][cl] + sh[0][3][cl]);
    part[((long)blockIdx.y * 2 + 1) * C + c] = (sh[1][0][cl] + sh[1][1][cl]) + (sh[1][2][cl] + sh[1][3][cl]);
  }
}
// Same sums with 16-byte accesses (C % 4 == 0): block = 16 column quads x 16 row lanes, 4 rows (8 loads) in flight per
// thread; the 16 row lanes are added in order through LDS.  The scalar form above streams the stem's 6.6 GB at 2.2 TB/s.
__global__ __launch_bounds__(256) void bn_bwd_reduce_partial_vec_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                                        const float* __restrict__ sub, const float* __restrict__ beta,
                                                                        long rows, int C, int rows_per, float* __restrict__ part) {
  __shared__ float4 sh[2][16][16];
  const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + cq * 4;
  const long r0 = (long)blockIdx.y * rows_per, r1 = min(rows, r0 + rows_per);
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
  if (c < C) {
    const float4 b = *reinterpret_cast<const float4*>(beta + c);
    auto acc = [&](const float4& d, const float4& yy, const float4& sb) {
      s0.x += d.x; s0.y += d.y; s0.z += d.z; s0.w += d.w;
      s1.x += d.x * (yy.x - sb.x - b.x); s1.y += d.y * (yy.y - sb.y - b.y); s1.z += d.z * (yy.z - sb.z - b.z); s1.w += d.w * (yy.w - sb.w - b.w);
    };
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    long r = r0 + rl;
    for (; r + 48 < r1; r += 64) {
      float4 d[4], yy[4], sb[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        d[u] = *reinterpret_cast<const float4*>(dy + (r + 16 * u) * C + c);
        yy[u] = *reinterpret_cast<const float4*>(y + (r + 16 * u) * C + c);
        sb[u] = sub ? *reinterpret_cast<const float4*>(sub + (r + 16 * u) * C + c) : z4;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc(d[u], yy[u], sb[u]);
    }
    for (; r < r1; r += 16)
      acc(*reinterpret_cast<const float4*>(dy + r * C + c), *reinterpret_cast<const float4*>(y + r * C + c),
          sub ? *reinterpret_cast<const float4*>(sub + r * C + c) : z4);
  }
  sh[0][rl][cq] = s0; sh[1][rl][cq] = s1;
  __syncthreads();
  if (rl < 2 && c < C) {   // row lane 0 finishes sum 0, row lane 1 finishes sum 1
    float4 t = sh[rl][0][cq];
#pragma unroll
    for (int i = 1; i < 16; ++i) { const float4 a = sh[rl][i][cq]; t.x += a.x; t.y += a.y; t.z += a.z; t.w += a.w; }
    *reinterpret_cast<float4*>(part + ((long)blockIdx.y * 2 + rl) * C + c) = t;
  }
}
// stage 2: block = 64 columns x 4 part-lanes
__global__ __launch_bounds__(256) void bn_bwd_reduce_final_kernel(const float* __restrict__ part, int nparts, int C,
                                                                  float* __restrict__ sumdy, float* __restrict__ sumdyy) {
  __shared__ float sh[2][4][64];
  const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float a = 0.f, b = 0.f;
  if (c < C)
    for (int p = pl; p < nparts; p += 4) { a += part[((long)p * 2) * C + c]; b += part[((long)p * 2 + 1) * C + c]; }
  sh[0][pl][cl] = a; sh[1][pl][cl] = b;
  __syncthreads();
  if (pl == 0 && c < C) {
    sumdy[c] = (sh[0][0][cl] + sh[0][1][cl]) + (sh[0][2][cl] + sh[0][3][cl]);
    sumdyy[c] = (sh[1][0][cl] + sh[1][1][cl]) + (sh[1][2][cl] + sh[1][3][cl]);
  }
}

// out[z][k][c] = sum over the z-th chunk of parts of part[p][k][c], k = 0..2; block = 64 columns x 16 part-lanes.
// Run twice (chunks -> 1) so the long partial lists of the 56x56 layers are reduced by many blocks, deterministically.
__global__ __launch_bounds__(1024) void bn_part_final_kernel(const float* __restrict__ part, int nparts, int chunk, int C,
                                                             float* __restrict__ out) {
  __shared__ float sh[16][64];
  const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl, k = blockIdx.y, z = blockIdx.z;
  const int p0 = z * chunk, p1 = min(nparts, p0 + chunk);
  float a = 0.f;
  if (c < C)
    for (int p = p0 + pl; p < p1; p += 16) a += part[((long)p * 3 + k) * C + c];
  sh[pl][cl] = a;
  __syncthreads();
  if (pl == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += sh[i][cl];
    out[((long)z * 3 + k) * C + c] = t;
  }
}

}  // namespace

static int bn_reduce_parts(long rows, int C) {
  const int cb = ceil_div(C, 64);
  long np = 1024 / cb; if (np < 16) np = 16; if (np > 512) np = 512;
  const long maxp = (rows + 63) / 64; if (np > maxp) np = maxp;
  if (np < 1) np = 1;
  return (int)np;
}
extern "C" size_t cxrk_bn_bwd_reduce_ws_bytes(long rows, int C) { return (size_t)bn_reduce_parts(rows, C) * 2 * C * sizeof(float); }

extern "C" int cxrk_bn_bwd_reduce(const float* dy, const float* y, const float* sub, const float* beta, long rows, int C,
                                  float* sumdy, float* sumdyy, float* ws, size_t ws_bytes, hipStream_t stream) {
  CXRK_CHECK_ARG(dy && y && beta && sumdy && sumdyy && rows > 0 && C > 0);
  int np = bn_reduce_parts(rows, C);
  if (ws == nullptr || ws_bytes < (size_t)np * 2 * C * sizeof(float)) return CXRK_ERR_WS;
  const int rows_per = (int)((rows + np - 1) / np);
  np = (int)((rows + rows_per - 1) / rows_per);
  if (C % 4 == 0 && aligned16(dy) && aligned16(y) && aligned16(beta) && aligned16(ws) && (!sub || aligned16(sub)))
    hipLaunchKernelGGL(bn_bwd_reduce_partial_vec_kernel, dim3(ceil_div(C, 64), np), dim3(256), 0, stream, dy, y, sub, beta, rows, C,
                       rows_per, ws);
  else
    hipLaunchKernelGGL(bn_bwd_reduce_partial_kernel, dim3(ceil_div(C, 64), np), dim3(256), 0, stream, dy, y, sub, beta, rows, C,
                       rows_per, ws);
  CXRK_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_bwd_reduce_final_kernel, dim3(ceil_div(C, 64)), dim3(256), 0, stream, ws, np, C, sumdy, sumdyy);
  CXRK_LAUNCH_CHECK();
  return CXRK_OK;
}

extern "C" int cxrk_bn_fold(const float* w, const float* gamma, const float* beta, const float* rmean, const float* rvar,
                            float eps, int Ko, int taps, int C, int Cpad, float* w_scaled, float* scale, float* shift,
                            float* rstd, hipStream_t stream) {
  CXRK_CHECK_ARG(w && gamma && beta && rmean && rvar && w_scaled && scale && shift && rstd && Ko > 0 && taps > 0 && C > 0 && Cpad >= C);
  hipLaunchKernelGGL(bn_fold_kernel, dim3(Ko), dim3(256), 0, stream, w, gamma, beta, rmean, rvar, eps, Ko, taps, C, Cpad,
                     w_scaled, scale, shift, rstd);
  CXRK_LAUNCH_CHECK();
  return CXRK_OK;
}

// The convolution gathers address a block's rows with 32-bit byte offsets from the first image the tile touches
// (gemm_loaders.h): a 256-row tile spans at most 256 / (Ho*Wo) + 2 images of the gathered tensor, which must stay < 2 GiB.
static bool tile_span_ok(long rows_per_image, long image_elems) {
  const long images = 256 / (rows_per_image > 0 ? rows_per_image : 1) + 2;
  return images * image_elems * 4 < (1L << 31);
}

extern "C" int cxrk_conv_bn_act_fwd(const float* x, const float* w_scaled, const float* shift, const float* residual,
                                    float* y, int N, int H, int W, int C, int Ko, int R, int S, int stride, int pad,
                                    int relu, hipStream_t stream) {
  CXRK_CHECK_ARG(x && w_scaled && y && N > 0 && C % 4 == 0 && aligned16(x) && aligned16(w_scaled));
  CXRK_CHECK_ARG(stride == 1 || stride == 2);
  const ConvGeom g = make_geom(N, H, W, C, Ko, R, S, stride, pad);
  CXRK_CHECK_ARG(g.Ho > 0 && g.Wo > 0);
  const long Ml = (long)N * g.Ho * g.Wo;
  CXRK_CHECK_ARG(Ml < (1L << 31));
  const int M = (int)Ml, K = R * S * C;
  if (!tile_span_ok((long)g.Ho * g.Wo, (long)H * W * C)) return CXRK_ERR_UNSUPPORTED;
  EpiParams ep{};
  ep.C = y; ep.ldc = Ko; ep.bias = shift; ep.R = residual; ep.ldr = Ko; ep.act = relu ? 1 : 0; ep.alpha = 1.f;
  int rc;
  const bool tapwise = (C % BK == 0) && R * S <= 32 && R <= 8;  // a K-tile inside one filter tap (everything but the stem)
  if (tapwise && use_wide256(M, Ko, K, 1, false, WIDE_MINK_FPROP)) {
    ConvIm2colKC<256, true, NT_WIDE>::P pa{x, g, M, K}; DenseKC<256, NT_WIDE>::P pb{w_scaled, (long)K, Ko, K};
    rc = launch_gemm_wide<ConvIm2colKC<256, true, NT_WIDE>, DenseKC<256, NT_WIDE>>(pa, pb, ep, M, Ko, K, 1, stream);
  } else if (tapwise) {
    if (Ko <= 64) {
      ConvIm2colKC<256>::P pa{x, g, M, K}; DenseKC<64>::P pb{w_scaled, (long)K, Ko, K};
      rc = launch_gemm<ConvIm2colKC<256>, DenseKC<64>, 4, 1>(pa, pb, ep, M, Ko, K, 1, stream);
    } else {
      ConvIm2colKC<128>::P pa{x, g, M, K}; DenseKC<128>::P pb{w_scaled, (long)K, Ko, K};
      rc = launch_gemm<ConvIm2colKC<128>, DenseKC<128>, 2, 2>(pa, pb, ep, M, Ko, K, 1, stream);
    }
  } else {
    if (Ko <= 64) {
      ConvIm2colKC<256, false>::P pa{x, g, M, K}; DenseKC<64>::P pb{w_scaled, (long)K, Ko, K};
      rc = launch_gemm<ConvIm2colKC<256, false>, DenseKC<64>, 4, 1>(pa, pb, ep, M, Ko, K, 1, stream);
    } else {
      ConvIm2colKC<128, false>::P pa{x, g, M, K}; DenseKC<128>::P pb{w_scaled, (long)K, Ko, K};
      rc = launch_gemm<ConvIm2colKC<128, false>, DenseKC<128>, 2, 2>(pa, pb, ep, M, Ko, K, 1, stream);
    }
  }
  return rc < 0 ? rc : CXRK_OK;
}

// dx[n][hi][wi][c] = mask( sum_{r,s,ko} dy[n][(hi+pad-r)/st][(wi+pad-s)/st][ko] * w_scaled[ko][r][s][c] + residual )
// 64-row slabs of partial sums one data-gradient launch writes: its row tile (256 with the 256x256 / 256x64 tiles, else 128)
// rounded up, in slabs.  K = contraction length of that launch (selects the tile exactly as the launch does).
static int dgrad_tiles(int rows, int C, long K) {
  const bool t256 = use_wide256(rows, C, K, 1, false, WIDE_MINK_DGRAD) || C <= 64;
  return ceil_div(rows, t256 ? 256 : 128) * (t256 ? 4 : 2);
}

// number of per-wave partial rows the fused BN reduction of a data-gradient launch produces (all parity classes)
static long dgrad_bn_parts(int N, int H, int W, int C, int stride, int Ko, int R, int S, int pad) {
  if (stride == 1) return dgrad_tiles(N * H * W, C, (long)R * S * Ko);
  long t = 0;
  for (int ph = 0; ph < 2; ++ph)
    for (int pw = 0; pw < 2; ++pw) {
      const int Hs = (H - ph + 1) / 2, Ws = (W - pw + 1) / 2;
      int nr = 0, ns = 0;
      for (int r = 0; r < R; ++r) if (((ph + pad - r) & 1) == 0) ++nr;
      for (int q = 0; q < S; ++q) if (((pw + pad - q) & 1) == 0) ++ns;
      if (Hs > 0 && Ws > 0 && nr > 0 && ns > 0) t += dgrad_tiles(N * Hs * Ws, C, (long)nr * ns * Ko);
    }
  return t;
}

static int conv_bwd_data_impl(const float* dy, const float* w_scaled, const float* residual, const float* relu_src, float* dx,
                              int N, int H, int W, int C, int Ko, int R, int S, int stride, int pad, float* bn_part,
                              const float* bn_sub, const float* bn_beta, const float* bn_beta2, hipStream_t stream) {
  CXRK_CHECK_ARG(dy && w_scaled && dx && N > 0 && C % 4 == 0 && Ko % 4 == 0 && aligned16(dy) && aligned16(w_scaled));
  CXRK_CHECK_ARG(stride == 1 || stride == 2);
  if (Ko % BK != 0 || R * S > 32 || R > 8) return CXRK_ERR_UNSUPPORTED;  // the gather keeps a K-tile inside one filter tap
  const ConvGeom g = make_geom(N, H, W, C, Ko, R, S, stride, pad);
  const long Ml = (long)N * H * W;
  CXRK_CHECK_ARG(Ml < (1L << 31));
  const int M = (int)Ml, K = R * S * Ko;
  if (!tile_span_ok((long)(H / stride) * (W / stride), (long)g.Ho * g.Wo * Ko)) return CXRK_ERR_UNSUPPORTED;
  EpiParams ep{};
  ep.C = dx; ep.ldc = C; ep.R = residual; ep.ldr = C; ep.alpha = 1.f;
  if (relu_src) { ep.aux = relu_src; ep.ldaux = C; ep.auxmode = 1; }
  ep.bn_part = bn_part; ep.bn_sub = bn_sub; ep.bn_ldsub = C; ep.bn_beta = bn_beta; ep.bn_beta2 = bn_beta2;
  int rc = 0;
  if (stride == 1) {
    if (use_wide256(M, C, K, 1, false, WIDE_MINK_DGRAD)) {
      ConvDgradKC<256, NT_WIDE>::P pa{dy, g, M, K}; ConvFilterMC<256, NT_WIDE>::P pb{w_scaled, g, C, K};
      rc = launch_gemm_wide<ConvDgradKC<256, NT_WIDE>, ConvFilterMC<256, NT_WIDE>>(pa, pb, ep, M, C, K, 1, stream);
    } else if (C <= 64) {
      ConvDgradKC<256>::P pa{dy, g, M, K}; ConvFilterMC<64>::P pb{w_scaled, g, C, K};
      rc = launch_gemm<ConvDgradKC<256>, ConvFilterMC<64>, 4, 1>(pa, pb, ep, M, C, K, 1, stream);
    } else {
      ConvDgradKC<128>::P pa{dy, g, M, K}; ConvFilterMC<128>::P pb{w_scaled, g, C, K};
      rc = launch_gemm<ConvDgradKC<128>, ConvFilterMC<128>, 2, 2>(pa, pb, ep, M, C, K, 1, stream);
    }
    return rc < 0 ? rc : CXRK_OK;
  }
  // stride 2: one launch per output-parity class, only over the taps that reach it
  CXRK_CHECK_ARG(R <= 3 && S <= 3);
  long part_off = 0;
  bool zeroed = false;
  for (int ph = 0; ph < 2; ++ph) {
    for (int pw = 0; pw < 2; ++pw) {
      S2Taps t{};
      for (int r = 0; r < R; ++r) if (((ph + pad - r) & 1) == 0) { t.r[t.nr] = r; t.dr[t.nr] = (ph + pad - r) / 2; ++t.nr; }
      for (int q = 0; q < S; ++q) if (((pw + pad - q) & 1) == 0) { t.s[t.ns] = q; t.ds[t.ns] = (pw + pad - q) / 2; ++t.ns; }
      const int Hs = (H - ph + 1) / 2, Ws = (W - pw + 1) / 2;
      if (Hs <= 0 || Ws <= 0) continue;
      if (t.nr == 0 || t.ns == 0) {
        // no tap reaches this class: its pixels are exactly zero.  Only the plain form is supported there.
        CXRK_CHECK_ARG(residual == nullptr && relu_src == nullptr);
        if (!zeroed) {
          if (hipMemsetAsync(dx, 0, (size_t)M * C * sizeof(float), stream) != hipSuccess) return CXRK_ERR_LAUNCH;
          zeroed = true;
        }
      }
    }
  }
  for (int ph = 0; ph < 2; ++ph) {
    for (int pw = 0; pw < 2; ++pw) {
      S2Taps t{};
      for (int r = 0; r < R; ++r) if (((ph + pad - r) & 1) == 0) { t.r[t.nr] = r; t.dr[t.nr] = (ph + pad - r) / 2; ++t.nr; }
      for (int q = 0; q < S; ++q) if (((pw + pad - q) & 1) == 0) { t.s[t.ns] = q; t.ds[t.ns] = (pw + pad - q) / 2; ++t.ns; }
      const int Hs = (H - ph + 1) / 2, Ws = (W - pw + 1) / 2;
      if (Hs <= 0 || Ws <= 0 || t.nr == 0 || t.ns == 0) continue;
      const int Ms = N * Hs * Ws, Ks = t.nr * t.ns * Ko;
      EpiParams e2 = ep;
      if (bn_part) { e2.bn_part = bn_part + part_off * 3 * C; part_off += dgrad_tiles(Ms, C, Ks); }
      e2.rm_on = 1; e2.rm_Hs = Hs; e2.rm_Ws = Ws; e2.rm_H = H; e2.rm_W = W; e2.rm_ph = ph; e2.rm_pw = pw;
      if (use_wide256(Ms, C, Ks, 1, false, WIDE_MINK_DGRAD)) {
        ConvDgradS2KC<256, NT_WIDE>::P pa{dy, g, t, Hs, Ws, Ms, Ks}; ConvFilterS2MC<256, NT_WIDE>::P pb{w_scaled, g, t, C, Ks};
        rc = launch_gemm_wide<ConvDgradS2KC<256, NT_WIDE>, ConvFilterS2MC<256, NT_WIDE>>(pa, pb, e2, Ms, C, Ks, 1, stream);
      } else if (C <= 64) {
        ConvDgradS2KC<256>::P pa{dy, g, t, Hs, Ws, Ms, Ks}; ConvFilterS2MC<64>::P pb{w_scaled, g, t, C, Ks};
        rc = launch_gemm<ConvDgradS2KC<256>, ConvFilterS2MC<64>, 4, 1>(pa, pb, e2, Ms, C, Ks, 1, stream);
      } else {
        ConvDgradS2KC<128>::P pa{dy, g, t, Hs, Ws, Ms, Ks}; ConvFilterS2MC<128>::P pb{w_scaled, g, t, C, Ks};
        rc = launch_gemm<ConvDgradS2KC<128>, ConvFilterS2MC<128>, 2, 2>(pa, pb, e2, Ms, C, Ks, 1, stream);
      }
      if (rc < 0) return rc;
    }
  }
  return CXRK_OK;
}

extern "C" int cxrk_conv_bn_act_bwd_data(const float* dy, const float* w_scaled, const float* residual,
                                         const float* relu_src, float* dx, int N, int H, int W, int C, int Ko, int R, int S,
                                         int stride, int pad, hipStream_t stream) {
  return conv_bwd_data_impl(dy, w_scaled, residual, relu_src, dx, N, H, W, C, Ko, R, S, stride, pad, nullptr, nullptr, nullptr,
                            nullptr, stream);
}

extern "C" size_t cxrk_conv_bwd_data_bnsum_ws_bytes(int N, int H, int W, int C, int stride) {
  // upper bound over the tile choices (the exact count needs the filter shape): 256-row tiles, 4 slabs each
  long parts = 0;
  if (stride == 1) parts = (long)ceil_div((long)N * H * W, 256) * 4;
  else
    for (int ph = 0; ph < 2; ++ph)
      for (int pw = 0; pw < 2; ++pw) { const int Hs = (H - ph + 1) / 2, Ws = (W - pw + 1) / 2; if (Hs > 0 && Ws > 0) parts += (long)ceil_div((long)N * Hs * Ws, 256) * 4; }
  return (size_t)(parts + 64) * 3 * C * sizeof(float);
}

// Data gradient + the BatchNorm-backward channel sums of the unit that PRODUCED relu_src, in one pass:
//   sums[0][c] = sum dx,  sums[1][c] = sum dx*(relu_src - bn_sub - bn_beta[c]),  sums[2][c] = sum dx*(bn_sub - bn_beta2[c])
// (dx = the masked gradient this call writes = that unit's dy; relu_src - bn_sub = its BN output where dx != 0).
extern "C" int cxrk_conv_bn_act_bwd_data_bnsum(const float* dy, const float* w_scaled, const float* residual,
                                               const float* relu_src, float* dx, int N, int H, int W, int C, int Ko, int R,
                                               int S, int stride, int pad, const float* bn_sub, const float* bn_beta,
                                               const float* bn_beta2, float* sums, float* ws, size_t ws_bytes,
                                               hipStream_t stream) {
  CXRK_CHECK_ARG(relu_src && bn_beta && sums && aligned16(bn_beta) && (!bn_beta2 || aligned16(bn_beta2)) && (!bn_sub || aligned16(bn_sub)));
  CXRK_CHECK_ARG(!(R == 1 && stride == 2));
  const long np = dgrad_bn_parts(N, H, W, C, stride, Ko, R, S, pad);
  if (ws == nullptr || ws_bytes < (size_t)(np + 64) * 3 * C * sizeof(float)) return CXRK_ERR_WS;
  const int rc = conv_bwd_data_impl(dy, w_scaled, residual, relu_src, dx, N, H, W, C, Ko, R, S, stride, pad, ws, bn_sub, bn_beta,
                                    bn_beta2, stream);
  if (rc != CXRK_OK) return rc;
  int Z = ceil_div(np, 512); if (Z > 64) Z = 64; if (Z < 1) Z = 1;
  const int chunk = ceil_div(np, Z);
  Z = ceil_div(np, chunk);
  float* tmp = ws + np * 3 * C;
  hipLaunchKernelGGL(bn_part_final_kernel, dim3(ceil_div(C, 64), 3, Z), dim3(1024), 0, stream, ws, (int)np, chunk, C, Z > 1 ? tmp : sums);
  CXRK_LAUNCH_CHECK();
  if (Z > 1) {
    hipLaunchKernelGGL(bn_part_final_kernel, dim3(ceil_div(C, 64), 3, 1), dim3(1024), 0, stream, tmp, Z, Z, C, sums);
    CXRK_LAUNCH_CHECK();
  }
  return CXRK_OK;
}

static int wgrad_splitk(int Ko, int Ncols, long Kred) { return cxrk_gemm_wgrad_splitk(Ko, Ncols, (int)Kred); }

extern "C" size_t cxrk_conv_wgrad_ws_bytes(int N, int H, int W, int C, int Ko, int R, int S, int stride, int pad) {
  const ConvGeom g = make_geom(N, H, W, C, Ko, R, S, stride, pad);
  const int sk = wgrad_splitk(Ko, R * S * C, (long)N * g.Ho * g.Wo);
  return (size_t)sk * (size_t)Ko * (size_t)(R * S * C) * sizeof(float);
}

// dW (+ BN parameter gradients).  x: conv input [N,H,W,Cpad]; dy: gradient w.r.t. the BN output, already ReLU-masked.
// w: raw (unscaled) filter [Ko][R][S][C]; sumdy[ko] = sum of dy over (n,ho,wo) (cxrk_colsum).  C may be < Cpad (stem).
extern "C" int cxrk_conv_bn_act_bwd_params(const float* x, const float* dy, const float* w, const float* scale,
                                           const float* rstd, const float* rmean, const float* sumdy,
                                           const float* gamma, const float* sumdyy, float* dw,
                                           float* dgamma, float* dbeta, int accumulate, int N, int H, int W, int C,
                                           int Cpad, int Ko, int R, int S, int stride, int pad, float* ws, size_t ws_bytes,
                                           hipStream_t stream) {
  CXRK_CHECK_ARG(x && dy && dw && N > 0 && Cpad % 4 == 0 && Ko % 4 == 0 && aligned16(x) && aligned16(dy));
  CXRK_CHECK_ARG(!(dgamma && !(w && rstd && rmean && sumdy && dbeta)));
  const ConvGeom g = make_geom(N, H, W, Cpad, Ko, R, S, stride, pad);
  const long Kl = (long)N * g.Ho * g.Wo;
  CXRK_CHECK_ARG(Kl < (1L << 31));
  const int Kred = (int)Kl, Nc = R * S * Cpad;
  if ((long)H * W * Cpad * 4 * 3 >= (1L << 31)) return CXRK_ERR_UNSUPPORTED;  // a K-tile of 32 pixels spans <= 3 images
  int sk = wgrad_splitk(Ko, Nc, Kl);
  if (ws == nullptr || ws_bytes < (size_t)sk * Ko * Nc * sizeof(float)) return CXRK_ERR_WS;
  EpiParams ep{};
  ep.C = ws; ep.ldc = Nc; ep.alpha = 1.f; ep.slab_stride = (long)Ko * Nc;
  int rc;
  if (use_wide256(Ko, Nc, Kred, sk, Cpad <= 4)) {
    DenseMC<256, NT_WIDE>::P pa{dy, (long)Ko, Ko, Kred}; ConvIm2colMC<256, NT_WIDE>::P pb{x, g, Nc, Kred};
    rc = launch_gemm_wide<DenseMC<256, NT_WIDE>, ConvIm2colMC<256, NT_WIDE>>(pa, pb, ep, Ko, Nc, Kred, sk, stream);
  } else if (Ko <= 64) {
    DenseMC<64>::P pa{dy, (long)Ko, Ko, Kred}; ConvIm2colMC<256>::P pb{x, g, Nc, Kred};
    rc = launch_gemm<DenseMC<64>, ConvIm2colMC<256>, 1, 4>(pa, pb, ep, Ko, Nc, Kred, sk, stream, Cpad <= 4);
  } else {
    DenseMC<128>::P pa{dy, (long)Ko, Ko, Kred}; ConvIm2colMC<128>::P pb{x, g, Nc, Kred};
    rc = launch_gemm<DenseMC<128>, ConvIm2colMC<128>, 2, 2>(pa, pb, ep, Ko, Nc, Kred, sk, stream, Cpad <= 4);
  }
  if (rc < 0) return rc;
  const int sg_log2 = rc >= 64 ? 4 : (rc >= 8 ? 2 : 0);  // slab groups per block: 16 / 4 / 1
  hipLaunchKernelGGL(wgrad_reduce_bn_kernel, dim3(Ko, ceil_div(Nc, 4 * (256 >> sg_log2))), dim3(256), 0, stream, ws, rc, (long)Ko * Nc, R * S, C, Cpad, w,
                     scale, rstd, rmean, sumdy, gamma, sumdyy, dw, dgamma, dbeta, accumulate, sg_log2);
  CXRK_LAUNCH_CHECK();
  return CXRK_OK;
}

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
