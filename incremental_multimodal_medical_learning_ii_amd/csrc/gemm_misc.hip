// Policy helpers, column sums (bias gradients) and the fp32 <-> planes conversions.
#include "cxrk.h"
#include "gemm_core.h"

using namespace cxrk;

namespace {
// Stage 1 of a deterministic column sum: block (bx, by) sums rows [by*rows_per, ...) of columns bx*256..+255.
// cols % 4 == 0: a thread owns 4 columns (16-byte loads) and every fourth row of the range, 8 loads in flight; the four
// row lanes are added in order through LDS.  (One dword column per thread kept 12 KB in flight per CU: 3 TB/s.)
__global__ __launch_bounds__(256) void colsum_partial_vec_kernel(const float* __restrict__ X, long ldx, long rows, int cols,
                                                                 int rows_per, float* __restrict__ part) {
  __shared__ float4 sh[4][64];
  const int cq = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int col = blockIdx.x * 256 + cq * 4;
  const long r0 = (long)blockIdx.y * rows_per;
  const long r1 = min(rows, r0 + rows_per);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col < cols) {
    const float* p = X + col;
    long r = r0 + rl;
    for (; r + 28 < r1; r += 32) {
      float4 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const float4*>(p + (r + 4 * u) * ldx);
#pragma unroll
      for (int u = 0; u < 8; ++u) { s.x += t[u].x; s.y += t[u].y; s.z += t[u].z; s.w += t[u].w; }
    }
    for (; r < r1; r += 4) { const float4 t = *reinterpret_cast<const float4*>(p + r * ldx); s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
  }
  sh[rl][cq] = s;
  __syncthreads();
  if (rl == 0 && col < cols) {
    const float4 a = sh[1][cq], b = sh[2][cq], c = sh[3][cq];
    s.x = (s.x + a.x) + (b.x + c.x); s.y = (s.y + a.y) + (b.y + c.y); s.z = (s.z + a.z) + (b.z + c.z); s.w = (s.w + a.w) + (b.w + c.w);
    *reinterpret_cast<float4*>(part + (long)blockIdx.y * cols + col) = s;
  }
}
// The same sum over a planes tensor (value = hi + lo): a thread owns 8 columns (16 bytes of each plane), 4 rows in flight.
__global__ __launch_bounds__(256) void colsum_partial_pl_kernel(const unsigned short* __restrict__ X, long ldx, long plane, long rows,
                                                                int cols, int rows_per, float* __restrict__ part) {
  __shared__ float sh[8][32][8];
  const int cq = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int col = blockIdx.x * 256 + cq * 8;
  const long r0 = (long)blockIdx.y * rows_per;
  const long r1 = min(rows, r0 + rows_per);
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (col < cols) {
    long r = r0 + rl;
    for (; r + 24 < r1; r += 32) {
      uint4 h[4], l[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        h[u] = *reinterpret_cast<const uint4*>(X + (r + 8 * u) * ldx + col);
        l[u] = *reinterpret_cast<const uint4*>(X + plane + (r + 8 * u) * ldx + col);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float v[8]; planes_unpack8(h[u], l[u], v);
#pragma unroll
        for (int q = 0; q < 8; ++q) s[q] += v[q];
      }
    }
    for (; r < r1; r += 8) {
      float v[8]; planes_load8(X, plane, r * ldx + col, v);
#pragma unroll
      for (int q = 0; q < 8; ++q) s[q] += v[q];
    }
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) sh[rl][cq][q] = s[q];
  __syncthreads();
  if (rl == 0 && col < cols) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float t = s[q];
#pragma unroll
      for (int i = 1; i < 8; ++i) t += sh[i][cq][q];
      part[(long)blockIdx.y * cols + col + q] = t;
    }
  }
}
// fp32 -> planes (hi = bf16(x), lo = bf16(x - hi)), 8 elements per thread; n % 8 == 0
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, long n8, unsigned short* __restrict__ out, long plane) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const float4 a = *reinterpret_cast<const float4*>(x + i * 8), b = *reinterpret_cast<const float4*>(x + i * 8 + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    planes_store8(out, plane, i * 8, v);
  }
}
__global__ __launch_bounds__(256) void merge_planes_kernel(const unsigned short* __restrict__ x, long plane, long n8, float* __restrict__ out) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    float v[8]; planes_load8(x, plane, i * 8, v);
    *reinterpret_cast<float4*>(out + i * 8) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(out + i * 8 + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
}

__global__ void colsum_partial_kernel(const float* __restrict__ X, long ldx, long rows, int cols, int rows_per,
                                      float* __restrict__ part) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= cols) return;
  const long r0 = (long)blockIdx.y * rows_per;
  const long r1 = min(rows, r0 + rows_per);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  long r = r0;
  for (; r + 3 < r1; r += 4) {
    s0 += X[r * ldx + col]; s1 += X[(r + 1) * ldx + col]; s2 += X[(r + 2) * ldx + col]; s3 += X[(r + 3) * ldx + col];
  }
  for (; r < r1; ++r) s0 += X[r * ldx + col];
  part[(long)blockIdx.y * cols + col] = (s0 + s1) + (s2 + s3);
}
// Stage 1 of the column sum of squared deviations sum_r (x[r][c] - mean[c])^2 (the second pass of a two-pass variance): fp32 (plane
// == 0) or planes input, one column per thread, four rows in flight.  Used by the BatchNorm calibration pass only.
__global__ void colvar_partial_kernel(const float* __restrict__ X, const unsigned short* __restrict__ Xp, long ldx, long plane, long rows,
                                      int cols, int rows_per, const float* __restrict__ mean, float* __restrict__ part) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= cols) return;
  const long r0 = (long)blockIdx.y * rows_per;
  const long r1 = min(rows, r0 + rows_per);
  const float m = mean[col];
  auto val = [&](long r) {
    if (Xp) return __builtin_bit_cast(float, (unsigned)Xp[r * ldx + col] << 16) + __builtin_bit_cast(float, (unsigned)Xp[plane + r * ldx + col] << 16);
    return X[r * ldx + col];
  };
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  long r = r0;
  for (; r + 3 < r1; r += 4) {
    const float a = val(r) - m, b = val(r + 1) - m, c = val(r + 2) - m, d = val(r + 3) - m;
    s0 = fmaf(a, a, s0); s1 = fmaf(b, b, s1); s2 = fmaf(c, c, s2); s3 = fmaf(d, d, s3);
  }
  for (; r < r1; ++r) { const float a = val(r) - m; s0 = fmaf(a, a, s0); }
  part[(long)blockIdx.y * cols + col] = (s0 + s1) + (s2 + s3);
}
__global__ void colsum_final_kernel(const float* __restrict__ part, int nparts, int cols, float* __restrict__ out,
                                    float alpha, int accumulate) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= cols) return;
  float s = 0.f;
  for (int p = 0; p < nparts; ++p) s += part[(long)p * cols + col];
  out[col] = accumulate ? out[col] + alpha * s : alpha * s;
}

}  // namespace

// Split-K factor for a weight-gradient shaped GEMM (small M x N output, very long K): enough slabs to fill the chip with
// the tile the launch will use (256x256, planes operands only: one round of 256 blocks; 128x128: about six blocks per CU),
// each slab at least 8 K-tiles long.
int cxrk::wgrad_splitk_policy(int M, int N, int K, bool planes) {
  const long maxk = K / (8 * BK) > 0 ? K / (8 * BK) : 1;
  if (planes && wide_mode() != 0 && M >= 256 && N >= 256) {
    const long tiles = (long)ceil_div(M, 256) * ceil_div(N, 256);
    // ONE round of 256 blocks (floor: one block over would cost a second round).  Two rounds (512) were the round-1 choice for
    // the register-staged kernel; with the pipelined kernel half as many, twice as long slabs win: fewer prologues / epilogues
    // per CU and half the slab traffic of the reduction (r2q: weight-gradient launches 27.3 -> 25.3 ms per step).
    static const long blocks = env_long("CXRK_WGRAD_BLOCKS", 256);   // tuning override
    long sk = blocks / tiles;
    if (sk > maxk) sk = maxk;
    if (sk < 1) sk = 1;
    if (sk > 512) sk = 512;
    if (use_wide256(M, N, K, (int)sk, true)) return (int)sk;
  }
  const long tiles = (long)ceil_div(M, 128) * ceil_div(N, 128);
  static const long blocks128 = env_long("CXRK_WGRAD_BLOCKS128", 1536);   // tuning override (128x128-class tiles: ~6 blocks per CU)
  long sk = (blocks128 + tiles - 1) / tiles;
  if (sk > K / 256) sk = K / 256;
  if (sk > 512) sk = 512;
  if (sk < 1) sk = 1;
  return (int)sk;
}
extern "C" int cxrk_gemm_wgrad_splitk(int M, int N, int K, int planes) { return wgrad_splitk_policy(M, N, K, planes != 0); }

// 1 when a launch of this shape on planes operands takes the 256x256 kernel.  kind: 0 / 3 = dense layer or weight gradient
// (3 = with a fused epilogue; same policy), 1 = convolution forward, 2 = convolution data gradient.
// (Reporting only: lets the host label its launch timings by mainloop.)
extern "C" int cxrk_gemm_wide_tile(int M, int N, long K, int splitk, int kind) {
  return use_wide256(M, N, K, splitk, true, kind == 1 ? WIDE_MINK_FPROP : (kind == 2 ? WIDE_MINK_DGRAD : WIDE_MINK_PLAIN));
}

extern "C" size_t cxrk_gemm_splitk_ws_bytes(int M, int N, int splitk) {
  return splitk > 1 ? (size_t)splitk * (size_t)M * (size_t)N * sizeof(float) : 0;
}

extern "C" size_t cxrk_colsum_ws_bytes(long rows, int cols) {
  int nparts = (int)((rows + 511) / 512);
  if (nparts > 512) nparts = 512;
  if (nparts < 1) nparts = 1;
  return (size_t)nparts * (size_t)cols * sizeof(float);
}

// fp32 tensor -> its split-bf16 planes (weights once per step, inputs of the path) and back (tests, host-side consumers).
extern "C" int cxrk_split_planes(const float* x, long n, void* out, long plane, hipStream_t stream) {
  CXRK_CHECK_ARG(x && out && n > 0 && (n % 8) == 0 && aligned16(x) && aligned16(out) && (plane % 8) == 0 && plane >= n);
  const long n8 = n / 8;
  long nb = (n8 + 255) / 256; if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)nb), dim3(256), 0, stream, x, n8, static_cast<unsigned short*>(out), plane);
  CXRK_LAUNCH_CHECK();
  return CXRK_OK;
}
extern "C" int cxrk_merge_planes(const void* x, long plane, long n, float* out, hipStream_t stream) {
  CXRK_CHECK_ARG(x && out && n > 0 && (n % 8) == 0 && aligned16(x) && aligned16(out) && (plane % 8) == 0);
  const long n8 = n / 8;
  long nb = (n8 + 255) / 256; if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(merge_planes_kernel, dim3((unsigned)nb), dim3(256), 0, stream, static_cast<const unsigned short*>(x), plane, n8, out);
  CXRK_LAUNCH_CHECK();
  return CXRK_OK;
}

extern "C" int cxrk_colsum_pl(const void* X, long ldx, long plane, long rows, int cols, float* out, float alpha, int accumulate,
                              float* ws, size_t ws_bytes, hipStream_t stream) {
  CXRK_CHECK_ARG(X && out && rows > 0 && cols > 0 && (cols % 8) == 0 && (ldx % 8) == 0 && (plane % 8) == 0 && aligned16(X));
  int nparts = (int)((rows + 511) / 512);
  if (nparts > 512) nparts = 512;
  if (nparts < 1) nparts = 1;
  if (ws == nullptr || ws_bytes < (size_t)nparts * cols * sizeof(float)) return CXRK_ERR_WS;
  const int rows_per = (int)((rows + nparts - 1) / nparts);
  nparts = (int)((rows + rows_per - 1) / rows_per);
  hipLaunchKernelGGL(colsum_partial_pl_kernel, dim3(ceil_div(cols, 256), nparts), dim3(256), 0, stream,
                     static_cast<const unsigned short*>(X), ldx, plane, rows, cols, rows_per, ws);
  CXRK_LAUNCH_CHECK();
  hipLaunchKernelGGL(colsum_final_kernel, dim3(ceil_div(cols, 256)), dim3(256), 0, stream, ws, nparts, cols, out, alpha, accumulate);
  CXRK_LAUNCH_CHECK();
  return CXRK_OK;
}

// out[c] = alpha * sum_r (X[r][c] - mean[c])^2; X fp32 (plane == 0) or planes (plane > 0: hi at X, lo `plane` elements behind).
extern "C" int cxrk_colvar(const void* X, long ldx, long plane, long rows, int cols, const float* mean, float* out, float alpha,
                           float* ws, size_t ws_bytes, hipStream_t stream) {
  CXRK_CHECK_ARG(X && mean && out && rows > 0 && cols > 0 && plane >= 0);
  int nparts = (int)((rows + 511) / 512);
  if (nparts > 512) nparts = 512;
  if (nparts < 1) nparts = 1;
  if (ws == nullptr || ws_bytes < (size_t)nparts * cols * sizeof(float)) return CXRK_ERR_WS;
  const int rows_per = (int)((rows + nparts - 1) / nparts);
  nparts = (int)((rows + rows_per - 1) / rows_per);
  hipLaunchKernelGGL(colvar_partial_kernel, dim3(ceil_div(cols, 256), nparts), dim3(256), 0, stream,
                     plane ? nullptr : static_cast<const float*>(X), plane ? static_cast<const unsigned short*>(X) : nullptr, ldx, plane, rows,
                     cols, rows_per, mean, ws);
  CXRK_LAUNCH_CHECK();
  hipLaunchKernelGGL(colsum_final_kernel, dim3(ceil_div(cols, 256)), dim3(256), 0, stream, ws, nparts, cols, out, alpha, 0);
  CXRK_LAUNCH_CHECK();
  return CXRK_OK;
}

extern "C" int cxrk_colsum(const float* X, long ldx, long rows, int cols, float* out, float alpha, int accumulate,
                           float* ws, size_t ws_bytes, hipStream_t stream) {
  CXRK_CHECK_ARG(X && out && rows > 0 && cols > 0);
  int nparts = (int)((rows + 511) / 512);
  if (nparts > 512) nparts = 512;
  if (nparts < 1) nparts = 1;
  if (ws == nullptr || ws_bytes < (size_t)nparts * cols * sizeof(float)) return CXRK_ERR_WS;
  const int rows_per = (int)((rows + nparts - 1) / nparts);
  nparts = (int)((rows + rows_per - 1) / rows_per);
  if (cols % 4 == 0 && ldx % 4 == 0 && aligned16(X) && aligned16(ws))
    hipLaunchKernelGGL(colsum_partial_vec_kernel, dim3(ceil_div(cols, 256), nparts), dim3(256), 0, stream, X, ldx, rows, cols,
                       rows_per, ws);
  else
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(ceil_div(cols, 256), nparts), dim3(256), 0, stream, X, ldx, rows, cols,
                       rows_per, ws);
  CXRK_LAUNCH_CHECK();
  hipLaunchKernelGGL(colsum_final_kernel, dim3(ceil_div(cols, 256)), dim3(256), 0, stream, ws, nparts, cols, out, alpha,
                     accumulate);
  CXRK_LAUNCH_CHECK();
  return CXRK_OK;
}
