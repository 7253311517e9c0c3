// Dense fp32 GEMM entry points (Linear fprop / dgrad / wgrad of CXR-BERT, the projection heads and the adapters)
// plus the generic column-sum (bias gradients) and split-K slab reduction.
#include "cxrk.h"
#include "gemm_core.h"

using namespace cxrk;

namespace {

__global__ void splitk_reduce_kernel(const float* __restrict__ ws, int nslab, long slab, float* __restrict__ C, long ldc,
                                     int N, float alpha, int accumulate) {
  const long i4 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 >= slab) return;
  float4 s = *reinterpret_cast<const float4*>(ws + i4);
  for (int z = 1; z < nslab; ++z) {
    const float4 t = *reinterpret_cast<const float4*>(ws + (long)z * slab + i4);
    s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
  }
  const long row = i4 / N; const int col = (int)(i4 - row * N);  // N % 4 == 0 -> the 4 values share a row
  float* c = C + row * ldc + col;
  if (accumulate) { c[0] += alpha * s.x; c[1] += alpha * s.y; c[2] += alpha * s.z; c[3] += alpha * s.w; }
  else { c[0] = alpha * s.x; c[1] = alpha * s.y; c[2] = alpha * s.z; c[3] = alpha * s.w; }
}

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
// the tile the launch will use (256x256: about two rounds of 256 blocks; 128x128: about six blocks per CU), each slab at
// least 8 K-tiles long.
extern "C" int cxrk_gemm_wgrad_splitk(int M, int N, int K) {
  const long maxk = K / (8 * BK) > 0 ? K / (8 * BK) : 1;
  if (gemm_precision_mode() == 1 && wide_mode() != 0 && M >= 256 && N >= 256) {
    const long tiles = (long)ceil_div(M, 256) * ceil_div(N, 256);
    long sk = 512 / tiles;   // floor: two full rounds of 256 blocks at most (one block over would cost a third round)
    if (sk > maxk) sk = maxk;
    if (sk < 1) sk = 1;
    if (sk > 512) sk = 512;
    if (use_wide256(M, N, K, (int)sk)) return (int)sk;
  }
  const long tiles = (long)ceil_div(M, 128) * ceil_div(N, 128);
  long sk = (1536 + tiles - 1) / tiles;
  if (sk > K / 256) sk = K / 256;
  if (sk > 512) sk = 512;
  if (sk < 1) sk = 1;
  return (int)sk;
}

// 1 when a launch of this shape takes the 256x256 tile in the current precision mode.  kind: 0 / 3 = dense layer or weight
// gradient (3 = with a fused epilogue; same policy), 1 = convolution forward, 2 = convolution data gradient.
// (Reporting only: lets the host label its launch timings by mainloop.)
extern "C" int cxrk_gemm_wide_tile(int M, int N, long K, int splitk, int kind) {
  return use_wide256(M, N, K, splitk, false, kind == 1 ? WIDE_MINK_FPROP : (kind == 2 ? WIDE_MINK_DGRAD : WIDE_MINK_PLAIN));
}

extern "C" size_t cxrk_gemm_splitk_ws_bytes(int M, int N, int splitk) {
  return splitk > 1 ? (size_t)splitk * (size_t)M * (size_t)N * sizeof(float) : 0;
}

extern "C" int cxrk_gemm_f32(int transA, int transB, int M, int N, int K, const float* A, long lda, const float* B,
                             long ldb, float* C, long ldc, const float* bias, const float* R, long ldr,
                             const float* aux, long ldaux, int auxmode, float* C2, long ldc2, int act, float alpha,
                             int accumulate, int splitk, float* ws, size_t ws_bytes, hipStream_t stream) {
  CXRK_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0);
  CXRK_CHECK_ARG(aligned16(A) && aligned16(B) && (lda % 4 == 0) && (ldb % 4 == 0));
  CXRK_CHECK_ARG(transA ? (M % 4 == 0) : (K % 4 == 0));
  CXRK_CHECK_ARG(transB ? (K % 4 == 0) : (N % 4 == 0));
  CXRK_CHECK_ARG(!(auxmode != 0 && aux == nullptr));
  // the loaders address a tile with 32-bit byte offsets from its origin (gemm_loaders.h): 256 rows x ld must stay < 2 GiB
  if (lda >= (1L << 20) || ldb >= (1L << 20)) return CXRK_ERR_UNSUPPORTED;
  if (splitk < 1) splitk = 1;
  EpiParams ep{};
  ep.alpha = alpha; ep.slab_stride = 0;
  const bool plain = !bias && !R && !aux && !C2 && act == 0;
  if (splitk > 1) {
    CXRK_CHECK_ARG(plain && (N % 4 == 0));
    if (ws == nullptr || ws_bytes < cxrk_gemm_splitk_ws_bytes(M, N, splitk)) return CXRK_ERR_WS;
    ep.C = ws; ep.ldc = N; ep.alpha = 1.f; ep.slab_stride = (long)M * N;
  } else {
    if (accumulate) { CXRK_CHECK_ARG(R == nullptr); ep.R = C; ep.ldr = ldc; }
    else { ep.R = R; ep.ldr = ldr; }
    ep.C = C; ep.ldc = ldc; ep.bias = bias; ep.aux = aux; ep.ldaux = ldaux; ep.auxmode = auxmode;
    ep.C2 = C2; ep.ldc2 = ldc2; ep.act = act;
  }
  int rc;
  // 256x256 tile wherever the policy allows it.  (Alone on the GPU, launches with a fused bias / GELU / residual epilogue ran
  // 3-7 % slower on it than on 128x128 -- one block per CU exposes the epilogue --, but in the step the other encoder's
  // stream fills those phases and the larger tile wins: -1.4 % step time.)
#define CXRK_TILES(LAT, LBT, pa_expr, pb_expr)                                                                    \
  if (use_wide256(M, N, K, splitk)) { LAT<256, NT_WIDE>::P pa = pa_expr; LBT<256, NT_WIDE>::P pb = pb_expr;       \
    rc = launch_gemm_wide<LAT<256, NT_WIDE>, LBT<256, NT_WIDE>>(pa, pb, ep, M, N, K, splitk, stream); }           \
  else if (N <= 64) { LAT<256>::P pa = pa_expr; LBT<64>::P pb = pb_expr;                                               \
    rc = launch_gemm<LAT<256>, LBT<64>, 4, 1>(pa, pb, ep, M, N, K, splitk, stream); }                             \
  else if (M <= 64) { LAT<64>::P pa = pa_expr; LBT<256>::P pb = pb_expr;                                          \
    rc = launch_gemm<LAT<64>, LBT<256>, 1, 4>(pa, pb, ep, M, N, K, splitk, stream); }                             \
  else { LAT<128>::P pa = pa_expr; LBT<128>::P pb = pb_expr;                                                      \
    rc = launch_gemm<LAT<128>, LBT<128>, 2, 2>(pa, pb, ep, M, N, K, splitk, stream); }
#define PA_KC {A, lda, M, K}
#define PA_MC {A, lda, M, K}
#define PB_KC {B, ldb, N, K}
#define PB_MC {B, ldb, N, K}
  if (!transA && transB) { CXRK_TILES(DenseKC, DenseKC, PA_KC, PB_KC) }
  else if (!transA && !transB) { CXRK_TILES(DenseKC, DenseMC, PA_KC, PB_MC) }
  else if (transA && !transB) { CXRK_TILES(DenseMC, DenseMC, PA_MC, PB_MC) }
  else { CXRK_TILES(DenseMC, DenseKC, PA_MC, PB_KC) }
#undef CXRK_TILES
  if (rc < 0) return rc;
  if (splitk > 1) {
    const long slab = (long)M * N;
    const int nblk = ceil_div(slab / 4, 256);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(nblk), dim3(256), 0, stream, ws, rc, slab, C, ldc, N, alpha, accumulate);
    CXRK_LAUNCH_CHECK();
  }
  return CXRK_OK;
}

extern "C" size_t cxrk_colsum_ws_bytes(long rows, int cols) {
  int nparts = (int)((rows + 511) / 512);
  if (nparts > 512) nparts = 512;
  if (nparts < 1) nparts = 1;
  return (size_t)nparts * (size_t)cols * sizeof(float);
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
