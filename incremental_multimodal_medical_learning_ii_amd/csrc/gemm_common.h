// Shared by gemm_f32.hip / gemm_pl.hip: tile dispatch on the 128x128-class tiles and the split-K slab reduction.
#pragma once
#include "cxrk.h"
#include "gemm_core.h"

namespace cxrk {

static __global__ void splitk_reduce_kernel(const float* __restrict__ ws, int nslab, long slab, float* __restrict__ C, long ldc,
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


// Tile dispatch on the 128x128-class tiles (fp32 or planes operands).
template <class FMT, template <int, class, int> class LAT, template <int, class, int> class LBT>
static int dense_small(const typename FMT::T* A, long lda, long aplane, const typename FMT::T* B, long ldb, long bplane,
                       const EpiParams& ep, int M, int N, int K, int splitk, hipStream_t stream) {
  if (N <= 64) {
    typename LAT<256, FMT, NTHREADS>::P pa{A, lda, M, K, aplane}; typename LBT<64, FMT, NTHREADS>::P pb{B, ldb, N, K, bplane};
    return launch_gemm<LAT<256, FMT, NTHREADS>, LBT<64, FMT, NTHREADS>, 4, 1>(pa, pb, ep, M, N, K, splitk, stream);
  }
  if (M <= 64) {
    typename LAT<64, FMT, NTHREADS>::P pa{A, lda, M, K, aplane}; typename LBT<256, FMT, NTHREADS>::P pb{B, ldb, N, K, bplane};
    return launch_gemm<LAT<64, FMT, NTHREADS>, LBT<256, FMT, NTHREADS>, 1, 4>(pa, pb, ep, M, N, K, splitk, stream);
  }
  typename LAT<128, FMT, NTHREADS>::P pa{A, lda, M, K, aplane}; typename LBT<128, FMT, NTHREADS>::P pb{B, ldb, N, K, bplane};
  return launch_gemm<LAT<128, FMT, NTHREADS>, LBT<128, FMT, NTHREADS>, 2, 2>(pa, pb, ep, M, N, K, splitk, stream);
}
static int finish_splitk(int rc, int splitk, int M, int N, float* ws, float* C, long ldc, float alpha, int accumulate,
                         hipStream_t stream) {
  if (rc < 0) return rc;
  if (splitk > 1) {
    const long slab = (long)M * N;
    const int nblk = ceil_div(slab / 4, 256);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(nblk), dim3(256), 0, stream, ws, rc, slab, C, ldc, N, alpha, accumulate);
    CXRK_LAUNCH_CHECK();
  }
  return CXRK_OK;
}


}  // namespace cxrk
