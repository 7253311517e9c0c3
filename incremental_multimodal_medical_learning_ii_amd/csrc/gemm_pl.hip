// Dense GEMM entry point on split-bf16 planes tensors (the encoders in split-bf16 mode), always on the LDS-DMA pipelined kernel
// (gemm_pw.h): 256x256 tiles where the policy picks them, 256x64 / 64x256 for outputs with <= 64 columns / rows, 128x128 otherwise.
#include "gemm_common.h"

using namespace cxrk;

template <template <int, int> class DLA, template <int, int> class DLB>
static int dense_planes(const unsigned short* A, long lda, long aplane, const unsigned short* B, long ldb, long bplane, const EpiParams& ep,
                        int M, int N, int K, int splitk, hipStream_t stream) {
  if (use_wide256(M, N, K, splitk, true)) {
    typename DLA<256, 8>::P pa{A, lda, M, K, aplane}; typename DLB<256, 8>::P pb{B, ldb, N, K, bplane};
    return launch_gemm_pw<Pw256, DLA<256, 8>, DLB<256, 8>>(pa, pb, ep, M, N, K, splitk, stream);
  }
  if (N <= 64) {
    typename DLA<256, 4>::P pa{A, lda, M, K, aplane}; typename DLB<64, 4>::P pb{B, ldb, N, K, bplane};
    return launch_gemm_pw<Pw256x64, DLA<256, 4>, DLB<64, 4>>(pa, pb, ep, M, N, K, splitk, stream);
  }
  if (M <= 64) {
    typename DLA<64, 4>::P pa{A, lda, M, K, aplane}; typename DLB<256, 4>::P pb{B, ldb, N, K, bplane};
    return launch_gemm_pw<Pw64x256, DLA<64, 4>, DLB<256, 4>>(pa, pb, ep, M, N, K, splitk, stream);
  }
  typename DLA<128, 4>::P pa{A, lda, M, K, aplane}; typename DLB<128, 4>::P pb{B, ldb, N, K, bplane};
  return launch_gemm_pw<Pw128, DLA<128, 4>, DLB<128, 4>>(pa, pb, ep, M, N, K, splitk, stream);
}

// Row tile the dispatch above uses (the fused column sums come as one partial row per 64 output rows of the padded row tiling).
static int dense_planes_row_tile(int M, int N, int K, int splitk) {
  if (use_wide256(M, N, K, splitk, true) || N <= 64) return 256;
  return M <= 64 ? 64 : 128;
}
extern "C" size_t cxrk_gemm_pl_colsum_ws_bytes(int M, int N) { return (size_t)(ceil_div(M, 64) + 4) * N * sizeof(float); }

namespace {
// out[c] (+)= sum_p part[p][c]: 64 columns x 16 part lanes per block, lanes added in order (deterministic)
__global__ __launch_bounds__(1024) void colsum_parts_kernel(const float* __restrict__ part, int nparts, int N, float* __restrict__ out,
                                                            int accumulate) {
  __shared__ float sh[16][64];
  const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float a = 0.f;
  if (c < N)
    for (int p = pl; p < nparts; p += 16) a += part[(long)p * N + c];
  sh[pl][cl] = a;
  __syncthreads();
  if (pl == 0 && c < N) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += sh[i][cl];
    out[c] = accumulate ? out[c] + t : t;
  }
}
}  // namespace

// Same contraction on pre-split ("planes") operands: A and B are bf16 hi/lo plane pairs (lo plane `aplane` / `bplane` elements
// behind the hi plane), the output is either fp32 (C) or planes (Cp, `cplane`), the residual either fp32 (R) or planes (Rp).
// auxmode 2 = multiply by gelu'(aux) (aux fp32), 3 = multiply by the ReLU decision bits `maskin` (byte [row][col / 8]);
// `maskout` (with act 1) receives the decision bits of this launch's own ReLU.  `colsum` (optional, [N], no split-K): column sums of
// the stored output, reduced in the epilogue (per-wave partial rows in `ws`, cxrk_gemm_pl_colsum_ws_bytes) and finished here.
extern "C" int cxrk_gemm_pl(int transA, int transB, int M, int N, int K, const void* A, long lda, long aplane, const void* B,
                            long ldb, long bplane, float* C, void* Cp, long ldc, long cplane, const float* bias, const float* R,
                            const void* Rp, long ldr, long rplane, const float* aux, long ldaux, int auxmode,
                            const unsigned char* maskin, long ldmaskin, unsigned char* maskout, long ldmaskout, float* C2,
                            long ldc2, int act, float alpha, int accumulate, int splitk, float* colsum, int colsum_accumulate,
                            float* ws, size_t ws_bytes, hipStream_t stream) {
  CXRK_CHECK_ARG(A && B && (C || Cp) && M > 0 && N > 0 && K > 0);
  CXRK_CHECK_ARG(aligned16(A) && aligned16(B) && (lda % 8 == 0) && (ldb % 8 == 0) && (aplane % 8 == 0) && (bplane % 8 == 0));
  CXRK_CHECK_ARG(transA ? (M % 8 == 0) : (K % 8 == 0));
  CXRK_CHECK_ARG(transB ? (K % 8 == 0) : (N % 8 == 0));
  CXRK_CHECK_ARG(!(transA && transB));                      // not needed on this path
  CXRK_CHECK_ARG(auxmode == 0 || (auxmode == 2 && aux) || (auxmode == 3 && maskin));
  if (lda >= (1L << 20) || ldb >= (1L << 20)) return CXRK_ERR_UNSUPPORTED;
  if (splitk < 1) splitk = 1;
  EpiParams ep{};
  ep.alpha = alpha;
  const bool plain = !bias && !R && !Rp && !aux && !maskin && !maskout && !C2 && act == 0;
  if (colsum) {
    CXRK_CHECK_ARG(splitk == 1 && (N % 8) == 0);
    if (ws == nullptr || ws_bytes < cxrk_gemm_pl_colsum_ws_bytes(M, N)) return CXRK_ERR_WS;
  }
  if (splitk > 1) {
    CXRK_CHECK_ARG(plain && C && !Cp && (N % 4 == 0));
    if (ws == nullptr || ws_bytes < cxrk_gemm_splitk_ws_bytes(M, N, splitk)) return CXRK_ERR_WS;
    ep.C = ws; ep.ldc = N; ep.alpha = 1.f; ep.slab_stride = (long)M * N;
  } else {
    if (accumulate) { CXRK_CHECK_ARG(R == nullptr && Rp == nullptr && C); ep.R = C; ep.ldr = ldc; }
    else { ep.R = R; ep.Rp = static_cast<const unsigned short*>(Rp); ep.ldr = ldr; ep.rplane = rplane; }
    ep.C = C; ep.Cp = static_cast<unsigned short*>(Cp); ep.ldc = ldc; ep.cplane = cplane; ep.bias = bias;
    ep.aux = aux; ep.ldaux = ldaux; ep.auxmode = auxmode; ep.maskin = maskin; ep.ldmaskin = ldmaskin;
    ep.maskout = maskout; ep.ldmaskout = ldmaskout; ep.C2 = C2; ep.ldc2 = ldc2; ep.act = act;
    ep.colsum_part = colsum ? ws : nullptr;
  }
  const unsigned short* Ap = static_cast<const unsigned short*>(A);
  const unsigned short* Bp = static_cast<const unsigned short*>(B);
  int rc;
  if (!transA && transB) rc = dense_planes<DmaDenseKC, DmaDenseKC>(Ap, lda, aplane, Bp, ldb, bplane, ep, M, N, K, splitk, stream);
  else if (!transA && !transB) rc = dense_planes<DmaDenseKC, DmaDenseMC>(Ap, lda, aplane, Bp, ldb, bplane, ep, M, N, K, splitk, stream);
  else rc = dense_planes<DmaDenseMC, DmaDenseMC>(Ap, lda, aplane, Bp, ldb, bplane, ep, M, N, K, splitk, stream);
  if (colsum) {
    if (rc < 0) return rc;
    const int tm = dense_planes_row_tile(M, N, K, splitk);
    const int nparts = ceil_div(M, tm) * (tm / 64);
    hipLaunchKernelGGL(colsum_parts_kernel, dim3(ceil_div(N, 64)), dim3(1024), 0, stream, ws, nparts, N, colsum, colsum_accumulate);
    CXRK_LAUNCH_CHECK();
    return CXRK_OK;
  }
  return finish_splitk(rc, splitk, M, N, ws, C, ldc, alpha, accumulate, stream);
}

