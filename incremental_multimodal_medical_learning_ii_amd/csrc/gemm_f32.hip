// Dense GEMM entry point on fp32 tensors (Linear fprop / dgrad / wgrad of the adapters, heads and of the encoders in exact-fp32 mode).
#include "gemm_common.h"

using namespace cxrk;

extern "C" int cxrk_gemm_f32(int transA, int transB, int M, int N, int K, const float* A, long lda, const float* B,
                             long ldb, float* C, long ldc, const float* bias, const float* R, long ldr,
                             const float* aux, long ldaux, int auxmode, float* C2, long ldc2, int act, float alpha,
                             int accumulate, int splitk, float* ws, size_t ws_bytes, hipStream_t stream) {
  CXRK_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0);
  CXRK_CHECK_ARG(aligned16(A) && aligned16(B) && (lda % 4 == 0) && (ldb % 4 == 0));
  CXRK_CHECK_ARG(transA ? (M % 4 == 0) : (K % 4 == 0));
  CXRK_CHECK_ARG(transB ? (K % 4 == 0) : (N % 4 == 0));
  CXRK_CHECK_ARG(!(auxmode != 0 && aux == nullptr) && auxmode >= 0 && auxmode <= 2);
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
  if (!transA && transB) rc = dense_small<F32, DenseKC, DenseKC>(A, lda, 0, B, ldb, 0, ep, M, N, K, splitk, stream);
  else if (!transA && !transB) rc = dense_small<F32, DenseKC, DenseMC>(A, lda, 0, B, ldb, 0, ep, M, N, K, splitk, stream);
  else if (transA && !transB) rc = dense_small<F32, DenseMC, DenseMC>(A, lda, 0, B, ldb, 0, ep, M, N, K, splitk, stream);
  else rc = dense_small<F32, DenseMC, DenseKC>(A, lda, 0, B, ldb, 0, ep, M, N, K, splitk, stream);
  return finish_splitk(rc, splitk, M, N, ws, C, ldc, alpha, accumulate, stream);
}

