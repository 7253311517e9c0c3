// Ablation driver for the MFMA mainloop (dense NT, 128x128 tile).  Build with -DCXRK_ABL=<n>:
//  0 full kernel  1 no global loads in the loop  2 no LDS stores in the loop  3 no barriers in the loop (results wrong)
//  4 MFMAs only (operands from registers, no LDS reads)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "gemm_core.h"
using namespace cxrk;
int main(int argc, char** argv) {
  const int M = 32768, N = 3072, K = 768;
  float *A, *B, *C;
  hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&B, (size_t)N * K * 4); hipMalloc(&C, (size_t)M * N * 4);
  std::vector<float> h((size_t)M * K);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
  hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice);
  hipMemcpy(B, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
  EpiParams ep{}; ep.C = C; ep.ldc = N; ep.alpha = 1.f;
#ifdef CXRK_TUNE_SPLIT
  gemm_precision_mode() = 1;
#endif
  DenseKC<128>::P pa{A, K, M, K}; DenseKC<128>::P pb{B, K, N, K};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch_gemm<DenseKC<128>, DenseKC<128>, 2, 2>(pa, pb, ep, M, N, K, 1, 0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) launch_gemm<DenseKC<128>, DenseKC<128>, 2, 2>(pa, pb, ep, M, N, K, 1, 0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  printf("ABL=%d  %.3f ms  %.1f TFLOP/s\n", CXRK_ABL, ms, 2.0 * M * N * K / ms / 1e9);
  return 0;
}
