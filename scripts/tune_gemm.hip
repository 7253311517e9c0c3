// Tuning driver for the split-bf16 mainloops (dense NT): times the 128x128 kernel (3 blocks per CU) and the 256x256 kernel
// (1 block per CU, software-pipelined) on the same operands in one process and checks that they agree.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -I<csrc> [-DCXRK_ABL=n] [-DCXRK_STAGGER=n] scripts/tune_gemm.hip -o build/tune
//   build/tune [M N K]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "gemm_core.h"
using namespace cxrk;

template <class KERN>
static float time_kernel(KERN launch, int reps) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) launch();
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main(int argc, char** argv) {
  const int M = argc > 3 ? atoi(argv[1]) : 32768, N = argc > 3 ? atoi(argv[2]) : 3072, K = argc > 3 ? atoi(argv[3]) : 768;
  float *A, *B, *C0, *C2;
  (void)hipMalloc(&A, (size_t)M * K * 4); (void)hipMalloc(&B, (size_t)N * K * 4);
  (void)hipMalloc(&C0, (size_t)M * N * 4); (void)hipMalloc(&C2, (size_t)M * N * 4);
  std::vector<float> h((size_t)(M > N ? M : N) * K);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
  (void)hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 40503u + 17) % 1000) / 500.f - 1.f;
  (void)hipMemcpy(B, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
  EpiParams ep{}; ep.ldc = N; ep.alpha = 1.f; ep.vec = 1; ep.nt = stream_output(M, N, 1);
  const double fl = 2.0 * M * N * K;
  EpiParams e0 = ep; e0.C = C0;
  EpiParams e2 = ep; e2.C = C2;
  DenseKC<128>::P pa{A, K, M, K}; DenseKC<128>::P pb{B, K, N, K};
  const int nMt = ceil_div(M, 128), nNt = ceil_div(N, 128);
  const float t0 = time_kernel([&] { hipLaunchKernelGGL((gemm_x3_kernel<DenseKC<128>, DenseKC<128>, 2, 2>), dim3(nMt * nNt), dim3(NTHREADS), 0, 0, pa, pb, e0, M, N, K, nMt, nNt, K); }, 20);
  DenseKC<256, NT_WIDE>::P pa2{A, K, M, K}; DenseKC<256, NT_WIDE>::P pb2{B, K, N, K};
  const int nMt2 = ceil_div(M, 256), nNt2 = ceil_div(N, 256);
  const float t2 = time_kernel([&] { hipLaunchKernelGGL((gemm_x3w_kernel<DenseKC<256, NT_WIDE>, DenseKC<256, NT_WIDE>>), dim3(nMt2 * nNt2), dim3(NT_WIDE), 0, 0, pa2, pb2, e2, M, N, K, nMt2, nNt2, K); }, 20);
  std::vector<float> c0((size_t)M * N), c2((size_t)M * N);
  (void)hipMemcpy(c0.data(), C0, c0.size() * 4, hipMemcpyDeviceToHost);
  (void)hipMemcpy(c2.data(), C2, c2.size() * 4, hipMemcpyDeviceToHost);
  double md = 0, mx = 0;
  for (size_t i = 0; i < c0.size(); ++i) { md = fmax(md, fabs((double)c0[i] - c2[i])); mx = fmax(mx, fabs((double)c0[i])); }
  printf("%dx%dx%d  128x128 %.3f ms %.1f TFLOP/s | 256x256 %.3f ms %.1f TFLOP/s | max|diff| %.3g (max|c| %.3g)\n", M, N, K,
         t0, fl / t0 / 1e9, t2, fl / t2 / 1e9, md, mx);
  return 0;
}
