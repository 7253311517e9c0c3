// Tuning driver for the split-bf16 mainloops (dense NT, 128x128 tile): times the single-buffer kernel and the pipelined
// kernel on the same operands in one process and checks that they agree.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -I<csrc> [-DCXRK_PIPE_SCHED=0] scripts/tune_gemm.hip -o build/tune
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
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main(int argc, char** argv) {
  const int M = argc > 3 ? atoi(argv[1]) : 32768, N = argc > 3 ? atoi(argv[2]) : 3072, K = argc > 3 ? atoi(argv[3]) : 768;
  float *A, *B, *C0, *C1;
  hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&B, (size_t)N * K * 4); hipMalloc(&C0, (size_t)M * N * 4); hipMalloc(&C1, (size_t)M * N * 4);
  std::vector<float> h((size_t)(M > N ? M : N) * K);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
  hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 40503u + 17) % 1000) / 500.f - 1.f;
  hipMemcpy(B, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
  EpiParams ep{}; ep.ldc = N; ep.alpha = 1.f; ep.vec = 1;
  DenseKC<128>::P pa{A, K, M, K}; DenseKC<128>::P pb{B, K, N, K};
  const int nMt = ceil_div(M, 128), nNt = ceil_div(N, 128);
  dim3 grid(nMt * nNt, 1, 1);
  const double fl = 2.0 * M * N * K;
  EpiParams e0 = ep; e0.C = C0;
  EpiParams e1 = ep; e1.C = C1;
  const float t0 = time_kernel([&] { hipLaunchKernelGGL((gemm_x3_kernel<DenseKC<128>, DenseKC<128>, 2, 2>), grid, dim3(NTHREADS), 0, 0, pa, pb, e0, M, N, K, nMt, nNt, K); }, 20);
  const float t1 = time_kernel([&] { hipLaunchKernelGGL((gemm_x3p_kernel<DenseKC<128>, DenseKC<128>, 2, 2>), grid, dim3(NTHREADS), 0, 0, pa, pb, e1, M, N, K, nMt, nNt, K); }, 20);
  std::vector<float> c0((size_t)M * N), c1((size_t)M * N);
  hipMemcpy(c0.data(), C0, c0.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(c1.data(), C1, c1.size() * 4, hipMemcpyDeviceToHost);
  double md = 0, mx = 0;
  for (size_t i = 0; i < c0.size(); ++i) { md = fmax(md, fabs((double)c0[i] - c1[i])); mx = fmax(mx, fabs((double)c0[i])); }
  printf("%dx%dx%d  single-buffer %.3f ms %.1f TFLOP/s | pipelined(sched=%d) %.3f ms %.1f TFLOP/s | max|diff| %.3g (max|c| %.3g)\n", M, N, K,
         t0, fl / t0 / 1e9, CXRK_PIPE_SCHED, t1, fl / t1 / 1e9, md, mx);
  return 0;
}
