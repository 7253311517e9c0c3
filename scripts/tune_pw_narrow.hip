// Narrow-output tiles of the LDS-DMA pipelined planes kernel (256x64 and 64x256, two blocks per CU) against the register-staged
// split-bf16 kernel on the same tiles (dense NT): correctness + timing.   tune_pw_narrow M N K   (N == 64 or M == 64)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "gemm_core.h"
using namespace cxrk;

__global__ void split_k(const float* x, long n8, unsigned short* out, long plane) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const float4 a = *reinterpret_cast<const float4*>(x + i * 8), b = *reinterpret_cast<const float4*>(x + i * 8 + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    planes_store8(out, plane, i * 8, v);
  }
}
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
  const int M = argc > 3 ? atoi(argv[1]) : 3211264, N = argc > 3 ? atoi(argv[2]) : 64, K = argc > 3 ? atoi(argv[3]) : 576;
  const size_t nA = (size_t)M * K, nB = (size_t)N * K, nC = (size_t)M * N;
  float *A, *B, *C0, *C1; unsigned short *Ap, *Bp;
  (void)hipMalloc(&A, nA * 4); (void)hipMalloc(&B, nB * 4); (void)hipMalloc(&C0, nC * 4); (void)hipMalloc(&C1, nC * 4);
  (void)hipMalloc(&Ap, nA * 4); (void)hipMalloc(&Bp, nB * 4);
  std::vector<float> h(nA > nB ? nA : nB);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 100003) / 50001.f - 1.f;
  (void)hipMemcpy(A, h.data(), nA * 4, hipMemcpyHostToDevice);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 40503u + 17) % 100019) / 50009.f - 1.f;
  (void)hipMemcpy(B, h.data(), nB * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(split_k, dim3(2048), dim3(256), 0, 0, A, (long)(nA / 8), Ap, (long)nA);
  hipLaunchKernelGGL(split_k, dim3(2048), dim3(256), 0, 0, B, (long)(nB / 8), Bp, (long)nB);
  const double fl = 2.0 * M * N * K, by = 4.0 * (nA + nB + nC);
  unsigned short* Cp; (void)hipMalloc(&Cp, nC * 4);
  EpiParams ep{}; ep.ldc = N; ep.alpha = 1.f;
  EpiParams e0 = ep; e0.C = C0; EpiParams e1 = ep; e1.C = C1; EpiParams e2 = ep; e2.Cp = Cp; e2.cplane = (long)nC;
  float t0, t1, t2;
  (void)hipMemset(C1, 0xff, nC * 4);
  if (N <= 64) {
    DenseKC<256, PL>::P pa{Ap, K, M, K, (long)nA}; DenseKC<64, PL>::P pb{Bp, K, N, K, (long)nB};
    DmaDenseKC<256, 4>::P da{Ap, K, M, K, (long)nA}; DmaDenseKC<64, 4>::P db{Bp, K, N, K, (long)nB};
    t0 = time_kernel([&] { launch_gemm<DenseKC<256, PL>, DenseKC<64, PL>, 4, 1>(pa, pb, e0, M, N, K, 1, 0); }, 10);
    t1 = time_kernel([&] { launch_gemm_pw<Pw256x64, DmaDenseKC<256, 4>, DmaDenseKC<64, 4>>(da, db, e1, M, N, K, 1, 0); }, 10);
    t2 = time_kernel([&] { launch_gemm_pw<Pw256x64, DmaDenseKC<256, 4>, DmaDenseKC<64, 4>>(da, db, e2, M, N, K, 1, 0); }, 10);
  } else {
    DenseKC<64, PL>::P pa{Ap, K, M, K, (long)nA}; DenseKC<256, PL>::P pb{Bp, K, N, K, (long)nB};
    DmaDenseKC<64, 4>::P da{Ap, K, M, K, (long)nA}; DmaDenseKC<256, 4>::P db{Bp, K, N, K, (long)nB};
    t0 = time_kernel([&] { launch_gemm<DenseKC<64, PL>, DenseKC<256, PL>, 1, 4>(pa, pb, e0, M, N, K, 1, 0); }, 10);
    t1 = time_kernel([&] { launch_gemm_pw<Pw64x256, DmaDenseKC<64, 4>, DmaDenseKC<256, 4>>(da, db, e1, M, N, K, 1, 0); }, 10);
    t2 = time_kernel([&] { launch_gemm_pw<Pw64x256, DmaDenseKC<64, 4>, DmaDenseKC<256, 4>>(da, db, e2, M, N, K, 1, 0); }, 10);
  }
  std::vector<float> c0(nC), c1(nC);
  (void)hipMemcpy(c0.data(), C0, nC * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(c1.data(), C1, nC * 4, hipMemcpyDeviceToHost);
  double md = 0, mx = 0; size_t bad = 0;
  for (size_t i = 0; i < nC; ++i) { const double d = fabs((double)c0[i] - c1[i]); if (!(d <= 1e-3)) ++bad; md = fmax(md, d); mx = fmax(mx, fabs((double)c0[i])); }
  printf("NT %dx%dx%d  regstage %.3f ms %.0f TF %.2f TB/s | DMA-pipelined (2 blocks/CU) fp32 out %.3f ms %.0f TF %.2f TB/s | planes out %.3f ms %.2f TB/s | max diff %.3g bad %zu (max|c| %.3g)\n",
         M, N, K, t0, fl / t0 / 1e9, by / t0 / 1e9, t1, fl / t1 / 1e9, by / t1 / 1e9, t2, by / t2 / 1e9, md, bad, mx);
  return bad != 0;
}
