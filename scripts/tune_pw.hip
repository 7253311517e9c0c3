// Experimental driver for the LDS-DMA, software-pipelined 256x256 planes mainloop (dense NT): correctness against the
// register-staged planes kernel + timing.   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -I<csrc> scripts/tune_pw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
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
  const int M = argc > 3 ? atoi(argv[1]) : 32768, N = argc > 3 ? atoi(argv[2]) : 3072, K = argc > 3 ? atoi(argv[3]) : 768;
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
  const double fl = 2.0 * M * N * K;
  EpiParams ep{}; ep.ldc = N; ep.alpha = 1.f;
  EpiParams e0 = ep; e0.C = C0; EpiParams e1 = ep; e1.C = C1;
  DenseKC<128, PL>::P pa{Ap, K, M, K, (long)nA}; DenseKC<128, PL>::P pb{Bp, K, N, K, (long)nB};
  DmaDenseKC<128, 4>::P qa{Ap, K, M, K, (long)nA}; DmaDenseKC<128, 4>::P qb{Bp, K, N, K, (long)nB};
  DmaDenseKC<256, 8>::P da{Ap, K, M, K, (long)nA}; DmaDenseKC<256, 8>::P db{Bp, K, N, K, (long)nB};
  const float t0 = time_kernel([&] { launch_gemm<DenseKC<128, PL>, DenseKC<128, PL>, 2, 2>(pa, pb, e0, M, N, K, 1, 0); }, 10);
  (void)hipMemset(C1, 0xff, nC * 4);
  const float t1 = time_kernel([&] { launch_gemm_pw<Pw256, DmaDenseKC<256, 8>, DmaDenseKC<256, 8>>(da, db, e1, M, N, K, 1, 0); }, 10);
  float* C2; (void)hipMalloc(&C2, nC * 4); (void)hipMemset(C2, 0xff, nC * 4);
  EpiParams e4 = ep; e4.C = C2;
  const float t2 = time_kernel([&] { launch_gemm_pw<Pw128, DmaDenseKC<128, 4>, DmaDenseKC<128, 4>>(qa, qb, e4, M, N, K, 1, 0); }, 10);
  auto stamp_run = [&](const char* what, EpiParams ebase) {  // where a tile's time goes: s_memtime stamps of wave 0 (diagnostic launch, not timed)
    const int nblk = ((M + 255) / 256) * ((N + 255) / 256);
    unsigned long long* st; (void)hipMalloc(&st, (size_t)nblk * 64); (void)hipMemset(st, 0, (size_t)nblk * 64);
    EpiParams es = ebase; es.stamps = st;
    launch_gemm_pw<Pw256, DmaDenseKC<256, 8>, DmaDenseKC<256, 8>>(da, db, es, M, N, K, 1, 0);
    std::vector<unsigned long long> hs((size_t)nblk * 8);
    (void)hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> d[5];
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int b = 0; b < nblk; ++b) {
      const unsigned long long* q = &hs[(size_t)b * 8];
      d[0].push_back((double)(q[1] - q[0])); d[1].push_back((double)(q[2] - q[1])); d[2].push_back((double)(q[3] - q[2])); d[3].push_back((double)(q[4] - q[3]));
      d[4].push_back((double)(q[4] - q[0]) / ((double)(q[6] - q[5]) * 10.0));   // cycles per ns -> GHz (s_memrealtime ticks at 100 MHz)
      tmin = q[5] < tmin ? q[5] : tmin; tmax = q[6] > tmax ? q[6] : tmax;
    }
    auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    const double nk = (K + 31) / 32;
    printf("  [%s] stamps (median over %d blocks, shader cycles): prologue %.0f | mainloop %.0f (%.0f per K-tile) | epilogue issue %.0f | store drain %.0f | clock %.2f GHz | kernel span %.1f us\n",
           what, nblk, med(d[0]), med(d[1]), med(d[1]) / nk, med(d[2]), med(d[3]), med(d[4]), (double)(tmax - tmin) / 100.0);
    (void)hipFree(st);
  };
  stamp_run("fp32 out", e1);
  { unsigned short* Cp; (void)hipMalloc(&Cp, nC * 4); EpiParams e2 = ep; e2.Cp = Cp; e2.cplane = (long)nC; stamp_run("planes out", e2);
    float* bias; (void)hipMalloc(&bias, (size_t)N * 4); (void)hipMemset(bias, 0, (size_t)N * 4);
    EpiParams e3 = e2; e3.bias = bias; e3.act = 2; e3.C2 = C0; e3.ldc2 = N; stamp_run("planes out + bias + gelu + preact copy", e3); (void)hipFree(Cp); }
  std::vector<float> c0(nC), c1(nC), c2(nC);
  (void)hipMemcpy(c0.data(), C0, nC * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(c1.data(), C1, nC * 4, hipMemcpyDeviceToHost);
  (void)hipMemcpy(c2.data(), C2, nC * 4, hipMemcpyDeviceToHost);
  double md = 0, mx = 0; size_t bad = 0;
  for (size_t i = 0; i < nC; ++i) {
    const double d = fmax(fabs((double)c0[i] - c1[i]), fabs((double)c0[i] - c2[i]));
    if (!(d <= 1e-3)) ++bad; md = fmax(md, d); mx = fmax(mx, fabs((double)c0[i]));
  }
  printf("NT %dx%dx%d  planes regstage 128^2 %.3f ms %.0f TF | DMA-pipelined 256^2 %.3f ms %.0f TF | DMA-pipelined 128^2 (2 blocks/CU) %.3f ms %.0f TF | max diff %.3g bad %zu (max|c| %.3g)\n",
         M, N, K, t0, fl / t0 / 1e9, t1, fl / t1 / 1e9, t2, fl / t2 / 1e9, md, bad, mx);
  return bad != 0;
}
