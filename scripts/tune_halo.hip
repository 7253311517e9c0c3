// Stand-alone timing of the window-resident 3x3 kernel (csrc/conv_halo.h) against the implicit-GEMM form on the same layer, with
// s_memtime stamps of its phases.   tune_halo [N images] [H] [W]     (64 -> 64 channels, planes operands, planes output)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
#include "conv_common.h"
#include "conv_halo.h"
using namespace cxrk;

__global__ void fill_planes(unsigned short* out, long plane, long n8, unsigned seed, float scale) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    float v[8];
    for (int q = 0; q < 8; ++q) { unsigned h = (unsigned)(i * 8 + q) * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; v[q] = ((float)(h & 0xffff) / 32768.f - 1.f) * scale; }
    planes_store8(out, plane, i * 8, v);
  }
}
static double pl_value(unsigned short h, unsigned short l) {
  const unsigned u = (unsigned)h << 16, v = (unsigned)l << 16; float p, q; std::memcpy(&p, &u, 4); std::memcpy(&q, &v, 4); return (double)p + q;
}
template <class F> static float time_it(F f, int reps) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 2; ++i) f();
  (void)hipDeviceSynchronize(); (void)hipEventRecord(a);
  for (int i = 0; i < reps; ++i) f();
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main(int argc, char** argv) {
  const int NI = argc > 1 ? atoi(argv[1]) : 1024, H = argc > 2 ? atoi(argv[2]) : 56, W = argc > 3 ? atoi(argv[3]) : 56;
  const int C = 64, Ko = 64, K = 9 * C;
  const long Ml = (long)NI * H * W; const int M = (int)Ml;
  const size_t nx = (size_t)M * C, nw = (size_t)Ko * K, ny = (size_t)M * Ko;
  unsigned short *x, *w, *y0, *y1;
  (void)hipMalloc(&x, nx * 4); (void)hipMalloc(&w, nw * 4); (void)hipMalloc(&y0, ny * 4); (void)hipMalloc(&y1, ny * 4);
  hipLaunchKernelGGL(fill_planes, dim3(2048), dim3(256), 0, 0, x, (long)nx, (long)(nx / 8), 1u, 1.0f);
  hipLaunchKernelGGL(fill_planes, dim3(64), dim3(256), 0, 0, w, (long)nw, (long)(nw / 8), 7u, 0.05f);
  const ConvGeom g = make_geom(NI, H, W, C, Ko, 3, 3, 1, 1);
  EpiParams e0{}; e0.ldc = Ko; e0.alpha = 1.f; e0.Cp = y0; e0.cplane = (long)ny;
  EpiParams e1 = e0; e1.Cp = y1;
  DmaConvIm2colKC<256, 4>::P pa{x, g, M, K, (long)nx}; DmaDenseKC<64, 4>::P pb{w, (long)K, Ko, K, (long)nw};
  const float t0 = time_it([&] { launch_gemm_pw<Pw256x64, DmaConvIm2colKC<256, 4>, DmaDenseKC<64, 4>>(pa, pb, e0, M, Ko, K, 1, 0); }, 10);
  const float t1 = time_it([&] { launch_conv3x3_halo<DmaDenseKC<64, 4>, false>(x, (long)nx, pb, e1, M, Ko, H, W, 0); }, 10);
  const double fl = 2.0 * M * Ko * (double)K;
  std::vector<unsigned short> a(ny * 2), b(ny * 2);
  (void)hipMemcpy(a.data(), y0, ny * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(b.data(), y1, ny * 4, hipMemcpyDeviceToHost);
  size_t bad = 0; double md = 0;
  for (size_t i = 0; i < ny; ++i) {
    const double dd = fabs(pl_value(a[i], a[ny + i]) - pl_value(b[i], b[ny + i]));
    md = fmax(md, dd); if (!(dd <= 1e-4)) ++bad;
  }
  // stamps, for the plain planes store and for the two epilogues of the training step
  const int nblk = ceil_div(M, HALO_TM);
  unsigned long long* st; (void)hipMalloc(&st, (size_t)nblk * 64);
  float* bias; (void)hipMalloc(&bias, 64 * 4); (void)hipMemset(bias, 0, 64 * 4);
  unsigned char *mout, *min_; (void)hipMalloc(&mout, (size_t)M * 8); (void)hipMalloc(&min_, (size_t)M * 8); (void)hipMemset(min_, 0x5a, (size_t)M * 8);
  float* parts; (void)hipMalloc(&parts, ((size_t)nblk * 8 + 64) * 64 * 4);
  auto stamp_run = [&](const char* what, EpiParams es) {
    (void)hipMemset(st, 0, (size_t)nblk * 64);
    es.stamps = nullptr;
    const float t = time_it([&] { launch_conv3x3_halo<DmaDenseKC<64, 4>, false>(x, (long)nx, pb, es, M, Ko, H, W, 0); }, 10);
    es.stamps = st;
    launch_conv3x3_halo<DmaDenseKC<64, 4>, false>(x, (long)nx, pb, es, M, Ko, H, W, 0);
    std::vector<unsigned long long> hs((size_t)nblk * 8);
    (void)hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> d[5];
    for (int b = 0; b < nblk; ++b) {
      const unsigned long long* q = &hs[(size_t)b * 8];
      for (int i = 0; i < 4; ++i) d[i].push_back((double)(q[i + 1] - q[i]));
      d[4].push_back((double)(q[4] - q[0]) / ((double)(q[6] - q[5]) * 10.0));
    }
    auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    printf("  [%s] %.3f ms | stamps (median over %d blocks, shader cycles): prologue %.0f | mainloop %.0f (%.0f per K-tile) | epilogue issue %.0f | store drain %.0f | clock %.2f GHz\n",
           what, t, nblk, med(d[0]), med(d[1]), med(d[1]) / 18.0, med(d[2]), med(d[3]), med(d[4]));
  };
  stamp_run("planes out", e1);
  { EpiParams e2 = e1; e2.bias = bias; e2.act = 1; e2.maskout = mout; e2.ldmaskout = Ko / 8; stamp_run("forward: shift + ReLU + decision bits", e2); }
  { EpiParams e3 = e1; e3.maskin = min_; e3.ldmaskin = Ko / 8; e3.auxmode = 3; e3.colsum_part = parts; stamp_run("data gradient: mask + column sums", e3); }
  printf("3x3 s1 64->64 %dx%dx%d: implicit GEMM 256x64 %.3f ms %.0f TF | window-resident %.3f ms %.0f TF | max diff %.3g bad %zu\n", NI, H, W, t0,
         fl / t0 / 1e9, t1, fl / t1 / 1e9, md, bad);
  return bad != 0;
}
