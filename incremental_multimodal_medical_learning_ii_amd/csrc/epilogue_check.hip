// Stand-alone check of the epilogue lane maps (gemm_epilogue.h) on small / ragged shapes, with guard regions around every
// buffer: a store that leaves its tensor lands in a guard and is REPORTED instead of faulting.  Exact-fp32 mainloop, CPU fp64
// reference.   Built by the Makefile next to the library (`epilogue_check`); `epilogue_check M N K` runs one shape verbosely.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
#include "gemm_core.h"
using namespace cxrk;

static const size_t GUARD = 1 << 20;   // floats on either side
struct Buf {
  float* d = nullptr; size_t n = 0; std::vector<float> h;
  void alloc(size_t n_, bool fill_random, unsigned seed) {
    n = n_; h.assign(n + 2 * GUARD, 0.f);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < h.size(); ++i) {
      s = s * 1664525u + 1013904223u;
      h[i] = (i < GUARD || i >= GUARD + n || !fill_random) ? -12345.f : (float)((int)(s >> 9) % 2001 - 1000) / 1000.f;
    }
    (void)hipMalloc(&d, h.size() * 4); (void)hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  }
  float* ptr() const { return d + GUARD; }
  const float* host() const { return h.data() + GUARD; }
  std::vector<float> back() const { std::vector<float> o(n + 2 * GUARD); (void)hipMemcpy(o.data(), d, o.size() * 4, hipMemcpyDeviceToHost); return o; }
};

static int run_case(int M, int N, int K, long ldc, bool bias, bool res, int act, bool inplace, bool auxsign = false, bool colsum = false, bool c2 = false) {
  Buf A, B, C, R, Bi;
  A.alloc((size_t)M * K, true, 1); B.alloc((size_t)N * K, true, 2); C.alloc((size_t)M * ldc, false, 3);
  R.alloc((size_t)M * ldc, true, 4); Bi.alloc((size_t)N, true, 5);
  if (inplace) (void)hipMemcpy(C.ptr(), R.host(), (size_t)M * ldc * 4, hipMemcpyHostToDevice);
  EpiParams ep{}; ep.C = C.ptr(); ep.ldc = ldc; ep.alpha = 0.5f; ep.act = act;
  if (bias) ep.bias = Bi.ptr();
  if (res) { ep.R = inplace ? C.ptr() : R.ptr(); ep.ldr = ldc; }
  Buf AX, CS, C2;
  const int tm = N <= 64 ? 256 : (M <= 64 ? 64 : 128);
  const int parts = ((M + tm - 1) / tm) * (tm / 64);
  AX.alloc((size_t)M * ldc, true, 6); CS.alloc((size_t)parts * N, false, 7); C2.alloc((size_t)M * ldc, false, 8);
  if (auxsign) { ep.aux = AX.ptr(); ep.ldaux = ldc; ep.auxmode = 1; }
  if (colsum) ep.colsum_part = CS.ptr();
  if (c2) { ep.C2 = C2.ptr(); ep.ldc2 = ldc; }
  int rc;
  if (N <= 64) { DenseKC<256>::P pa{A.ptr(), K, M, K, 0}; DenseKC<64>::P pb{B.ptr(), K, N, K, 0}; rc = launch_gemm<DenseKC<256>, DenseKC<64>, 4, 1>(pa, pb, ep, M, N, K, 1, 0); }
  else if (M <= 64) { DenseKC<64>::P pa{A.ptr(), K, M, K, 0}; DenseKC<256>::P pb{B.ptr(), K, N, K, 0}; rc = launch_gemm<DenseKC<64>, DenseKC<256>, 1, 4>(pa, pb, ep, M, N, K, 1, 0); }
  else { DenseKC<128>::P pa{A.ptr(), K, M, K, 0}; DenseKC<128>::P pb{B.ptr(), K, N, K, 0}; rc = launch_gemm<DenseKC<128>, DenseKC<128>, 2, 2>(pa, pb, ep, M, N, K, 1, 0); }
  if (hipDeviceSynchronize() != hipSuccess || rc < 0) { printf("  launch failed rc=%d\n", rc); return 1; }
  const std::vector<float> out = C.back();
  int bad = 0; size_t guard_hits = 0; double md = 0;
  std::vector<double> cs((size_t)N, 0.0);
  const std::vector<float> out2 = C2.back(), outcs = CS.back();
  for (size_t i = 0; i < GUARD; ++i) guard_hits += (out2[i] != -12345.f) + (out2[GUARD + C2.n + i] != -12345.f) + (outcs[i] != -12345.f) + (outcs[GUARD + CS.n + i] != -12345.f);
  for (size_t i = 0; i < GUARD; ++i) guard_hits += (out[i] != -12345.f) + (out[GUARD + C.n + i] != -12345.f);
  for (int i = 0; i < M; ++i)
    for (long j = 0; j < ldc; ++j) {
      const float got = out[GUARD + (size_t)i * ldc + j];
      if (j >= N) { if (got != (inplace ? R.host()[(size_t)i * ldc + j] : -12345.f)) ++guard_hits; continue; }   // padding columns of a strided output
      double s = 0;
      for (int k = 0; k < K; ++k) s += (double)A.host()[(size_t)i * K + k] * B.host()[(size_t)j * K + k];
      s = 0.5 * s + (bias ? Bi.host()[j] : 0.0) + (res ? R.host()[(size_t)i * ldc + j] : 0.0);
      if (c2) { const double d2 = fabs(s - out2[GUARD + (size_t)i * ldc + j]); md = fmax(md, d2); if (!(d2 < 1e-3)) ++bad; }
      if (act == 1) s = s > 0 ? s : 0;
      if (act == 2) s = 0.5 * s * (1.0 + erf(s * 0.7071067811865476));
      if (auxsign) s = AX.host()[(size_t)i * ldc + j] > 0.f ? s : 0.0;
      cs[j] += s;
      const double d = fabs(s - got); md = fmax(md, d);
      if (!(d < 1e-3)) { if (bad < 12 && getenv("EPI_VERBOSE")) printf("    wrong C[%d][%ld] = %g, expected %g\n", i, j, got, s); ++bad; }
    }
  if (getenv("EPI_VERBOSE")) {
    size_t gc = 0, g2 = 0, gs = 0; long firstc = -1;
    for (size_t i = 0; i < GUARD; ++i) { if (out[i] != -12345.f || out[GUARD + C.n + i] != -12345.f) { ++gc; if (firstc < 0) firstc = (out[GUARD + C.n + i] != -12345.f) ? (long)i : -(long)(GUARD - i); } g2 += (out2[i] != -12345.f) + (out2[GUARD + C2.n + i] != -12345.f); gs += (outcs[i] != -12345.f) + (outcs[GUARD + CS.n + i] != -12345.f); }
    printf("    guard hits: C %zu (first at offset %ld past the end), C2 %zu, colsum %zu\n", gc, firstc, g2, gs);
  }
  if (colsum)
    for (int j = 0; j < N; ++j) {
      double t = 0;
      for (int p_ = 0; p_ < parts; ++p_) { const float v = outcs[GUARD + (size_t)p_ * N + j]; if (v != -12345.f) t += v; }
      const double d = fabs(t - cs[j]); md = fmax(md, d);
      if (!(d < 1e-2)) { if (getenv("EPI_VERBOSE") && bad < 24) printf("    wrong colsum[%d] = %g, expected %g\n", j, t, cs[j]); ++bad; }
    }
  auto chk = [&](const Buf& b, const char* nm) { const std::vector<float> o = b.back(); size_t hits = 0; for (size_t i = 0; i < o.size(); ++i) hits += memcmp(&o[i], &b.h[i], 4) != 0; if (hits) printf("  INPUT %s modified in %zu places\n", nm, hits); return hits; };
  size_t in_hits = chk(A, "A") + chk(B, "B") + chk(Bi, "bias") + (inplace ? 0 : chk(R, "R"));
  printf("%s M=%d N=%d K=%d ldc=%ld bias=%d res=%d(inplace %d) act=%d: max err %.2e, wrong %d, stray writes %zu\n",
         (bad || guard_hits || in_hits) ? "FAIL" : "ok  ", M, N, K, ldc, bias, res, inplace, act, md, bad, guard_hits + in_hits);
  (void)hipFree(A.d); (void)hipFree(B.d); (void)hipFree(C.d); (void)hipFree(R.d); (void)hipFree(Bi.d); (void)hipFree(AX.d); (void)hipFree(CS.d); (void)hipFree(C2.d);
  return (bad || guard_hits || in_hits) ? 1 : 0;
}

int main(int argc, char** argv) {
  int fails = 0;
  if (argc > 3) {   // one shape, the conv data-gradient + identity kind, verbose
    setenv("EPI_VERBOSE", "1", 1);
    return run_case(atoi(argv[1]), atoi(argv[2]), atoi(argv[3]), atoi(argv[2]), false, true, 0, false, true, true);
  }
  const int shapes[][3] = {{4, 8, 128}, {4, 128, 8}, {16, 2048, 512}, {70, 40, 64}, {8, 128, 128}, {130, 136, 72}, {64, 64, 32}, {300, 72, 40}, {16, 512, 2048}};
  for (auto& s : shapes)
    for (int cfg = 0; cfg < 6; ++cfg) {
      const bool bias = cfg == 1 || cfg == 2 || cfg == 4, res = cfg == 2 || cfg == 3 || cfg == 5, inplace = cfg == 5;
      const int act = cfg == 4 ? 1 : 0;
      fails += run_case(s[0], s[1], s[2], s[1], bias, res, act, inplace);
      if (cfg == 3) fails += run_case(s[0], s[1], s[2], (long)s[1] * 3 + 8, bias, res, act, false);   // strided output rows
      if (cfg == 0) {
        fails += run_case(s[0], s[1], s[2], s[1], false, false, 0, false, true, true);    // conv data gradient, fp32 mode (kind 14)
        fails += run_case(s[0], s[1], s[2], s[1], false, true, 0, false, true, true);     // + identity gradient (kind 15)
        fails += run_case(s[0], s[1], s[2], s[1], true, false, 2, false, false, false, true);   // FFN up, fp32 mode (kind 17)
        fails += run_case(s[0], s[1], s[2], s[1], false, false, 0, false, false, true);   // generic: plain + column sums
      }
    }
  printf("%d failing cases\n", fails);
  return fails != 0;
}
