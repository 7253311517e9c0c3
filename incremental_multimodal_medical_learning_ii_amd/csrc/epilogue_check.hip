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


// ---- planes operands / outputs (the split-bf16 product path: gemm_pw kernels) -------------------------------------------------------
static unsigned short bf16_rne(float x) { unsigned u; memcpy(&u, &x, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }
static float bf16_f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
struct PBuf {   // a [2][n] bf16 planes tensor with guards around the whole allocation and BETWEEN the planes
  unsigned short* d = nullptr; size_t n = 0; std::vector<unsigned short> h; std::vector<float> val;
  static constexpr unsigned short SENT = 0xBEEF;
  size_t plane() const { return n + GUARD; }           // elements from the hi plane to the lo plane
  void alloc(size_t n_, bool fill_random, unsigned seed) {
    n = n_; h.assign(3 * GUARD + 2 * n, SENT); val.assign(n, 0.f);
    unsigned s = seed * 2654435761u + 99u;
    if (fill_random)
      for (size_t i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        const float x = (float)((int)(s >> 9) % 2001 - 1000) / 1000.f;
        const unsigned short hi = bf16_rne(x), lo = bf16_rne(x - bf16_f(hi));
        h[GUARD + i] = hi; h[GUARD + plane() + i] = lo; val[i] = bf16_f(hi) + bf16_f(lo);
      }
    (void)hipMalloc(&d, h.size() * 2); (void)hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  }
  unsigned short* ptr() const { return d + GUARD; }
  std::vector<unsigned short> back() const { std::vector<unsigned short> o(h.size()); (void)hipMemcpy(o.data(), d, o.size() * 2, hipMemcpyDeviceToHost); return o; }
  // stray writes outside the two planes (only meaningful for an output that was allocated with fill_random = false)
  static size_t guard_hits(const std::vector<unsigned short>& o, size_t n) {
    size_t hits = 0;
    for (size_t i = 0; i < GUARD; ++i) hits += (o[i] != SENT) + (o[GUARD + n + i] != SENT) + (o[2 * GUARD + 2 * n + i] != SENT);
    return hits;
  }
};

// out = act(0.5 * A B^T + bias + res) [* mask bits]; planes or fp32 output; optional ReLU decision bits out, column sums
static int run_planes_case(int M, int N, int K, bool outpl, bool bias, bool respl, int act, bool maskin, bool maskout, bool colsum) {
  PBuf A, B, Cp, R; Buf C, Bi, CS;
  A.alloc((size_t)M * K, true, 11); B.alloc((size_t)N * K, true, 12); Cp.alloc((size_t)M * N, false, 13); R.alloc((size_t)M * N, true, 14);
  C.alloc((size_t)M * N, false, 15); Bi.alloc((size_t)N, true, 16);
  const bool wide = M >= 256 && N >= 256;
  const int tm = wide ? 256 : (N <= 64 ? 256 : (M <= 64 ? 64 : 128));
  const int parts = ((M + tm - 1) / tm) * (tm / 64);
  CS.alloc((size_t)parts * N, false, 17);
  const size_t mb = (size_t)M * (N / 8);
  std::vector<unsigned char> hmin(mb + 2 * GUARD, 0xA5), hmout(mb + 2 * GUARD, 0xA5);
  unsigned s = 777u;
  for (size_t i = 0; i < mb; ++i) { s = s * 1664525u + 1013904223u; hmin[GUARD + i] = (unsigned char)(s >> 13); }
  unsigned char *dmin, *dmout;
  (void)hipMalloc(&dmin, hmin.size()); (void)hipMalloc(&dmout, hmout.size());
  (void)hipMemcpy(dmin, hmin.data(), hmin.size(), hipMemcpyHostToDevice); (void)hipMemcpy(dmout, hmout.data(), hmout.size(), hipMemcpyHostToDevice);
  EpiParams ep{}; ep.ldc = N; ep.alpha = 0.5f; ep.act = act;
  if (outpl) { ep.Cp = Cp.ptr(); ep.cplane = (long)Cp.plane(); } else ep.C = C.ptr();
  if (bias) ep.bias = Bi.ptr();
  if (respl) { ep.Rp = R.ptr(); ep.ldr = N; ep.rplane = (long)R.plane(); }
  if (maskin) { ep.maskin = dmin + GUARD; ep.ldmaskin = N / 8; ep.auxmode = 3; }
  if (maskout) { ep.maskout = dmout + GUARD; ep.ldmaskout = N / 8; }
  if (colsum) ep.colsum_part = CS.ptr();
  int rc;
  const long ap = (long)A.plane(), bp = (long)B.plane();
  if (wide) { DmaDenseKC<256, 8>::P pa{A.ptr(), K, M, K, ap}; DmaDenseKC<256, 8>::P pb{B.ptr(), K, N, K, bp}; rc = launch_gemm_pw<Pw256, DmaDenseKC<256, 8>, DmaDenseKC<256, 8>>(pa, pb, ep, M, N, K, 1, 0); }
  else if (N <= 64) { DmaDenseKC<256, 4>::P pa{A.ptr(), K, M, K, ap}; DmaDenseKC<64, 4>::P pb{B.ptr(), K, N, K, bp}; rc = launch_gemm_pw<Pw256x64, DmaDenseKC<256, 4>, DmaDenseKC<64, 4>>(pa, pb, ep, M, N, K, 1, 0); }
  else if (M <= 64) { DmaDenseKC<64, 4>::P pa{A.ptr(), K, M, K, ap}; DmaDenseKC<256, 4>::P pb{B.ptr(), K, N, K, bp}; rc = launch_gemm_pw<Pw64x256, DmaDenseKC<64, 4>, DmaDenseKC<256, 4>>(pa, pb, ep, M, N, K, 1, 0); }
  else { DmaDenseKC<128, 4>::P pa{A.ptr(), K, M, K, ap}; DmaDenseKC<128, 4>::P pb{B.ptr(), K, N, K, bp}; rc = launch_gemm_pw<Pw128, DmaDenseKC<128, 4>, DmaDenseKC<128, 4>>(pa, pb, ep, M, N, K, 1, 0); }
  if (hipDeviceSynchronize() != hipSuccess || rc < 0) { printf("  planes launch failed rc=%d\n", rc); return 1; }
  const std::vector<unsigned short> op = Cp.back(); const std::vector<float> of = C.back(), ocs = CS.back();
  std::vector<unsigned char> omout(hmout.size()); (void)hipMemcpy(omout.data(), dmout, omout.size(), hipMemcpyDeviceToHost);
  size_t stray = outpl ? PBuf::guard_hits(op, Cp.n) : 0;
  for (size_t i = 0; i < GUARD; ++i) {
    stray += (of[i] != -12345.f) + (of[GUARD + C.n + i] != -12345.f) + (ocs[i] != -12345.f) + (ocs[GUARD + CS.n + i] != -12345.f);
    stray += (omout[i] != 0xA5) + (omout[GUARD + mb + i] != 0xA5);
  }
  if (!outpl) for (size_t i = 0; i < 3 * GUARD + 2 * Cp.n; ++i) stray += op[i] != PBuf::SENT;          // untouched planes buffer
  else for (size_t i = 0; i < C.n; ++i) stray += of[GUARD + i] != -12345.f;                            // untouched fp32 buffer
  if (!maskout) for (size_t i = 0; i < mb; ++i) stray += omout[GUARD + i] != 0xA5;
  int bad = 0; double md = 0; std::vector<double> cs((size_t)N, 0.0);
  for (int i = 0; i < M; ++i)
    for (int j = 0; j < N; ++j) {
      double v = 0;
      for (int k = 0; k < K; ++k) v += (double)A.val[(size_t)i * K + k] * B.val[(size_t)j * K + k];
      v = 0.5 * v + (bias ? Bi.host()[j] : 0.0) + (respl ? R.val[(size_t)i * N + j] : 0.0);
      const bool pos = v > 0;
      if (act == 1) v = pos ? v : 0;
      if (maskin && !((hmin[GUARD + (size_t)i * (N / 8) + j / 8] >> (j % 8)) & 1)) v = 0;
      cs[j] += v;
      const double got = outpl ? (double)bf16_f(op[GUARD + (size_t)i * N + j]) + bf16_f(op[GUARD + Cp.plane() + (size_t)i * N + j]) : (double)of[GUARD + (size_t)i * N + j];
      const double d = fabs(got - v); md = fmax(md, d);
      if (!(d < 2e-3)) ++bad;
      if (maskout && fabs(v) > 1e-3 && (((omout[GUARD + (size_t)i * (N / 8) + j / 8] >> (j % 8)) & 1) != (pos ? 1 : 0))) ++bad;
    }
  if (colsum)
    for (int j = 0; j < N; ++j) {
      double t = 0;
      for (int p_ = 0; p_ < parts; ++p_) { const float v = ocs[GUARD + (size_t)p_ * N + j]; if (v != -12345.f) t += v; }
      if (!(fabs(t - cs[j]) < 2e-2)) ++bad;
    }
  printf("%s planes M=%d N=%d K=%d out=%s bias=%d res_pl=%d act=%d maskin=%d maskout=%d colsum=%d: max err %.2e, wrong %d, stray writes %zu\n",
         (bad || stray) ? "FAIL" : "ok  ", M, N, K, outpl ? "planes" : "fp32", bias, respl, act, maskin, maskout, colsum, md, bad, stray);
  (void)hipFree(A.d); (void)hipFree(B.d); (void)hipFree(Cp.d); (void)hipFree(R.d); (void)hipFree(C.d); (void)hipFree(Bi.d); (void)hipFree(CS.d);
  (void)hipFree(dmin); (void)hipFree(dmout);
  return (bad || stray) ? 1 : 0;
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
  // planes operands: every tile configuration, N % 64 == 0 (bit masks need it) and ragged M
  const int pshapes[][3] = {{300, 256, 64}, {130, 128, 96}, {200, 64, 72}, {64, 320, 40}, {257, 64, 32}, {40, 128, 64}, {520, 512, 32}};
  for (auto& s : pshapes) {
    fails += run_planes_case(s[0], s[1], s[2], true, false, false, 0, false, false, false);    // 11 plain planes
    fails += run_planes_case(s[0], s[1], s[2], true, true, false, 1, false, true, false);      //  6 conv + BN + ReLU + mask out
    fails += run_planes_case(s[0], s[1], s[2], true, true, true, 1, false, true, false);       //  7 + identity
    fails += run_planes_case(s[0], s[1], s[2], true, false, false, 0, true, false, true);      //  9 data gradient: mask in + column sums
    fails += run_planes_case(s[0], s[1], s[2], true, false, true, 0, true, false, true);       // 10 + identity gradient
    fails += run_planes_case(s[0], s[1], s[2], false, false, true, 0, false, false, false);    //  5 fp32 out + planes residual (8-byte loads)
    fails += run_planes_case(s[0], s[1], s[2], false, true, true, 0, false, false, false);     //  2 bias + planes residual -> fp32
    fails += run_planes_case(s[0], s[1], s[2], false, false, false, 0, false, false, false);   //  0 plain fp32 from planes operands
  }
  printf("%d failing cases\n", fails);
  return fails != 0;
}
