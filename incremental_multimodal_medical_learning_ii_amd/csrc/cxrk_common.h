// Shared device/host helpers for the cxrk kernels (gfx950 / CDNA4 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CXRK_OK 0
#define CXRK_ERR_ARG (-1)      // bad shape / alignment / null pointer
#define CXRK_ERR_WS (-2)       // workspace too small
#define CXRK_ERR_LAUNCH (-3)   // hipGetLastError() after launch
#define CXRK_ERR_UNSUPPORTED (-4)

#define CXRK_CHECK_ARG(cond) do { if (!(cond)) return CXRK_ERR_ARG; } while (0)
#define CXRK_LAUNCH_CHECK() do { if (hipGetLastError() != hipSuccess) return CXRK_ERR_LAUNCH; } while (0)

namespace cxrk {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WAVE = 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}

// Block-wide sum for blockDim.x <= 1024 (multiple of 64). `sh` needs >= 16 floats. All threads get the result.
__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63, nw = blockDim.x >> 6;
  __syncthreads();
  if (l == 0) sh[w] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += sh[i];
  return r;
}
__device__ __forceinline__ float block_max(float v, float* sh) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63, nw = blockDim.x >> 6;
  __syncthreads();
  if (l == 0) sh[w] = v;
  __syncthreads();
  float r = sh[0];
  for (int i = 1; i < nw; ++i) r = fmaxf(r, sh[i]);
  return r;
}
__device__ __forceinline__ float block_min(float v, float* sh) {
  v = wave_min(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63, nw = blockDim.x >> 6;
  __syncthreads();
  if (l == 0) sh[w] = v;
  __syncthreads();
  float r = sh[0];
  for (int i = 1; i < nw; ++i) r = fminf(r, sh[i]);
  return r;
}

// erf-GELU (HF "gelu", what CXR-BERT uses) and its derivative, branch-free:  Phi(x) = 1 - E/2 (x >= 0), E/2 (x < 0) with
//   E = erfc(|x| / sqrt 2) ~ t (a1 + t (a2 + t (a3 + t (a4 + t a5)))) exp(-x^2 / 2),  t = 1 / (1 + p |x| / sqrt 2)
// (Abramowitz-Stegun 7.1.26, |error| <= 1.5e-7 on erf).  In fp32 its error on gelu(x) and gelu'(x) is 4.2e-7 / 3.1e-7 over
// [-9, 9] — the same as the erff() form's own rounding (4.5e-7) — at 14 vector instructions (one v_rcp_f32, one v_exp_f32) instead
// of ~30, and the derivative reuses exp(-x^2 / 2) for the density term.  The FFN-up GEMM epilogue spent 30 of its 50 thousand
// cycles per tile in erff().
struct GeluParts { float phi, e; };   // Phi(x), exp(-x^2 / 2)
__device__ __forceinline__ GeluParts gelu_parts(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(t, 1.061405429f, -1.453152027f);
  p = fmaf(t, p, 1.421413741f);
  p = fmaf(t, p, -0.284496736f);
  p = fmaf(t, p, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-(z * z) * 1.4426950408889634f);
  const float half_e = 0.5f * (t * p) * e;
  return {x >= 0.f ? 1.0f - half_e : half_e, e};
}
__device__ __forceinline__ float gelu_erf(float x) { return x * gelu_parts(x).phi; }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const GeluParts g = gelu_parts(x);
  return fmaf(x * 0.39894228040143267794f, g.e, g.phi);
}

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// ---- split-bf16 ("planes") storage -----------------------------------------------------------------------------------
// x = hi + lo + O(2^-17 |x|) with hi = bf16(x), lo = bf16(x - hi).
// 5 vector instructions per pair: v_cvt_pk_bf16_f32 (hi, round-to-nearest-even), shift / mask back to f32, one packed
// subtract, v_cvt_pk_bf16_f32 (lo).  The first conversion is inline asm so that the compiler keeps the packed result
// instead of converting each element a second time on its own.
__device__ __forceinline__ void split2(float x0, float x1, unsigned& hi, unsigned& lo) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  unsigned hp;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hp) : "v"(x0), "v"(x1));
  const f32x2 x = {x0, x1};
  const f32x2 hf = {__builtin_bit_cast(float, hp << 16), __builtin_bit_cast(float, hp & 0xffff0000u)};
  const f32x2 l = x - hf;
  const bf16x2 lb = {(__bf16)l[0], (__bf16)l[1]};
  hi = hp; lo = __builtin_bit_cast(unsigned, lb);
}
// ---- planes helpers (device side of the split-bf16 storage format) -----------------------------------------------
// 8 consecutive elements: hi plane word pairs h.x..h.w (2 bf16 each), same for lo; value = hi + lo.
__device__ __forceinline__ void planes_unpack8(const uint4& h, const uint4& l, float (&v)[8]) {
  const unsigned hw[4] = {h.x, h.y, h.z, h.w}, lw[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    v[2 * q] = __builtin_bit_cast(float, hw[q] << 16) + __builtin_bit_cast(float, lw[q] << 16);
    v[2 * q + 1] = __builtin_bit_cast(float, hw[q] & 0xffff0000u) + __builtin_bit_cast(float, lw[q] & 0xffff0000u);
  }
}
__device__ __forceinline__ void planes_pack8(const float (&v)[8], uint4& h, uint4& l) {
  split2(v[0], v[1], h.x, l.x); split2(v[2], v[3], h.y, l.y); split2(v[4], v[5], h.z, l.z); split2(v[6], v[7], h.w, l.w);
}
__device__ __forceinline__ void planes_load8(const unsigned short* hi, long plane, long off, float (&v)[8]) {
  const uint4 h = *reinterpret_cast<const uint4*>(hi + off);
  const uint4 l = *reinterpret_cast<const uint4*>(hi + plane + off);
  planes_unpack8(h, l, v);
}
__device__ __forceinline__ void planes_store8(unsigned short* hi, long plane, long off, const float (&v)[8], bool nt = false) {
  uint4 h, l;
  planes_pack8(v, h, l);
  if (nt) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 hv = {h.x, h.y, h.z, h.w}, lv = {l.x, l.y, l.z, l.w};
    __builtin_nontemporal_store(hv, reinterpret_cast<u32x4*>(hi + off));
    __builtin_nontemporal_store(lv, reinterpret_cast<u32x4*>(hi + plane + off));
  } else {
    *reinterpret_cast<uint4*>(hi + off) = h;
    *reinterpret_cast<uint4*>(hi + plane + off) = l;
  }
}

// 4 consecutive elements (8 bytes of each plane)
__device__ __forceinline__ void planes_store4(unsigned short* hi, long plane, long off, const float (&v)[4]) {
  uint2 h, l;
  split2(v[0], v[1], h.x, l.x); split2(v[2], v[3], h.y, l.y);
  *reinterpret_cast<uint2*>(hi + off) = h;
  *reinterpret_cast<uint2*>(hi + plane + off) = l;
}

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace cxrk
