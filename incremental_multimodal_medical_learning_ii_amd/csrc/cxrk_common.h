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

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
  return cdf + x * pdf;
}

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace cxrk
