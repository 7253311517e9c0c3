// Probe: what does an out-of-range `buffer_load ... lds` (LDS-DMA) lane write into LDS: zero, or nothing (stale bytes)?
// Also checks the per-lane SOURCE gather + lane-linear destination rule.   hipcc --offload-arch=gfx950 -O2 probe_ldsdma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(const float* src, int nbytes, float* out) {
  __shared__ __attribute__((aligned(16))) float lds[64 * 4];
  const int lane = threadIdx.x;
  for (int i = 0; i < 4; ++i) lds[lane * 4 + i] = -7.f;   // poison
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, nbytes, 0x00020000);
  // odd lanes are pushed out of range (bit 31), even lanes gather row (63 - lane)
  unsigned voff = (lane & 1) ? 0x80000000u : (unsigned)((63 - lane) * 16);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = 0; i < 4; ++i) out[lane * 4 + i] = lds[lane * 4 + i];
}
int main() {
  float *src, *out;
  (void)hipMalloc(&src, 64 * 16); (void)hipMalloc(&out, 64 * 16);
  std::vector<float> h(256);
  for (int i = 0; i < 256; ++i) h[i] = (float)i;
  (void)hipMemcpy(src, h.data(), 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src, 1024, out);
  (void)hipMemcpy(h.data(), out, 1024, hipMemcpyDeviceToHost);
  printf("lane0 (in range, gathers row 63): %g %g %g %g\n", h[0], h[1], h[2], h[3]);
  printf("lane1 (out of range): %g %g %g %g   [0 = zero-filled, -7 = not written]\n", h[4], h[5], h[6], h[7]);
  printf("lane2 (row 61): %g ; lane3 (oob): %g\n", h[8], h[12]);
  return 0;
}
