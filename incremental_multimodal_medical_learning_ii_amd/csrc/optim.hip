// Optimiser + continual-learning weight-reset kernels (HBM-bound elementwise / reduction work).
//  * fused Adam, torch.optim.Adam defaults (Trainer.py:172-178): 16 B read + 12 B written per parameter
//  * SGD (Trainer.py:176-178)
//  * Trainer.myIncremental (Trainer.py:1556-1587): per-tensor min/max of |new-old|, thresholded restore + counters
#include "cxrk.h"
#include "cxrk_common.h"

using namespace cxrk;

namespace {

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                                                   float weight_decay, float bc1, float bc2_sqrt, float grad_scale) {
  const long stride = (long)gridDim.x * blockDim.x * 4;
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 3 < n) {
      float4 pv = *reinterpret_cast<float4*>(p + i);
      float4 gv = *reinterpret_cast<const float4*>(g + i);
      float4 mv = *reinterpret_cast<float4*>(m + i);
      float4 vv = *reinterpret_cast<float4*>(v + i);
      float* pp = &pv.x; float* gp = &gv.x; float* mp = &mv.x; float* vp = &vv.x;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float gg = gp[k] * grad_scale;
        if (weight_decay != 0.f) gg += weight_decay * pp[k];
        mp[k] = mp[k] + (gg - mp[k]) * (1.f - b1);            // exp_avg.lerp_(grad, 1-beta1)
        vp[k] = vp[k] * b2 + (1.f - b2) * gg * gg;            // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1-beta2)
        const float denom = sqrtf(vp[k]) / bc2_sqrt + eps;
        pp[k] = pp[k] - (lr / bc1) * (mp[k] / denom);         // param.addcdiv_(exp_avg, denom, value=-step_size)
      }
      *reinterpret_cast<float4*>(p + i) = pv;
      *reinterpret_cast<float4*>(m + i) = mv;
      *reinterpret_cast<float4*>(v + i) = vv;
    } else {
      for (long j = i; j < n; ++j) {
        float gg = g[j] * grad_scale;
        if (weight_decay != 0.f) gg += weight_decay * p[j];
        const float mm = m[j] + (gg - m[j]) * (1.f - b1);
        const float vv = v[j] * b2 + (1.f - b2) * gg * gg;
        m[j] = mm; v[j] = vv;
        p[j] = p[j] - (lr / bc1) * (mm / (sqrtf(vv) / bc2_sqrt + eps));
      }
    }
  }
}

__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, long n, float lr, float weight_decay, float grad_scale) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float gg = g[i] * grad_scale;
    if (weight_decay != 0.f) gg += weight_decay * p[i];
    p[i] -= lr * gg;
  }
}

// stage 1: per-block min/max of |a-b|
__global__ __launch_bounds__(256) void absdiff_minmax_kernel(const float* __restrict__ a, const float* __restrict__ b, long n,
                                                             float* __restrict__ part) {
  __shared__ float sh[16];
  float mn = INFINITY, mx = -INFINITY;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float d = fabsf(a[i] - b[i]);
    mn = fminf(mn, d); mx = fmaxf(mx, d);
  }
  mn = block_min(mn, sh); mx = block_max(mx, sh);
  if (threadIdx.x == 0) { part[2 * blockIdx.x] = mn; part[2 * blockIdx.x + 1] = mx; }
}
// stage 2: every block re-derives the threshold from the partials, restores, counts.
__global__ __launch_bounds__(256) void weight_reset_kernel(float* __restrict__ pnew, const float* __restrict__ pold, long n,
                                                           const float* __restrict__ part, int nparts, float threshold,
                                                           unsigned long long* __restrict__ counters) {
  float mn = INFINITY, mx = -INFINITY;
  for (int i = 0; i < nparts; ++i) { mn = fminf(mn, part[2 * i]); mx = fmaxf(mx, part[2 * i + 1]); }
  const float to_reset = mn + threshold * (mx - mn);
  unsigned int cnt = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float o = pold[i];
    const float d = fabsf(pnew[i] - o);
    if (d < to_reset) { pnew[i] = o; ++cnt; }
  }
  __shared__ float sh[16];
  const float tot = block_sum((float)cnt, sh);  // exact below 2^24 per block
  if (threadIdx.x == 0) {
    atomicAdd(&counters[0], (unsigned long long)(tot + 0.5f));
  }
}

}  // namespace

extern "C" int cxrk_adam_fused(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                               float eps, float weight_decay, int step, float grad_scale, hipStream_t stream) {
  CXRK_CHECK_ARG(p && g && m && v && n > 0 && step >= 1);
  CXRK_CHECK_ARG(aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v));
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  long nb = (n / 4 + 255) / 256; if (nb > 4096) nb = 4096; if (nb < 1) nb = 1;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)nb), dim3(256), 0, stream, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay,
                     (float)bc1, (float)sqrt(bc2), grad_scale);
  CXRK_LAUNCH_CHECK();
  return CXRK_OK;
}

extern "C" int cxrk_sgd(float* p, const float* g, long n, float lr, float weight_decay, float grad_scale, hipStream_t stream) {
  CXRK_CHECK_ARG(p && g && n > 0);
  long nb = (n + 255) / 256; if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)nb), dim3(256), 0, stream, p, g, n, lr, weight_decay, grad_scale);
  CXRK_LAUNCH_CHECK();
  return CXRK_OK;
}

extern "C" size_t cxrk_weight_reset_ws_bytes(void) { return 2 * 256 * sizeof(float); }

// counters[0] += number of restored elements (caller derives updated = n - reset).
extern "C" int cxrk_weight_reset(float* pnew, const float* pold, long n, float threshold, unsigned long long* counters,
                                 float* ws, size_t ws_bytes, hipStream_t stream) {
  CXRK_CHECK_ARG(pnew && pold && counters && n > 0);
  if (ws == nullptr || ws_bytes < 2 * 256 * sizeof(float)) return CXRK_ERR_WS;
  long nb = (n + 255) / 256; if (nb > 256) nb = 256;
  hipLaunchKernelGGL(absdiff_minmax_kernel, dim3((unsigned)nb), dim3(256), 0, stream, pnew, pold, n, ws);
  CXRK_LAUNCH_CHECK();
  hipLaunchKernelGGL(weight_reset_kernel, dim3((unsigned)nb), dim3(256), 0, stream, pnew, pold, n, ws, (int)nb, threshold, counters);
  CXRK_LAUNCH_CHECK();
  return CXRK_OK;
}
