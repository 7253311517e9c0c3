// ResNet-50 image-encoder kernels, data gradient: implicit-GEMM transposed convolution (stride-2 layers by output-parity classes),
// ReLU mask, identity-branch gradient and the fused BatchNorm beta-gradient column sums in the epilogue.
#include "conv_common.h"
#include "conv_halo.h"

using namespace cxrk;

namespace {
// out[z][c] = sum over the z-th chunk of parts of part[p][c]; block = 64 columns x 16 part-lanes.
// Run twice (chunks -> 1) so the long partial lists of the 56x56 layers are reduced by many blocks, deterministically.
__global__ __launch_bounds__(1024) void colsum_part_final_kernel(const float* __restrict__ part, int nparts, int chunk, int C,
                                                                 float* __restrict__ out) {
  __shared__ float sh[16][64];
  const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl, z = blockIdx.z;
  const int p0 = z * chunk, p1 = min(nparts, p0 + chunk);
  float a = 0.f;
  if (c < C)
    for (int p = p0 + pl; p < p1; p += 16) a += part[(long)p * C + c];
  sh[pl][cl] = a;
  __syncthreads();
  if (pl == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += sh[i][cl];
    out[(long)z * C + c] = t;
  }
}

}  // namespace

// ---- data gradient ---------------------------------------------------------------------------------------------------
// dx[n][hi][wi][c] = mask( sum_{r,s,ko} dy[n][(hi+pad-r)/st][(wi+pad-s)/st][ko] * w_scaled[ko][r][s][c] + residual )
// 64-row slabs of column-sum partials one data-gradient launch writes: its row tile (256 with the 256x256 / 256x64 tiles, else
// 128) rounded up, in slabs.  K = contraction length of that launch (selects the tile exactly as the launch does).
static int dgrad_tiles(int rows, int C, long K, bool planes) {
  const bool t256 = use_wide256(rows, C, K, 1, planes, WIDE_MINK_DGRAD) || C <= 64;
  return ceil_div(rows, t256 ? 256 : 128) * (t256 ? 4 : 2);
}

// number of per-wave partial rows the fused column sum of a data-gradient launch produces (all parity classes)
static long dgrad_colsum_parts(int N, int H, int W, int C, int stride, int Ko, int R, int S, int pad, bool planes) {
  if (planes && halo_applies(H, W, Ko, C, R, S, stride, pad) && !use_wide256(N * H * W, C, (long)R * S * Ko, 1, planes, WIDE_MINK_DGRAD))
    return (long)ceil_div(N * H * W, HALO_TM) * 8;      // conv_halo.h: one partial row per 32-row half of a 64-row block
  if (stride == 1) return dgrad_tiles(N * H * W, C, (long)R * S * Ko, planes);
  long t = 0;
  for (int ph = 0; ph < 2; ++ph)
    for (int pw = 0; pw < 2; ++pw) {
      const int Hs = (H - ph + 1) / 2, Ws = (W - pw + 1) / 2;
      int nr = 0, ns = 0;
      for (int r = 0; r < R; ++r) if (((ph + pad - r) & 1) == 0) ++nr;
      for (int q = 0; q < S; ++q) if (((pw + pad - q) & 1) == 0) ++ns;
      if (Hs > 0 && Ws > 0 && nr > 0 && ns > 0) t += dgrad_tiles(N * Hs * Ws, C, (long)nr * ns * Ko, planes);
    }
  return t;
}

// `ep` carries output / residual / mask operands in either storage format; ep.colsum_part (optional) = partial buffer.
template <class FMT>
static int conv_bwd_data_impl(const typename FMT::T* dy, long dyplane, const typename FMT::T* w, long wplane, EpiParams ep,
                              bool has_side, int N, int H, int W, int C, int Ko, int R, int S, int stride, int pad,
                              hipStream_t stream) {
  CXRK_CHECK_ARG(dy && w && N > 0 && C % FMT::EPL == 0 && Ko % FMT::EPL == 0 && aligned16(dy) && aligned16(w));
  CXRK_CHECK_ARG(stride == 1 || stride == 2);
  if (Ko % BK != 0 || R * S > 32 || R > 8) return CXRK_ERR_UNSUPPORTED;  // the gather keeps a K-tile inside one filter tap
  const ConvGeom g = make_geom(N, H, W, C, Ko, R, S, stride, pad);
  const long Ml = (long)N * H * W;
  CXRK_CHECK_ARG(Ml < (1L << 31));
  const int M = (int)Ml, K = R * S * Ko;
  if (!tile_span_ok((long)(H / stride) * (W / stride), (long)g.Ho * g.Wo * Ko)) return CXRK_ERR_UNSUPPORTED;
  int rc = 0;
  if (stride == 1) {
    if (use_wide256(M, C, K, 1, FMT::PLANES, WIDE_MINK_DGRAD)) {
      if constexpr (FMT::PLANES) {
        DmaConvDgradKC<256, 8>::P pa{dy, g, M, K, dyplane}; DmaConvFilterMC<256, 8>::P pb{w, g, C, K, wplane};
        rc = launch_gemm_pw<Pw256, DmaConvDgradKC<256, 8>, DmaConvFilterMC<256, 8>>(pa, pb, ep, M, C, K, 1, stream);
      } else return CXRK_ERR_UNSUPPORTED;
    } else if (FMT::PLANES && halo_applies(H, W, Ko, C, R, S, stride, pad)) {   // 64 -> 64: the dy window resident in LDS (conv_halo.h)
      if constexpr (FMT::PLANES) {
        DmaConvFilterMC<64, 4>::P pb{w, g, C, K, wplane};
        rc = launch_conv3x3_halo<DmaConvFilterMC<64, 4>, true>(dy, dyplane, pb, ep, M, C, H, W, stream);
      } else rc = CXRK_ERR_UNSUPPORTED;
    } else if (C <= 64) {
      if constexpr (FMT::PLANES) {
        DmaConvDgradKC<256, 4>::P pa{dy, g, M, K, dyplane}; DmaConvFilterMC<64, 4>::P pb{w, g, C, K, wplane};
        rc = launch_gemm_pw<Pw256x64, DmaConvDgradKC<256, 4>, DmaConvFilterMC<64, 4>>(pa, pb, ep, M, C, K, 1, stream);
      } else {
        typename ConvDgradKC<256, FMT>::P pa{dy, g, M, K, dyplane}; typename ConvFilterMC<64, FMT>::P pb{w, g, C, K, wplane};
        rc = launch_gemm<ConvDgradKC<256, FMT>, ConvFilterMC<64, FMT>, 4, 1>(pa, pb, ep, M, C, K, 1, stream);
      }
    } else if constexpr (FMT::PLANES) {
      DmaConvDgradKC<128, 4>::P pa{dy, g, M, K, dyplane}; DmaConvFilterMC<128, 4>::P pb{w, g, C, K, wplane};
      rc = launch_gemm_pw<Pw128, DmaConvDgradKC<128, 4>, DmaConvFilterMC<128, 4>>(pa, pb, ep, M, C, K, 1, stream);
    } else {
      typename ConvDgradKC<128, FMT>::P pa{dy, g, M, K, dyplane}; typename ConvFilterMC<128, FMT>::P pb{w, g, C, K, wplane};
      rc = launch_gemm<ConvDgradKC<128, FMT>, ConvFilterMC<128, FMT>, 2, 2>(pa, pb, ep, M, C, K, 1, stream);
    }
    return rc < 0 ? rc : CXRK_OK;
  }
  // stride 2: one launch per output-parity class, only over the taps that reach it
  CXRK_CHECK_ARG(R <= 3 && S <= 3);
  long part_off = 0;
  bool zeroed = false;
  for (int pass = 0; pass < 2; ++pass) {
    for (int ph = 0; ph < 2; ++ph) {
      for (int pw = 0; pw < 2; ++pw) {
        S2Taps t{};
        for (int r = 0; r < R; ++r) if (((ph + pad - r) & 1) == 0) { t.r[t.nr] = r; t.dr[t.nr] = (ph + pad - r) / 2; ++t.nr; }
        for (int q = 0; q < S; ++q) if (((pw + pad - q) & 1) == 0) { t.s[t.ns] = q; t.ds[t.ns] = (pw + pad - q) / 2; ++t.ns; }
        const int Hs = (H - ph + 1) / 2, Ws = (W - pw + 1) / 2;
        if (Hs <= 0 || Ws <= 0) continue;
        if (pass == 0) {
          if (t.nr == 0 || t.ns == 0) {
            // no tap reaches this class: its pixels are exactly zero.  Only the plain form is supported there.
            CXRK_CHECK_ARG(!has_side);
            if (!zeroed) {
              void* base = ep.Cp ? static_cast<void*>(ep.Cp) : static_cast<void*>(ep.C);
              if (hipMemsetAsync(base, 0, (size_t)M * C * (ep.Cp ? 2 : 4), stream) != hipSuccess) return CXRK_ERR_LAUNCH;
              if (ep.Cp && hipMemsetAsync(ep.Cp + ep.cplane, 0, (size_t)M * C * 2, stream) != hipSuccess) return CXRK_ERR_LAUNCH;
              zeroed = true;
            }
          }
          continue;
        }
        if (t.nr == 0 || t.ns == 0) continue;
        const int Ms = N * Hs * Ws, Ks = t.nr * t.ns * Ko;
        EpiParams e2 = ep;
        if (ep.colsum_part) { e2.colsum_part = ep.colsum_part + part_off * C; part_off += dgrad_tiles(Ms, C, Ks, FMT::PLANES); }
        e2.rm_on = 1; e2.rm_Hs = Hs; e2.rm_Ws = Ws; e2.rm_H = H; e2.rm_W = W; e2.rm_ph = ph; e2.rm_pw = pw;
        if (use_wide256(Ms, C, Ks, 1, FMT::PLANES, WIDE_MINK_DGRAD)) {
          if constexpr (FMT::PLANES) {
            DmaConvDgradS2KC<256, 8>::P pa{dy, g, t, Hs, Ws, Ms, Ks, dyplane}; DmaConvFilterS2MC<256, 8>::P pb{w, g, t, C, Ks, wplane};
            rc = launch_gemm_pw<Pw256, DmaConvDgradS2KC<256, 8>, DmaConvFilterS2MC<256, 8>>(pa, pb, e2, Ms, C, Ks, 1, stream);
          } else return CXRK_ERR_UNSUPPORTED;
        } else if (C <= 64) {
          if constexpr (FMT::PLANES) {
            DmaConvDgradS2KC<256, 4>::P pa{dy, g, t, Hs, Ws, Ms, Ks, dyplane}; DmaConvFilterS2MC<64, 4>::P pb{w, g, t, C, Ks, wplane};
            rc = launch_gemm_pw<Pw256x64, DmaConvDgradS2KC<256, 4>, DmaConvFilterS2MC<64, 4>>(pa, pb, e2, Ms, C, Ks, 1, stream);
          } else {
            typename ConvDgradS2KC<256, FMT>::P pa{dy, g, t, Hs, Ws, Ms, Ks, dyplane}; typename ConvFilterS2MC<64, FMT>::P pb{w, g, t, C, Ks, wplane};
            rc = launch_gemm<ConvDgradS2KC<256, FMT>, ConvFilterS2MC<64, FMT>, 4, 1>(pa, pb, e2, Ms, C, Ks, 1, stream);
          }
        } else if constexpr (FMT::PLANES) {
          DmaConvDgradS2KC<128, 4>::P pa{dy, g, t, Hs, Ws, Ms, Ks, dyplane}; DmaConvFilterS2MC<128, 4>::P pb{w, g, t, C, Ks, wplane};
          rc = launch_gemm_pw<Pw128, DmaConvDgradS2KC<128, 4>, DmaConvFilterS2MC<128, 4>>(pa, pb, e2, Ms, C, Ks, 1, stream);
        } else {
          typename ConvDgradS2KC<128, FMT>::P pa{dy, g, t, Hs, Ws, Ms, Ks, dyplane}; typename ConvFilterS2MC<128, FMT>::P pb{w, g, t, C, Ks, wplane};
          rc = launch_gemm<ConvDgradS2KC<128, FMT>, ConvFilterS2MC<128, FMT>, 2, 2>(pa, pb, e2, Ms, C, Ks, 1, stream);
        }
        if (rc < 0) return rc;
      }
    }
  }
  return CXRK_OK;
}

static int finish_colsum(float* ws, long np, int C, float* sums, hipStream_t stream) {
  int Z = ceil_div(np, 512); if (Z > 64) Z = 64; if (Z < 1) Z = 1;
  const int chunk = ceil_div(np, Z);
  Z = ceil_div(np, chunk);
  float* tmp = ws + np * C;
  hipLaunchKernelGGL(colsum_part_final_kernel, dim3(ceil_div(C, 64), 1, Z), dim3(1024), 0, stream, ws, (int)np, chunk, C, Z > 1 ? tmp : sums);
  CXRK_LAUNCH_CHECK();
  if (Z > 1) {
    hipLaunchKernelGGL(colsum_part_final_kernel, dim3(ceil_div(C, 64), 1, 1), dim3(1024), 0, stream, tmp, Z, Z, C, sums);
    CXRK_LAUNCH_CHECK();
  }
  return CXRK_OK;
}

extern "C" size_t cxrk_conv_bwd_data_colsum_ws_bytes(int N, int H, int W, int C, int stride) {
  // upper bound over the tile choices (the exact count needs the filter shape): 256-row tiles, 4 slabs each
  long parts = 0;
  if (stride == 1) parts = (long)ceil_div((long)N * H * W, 256) * 8;   // 8: the window-resident 3x3 kernel's halves (conv_halo.h)
  else
    for (int ph = 0; ph < 2; ++ph)
      for (int pw = 0; pw < 2; ++pw) { const int Hs = (H - ph + 1) / 2, Ws = (W - pw + 1) / 2; if (Hs > 0 && Ws > 0) parts += (long)ceil_div((long)N * Hs * Ws, 256) * 4; }
  return (size_t)(parts + 64) * C * sizeof(float);
}

// fp32 storage.  `sums` (optional, [C]) receives the column sums of dx = the BatchNorm beta gradient of the unit that produced
// relu_src (dx is that unit's masked output gradient); needs `ws` of cxrk_conv_bwd_data_colsum_ws_bytes().
extern "C" int cxrk_conv_bn_act_bwd_data(const float* dy, const float* w_scaled, const float* residual,
                                         const float* relu_src, float* dx, int N, int H, int W, int C, int Ko, int R, int S,
                                         int stride, int pad, float* sums, float* ws, size_t ws_bytes, hipStream_t stream) {
  CXRK_CHECK_ARG(dx);
  EpiParams ep{};
  ep.C = dx; ep.ldc = C; ep.R = residual; ep.ldr = C; ep.alpha = 1.f;
  if (relu_src) { ep.aux = relu_src; ep.ldaux = C; ep.auxmode = 1; }
  long np = 0;
  if (sums) {
    CXRK_CHECK_ARG(!(R == 1 && stride == 2));
    np = dgrad_colsum_parts(N, H, W, C, stride, Ko, R, S, pad, false);
    if (ws == nullptr || ws_bytes < (size_t)(np + 64) * C * sizeof(float)) return CXRK_ERR_WS;
    ep.colsum_part = ws;
  }
  const int rc = conv_bwd_data_impl<F32>(dy, 0, w_scaled, 0, ep, residual || relu_src, N, H, W, C, Ko, R, S, stride, pad, stream);
  if (rc != CXRK_OK || !sums) return rc;
  return finish_colsum(ws, np, C, sums, stream);
}

// planes storage: dy, w_scaled, residual (optional) and dx are planes; `maskin` (optional) = ReLU decision bits of the unit
// whose output gradient this is (byte [pixel][C / 8]).
static int conv_bwd_data_pl_impl(const void* dy, long dyplane, const void* w_scaled, long wplane, const void* residual, long rplane, int res_s2,
                                 const unsigned char* maskin, void* dx, long dxplane, int N, int H, int W, int C, int Ko, int R, int S,
                                 int stride, int pad, float* sums, float* ws, size_t ws_bytes, hipStream_t stream);
extern "C" int cxrk_conv_bn_act_bwd_data_pl(const void* dy, long dyplane, const void* w_scaled, long wplane, const void* residual,
                                            long rplane, const unsigned char* maskin, void* dx, long dxplane, int N, int H, int W,
                                            int C, int Ko, int R, int S, int stride, int pad, float* sums, float* ws,
                                            size_t ws_bytes, hipStream_t stream) {
  return conv_bwd_data_pl_impl(dy, dyplane, w_scaled, wplane, residual, rplane, 0, maskin, dx, dxplane, N, H, W, C, Ko, R, S, stride, pad, sums, ws,
                               ws_bytes, stream);
}
// The same with a COMPACT residual: planes [N, (H + 1) / 2, (W + 1) / 2, C] holding the identity-branch gradient at the pixels with
// even (h, w) — what the data gradient of a 1x1 / stride-2 projection shortcut is (a dense [pixels / 4, Ko] x [Ko, C] product);
// the other pixels receive nothing.  Saves zero-filling, writing and re-reading the 3/4 of that gradient that are zero.
extern "C" int cxrk_conv_bn_act_bwd_data_pl_s2res(const void* dy, long dyplane, const void* w_scaled, long wplane, const void* residual_s2,
                                                  long rplane, const unsigned char* maskin, void* dx, long dxplane, int N, int H, int W,
                                                  int C, int Ko, int R, int S, int stride, int pad, float* sums, float* ws,
                                                  size_t ws_bytes, hipStream_t stream) {
  CXRK_CHECK_ARG(residual_s2 && stride == 1);
  // 32-bit byte offsets into the compact tensor (gemm_epilogue.h)
  if ((long)N * ((H + 1) / 2) * ((W + 1) / 2) * C * 2 >= (1L << 31)) return CXRK_ERR_UNSUPPORTED;
  return conv_bwd_data_pl_impl(dy, dyplane, w_scaled, wplane, residual_s2, rplane, 1, maskin, dx, dxplane, N, H, W, C, Ko, R, S, stride, pad, sums,
                               ws, ws_bytes, stream);
}
static int conv_bwd_data_pl_impl(const void* dy, long dyplane, const void* w_scaled, long wplane, const void* residual, long rplane, int res_s2,
                                 const unsigned char* maskin, void* dx, long dxplane, int N, int H, int W, int C, int Ko, int R, int S,
                                 int stride, int pad, float* sums, float* ws, size_t ws_bytes, hipStream_t stream) {
  CXRK_CHECK_ARG(dx && (C % 8) == 0);
  EpiParams ep{};
  if (res_s2) { ep.rs2_on = 1; ep.rs2_H = H; ep.rs2_W = W; ep.rs2_Ho = (H + 1) / 2; ep.rs2_Wo = (W + 1) / 2; }
  ep.Cp = static_cast<unsigned short*>(dx); ep.cplane = dxplane; ep.ldc = C;
  ep.Rp = static_cast<const unsigned short*>(residual); ep.rplane = rplane; ep.ldr = C; ep.alpha = 1.f;
  if (maskin) { ep.maskin = maskin; ep.ldmaskin = C / 8; ep.auxmode = 3; }
  long np = 0;
  if (sums) {
    CXRK_CHECK_ARG(!(R == 1 && stride == 2));
    np = dgrad_colsum_parts(N, H, W, C, stride, Ko, R, S, pad, true);
    if (ws == nullptr || ws_bytes < (size_t)(np + 64) * C * sizeof(float)) return CXRK_ERR_WS;
    ep.colsum_part = ws;
  }
  const int rc = conv_bwd_data_impl<PL>(static_cast<const unsigned short*>(dy), dyplane, static_cast<const unsigned short*>(w_scaled), wplane,
                                        ep, residual || maskin, N, H, W, C, Ko, R, S, stride, pad, stream);
  if (rc != CXRK_OK || !sums) return rc;
  return finish_colsum(ws, np, C, sums, stream);
}

