// ResNet-50 image-encoder kernels, forward: NHWC implicit-GEMM convolution + eval-mode BatchNorm (folded) + residual + ReLU on the
// shared MFMA mainloops.  Reference semantics: health_multimodal/image/model/resnet.py:25-47, torchvision Bottleneck (v1.5).
#include "conv_common.h"
#include "conv_halo.h"

using namespace cxrk;

// ---- forward ------------------------------------------------------------------------------------------------------
template <class FMT>
static int conv_fwd_impl(const typename FMT::T* x, long xplane, const typename FMT::T* w, long wplane, EpiParams ep, int N, int H,
                         int W, int C, int Ko, int R, int S, int stride, int pad, hipStream_t stream) {
  CXRK_CHECK_ARG(x && w && N > 0 && C % FMT::EPL == 0 && aligned16(x) && aligned16(w));
  CXRK_CHECK_ARG(stride == 1 || stride == 2);
  const ConvGeom g = make_geom(N, H, W, C, Ko, R, S, stride, pad);
  CXRK_CHECK_ARG(g.Ho > 0 && g.Wo > 0);
  const long Ml = (long)N * g.Ho * g.Wo;
  CXRK_CHECK_ARG(Ml < (1L << 31));
  const int M = (int)Ml, K = R * S * C;
  if (!tile_span_ok((long)g.Ho * g.Wo, (long)H * W * C)) return CXRK_ERR_UNSUPPORTED;
  int rc;
  const bool tapwise = (C % BK == 0) && R * S <= 32 && R <= 8;  // a K-tile inside one filter tap (everything but the stem)
  if (tapwise && use_wide256(M, Ko, K, 1, FMT::PLANES, WIDE_MINK_FPROP)) {
    if constexpr (FMT::PLANES) {
      DmaConvIm2colKC<256, 8>::P pa{x, g, M, K, xplane}; DmaDenseKC<256, 8>::P pb{w, (long)K, Ko, K, wplane};
      rc = launch_gemm_pw<Pw256, DmaConvIm2colKC<256, 8>, DmaDenseKC<256, 8>>(pa, pb, ep, M, Ko, K, 1, stream);
    } else return CXRK_ERR_UNSUPPORTED;
  } else if (tapwise) {
    if (FMT::PLANES && halo_applies(H, W, C, Ko, R, S, stride, pad)) {   // the 56 x 56 64 -> 64 layers: window resident in LDS
      if constexpr (FMT::PLANES) {
        DmaDenseKC<64, 4>::P pb{w, (long)K, Ko, K, wplane};
        rc = launch_conv3x3_halo<DmaDenseKC<64, 4>, false>(x, xplane, pb, ep, M, Ko, H, W, stream);
      } else rc = CXRK_ERR_UNSUPPORTED;
    } else if (Ko <= 64) {
      if constexpr (FMT::PLANES) {
        DmaConvIm2colKC<256, 4>::P pa{x, g, M, K, xplane}; DmaDenseKC<64, 4>::P pb{w, (long)K, Ko, K, wplane};
        rc = launch_gemm_pw<Pw256x64, DmaConvIm2colKC<256, 4>, DmaDenseKC<64, 4>>(pa, pb, ep, M, Ko, K, 1, stream);
      } else {
        typename ConvIm2colKC<256, FMT>::P pa{x, g, M, K, xplane}; typename DenseKC<64, FMT>::P pb{w, (long)K, Ko, K, wplane};
        rc = launch_gemm<ConvIm2colKC<256, FMT>, DenseKC<64, FMT>, 4, 1>(pa, pb, ep, M, Ko, K, 1, stream);
      }
    } else if constexpr (FMT::PLANES) {
      DmaConvIm2colKC<128, 4>::P pa{x, g, M, K, xplane}; DmaDenseKC<128, 4>::P pb{w, (long)K, Ko, K, wplane};
      rc = launch_gemm_pw<Pw128, DmaConvIm2colKC<128, 4>, DmaDenseKC<128, 4>>(pa, pb, ep, M, Ko, K, 1, stream);
    } else {
      typename ConvIm2colKC<128, FMT>::P pa{x, g, M, K, xplane}; typename DenseKC<128, FMT>::P pb{w, (long)K, Ko, K, wplane};
      rc = launch_gemm<ConvIm2colKC<128, FMT>, DenseKC<128, FMT>, 2, 2>(pa, pb, ep, M, Ko, K, 1, stream);
    }
  } else {
    if constexpr (FMT::PLANES) return CXRK_ERR_UNSUPPORTED;   // the per-lane-tap gather (stem) reads fp32
    else {
      if (Ko <= 64) {
        ConvIm2colKC<256, F32, false>::P pa{x, g, M, K, 0}; DenseKC<64>::P pb{w, (long)K, Ko, K, 0};
        rc = launch_gemm<ConvIm2colKC<256, F32, false>, DenseKC<64>, 4, 1>(pa, pb, ep, M, Ko, K, 1, stream);
      } else {
        ConvIm2colKC<128, F32, false>::P pa{x, g, M, K, 0}; DenseKC<128>::P pb{w, (long)K, Ko, K, 0};
        rc = launch_gemm<ConvIm2colKC<128, F32, false>, DenseKC<128>, 2, 2>(pa, pb, ep, M, Ko, K, 1, stream);
      }
    }
  }
  return rc < 0 ? rc : CXRK_OK;
}

extern "C" int cxrk_conv_bn_act_fwd(const float* x, const float* w_scaled, const float* shift, const float* residual,
                                    float* y, int N, int H, int W, int C, int Ko, int R, int S, int stride, int pad,
                                    int relu, hipStream_t stream) {
  CXRK_CHECK_ARG(y);
  EpiParams ep{};
  ep.C = y; ep.ldc = Ko; ep.bias = shift; ep.R = residual; ep.ldr = Ko; ep.act = relu ? 1 : 0; ep.alpha = 1.f;
  return conv_fwd_impl<F32>(x, 0, w_scaled, 0, ep, N, H, W, C, Ko, R, S, stride, pad, stream);
}

// Planes output (y, optional residual and ReLU decision bits).  in_planes = 1: x and w_scaled are planes too (every unit but
// the stem); in_planes = 0: x and w_scaled are fp32 (the stem reads the fp32 image through the per-lane-tap gather).
extern "C" int cxrk_conv_bn_act_fwd_pl(const void* x, long xplane, const void* w_scaled, long wplane, int in_planes,
                                       const float* shift, const void* residual, long rplane, void* y, long yplane,
                                       unsigned char* maskout, int N, int H, int W, int C, int Ko, int R, int S, int stride,
                                       int pad, int relu, hipStream_t stream) {
  CXRK_CHECK_ARG(y && (Ko % 8) == 0 && !(maskout && !relu));
  EpiParams ep{};
  ep.Cp = static_cast<unsigned short*>(y); ep.cplane = yplane; ep.ldc = Ko; ep.bias = shift;
  ep.Rp = static_cast<const unsigned short*>(residual); ep.rplane = rplane; ep.ldr = Ko; ep.act = relu ? 1 : 0; ep.alpha = 1.f;
  ep.maskout = maskout; ep.ldmaskout = Ko / 8;
  if (in_planes)
    return conv_fwd_impl<PL>(static_cast<const unsigned short*>(x), xplane, static_cast<const unsigned short*>(w_scaled), wplane, ep, N, H,
                             W, C, Ko, R, S, stride, pad, stream);
  return conv_fwd_impl<F32>(static_cast<const float*>(x), 0, static_cast<const float*>(w_scaled), 0, ep, N, H, W, C, Ko, R, S, stride,
                            pad, stream);
}

