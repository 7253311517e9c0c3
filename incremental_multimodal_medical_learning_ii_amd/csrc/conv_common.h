// Shared by the convolution translation units (conv_fwd / conv_dgrad / conv_wgrad / conv_misc.hip): geometry helper and the
// 2-GiB tile-span check of the gathers.  The kernels live in the unit that launches them.
#pragma once
#include "cxrk.h"
#include "gemm_core.h"

namespace cxrk {

static inline ConvGeom make_geom(int N, int H, int W, int C, int Ko, int R, int S, int stride, int pad) {
  ConvGeom g;
  g.N = N; g.H = H; g.W = W; g.C = C; g.Ko = Ko; g.R = R; g.S = S; g.stride = stride; g.pad = pad;
  g.Ho = (H + 2 * pad - R) / stride + 1;
  g.Wo = (W + 2 * pad - S) / stride + 1;
  return g;
}

// The convolution gathers address a block's rows with 32-bit byte offsets from the first image the tile touches
// (gemm_loaders.h): a 256-row tile spans at most 256 / (Ho*Wo) + 2 images of the gathered tensor, which must stay < 2 GiB.
static bool tile_span_ok(long rows_per_image, long image_elems) {
  const long images = 256 / (rows_per_image > 0 ? rows_per_image : 1) + 2;
  return images * image_elems * 4 < (1L << 31);
}

}  // namespace cxrk
