#include "cxrk.h"
#include "gemm_core.h"
extern "C" const char* cxrk_version(void) { return "cxrk 0.1 gfx950"; }
extern "C" int cxrk_set_precision(int mode) {
  if (mode != 0 && mode != 1) return CXRK_ERR_ARG;
  cxrk::gemm_precision_mode() = mode;
  return CXRK_OK;
}
extern "C" int cxrk_get_precision(void) { return cxrk::gemm_precision_mode(); }
// 256x256-kernel policy: 0 never, 1 where it pays (default), 2 every planes launch (tests); returns the previous mode
extern "C" int cxrk_set_wide_mode(int mode) {
  if (mode < 0 || mode > 2) return CXRK_ERR_ARG;
  const int old = cxrk::wide_mode_ref();
  cxrk::wide_mode_ref() = mode;
  return old;
}
