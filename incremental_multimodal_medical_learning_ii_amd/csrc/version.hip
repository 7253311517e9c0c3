#include "cxrk.h"
extern "C" const char* cxrk_version(void) { return "cxrk 0.1 gfx950"; }
