// Library identification for libkaolin_amd.so.
#include "common.h"
#include "../../include/kaolin_amd.h"

extern "C" const char* kamd_version(void) { return "kaolin_amd 0.1.0 gfx950"; }
