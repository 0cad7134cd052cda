// Library identification for libocrf_hip.so.
#include <hip/hip_runtime.h>

#include "ocrf_hip.h"

extern "C" const char* ocrf_version(void) { return "ocrf_hip 0.1 gfx950"; }
