// info.cpp — build tag of libcream_amd.so.
#include "cream_amd.h"

#ifndef CREAM_BUILD_TAG
#define CREAM_BUILD_TAG "gfx950"
#endif

extern "C" const char* cream_build_info(void) { return CREAM_BUILD_TAG; }
