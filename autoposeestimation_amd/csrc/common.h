// Shared helpers for libape_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ape_hip.h"

namespace ape {

void set_last_error(const char* what);

static inline int check_launch(const char* what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_last_error(hipGetErrorString(e));
        (void)what;
        return APE_ELAUNCH;
    }
    return APE_OK;
}

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace ape
