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

#if defined(__HIPCC__)
// APE_ACT_NONE / RELU / PRELU without control flow: v > 0 ? v : (slope * v) & keep, slope = 1 / 1 / alpha, keep = ~0 / 0 / ~0 (RELU's
// negative side becomes the bits of +0.0f).  Same values bit for bit as `v > 0 ? v : 0`, `v > 0 ? v : alpha * v` and `v` (1 * v == v
// also for -0 and NaN).  slope and keep are made OPAQUE vector values once, outside the element loops: written as selects on the
// (wave-uniform) activation code, hipcc threads the conditions back into a scalar compare-and-branch cascade PER ELEMENT -- ~8 scalar
// instructions and two or three branches each, 1.8 k scalar instructions in the 256-element head epilogue of up_3.
struct ActFast { float slope; unsigned keep; };
__device__ __forceinline__ ActFast act_fast_make(int act, float alpha)
{
    ActFast a;
    a.slope = act == APE_ACT_PRELU ? alpha : 1.f;
    a.keep = act == APE_ACT_RELU ? 0u : 0xFFFFFFFFu;
    asm volatile("" : "+v"(a.slope), "+v"(a.keep));
    return a;
}
__device__ __forceinline__ float act_fast(float v, const ActFast& a)
{
    const float neg = __uint_as_float(__float_as_uint(a.slope * v) & a.keep);
    return v > 0.f ? v : neg;
}
#endif

}  // namespace ape
