// Shared helpers for libape_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
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

// Per-(kernel, device) set-up of the kernels with more than 64 KB of dynamic LDS: hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a
// per-DEVICE property and so is the compute-unit count the persistent grids are sized from -- a process may drive several devices
// (one `static DeviceOnce` per kernel instantiation; devices 0..63).  Returns APE_OK and the current device's CU count (256 if unknown).
struct DeviceOnce {
    std::atomic<unsigned long long> done{0};
    int ncu[64];
};
static inline int device_once(DeviceOnce& s, const void* kernel, int lds_bytes, int* ncu_out)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) { set_last_error("hipGetDevice"); return APE_ELAUNCH; }
    if (!((s.done.load(std::memory_order_acquire) >> dev) & 1ULL)) {
        if (lds_bytes > 0 && hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) {
            set_last_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
            return APE_ELAUNCH;
        }
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
        s.ncu[dev] = n;
        s.done.fetch_or(1ULL << dev, std::memory_order_release);
    }
    if (ncu_out) *ncu_out = s.ncu[dev];
    return APE_OK;
}

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
