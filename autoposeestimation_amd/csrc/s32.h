// Element access to "S32" (pre-split) activation tensors -- include/ape_hip.h: per pixel and 32-channel group 128 bytes
// [hi 32 x bf16 | lo 32 x bf16], value = hi + lo.  `c4` counts float4s (4 channels) inside the pixel, like the fp32 kernels do.
#pragma once
#include <hip/hip_runtime.h>

namespace ape {

typedef __bf16 s32_bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 s32_load4(const void* base, long pixel, int C4, int c4)
{
    const char* p = reinterpret_cast<const char*>(base) + (pixel * C4) * 16 + (c4 >> 3) * 128 + (c4 & 7) * 8;
    const s32_bf16x4 h = *reinterpret_cast<const s32_bf16x4*>(p), l = *reinterpret_cast<const s32_bf16x4*>(p + 64);
    return make_float4((float)h[0] + (float)l[0], (float)h[1] + (float)l[1], (float)h[2] + (float)l[2], (float)h[3] + (float)l[3]);
}

__device__ __forceinline__ void s32_store4(void* base, long pixel, int C4, int c4, const float4 v)
{
    char* p = reinterpret_cast<char*>(base) + (pixel * C4) * 16 + (c4 >> 3) * 128 + (c4 & 7) * 8;
    s32_bf16x4 h, l;
    h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
    l[0] = (__bf16)(v.x - (float)h[0]); l[1] = (__bf16)(v.y - (float)h[1]); l[2] = (__bf16)(v.z - (float)h[2]); l[3] = (__bf16)(v.w - (float)h[3]);
    *reinterpret_cast<s32_bf16x4*>(p) = h;
    *reinterpret_cast<s32_bf16x4*>(p + 64) = l;
}

}  // namespace ape
