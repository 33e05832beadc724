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

// The same store by lane PAIRS: the lanes of channel quads c4 = 2 k and 2 k + 1 (adjacent lanes, both active, same pixel) hold 8 + 8 adjacent bytes of
// the hi half and 8 + 8 of the lo half of one line.  They swap what the other needs (quad_perm [1, 0, 3, 2]) and store 16 bytes each -- the even lane
// the pair's hi bytes, the odd lane its lo bytes -- instead of two 8-byte stores per lane.  Same bytes in memory.
__device__ __forceinline__ void s32_store4_pair(void* base, long pixel, int C4, int c4, const float4 v)
{
    s32_bf16x4 h, l;
    h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
    l[0] = (__bf16)(v.x - (float)h[0]); l[1] = (__bf16)(v.y - (float)h[1]); l[2] = (__bf16)(v.z - (float)h[2]); l[3] = (__bf16)(v.w - (float)h[3]);
    const uint2 hu = __builtin_bit_cast(uint2, h), lu = __builtin_bit_cast(uint2, l);
    const bool odd = c4 & 1;
    const unsigned sx = odd ? hu.x : lu.x, sy = odd ? hu.y : lu.y;          // what the neighbour stores
    const unsigned gx = (unsigned)__builtin_amdgcn_update_dpp(0, (int)sx, 0xB1, 0xf, 0xf, false);
    const unsigned gy = (unsigned)__builtin_amdgcn_update_dpp(0, (int)sy, 0xB1, 0xf, 0xf, false);
    const uint4 val = odd ? make_uint4(gx, gy, lu.x, lu.y) : make_uint4(hu.x, hu.y, gx, gy);
    char* p = reinterpret_cast<char*>(base) + (pixel * C4) * 16 + (c4 >> 3) * 128 + (c4 & 6) * 8 + (odd ? 64 : 0);
    *reinterpret_cast<uint4*>(p) = val;
}

}  // namespace ape
