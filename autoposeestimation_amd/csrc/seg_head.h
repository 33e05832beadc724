// Shared device code of the segmentation head: final 1x1 conv (64 -> C <= 16 classes, pspnet.py:53-55) + softmax (+ softmax)
// + arg-max for a group of 16 pixels on the exact-fp32 matrix cores (v_mfma_f32_16x16x4_f32, D = W . X^T: rows = classes,
// columns = 16 pixels).  Lane (px = lane & 15, kq = lane >> 4) holds, for j = 0..3, the float4 of channels 16 j + 4 kq .. + 3
// of ITS pixel in x[j]; the 16 weight operands and 4 biases per lane are loaded once by seg_head_load_weights.  Used by
// seg_head_kernel (streaming over a feature tensor) and by the LDS-halo convolution's fused-head epilogue, so both paths
// produce bit-identical labels and scores.
#pragma once
#include "common.h"

namespace ape_seg {

typedef float f32x4h __attribute__((ext_vector_type(4)));

// Value of lane ^ 16 / lane ^ 32 (the other channel quads of the same pixel) through gfx950's v_permlane16_swap / v_permlane32_swap:
// with both operands the same register, the swap leaves the even rows (low half) of the value in one result and the odd rows
// (high half) in the other, so the partner's value is a select away -- two VALU operations instead of a ds_bpermute_b32 round
// trip through the LDS crossbar (eight dependent ones per 16-pixel group made the head latency-bound).
__device__ __forceinline__ unsigned lane_xor16_u(unsigned u, int lane)
{
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);    // r[0] = {row0,row0,row2,row2}, r[1] = {row1,row1,row3,row3}
    return (lane & 16) ? r[0] : r[1];
}
__device__ __forceinline__ unsigned lane_xor32_u(unsigned u, int lane)
{
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);    // r[0] = {lo,lo}, r[1] = {hi,hi}
    return (lane & 32) ? r[0] : r[1];
}
__device__ __forceinline__ float lane_xor16(float v, int lane) { return __uint_as_float(lane_xor16_u(__float_as_uint(v), lane)); }
__device__ __forceinline__ float lane_xor32(float v, int lane) { return __uint_as_float(lane_xor32_u(__float_as_uint(v), lane)); }

__device__ __forceinline__ float quad_sum(float v, int lane) { v += lane_xor16(v, lane); v += lane_xor32(v, lane); return v; }

__device__ __forceinline__ void seg_head_load_weights(const float* __restrict__ w, const float* __restrict__ bias, int C, int lane,
                                                      float (&wreg)[16], float (&breg)[4])
{
    const int px = lane & 15, kq = lane >> 4;
    // A operand (weights): class = lane&15, k = 16 j + 4 kq + e  -> wreg[j*4+e]; rows >= C are zero
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) wreg[j * 4 + e] = px < C ? w[px * 64 + 16 * j + 4 * kq + e] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) breg[r] = (kq * 4 + r < C && bias) ? bias[kq * 4 + r] : 0.f;
}

// Everything behind the 16 matrix instructions of seg_head_group: arg-max over the classes, softmax (+ softmax), tie rule.
// acc[r] = logit of class 4 * (lane >> 4) + r of pixel lane & 15 (C/D map of v_mfma_f32_16x16x4_f32).  Split out of seg_head_group so that a
// caller may run the matrix instructions itself (in seg_head_group's order: bit-identical labels and scores).
__device__ __forceinline__ void seg_head_finish(const f32x4h& acc, int C, int lane, int double_softmax, int& am_out, float& pm_out)
{
    const int kq = lane >> 4;
    // acc[r] = logit of class kq*4 + r for pixel px (C/D map of 16x16x4: row = 4*(lane>>4) + r, col = lane&15)
    float m = -__builtin_inff();
    int am = 0x7fffffff;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = kq * 4 + r;
        if (c < C && acc[r] > m) { m = acc[r]; am = c; }      // ascending c: first maximum of this lane's four
    }
    {
        const float om = lane_xor16(m, lane);
        const int oa = (int)lane_xor16_u((unsigned)am, lane);
        if (om > m || (om == m && oa < am)) { m = om; am = oa; }  // first maximum overall (torch.argmax on the CPU)
    }
    {
        const float om = lane_xor32(m, lane);
        const int oa = (int)lane_xor32_u((unsigned)am, lane);
        if (om > m || (om == m && oa < am)) { m = om; am = oa; }
    }
    // v_exp_f32-based exponentials and ONE reciprocal per softmax (a few ulp from expf / a true division, far inside the 1e-5
    // the score is compared at): the precise forms cost ~400 VALU instructions per 16-pixel group, 1.6 k per wave and tile of the
    // fused up_3 kernel, which is VALU-issue-bound
    float e[4], s = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { e[r] = (kq * 4 + r < C) ? __expf(acc[r] - m) : 0.f; s += e[r]; }
    s = quad_sum(s, lane);
    // The reference's arg-max runs over the PROBABILITIES it finally holds (pipeline/utils.py:430-435): after ONE softmax a class whose
    // logit lies so close below the maximum that exp(l - m) rounds to 1 has the same float32 probability; after TWO (the live path:
    // predict's activation, then F.softmax) the band is wider -- every class whose exp(p1 - p1max) rounds to 1.  torch.argmax returns
    // the LOWEST index of such a tie: take the lowest class whose last exponential is exactly 1 (the maximum itself always qualifies;
    // e == 1 implies p1 == p1max, so the two-softmax test contains the one-softmax one).
    // v_rcp_f32 (1 ulp): `1.f / s` is a ten-instruction IEEE division.  segpost.hip's seg_argmax_kernel (the unfused form over a logits
    // tensor) divides: at the edge of the tie band (a top probability below 0.5, where one ulp of p decides exp(p - pmax) == 1) the two can
    // name different classes of numerically equal probability; include/ape_hip.h says so at ape_seg_head_f32.
    float pm = __builtin_amdgcn_rcpf(s);
    float t[4];
    if (double_softmax) {
        const float inv = pm;
        float s2 = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { t[r] = (kq * 4 + r < C) ? __expf(e[r] * inv - pm) : 0.f; s2 += t[r]; }
        pm = __builtin_amdgcn_rcpf(quad_sum(s2, lane));
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = e[r];
    }
    {
        int tmin = am;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (t[r] == 1.f && kq * 4 + r < tmin) tmin = kq * 4 + r;
        const int o16 = (int)lane_xor16_u((unsigned)tmin, lane);
        tmin = o16 < tmin ? o16 : tmin;
        const int o32 = (int)lane_xor32_u((unsigned)tmin, lane);
        am = o32 < tmin ? o32 : tmin;
    }
    am_out = am;
    pm_out = pm;
}

// -> am (arg-max class, first maximum) and pm (its probability after one or two softmaxes), valid on every lane of the pixel
__device__ __forceinline__ void seg_head_group(const float4 (&x)[4], const float (&wreg)[16], const float (&breg)[4], int C, int lane,
                                               int double_softmax, int& am_out, float& pm_out)
{
    f32x4h acc = {breg[0], breg[1], breg[2], breg[3]};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[j * 4 + 0], x[j].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[j * 4 + 1], x[j].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[j * 4 + 2], x[j].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[j * 4 + 3], x[j].w, acc, 0, 0, 0);
    }
    seg_head_finish(acc, C, lane, double_softmax, am_out, pm_out);
}

}  // namespace ape_seg
