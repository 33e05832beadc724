// PSPUpsample with 64 output channels (DenseFusion/lib/pspnet.py:27-37: nn.Upsample(x2, bilinear, align_corners=True) -> Conv2d 3x3 pad 1
// -> PReLU; the segmentor's up_3, pspnet.py:51) as ONE kernel, optionally followed by the segmentation head (final 1x1 conv rows + softmax
// (+ softmax) + arg-max, pspnet.py:53-55, pipeline/utils.py:429-435) in the same launch.
//
// The direct form (conv3x3_halo.hip with the fused up-sampling) multiplies at HIGH resolution: 2 * 9 * Cin * 64 flop per output pixel.
// Conv and bilinear resize are both linear, so the channel mixing can run at LOW resolution -- z[p][tap][co] = sum_ci W[co][ci][tap] x[p][ci],
// 4x fewer flops -- and the output is  out(Y, X) = act(bias + sum_{ky,kx} lerp2d(z_{ky,kx})(Y + ky - 1, X + kx - 1))  (engine.UpConv:
// ape_conv_gemm_s32 + ape_upconv3x3_gather_f32 for up_1 / up_2).  For up_3 the tap tensor z would be 9 * 64 channels at 240 x 320:
// 11 GB per 64 frames.  Here it never leaves the CU:
//   * a workgroup (12 waves) owns a 16 x 24 output tile = the 10 x 16 low-resolution pixels under it (with the floor pattern of the
//     align_corners source index, iy0(q) = (q - 1) >> 1, the 18 up-sampled rows Y0 - 1 .. Y0 + 16 touch exactly ten low-resolution rows
//     and the 26 columns at most fourteen); their S32 rows are LDS-DMA'd once (XT, swizzled like conv3x3_halo_s32.hip's image);
//   * wave (kx, slab) multiplies the 160 pixels with the weights of the three taps (ky = 0..2, kx) of its 16 output channels:
//     v_mfma_f32_16x16x32_bf16, weights as the row operand, split-bf16 products in ape_conv_gemm_s32's order -- the accumulators ARE
//     z, bit for bit.  A lane holds ONE low-resolution column (lane & 15) and four channels in all ten rows, so the ROW interpolation and
//     the sum over ky  S_kx(Y, ix) = sum_ky lerp_y(z_{ky,kx}(., ix), Y + ky - 1)  is register arithmetic on the accumulators with
//     wave-uniform weights (static register indices: slot (t >> 1), (t >> 1) + 1 for t = y + ky);
//   * S goes to LDS (12 waves x 8 rows x 1 KB, the tile's top and bottom half in turn), and each wave then finishes one group of 16
//     output pixels x 64 channels per half: column interpolation + sum over kx from LDS, bias, activation, and either the store or the
//     head's sixteen exact-fp32 matrix instructions (seg_head.h, in seg_head_group's order) + soft-max + arg-max.
// Arithmetic = ape_conv_gemm_s32 followed by ape_upconv3x3_gather_ex(fma) operation for operation (tests/test_gpu_upfuse.py compares
// bit for bit); FMA picks the chained fused-multiply-add form of both interpolations.
// Persistent workgroups (one per CU) walk the tiles in an XCD-contiguous order; the next tile's pixels are requested as soon as the
// current tile's matrix phase is over and land during its gather phases.
#include <type_traits>
#include "common.h"
#include "s32.h"
#include "seg_head.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

struct UpFuseArgs {
    const char* x;          // S32 activations [B][h][w][G * 32]
    const char* w;          // S32K weights [9 * 64][G][hi 32 | lo 32], row = tap * 64 + co
    const float* bias;      // [64] or null
    char* y;                // !HEAD: out[B][2h][2w][64] in out_fmt
    const float* head_w;    // HEAD: [C][64]
    const float* head_b;    // HEAD: [C] or null
    uint8_t* label;         // HEAD: [B][2h][2w]
    float* score;
    int B, h, w_;
    int act;
    float alpha;
    float sh, sw;
    int out_fmt, head_c, head_dsm;
    int tiles_x, tiles_y;
};

constexpr int TY = 16, TX = 24;         // output tile
constexpr int ROWS = 10, COLS = 16;     // low-resolution pixels under it (slot r <-> row clamp(Y0 / 2 - 1 + r), slot s <-> column X0 / 2 - 1 + s)
constexpr int NW = 12;                  // waves: (kx = wave % 3, slab = wave / 3)
constexpr int SB_BYTES = NW * 8 * 1024; // S_kx rows of half a tile: [wave][y & 7][column slot 16][channel quad 4 (swizzled)] float4
constexpr int HW_BYTES = 16 * 64 * 4, HB_BYTES = 64, CB_BYTES = 256;
constexpr int xt_bytes(int G) { return G * ROWS * COLS * 128; }
constexpr int lds_bytes(int G) { return xt_bytes(G) + SB_BYTES + HW_BYTES + HB_BYTES + CB_BYTES; }

__device__ __forceinline__ float lane_value(float v, int l)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

template <int G, bool HEAD, bool FMA>
__global__ __launch_bounds__(NW * 64) void upconv_fused_kernel(const UpFuseArgs a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const XT = smem;
    char* const SB = smem + xt_bytes(G);
    float* const HW = reinterpret_cast<float*>(smem + xt_bytes(G) + SB_BYTES);
    float* const HB = HW + 16 * 64;
    float* const CB = HB + 16;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kx = wave % 3, slab = wave / 3;
    const int H2 = 2 * a.h, W2 = 2 * a.w_;

    // ---- constants of the gather phases: head weights (class-major, zero rows past C), head bias, conv bias
    if (HEAD) {
        for (int i = tid; i < 16 * 64; i += NW * 64) HW[i] = (i >> 6) < a.head_c ? a.head_w[i] : 0.f;
        if (tid < 16) HB[tid] = (tid < a.head_c && a.head_b) ? a.head_b[tid] : 0.f;
    }
    if (tid < 64) CB[tid] = a.bias ? a.bias[tid] : 0.f;
    const bool has_bias = a.bias != nullptr;

    // ---- tile walk: logical tile ids are dealt to the XCDs in contiguous runs (bijective remap of the dispatch id, as halo_s32_kernel)
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const int nt = a.B * tiles_per_img;
    const int q8 = nt / 8, r8 = nt % 8;
    const int grid = (int)gridDim.x;
    auto decode = [&](int orig, int& tb, int& ty0, int& tx0) {
        const int xcd = orig % 8;
        const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + orig / 8;
        tb = logical / tiles_per_img;
        const int trem = logical - tb * tiles_per_img;
        const int ty = trem / a.tiles_x;
        ty0 = ty * TY;
        tx0 = (trem - ty * a.tiles_x) * TX;
    };
    // the 10 x 16 pixels of a tile, 32-channel group by group: piece p = 8 pixels of one row (1 KB per wave-instruction); lanes 8 j .. 8 j + 7
    // fetch pixel j's 128-B line with the 16-B chunks permuted by the column swizzle (chunk c of column slot sp lives in slot c ^ ((sp >> 1) & 7));
    // columns outside the image lie outside the descriptor and arrive as zeros (no interpolation ever refers to them), rows are clamped
    const long frame = (long)a.h * a.w_ * G * 128;
    auto dma_tile = [&](int tb, int ty0, int tx0) {
        const int iyb = ty0 / 2 - 1, ixb = tx0 / 2 - 1;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (size_t)tb * frame), 0, (int)frame, 0x00020000);
        for (int p = wave; p < G * ROWS * 2; p += NW) {
            const int g = p / (ROWS * 2), rem = p - g * (ROWS * 2), rr = rem >> 1, half = rem & 1;
            int row = iyb + rr;
            row = row < 0 ? 0 : (row > a.h - 1 ? a.h - 1 : row);
            const int sp = half * 8 + (lane >> 3);
            const int col = ixb + sp;
            const bool ok = (unsigned)col < (unsigned)a.w_;
            const unsigned voff = ok ? (unsigned)(((row * a.w_ + col) * G + g) * 128 + (((lane & 7) ^ ((sp >> 1) & 7)) * 16)) : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(XT + p * 1024), 16, voff, 0, 0, 0);
        }
    };

    int orig = blockIdx.x;
    int b, Y0, X0;
    if (orig < nt) {
        decode(orig, b, Y0, X0);
        dma_tile(b, Y0, X0);
    }
    // the weights of this wave's (kx, slab): three taps x 16 rows, fetched through a descriptor (scalar base + ONE lane offset register)
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, 9 * 64 * G * 128, 0x00020000);

#pragma unroll 1
    for (; orig < nt; orig += grid) {
        decode(orig, b, Y0, X0);
        const int ixb = X0 / 2 - 1;
        // the tile's pixels (requested one tile ago) and, the first time, the constants
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

        // ---- matrix phase: z of this wave's three taps x 16 channels for the 10 x 16 pixels.  Every lane-derived address is formed INSIDE
        // the tile loop from an opaque copy of the lane id, per phase: hoisted in front of the loop (they are loop-invariant) they would
        // stay allocated through the 150-register matrix phase and spill (a scratch reload's vmcnt(0) would also drain the LDS-DMA)
        int ln_m = lane;
        asm volatile("" : "+v"(ln_m));
        const int s_m = ln_m & 15, fc_m = ln_m >> 4, sw7 = (s_m >> 1) & 7;
        const unsigned xoff_h = (unsigned)(s_m * 128 + ((fc_m ^ sw7) * 16)), xoff_l = (unsigned)(s_m * 128 + (((4 + fc_m) ^ sw7) * 16));
        const unsigned wv = (unsigned)(s_m * (G * 128) + fc_m * 16);
        f32x4 acc[3][ROWS];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int r = 0; r < ROWS; ++r) acc[ky][r] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < G; ++g) {
            bf16x8 wh[3], wl[3];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int soff = (((ky * 3 + kx) * 64 + slab * 16) * G + g) * 128;
                wh[ky] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, wv, soff, 0));
                wl[ky] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, wv + 64, soff, 0));
            }
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                const char* xp = XT + (g * ROWS + r) * (COLS * 128);
                const bf16x8 xh = *reinterpret_cast<const bf16x8*>(xp + xoff_h), xl = *reinterpret_cast<const bf16x8*>(xp + xoff_l);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    // weights as the row operand (D[channel 4 fc + e][pixel s]); product order of conv_gemm_s32.hip
                    acc[ky][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ky], xl, acc[ky][r], 0, 0, 0);
                    acc[ky][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[ky], xh, acc[ky][r], 0, 0, 0);
                    acc[ky][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ky], xh, acc[ky][r], 0, 0, 0);
                }
            }
        }

        // ---- row interpolation weights of the 18 up-sampled rows q = Y0 - 1 + t (lane t computes, everybody reads them as scalars)
        float l0v, l1v;
        bool okv;
        {
            const int qv = Y0 - 1 + lane;
            okv = lane < TY + 2 && qv >= 0 && qv < H2;
            const float fy = a.sh * (float)(okv ? qv : 0);
            const int iy0 = (int)fy;
            l1v = fy - (float)iy0;
            l0v = 1.f - l1v;
        }
        const unsigned okm = (unsigned)__ballot(okv);
        const bool top_tile = Y0 == 0;
        // S_kx(Y0 + y, column slot s) = sum_ky lerp_y(z_{ky,kx}(., s), q = Y0 + y + ky - 1): iy0(q) - (Y0 / 2 - 1) = t >> 1 for t = y + ky
        // (q = 0, the top tile's t = 1, is the one exception of the pattern: iy0 = 0 is slot 1 like slot 0 (clamped), iy1 = 1 is slot 2)
        auto ylerp_rows = [&](auto half_c) __attribute__((always_inline)) {
            constexpr int y_lo = decltype(half_c)::value * 8;
            int ln_y = lane;
            asm volatile("" : "+v"(ln_y));
            const int s_y = ln_y & 15, fc_y = ln_y >> 4;
            const unsigned s_wr = (unsigned)(wave * 8192 + s_y * 64 + ((fc_y ^ ((s_y >> 1) & 3)) * 16));    // this lane's float4 in row 0 of its wave's S block
#pragma unroll
            for (int yy = 0; yy < 8; ++yy) {
                const int y = y_lo + yy;
                f32x4 S = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int t = y + ky;
                    if ((okm >> t) & 1u) {
                        const float l0 = lane_value(l0v, t), l1 = lane_value(l1v, t);
                        const f32x4 v0 = acc[ky][t >> 1];
                        f32x4 v1 = acc[ky][(t >> 1) + 1];
                        if (t == 1 && top_tile) v1 = acc[ky][2];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (FMA) S[e] = __builtin_fmaf(l1, v1[e], __builtin_fmaf(l0, v0[e], S[e]));
                            else S[e] = S[e] + (l0 * v0[e] + l1 * v1[e]);
                        }
                    }
                }
                *reinterpret_cast<f32x4*>(SB + s_wr + yy * 1024) = S;
            }
        };
        // ---- one group of 16 output pixels x 64 channels per wave and half tile: column interpolation + sum over kx, bias, activation,
        // then the store or the head
        auto gather_half = [&](int hh) __attribute__((always_inline)) {
            int ln_g = lane;
            asm volatile("" : "+v"(ln_g));
            const int s = ln_g & 15, fc = ln_g >> 4;
            const ape::ActFast af = ape::act_fast_make(a.act, a.alpha);
            const int lin = wave * 16 + s;
            const int yl = lin / TX, Xl = lin - yl * TX;
            const int Y = Y0 + hh * 8 + yl;
            int X = X0 + Xl;
            const bool pix_ok = Y < H2 && X < W2;
            X = X < W2 ? X : W2 - 1;
            float lx0[3], lx1[3];
            unsigned o0[3], o1[3];
            bool okx[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int qx = X + k - 1;
                okx[k] = (unsigned)qx < (unsigned)W2;
                const float fx = a.sw * (float)(okx[k] ? qx : 0);
                const int ix0 = (int)fx, ix1 = ix0 + (ix0 < a.w_ - 1 ? 1 : 0);
                lx1[k] = fx - (float)ix0;
                lx0[k] = 1.f - lx1[k];
                const int s0 = (ix0 - ixb) & 15, s1 = (ix1 - ixb) & 15;
                o0[k] = (unsigned)(yl * 1024 + s0 * 64 + ((fc ^ ((s0 >> 1) & 3)) * 16));
                o1[k] = (unsigned)(yl * 1024 + s1 * 64 + ((fc ^ ((s1 >> 1) & 3)) * 16));
            }
            ape_seg::f32x4h hacc;
            if (HEAD) {
                const float4 hb = *reinterpret_cast<const float4*>(HB + fc * 4);
                hacc = ape_seg::f32x4h{hb.x, hb.y, hb.z, hb.w};
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const char* sp = SB + (j * 3 + k) * 8192;
                    const f32x4 a0 = *reinterpret_cast<const f32x4*>(sp + o0[k]), a1 = *reinterpret_cast<const f32x4*>(sp + o1[k]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t;
                        if (FMA) t = __builtin_fmaf(lx1[k], a1[e], __builtin_fmaf(lx0[k], a0[e], o[e]));
                        else t = o[e] + (lx0[k] * a0[e] + lx1[k] * a1[e]);
                        o[e] = okx[k] ? t : o[e];
                    }
                }
                if (has_bias) {
                    const float4 cb = *reinterpret_cast<const float4*>(CB + 16 * j + 4 * fc);
                    o[0] += cb.x; o[1] += cb.y; o[2] += cb.z; o[3] += cb.w;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = ape::act_fast(o[e], af);
                if (HEAD) {
                    const float4 wv = *reinterpret_cast<const float4*>(HW + s * 64 + 16 * j + 4 * fc);
                    hacc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.x, o[0], hacc, 0, 0, 0);
                    hacc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.y, o[1], hacc, 0, 0, 0);
                    hacc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.z, o[2], hacc, 0, 0, 0);
                    hacc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.w, o[3], hacc, 0, 0, 0);
                } else if (pix_ok) {
                    const size_t pix = ((size_t)b * H2 + Y) * W2 + X;
                    const float4 ov = make_float4(o[0], o[1], o[2], o[3]);
                    if (a.out_fmt == APE_FMT_S32) ape::s32_store4(a.y, (long)pix, 16, 4 * j + fc, ov);
                    else *reinterpret_cast<float4*>(a.y + (pix * 64 + 16 * j + 4 * fc) * 4) = ov;
                }
            }
            if (HEAD) {
                int am;
                float pm;
                ape_seg::seg_head_finish(hacc, a.head_c, ln_g, a.head_dsm, am, pm);
                if (fc == 0 && pix_ok) {
                    const size_t pix = ((size_t)b * H2 + Y) * W2 + X;
                    a.label[pix] = (uint8_t)am;
                    a.score[pix] = pm;
                }
            }
        };

        ylerp_rows(std::integral_constant<int, 0>{});
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // XT is free: the next tile's pixels travel during the gather phases
        if (orig + grid < nt) {
            int nb, ny0, nx0;
            decode(orig + grid, nb, ny0, nx0);
            dma_tile(nb, ny0, nx0);
        }
        gather_half(0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        ylerp_rows(std::integral_constant<int, 1>{});
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        gather_half(1);
    }
#endif
}

// the register form's floor pattern: (int)(scale * q) == (q - 1) >> 1 for q >= 1 (0 for q = 0), scale = (n - 1) / (2 n - 1) in fp32
bool floor_pattern_ok(int n)
{
    const float sc = 2 * n > 1 ? (float)(n - 1) / (float)(2 * n - 1) : 0.f;
    for (int q = 0; q < 2 * n; ++q) {
        const int want = q >= 1 ? (q - 1) >> 1 : 0;
        if ((int)(sc * (float)q) != want) return false;
    }
    return true;
}

bool fused_supported(int h, int w, int Cin, int Cout)
{
    if (Cin != 64 || Cout != 64 || h < 1 || w < 1) return false;
    if ((long)h * w * Cin * 4 >= (1L << 31)) return false;
    return floor_pattern_ok(h) && floor_pattern_ok(w);
}

template <int G, bool HEAD, bool FMA>
int launch_fused(const UpFuseArgs& a, hipStream_t st)
{
    auto kern = upconv_fused_kernel<G, HEAD, FMA>;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    static unsigned long long attr_set = 0;      // one bit per device (the attribute is per device, not per process)
    static int ncu_of[64];
    const int di = dev & 63;
    if (!((attr_set >> di) & 1ull)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes(G)) != hipSuccess) {
            ape::set_last_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
            return APE_ELAUNCH;
        }
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu < 8) ncu = 256;
        ncu_of[di] = ncu;
        attr_set |= 1ull << di;
    }
    const long nt = (long)a.B * a.tiles_x * a.tiles_y;
    int grid = (int)(nt < ncu_of[di] ? nt : ncu_of[di]);
    if (grid > 8) grid -= grid % 8;             // whole XCD rounds: dispatch id % 8 labels a workgroup's XCD for every tile it walks
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds_bytes(G), st, a);
    return ape::check_launch("ape_upconv3x3_fused");
}

int run_fused(const void* x, const void* w9, const float* bias, void* out, int out_fmt, int B, int h, int w, int Cin, int act, float alpha,
              int fma, const float* head_w, const float* head_b, int C, uint8_t* label, float* score, int dsm, bool head, void* stream)
{
    if (!x || !w9 || B < 0 || (fma != 0 && fma != 1)) return APE_EINVAL;
    if (!fused_supported(h, w, Cin, 64)) return APE_EINVAL;
    if (act < APE_ACT_NONE || act > APE_ACT_PRELU) return APE_EINVAL;
    if (head) {
        if (!head_w || !label || !score || C < 1 || C > 16) return APE_EINVAL;
    } else {
        if (!out || (out_fmt != APE_FMT_F32 && out_fmt != APE_FMT_S32)) return APE_EINVAL;
    }
    if ((long)B * 4 * h * w >= (1L << 31)) return APE_EINVAL;
    if (B == 0) return APE_OK;
    UpFuseArgs a;
    a.x = (const char*)x; a.w = (const char*)w9; a.bias = bias; a.y = (char*)out;
    a.head_w = head_w; a.head_b = head_b; a.label = label; a.score = score;
    a.B = B; a.h = h; a.w_ = w; a.act = act; a.alpha = alpha;
    a.sh = 2 * h > 1 ? (float)(h - 1) / (float)(2 * h - 1) : 0.f;
    a.sw = 2 * w > 1 ? (float)(w - 1) / (float)(2 * w - 1) : 0.f;
    a.out_fmt = out_fmt; a.head_c = C; a.head_dsm = dsm;
    a.tiles_x = ape::ceil_div(2 * w, TX); a.tiles_y = ape::ceil_div(2 * h, TY);
    hipStream_t st = (hipStream_t)stream;
    if (head) return fma ? launch_fused<2, true, true>(a, st) : launch_fused<2, true, false>(a, st);
    return fma ? launch_fused<2, false, true>(a, st) : launch_fused<2, false, false>(a, st);
}

}  // namespace

extern "C" int ape_upconv3x3_fused_supported(int h, int w, int Cin, int Cout) { return fused_supported(h, w, Cin, Cout) ? 1 : 0; }

extern "C" int ape_upconv3x3_fused_s32(const void* x_s32, const void* w9_s32k, const float* bias, void* out, int out_fmt, int B, int h, int w,
                                       int Cin, int act, float alpha, int fma, void* stream)
{
    return run_fused(x_s32, w9_s32k, bias, out, out_fmt, B, h, w, Cin, act, alpha, fma, nullptr, nullptr, 0, nullptr, nullptr, 0, false, stream);
}

extern "C" int ape_upconv3x3_fused_seghead_s32(const void* x_s32, const void* w9_s32k, const float* bias, int B, int h, int w, int Cin, int act,
                                               float alpha, int fma, const float* head_w, const float* head_b, int C, uint8_t* label, float* score,
                                               int double_softmax, void* stream)
{
    return run_fused(x_s32, w9_s32k, bias, nullptr, APE_FMT_F32, B, h, w, Cin, act, alpha, fma, head_w, head_b, C, label, score, double_softmax, true,
                     stream);
}
