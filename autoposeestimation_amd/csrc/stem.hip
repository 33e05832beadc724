// ResNet stem fused: Conv2d(3(4) -> 64, 7x7, stride 2, pad 3, no bias) + ReLU + MaxPool2d(3, 2, 1)   (extractors.py:82-85,
// 111-117) in ONE kernel on the bf16 matrix cores (split-bf16 x3 or plain bf16 operands, fp32 accumulate).
//
// Why: as two launches the stem wrote its 64-channel half-resolution activation (1.26 GB for 64 frames) and the pool read it
// back -- 0.8 + 0.34 ms for 0.12 TFLOP.  Here a workgroup owns a 4 x 8 tile of POOLED pixels: it stages the 23 x 40-pixel
// input patch under the 9 x 17 conv outputs the tile needs (split to bf16 hi/lo once, 16 B per pixel), multiplies straight
// out of the patch, pools in LDS and writes only the pooled tile (0.31 GB).
//
// GEMM view: M = 153 conv pixels (10 row blocks of 16), N = 64 (one 16-channel block per wave), K = 7 kernel rows x 8 kernel
// columns (column 7 has zero weights) x 4 channels = 7 k-steps of 32.  A lane's 8 consecutive k of a k-step are two
// ADJACENT input pixels of one patch row, i.e. 16 contiguous bytes of the patch: the A fragments are ds_read_b128s of the
// patch itself (no im2col image), and lanes that share a pixel pair read the same address (broadcast, conflict-free).
// The 7 x 2 weight fragments of a wave's 16 output channels stay in registers for the whole (persistent) kernel.
//
// U8 form (ape_stem_conv_pool_u8): the patch is read from the uint8 RGB frames themselves -- crop o = the Hc x Wc window of frame
// rects[o][0] at (rects[o][1], rects[o][2]); NULL rects: whole frames -- and ToTensor / Normalize (pipeline/utils.py:421-427, 556-560)
// run on the way into LDS with ape_preprocess_u8_nhwc4's own expressions, so the results are bit for bit those of the two launches
// while the fp32 NHWC4 image (16 B per pixel: 315 MB written and read back per 64 frames) never exists.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int PTY = 4, PTX = 8;                 // pooled tile
constexpr int CR = 2 * PTY + 1, CC = 2 * PTX + 1;   // conv outputs under it: 9 x 17
constexpr int NCV = CR * CC;                    // 153
constexpr int NRB = (NCV + 15) / 16;            // 10 row blocks
constexpr int NRW = NRB / 2;                    // row blocks per wave (a wave = a pair of channel blocks x half of the row blocks)
static_assert(NRB % 2 == 0, "the row blocks are dealt to two waves");
constexpr int PR = 2 * CR + 5, PC = 2 * CC + 5 + 1; // input patch 23 x 40 (one extra column for the zero-weight kernel column 7)
constexpr int NPATCH = PR * PC;                 // 920 pixels
constexpr int ELD = 68;                         // floats per staged conv pixel (64 channels + pad)
constexpr int PITEMS = (NPATCH + 255) / 256;    // patch pixels per thread (4)

template <int NSPLIT, bool U8>
__global__ __launch_bounds__(256, 2) void stem_pool_kernel(const void* __restrict__ xin, const int* __restrict__ rects, int Hf, int Wf, long npix, int div255,
                                                           const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y, int B, int H, int W,
                                                           int Ho, int Wo, int Hp, int Wp, int tiles_y, int tiles_x)
{
    const float4* const x = reinterpret_cast<const float4*>(xin);
    const uint8_t* const x8 = reinterpret_cast<const uint8_t*>(xin);
    // U8: the 3 x 256 possible normalised values, computed once with crop_normalize_kernel's own expressions (six fp32 divisions per pixel
    // in the patch load cost more than the launch they replace)
    __shared__ float lut[U8 ? 768 : 1];
    if (U8) {
        for (int i = threadIdx.x; i < 768; i += 256) {
            const int c = i >> 8;
            float v0 = (float)(i & 255);
            if (div255) v0 = v0 / 255.f;                                                        // torchvision ToTensor
            lut[i] = c == 0 ? (v0 - 0.485f) / 0.229f : c == 1 ? (v0 - 0.456f) / 0.224f : (v0 - 0.406f) / 0.225f;
        }
        __syncthreads();
    }
    constexpr int NPL = NSPLIT == 3 ? 2 : 1;
    __shared__ __attribute__((aligned(16))) char smem[NPL * NPATCH * 8 + NCV * ELD * 4];
    __bf16* const patch = reinterpret_cast<__bf16*>(smem);                 // [NPL][NPATCH][4]
    float* const stage = reinterpret_cast<float*>(smem + NPL * NPATCH * 8);   // [NCV][ELD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kq = lane >> 4;

    // ---- this wave's weights: 2 x 16 output channels x (7 kernel rows x [8 columns x 4 channels]) as MFMA B fragments.  A wave owns the
    // channel blocks 2 cp, 2 cp + 1 and HALF of the row blocks (mh): every A fragment it reads feeds six matrix instructions instead of
    // three, the four waves read the patch twice per tile instead of four times (it was 573 KB of ds_read_b128 per tile) ----------------
    const int cp = wave >> 1, mh = wave & 1;
    bf16x8 bh[2][7], bl[2][7];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int co = (2 * cp + cb) * 16 + r16;
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int kx = 2 * kq + (j >> 2), ch = j & 3;
                const float v = kx < 7 ? w[((co * 7 + ky) * 7 + kx) * 4 + ch] : 0.f;
                bh[cb][ky][j] = (__bf16)v;
                bl[cb][ky][j] = (__bf16)(v - (float)bh[cb][ky][j]);
            }
    }
    // ---- this lane's A fragment origin per row block: element offset of patch pixel (2 cy, 2 cx + 2 kq) ----------------------
    int abase[NRW];
#pragma unroll
    for (int i = 0; i < NRW; ++i) {
        int m = (mh * NRW + i) * 16 + r16;
        m = m < NCV ? m : NCV - 1;
        const int cy = m / CC, cx = m - cy * CC;
        abase[i] = ((2 * cy) * PC + 2 * cx + 2 * kq) * 4;
    }
    const float cbias0 = bias ? bias[(2 * cp) * 16 + r16] : 0.f, cbias1 = bias ? bias[(2 * cp + 1) * 16 + r16] : 0.f;

    const int ntiles = B * tiles_y * tiles_x;
    // U8: the pixel's raw bytes wait in `pu` (+ one validity bit each) and go through the table in store_patch(), i.e. AFTER the matrix
    // phase and the pooling they were requested under -- a table look-up in load_patch() made the wave wait for the loads right there
    float4 preg[U8 ? 1 : PITEMS];
    unsigned int pu[U8 ? PITEMS : 1];
    unsigned int pok = 0;
    auto load_patch = [&](int tile) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
        const int iy0 = 4 * ty * PTY - 5, ix0 = 4 * tx * PTX - 5;
#pragma unroll
        for (int i = 0; i < PITEMS; ++i) {
            const int e = tid + 256 * i;
            const int pr = e / PC, pc = e - pr * PC;
            const int iy = iy0 + pr, ix = ix0 + pc;
            const bool ok = e < NPATCH && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            const int cyc = iy < 0 ? 0 : (iy >= H ? H - 1 : iy), cxc = ix < 0 ? 0 : (ix >= W ? W - 1 : ix);
            if (U8) {
                const int fb = rects ? rects[b * 3] : b, fr = (rects ? rects[b * 3 + 1] : 0) + cyc, fc = (rects ? rects[b * 3 + 2] : 0) + cxc;
                // the pixel's three bytes by ONE (unaligned) 32-bit load; the very last pixel of the buffer is read one byte early and shifted
                const long pi = ((long)fb * Hf + fr) * Wf + fc;
                const int last = pi == npix - 1 ? 1 : 0;
                typedef unsigned int u32_unaligned __attribute__((aligned(1)));
                pu[i] = *reinterpret_cast<const u32_unaligned*>(x8 + pi * 3 - last) >> (8 * last);
                pok = (pok & ~(1u << i)) | ((ok ? 1u : 0u) << i);
            } else {
                const float4 v = x[((long)b * H + cyc) * W + cxc];          // unconditional (clamped), zeroed below: conv zero padding
                preg[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int i = 0; i < PITEMS; ++i) {
            const int e = tid + 256 * i;
            if (e >= NPATCH) continue;
            float4 v;
            if (U8) {
                const unsigned int u = pu[i];
                v = (pok >> i) & 1u ? make_float4(lut[u & 255u], lut[256 + ((u >> 8) & 255u)], lut[512 + ((u >> 16) & 255u)], 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                v = preg[i];
            }
            bf16x4 hi, lo;
            hi[0] = (__bf16)v.x; hi[1] = (__bf16)v.y; hi[2] = (__bf16)v.z; hi[3] = (__bf16)v.w;
            *reinterpret_cast<bf16x4*>(patch + e * 4) = hi;
            if (NPL == 2) {
                lo[0] = (__bf16)(v.x - (float)hi[0]); lo[1] = (__bf16)(v.y - (float)hi[1]);
                lo[2] = (__bf16)(v.z - (float)hi[2]); lo[3] = (__bf16)(v.w - (float)hi[3]);
                *reinterpret_cast<bf16x4*>(patch + NPATCH * 4 + e * 4) = lo;
            }
        }
    };

    int tile = blockIdx.x;
    if (tile < ntiles) load_patch(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        store_patch();
        __syncthreads();
        const int next = tile + gridDim.x;
        if (next < ntiles) load_patch(next);                 // flies under the MFMAs and the pooling of this tile

        f32x4 acc[NRW][2];
#pragma unroll
        for (int i = 0; i < NRW; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[i][0][e] = cbias0; acc[i][1][e] = cbias1; }
        // (per output element the same chain as ever: ky ascending, lo x hi, hi x lo, hi x hi)
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
            for (int i = 0; i < NRW; ++i) {
                const int o = abase[i] + ky * PC * 4;
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(patch + o);
                if (NPL == 2) {
                    const bf16x8 al = *reinterpret_cast<const bf16x8*>(patch + NPATCH * 4 + o);
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) {
                        acc[i][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[cb][ky], acc[i][cb], 0, 0, 0);
                        acc[i][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[cb][ky], acc[i][cb], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) acc[i][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[cb][ky], acc[i][cb], 0, 0, 0);
            }
        // ReLU'd conv outputs -> LDS [conv pixel][channel]   (C/D map: row = 4 kq + e is the pixel, column = r16 the channel)
#pragma unroll
        for (int i = 0; i < NRW; ++i)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int m = (mh * NRW + i) * 16 + kq * 4 + e;
                    if (m < NCV) stage[m * ELD + (2 * cp + cb) * 16 + r16] = fmaxf(acc[i][cb][e], 0.f);
                }
        __syncthreads();
        // 3x3 / stride 2 / pad 1 max-pool over the staged conv tile; conv pixels outside the conv image do not take part
        {
            const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
            const int py0 = ty * PTY, px0 = tx * PTX;
            for (int it = tid; it < PTY * PTX * 16; it += 256) {
                const int c4 = it & 15, pp = it >> 4;
                const int ly = pp / PTX, lx = pp - ly * PTX;
                const int py = py0 + ly, px = px0 + lx;
                if (py >= Hp || px >= Wp) continue;
                // branch-free: a window position outside the conv image reads the window's CENTRE instead (conv pixel (2 py, 2 px) always
                // exists and is in the maximum anyway), so the nine reads are issued back to back -- behind a `continue` each one sat in
                // its own block with its own wait
                float4 m4 = make_float4(-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff());
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int sy = (unsigned)(2 * py - 1 + dy) < (unsigned)Ho ? 2 * ly + dy : 2 * ly + 1;
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const int sx = (unsigned)(2 * px - 1 + dx) < (unsigned)Wo ? 2 * lx + dx : 2 * lx + 1;
                        const float4 v = *reinterpret_cast<const float4*>(&stage[(sy * CC + sx) * ELD + c4 * 4]);
                        m4.x = fmaxf(m4.x, v.x); m4.y = fmaxf(m4.y, v.y); m4.z = fmaxf(m4.z, v.z); m4.w = fmaxf(m4.w, v.w);
                    }
                }
                *reinterpret_cast<float4*>(y + (((long)b * Hp + py) * Wp + px) * 64 + c4 * 4) = m4;
            }
        }
        __syncthreads();      // the patch and the staging rows are free for the next tile
    }
}

}  // namespace

static int stem_run(const void* x, bool u8, const int* rects, int Hf, int Wf, long npix, int div255, const float* w, const float* bias, float* y,
                    int B, int H, int W, int nsplit, void* stream)
{
    if (!x || !w || !y || B < 0 || H < 1 || W < 1 || (nsplit != 1 && nsplit != 3)) return APE_EINVAL;
    if (B == 0) return APE_OK;
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    const int Hp = (Ho + 2 - 3) / 2 + 1, Wp = (Wo + 2 - 3) / 2 + 1;
    if (Ho < 1 || Wo < 1) return APE_EINVAL;
    const int tiles_y = ape::ceil_div(Hp, PTY), tiles_x = ape::ceil_div(Wp, PTX);
    const long ntiles = (long)B * tiles_y * tiles_x;
    if (ntiles >= (1L << 31) || (long)B * H * W >= (1L << 31)) return APE_EINVAL;
    const int grid = (int)(ntiles < 512 ? ntiles : 512);        // persistent: two workgroups per CU walk the tiles
    hipStream_t st = (hipStream_t)stream;
#define APE_STEM_LAUNCH(NS, U)                                                                                                              \
    hipLaunchKernelGGL((stem_pool_kernel<NS, U>), dim3(grid), dim3(256), 0, st, x, rects, Hf, Wf, npix, div255, w, bias, y, B, H, W, Ho, Wo, Hp, Wp, \
                       tiles_y, tiles_x)
    if (u8) { if (nsplit == 3) APE_STEM_LAUNCH(3, true); else APE_STEM_LAUNCH(1, true); }
    else { if (nsplit == 3) APE_STEM_LAUNCH(3, false); else APE_STEM_LAUNCH(1, false); }
#undef APE_STEM_LAUNCH
    return ape::check_launch("ape_stem_conv_pool");
}

/* x[B][H][W][4] f32 (RGB + zero pad), w[64][7][7][4] f32, bias[64] or NULL -> y[B][Hp][Wp][64],
 * Hp = ((H + 6 - 7) / 2 + 1 + 2 - 3) / 2 + 1 (the 7x7/s2/p3 conv followed by ReLU and the 3x3/s2/p1 max-pool) */
extern "C" int ape_stem_conv_pool_bf16(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int nsplit,
                                       void* stream)
{
    return stem_run(x, false, nullptr, H, W, 0, 0, w, bias, y, B, H, W, nsplit, stream);
}

/* the same on the uint8 frames: rgb[n_frames][Hf][Wf][3], crop o = the Hc x Wc window of frame rects[o][0] at row rects[o][1], column
 * rects[o][2] (rects NULL: n whole frames, Hc = Hf, Wc = Wf), ToTensor (div255) + Normalize fused into the patch load */
extern "C" int ape_stem_conv_pool_u8(const uint8_t* rgb, int n_frames, const int* rects, const float* w, const float* bias, float* y, int n, int Hf,
                                     int Wf, int Hc, int Wc, int div255, int nsplit, void* stream)
{
    if (n_frames < 1 || Hf < 1 || Wf < 1 || Hc > Hf || Wc > Wf || (!rects && (Hc != Hf || Wc != Wf || n != n_frames))) return APE_EINVAL;
    return stem_run(rgb, true, rects, Hf, Wf, (long)n_frames * Hf * Wf, div255, w, bias, y, n, Hc, Wc, nsplit, stream);
}
