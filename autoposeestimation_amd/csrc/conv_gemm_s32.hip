// 1x1 convolutions (row-major GEMMs) on PRE-SPLIT activations, operands streamed HBM -> LDS by LDS-DMA.
//
// "S32" activation layout (the split-bf16 operands of conv_gemm.hip / conv3x3_halo.hip, made once by the PRODUCER's epilogue
// instead of by every consumer tile): a pixel's C channels (C % 32 == 0) are C/32 groups of 128 bytes,
//      [ hi: 32 x bf16 | lo: 32 x bf16 ],     hi = bf16(v),  lo = bf16(v - hi)        (4 bytes per channel, like fp32)
// so that a 32-channel k-tile of a pixel is ONE 128-byte line that is already the MFMA operand pair.  Weights use the same
// grouping along K ("S32K": [Cout][K/32][hi 32 | lo 32]).  The MFMA inputs are bit-identical to what conv_gemm.hip derives from
// fp32 activations; only non-MFMA consumers (residual adds) see hi + lo instead of the fp32 value (2^-17 relative).
//
// Main loop (256 pixels x BN channels per workgroup, 8 waves = 2 x 4, wave tile 128 x BN/4, v_mfma_f32_16x16x32_bf16, weights as
// the row operand like conv_gemm.hip):
//   * staging costs NO registers and NO VALU: every wave issues 4 + BN/64 `buffer_load_dwordx4 ... lds` per k-tile (1 KB each =
//     8 rows x 128 B); rows past M / Cout are out of the buffer descriptor's range and arrive as zeros;
//   * LDS image: rows of 128 B, 16-B chunk c of row r (c = 0..3 hi, 4..7 lo) in slot c ^ ((r >> 1) & 7).  The DMA writes LDS
//     linearly (lanes 8 r .. 8 r + 7 = one row = one whole 128-B line: the permutation is on the per-lane SOURCE address, inside
//     the line); the 16 lanes of every ds_read_b128 lane group (rows r..r+3, r+12..r+15 of chunk k and rows r+4..r+11 of chunk
//     k+1) hit 16 distinct 16-B bank slots;
//   * LDS rings: three slots for the pixel tiles, two for the weight tiles (160 KB at BN = 256), counted vmcnt: up to 96 KB per
//     CU in flight across the barrier (with two stages and one tile in flight the DMA round trip, ~1.9 us per 64 KB, was as long
//     as the k-tile's MFMAs and every barrier waited for it);
//   * ONE barrier per k-tile; the k-tile is cut into four phases of (4 x TN/2 x 3) MFMAs -- (A0,B0) (A0,B1)
//     (A1,B1) (A1,B0) -- and the fragments of the NEXT phase are read from LDS while the current phase's MFMAs run (two A
//     register sets, two B sets, the B sets swap roles every k-tile), so no wave ever waits on the LDS port in front of its
//     MFMAs: the freed staging registers are what pays for the second fragment set.
#include <type_traits>
#include "common.h"

// timing-only ablation switches (results wrong when set): compiled IN only by `make ablations` (-DAPE_ABLATIONS -> ../libape_hip_abl.so, what
// tools/mb_*_abl.py load).  In the product build they cost: the run-time tests around the MFMA rows, waits and DMA statements of halo_s32
// were 1 ms of the 33 ms step (same-box A/B).
#ifdef APE_ABLATIONS
#define ABL(bit) (a.dbg & (bit))
#else
#define ABL(bit) 0
#endif

// S32 output: one 16-byte store per lane and (pixel block, channel block) instead of two 8-byte ones (round 6; same bytes at the same addresses:
// bit-identical tensors; up_2 mix -3.7 %, up_1 mix -1.8 %, tools/mb_gemm_s32.py).  APE_S32_WIDE_STORE=0 builds the two-store form.
#ifndef APE_S32_WIDE_STORE
#define APE_S32_WIDE_STORE 1
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

struct GemmS32Args {
    const char* x;          // S32 activations, row m at x + m * ldx * 4 (bytes)
    const char* w;          // S32K weights, row n at w + n * K * 4
    const float* bias;
    const char* res;        // residual (fp32 or S32, res_fmt) or null
    char* y;
    int M, K, Cout;
    int ldx, xoff, ldy, yoff, ldr, roff;
    int act;
    float alpha;
    int bias_bstride, rows_per_image;
    int out_fmt, res_fmt;   // APE_FMT_F32 / APE_FMT_S32
    int m_tiles, n_tiles, nk;
    int img_tiles;          // 0: M is one flat run of rows cut into 256-row tiles; > 0: every image (rows_per_image rows) is cut into img_tiles tiles
                            // of its own -- no tile holds rows of two images -- and ...
    long w_img_stride;      // ... image i multiplies with the weights at w + i * w_img_stride bytes (0: one weight matrix for all)
    int dbg;                // timing ablations (results wrong): 1 no in-loop DMA, 2 no in-loop barrier, 4 no MFMAs, 8 no fragment reads, 512 two of the three MFMAs;
                            // 16: per-workgroup k-tile rotation (valid results; tested against L2 hot-spotting on the shared weight lines: +-0);
                            // 32: no static priority for waves 4-7; 64: one tile per workgroup (no persistent walk)
};

// NONE / RELU / PRELU as selects on loop-invariant scalars (a `switch` per element compiled to a cascade of scalar compares and branches
// per element); same values bit for bit (1 * v == v, also for -0 and NaN)
__device__ __forceinline__ float act_fn(float v, int act, float alpha)
{
    if (act == APE_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
    const float neg = act == APE_ACT_RELU ? 0.f : (act == APE_ACT_PRELU ? alpha : 1.f) * v;
    return v > 0.f ? v : neg;
}


// (the host pass of hipcc instantiates kernel bodies too and silently drops a kernel whose body holds an asm it cannot check
// against the HOST target -- the library then lacks the kernel's stub -- so the statement exists in the device pass only)
#if defined(__HIP_DEVICE_COMPILE__)
#define APE_DS_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#else
#define APE_DS_READ(dst, addr, off) (void)(addr)
#endif
// four 16-row blocks x two planes of one operand half (block i at + i * 2048; the planes have their own base registers)
template <int OFF>
__device__ __forceinline__ void ds_read_frags4(u32x4 (&f)[4][2], unsigned hi, unsigned lo)
{
    APE_DS_READ(f[0][0], hi, OFF);
    APE_DS_READ(f[0][1], lo, OFF);
    APE_DS_READ(f[1][0], hi, OFF + 2048);
    APE_DS_READ(f[1][1], lo, OFF + 2048);
    APE_DS_READ(f[2][0], hi, OFF + 4096);
    APE_DS_READ(f[2][1], lo, OFF + 4096);
    APE_DS_READ(f[3][0], hi, OFF + 6144);
    APE_DS_READ(f[3][1], lo, OFF + 6144);
}
template <int OFF, int NJ, int TNH>
__device__ __forceinline__ void ds_read_frags(u32x4 (&f)[TNH][2], unsigned hi, unsigned lo)
{
    if constexpr (NJ >= 1) {
        APE_DS_READ(f[0][0], hi, OFF);
        APE_DS_READ(f[0][1], lo, OFF);
    }
    if constexpr (NJ >= 2) {
        APE_DS_READ(f[1][0], hi, OFF + 2048);
        APE_DS_READ(f[1][1], lo, OFF + 2048);
    }
}
// PP form: all N 16-row blocks x two planes of one operand (block i at + i * 2048)
template <int N, int I = 0>
__device__ __forceinline__ void ds_read_blocks(u32x4 (&f)[N][2], unsigned hi, unsigned lo)
{
    if constexpr (I < N) {
        APE_DS_READ(f[I][0], hi, I * 2048);
        APE_DS_READ(f[I][1], lo, I * 2048);
        ds_read_blocks<N, I + 1>(f, hi, lo);
    }
}
#undef APE_DS_READ

// keeps asm-read destinations allocated up to this point (a free function: asm operands cannot name variables captured by a generic
// lambda; device pass only: the host pass cannot check a "v" constraint)
__device__ __forceinline__ void keep_regs(const u32x4& a, const u32x4& b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" :: "v"(a), "v"(b));
#endif
}

constexpr int BM = 256;
constexpr int A_STAGE = BM * 128;        // bytes of the A image of one k-tile

// ---- epilogue straight from the registers (both schedules): lane (frow, fc) holds channels 16 j + 4 fc .. + 3 of pixel 16 i + frow ----------
template <int BN, bool RES>
__device__ __forceinline__ void gemm_s32_epilogue(const GemmS32Args& a, f32x4 (&acc)[8][BN / 64], int m0, int m_end, int n0, int wm, int wn, int lane)
{
    constexpr int TN = BN / 64;
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int frow_e = lane_e & 15, fc_e = lane_e >> 4;
    const int nq = n0 + wn * (TN * 16) + fc_e * 4;
    const ape::ActFast af = ape::act_fast_make(a.act, a.alpha);
    const bool sigm = a.act == APE_ACT_SIGMOID;
#if APE_S32_WIDE_STORE
    const bool wide = a.Cout % 8 == 0 && a.yoff % 8 == 0;          // (a pair of lanes = two adjacent channel quads: both inside Cout or both outside)
#endif
    float4 b4[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = nq + j * 16;
        b4[j] = (a.bias && !a.bias_bstride && n < a.Cout) ? *reinterpret_cast<const float4*>(a.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // The residual tile is requested FOUR pixel rows ahead of its use, as two 8-byte halves per (row, block) for both formats (S32: hi 4 x
    // bf16 | lo 4 x bf16, 64 B apart; fp32: floats 0,1 | 2,3): 4 x TN loads in flight per round trip instead of TN -- the stores of
    // row i may alias the loads of row i + 1 for all the compiler knows, so the row-at-a-time form paid eight serialised round trips
    // per tile (the PSP bottleneck, whose residual is the 1.26 GB prior sum, spent as long in its epilogue as in its k-loop).
    const bool rs32 = a.res_fmt == APE_FMT_S32;
    const long rsecond = rs32 ? 64 : 8;
#pragma unroll
    for (int i0 = 0; i0 < 8; i0 += 4) {
        uint2 rlo[4][TN], rhi[4][TN];
        if constexpr (RES) {
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int m = m0 + wm * 128 + (i0 + ii) * 16 + frow_e;
                const size_t mc = m < a.M ? m : a.M - 1;                       // (a clamped, never-used address for rows past M)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = nq + j * 16;
                    const int cr = a.roff + (n < a.Cout ? n : 0);
                    const char* rp = a.res + (rs32 ? mc * a.ldr * 4 + (size_t)((cr >> 5) * 128 + (cr & 31) * 2) : (mc * a.ldr + cr) * 4);
                    rlo[ii][j] = *reinterpret_cast<const uint2*>(rp);
                    rhi[ii][j] = *reinterpret_cast<const uint2*>(rp + rsecond);
                }
            }
        }
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const int i = i0 + ii;
            const int m = m0 + wm * 128 + i * 16 + frow_e;
            if (m >= m_end) continue;
            const float* brow = (a.bias && a.bias_bstride) ? a.bias + (size_t)(m / a.rows_per_image) * a.bias_bstride : nullptr;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = nq + j * 16;
                if (n >= a.Cout) continue;
                float vv[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                if (brow) { vv[0] += brow[n]; vv[1] += brow[n + 1]; vv[2] += brow[n + 2]; vv[3] += brow[n + 3]; }
                else { vv[0] += b4[j].x; vv[1] += b4[j].y; vv[2] += b4[j].z; vv[3] += b4[j].w; }
                if constexpr (RES) {
                    if (rs32) {
                        const bf16x4 h = __builtin_bit_cast(bf16x4, rlo[ii][j]), l = __builtin_bit_cast(bf16x4, rhi[ii][j]);
                        // (hi + lo first, then the add: the order of the row-at-a-time form)
                        vv[0] += (float)h[0] + (float)l[0]; vv[1] += (float)h[1] + (float)l[1];
                        vv[2] += (float)h[2] + (float)l[2]; vv[3] += (float)h[3] + (float)l[3];
                    } else {
                        vv[0] += __uint_as_float(rlo[ii][j].x); vv[1] += __uint_as_float(rlo[ii][j].y);
                        vv[2] += __uint_as_float(rhi[ii][j].x); vv[3] += __uint_as_float(rhi[ii][j].y);
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) vv[e] = sigm ? act_fn(vv[e], a.act, a.alpha) : ape::act_fast(vv[e], af);
                if (ABL(256)) continue;              // (timing only: no stores)
                if (a.out_fmt == APE_FMT_S32) {
                    const int cy = a.yoff + n;
                    char* yp = a.y + (size_t)m * a.ldy * 4 + (cy >> 5) * 128 + (cy & 31) * 2;
                    bf16x4 h, l;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { h[e] = (__bf16)vv[e]; l[e] = (__bf16)(vv[e] - (float)h[e]); }
#if APE_S32_WIDE_STORE
                    if (wide) {
                        // ONE 16-byte store per lane instead of two 8-byte ones: the lanes fc and fc ^ 1 of a pixel (16 lanes apart) hold adjacent
                        // channel quads; v_permlane16_swap exchanges the odd 16-lane rows of its first operand with the even rows of its
                        // second, so with (hi, lo) an even-fc lane ends up with the hi halves of BOTH quads (16 contiguous bytes at its own hi
                        // address) and the odd-fc lane with both lo halves (at the even lane's lo address = its own hi address - 8 + 64)
                        uint2 hu = __builtin_bit_cast(uint2, h), lu = __builtin_bit_cast(uint2, l);
                        const auto r0 = __builtin_amdgcn_permlane16_swap(hu.x, lu.x, false, false);
                        const auto r1 = __builtin_amdgcn_permlane16_swap(hu.y, lu.y, false, false);
                        *reinterpret_cast<uint4*>((fc_e & 1) ? yp + 56 : yp) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
                    } else
#endif
                    {
                        *reinterpret_cast<bf16x4*>(yp) = h;
                        *reinterpret_cast<bf16x4*>(yp + 64) = l;
                    }
                } else {
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.y) + (size_t)m * a.ldy + a.yoff + n) = make_float4(vv[0], vv[1], vv[2], vv[3]);
                }
            }
        }
    }
}

// RES: the build with the residual operand.  Its epilogue needs 64 more registers (the residual tile requested four rows ahead), which the
// tile walk below cannot spare (hipcc spills 116 VGPRs in the loop form), so it stays one tile per workgroup: has_next is constant false
// and the tile loop folds away.
template <int BN, bool RES>
__device__ __forceinline__ void gemm_s32_body(const GemmS32Args& a)
{
#if defined(__HIP_DEVICE_COMPILE__)      // (see APE_DS_READ: the host pass only needs the kernel's stub)
    constexpr int TN = BN / 64;                  // 16-channel blocks per wave
    constexpr int TNH = (TN + 1) / 2;            // ... in the larger B half
    constexpr int TN0 = TN / 2;                  // B0 = blocks [0, TN0), B1 = [TN0, TN)
    constexpr int NB = BN / 64;                  // B blocks (8 rows) each wave stages per k-tile
    constexpr int B_STAGE = BN * 128;
    constexpr int NA = 3;                        // A ring slots (pixels: 3 x 32 KB); the weights get two -- 160 KB in all for BN = 256
    constexpr int B_BASE = NA * A_STAGE;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // PERSISTENT workgroups: gridDim.x = G <= tiles; this workgroup walks the tiles first, first + G, ... (`first` = the XCD-aware
    // position of the block among the G resident ones, so at any time the chip works on G consecutive tiles: the channel tiles of a
    // pixel tile, and neighbouring pixel tiles, side by side on one XCD).  The k-tile stream runs THROUGH the tile boundary: the DMA
    // requests of the next tile's first k-tiles go out from the last k-tiles of this one (same ring discipline), so they land while
    // the epilogue stores this tile, instead of a cold start behind a drained workgroup.  G = tiles reproduces one tile per workgroup.
    const int nwg = a.m_tiles * a.n_tiles;
    const int G = gridDim.x;
    const int orig = blockIdx.x;
    const int xcd = orig % 8, q = G / 8, r = G % 8;
    int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    // ---- LDS-DMA sources: buffer descriptors whose range ends at the last valid row (rows beyond read as zeros) -----------
    // pixel tile mt -> first row, end of the rows it may touch (the image's end with per-image tiling: rows past it read as zeros and are
    // not stored), image index (selects the weights)
    auto tile_rows = [&](int mt, int& m0_, int& mend_, int& img_) {
        if (a.img_tiles > 0) {
            img_ = mt / a.img_tiles;
            m0_ = img_ * a.rows_per_image + (mt - img_ * a.img_tiles) * BM;
            mend_ = (img_ + 1) * a.rows_per_image;
        } else {
            img_ = 0;
            m0_ = mt * BM;
            mend_ = a.M;
        }
    };
    auto make_rs_a = [&](int m0_, int mend_) {
        const long a_bytes = ((long)(mend_ - m0_) * a.ldx - a.xoff) * 4;
        return __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + ((long)m0_ * a.ldx + a.xoff) * 4), 0,
                                                 (int)(a_bytes > 0xFFFFFFFFL ? 0xFFFFFFFFu : (unsigned)a_bytes), 0x00020000);
    };
    auto make_rs_b = [&](int n0_, int img_) {
        const long b_bytes = (long)(a.Cout - n0_) * a.K * 4;
        return __builtin_amdgcn_make_buffer_rsrc((void*)(a.w + (long)img_ * a.w_img_stride + (long)n0_ * a.K * 4), 0,
                                                 (int)(b_bytes > 0xFFFFFFFFL ? 0xFFFFFFFFu : (unsigned)b_bytes), 0x00020000);
    };
    int m0, m_end, img, n0 = (logical % a.n_tiles) * BN;
    tile_rows(logical / a.n_tiles, m0, m_end, img);
    __amdgpu_buffer_rsrc_t rs_a = make_rs_a(m0, m_end), rs_b = make_rs_b(n0, img);
    bool has_next = false;                                    // (the NEXT tile's descriptors are built where a k-tile index wraps into it)
    // one DMA = 8 rows x 128 B: lanes 8 r .. 8 r + 7 fetch ONE row (a whole 128-B line), its 16-B chunks permuted by the row's XOR
    // swizzle ((row >> 1) & 7: the rows of a block are 8-aligned, so this is ((lane >> 4) & 3) | parity-free bits of the block)
    // (every per-lane address below is rebuilt at the start of each tile from an opaque copy of the lane id: kept loop-invariant they would
    // stay live across the epilogue, which needs the registers)
    unsigned va[4], vb[NB];
    auto setup_dma_lane = [&](int ln) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (wave * 4 + i) * 8 + (ln >> 3);
            va[i] = (unsigned)(row * a.ldx * 4 + (((ln & 7) ^ ((row >> 1) & 7)) * 16));
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int row = (wave * NB + i) * 8 + (ln >> 3);
            vb[i] = (unsigned)(row * a.K * 4 + (((ln & 7) ^ ((row >> 1) & 7)) * 16));
        }
    };
    setup_dma_lane(lane);
    // experiment (dbg bit 16): every workgroup starts its k-loop at a different k-tile (sum order changes, results stay valid)
    const int krot = ((a.dbg & 16) && G == nwg) ? ((logical / a.n_tiles) * 5 + (logical % a.n_tiles) * 3) % a.nk : 0;
    auto koff = [&](int kt) { const int k = kt + krot; return (k >= a.nk ? k - a.nk : k) * 128; };
    auto dma_a = [&](int kt, int slot_bytes) {        // 4 pieces per wave into A ring slot `slot_bytes` / A_STAGE
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_void*)(smem + slot_bytes + (wave * 4 + i) * 1024), 16, va[i], koff(kt), 0, 0);
    };
    // (the in-loop pieces: `rs` = this tile's or the next tile's descriptor, chosen once per k-tile; k-tile index already reduced)
    auto dma_a_piece = [&](__amdgpu_buffer_rsrc_t rs, int kt, int slot_bytes, int i) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + slot_bytes + (wave * 4 + i) * 1024), 16, va[i], koff(kt), 0, 0);
    };
    auto dma_b_piece = [&](__amdgpu_buffer_rsrc_t rs, int kt, int stage, int i) {
        if (i < NB)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + B_BASE + stage * B_STAGE + (wave * NB + i) * 1024), 16, vb[i], koff(kt), 0, 0);
    };
    auto dma_b = [&](int kt, int stage) {             // NB pieces per wave
#pragma unroll
        for (int i = 0; i < NB; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_void*)(smem + B_BASE + stage * B_STAGE + (wave * NB + i) * 1024), 16, vb[i], koff(kt), 0, 0);
    };

    // ---- fragment addresses: lane (frow, fc) reads chunk fc (hi) / 4 + fc (lo) of row frow of a 16-row block ----------------
    int frow = lane & 15, fc = lane >> 4;
    // LDS image: rows of 128 B, 16-B chunk c of row r at slot c ^ ((r >> 1) & 7).  The hi chunk fc and the lo chunk 4 + fc of a lane
    // are 64 B apart, but on which side depends on the row: one base register per plane; the 16-row block (i | j) is at + 2048,
    // an immediate.  B: one pair per stage (compile-time stage); A: the ring slot of the current tile ([0]) and of the next one
    // ([1]) rotate, two v_add per k-tile.
    unsigned a_lane[2], a_addr[2][2], b_addr[2][2];
    auto setup_frag_lane = [&](int ln, int slot_cur) {       // slot_cur: byte offset of the A ring slot of the tile's first k-tile
        frow = ln & 15;
        fc = ln >> 4;
        const int swz = (frow >> 1) & 7;
        const int lane_hi = frow * 128 + ((fc ^ swz) * 16), lane_lo = frow * 128 + (((4 + fc) ^ swz) * 16);
        a_lane[0] = (unsigned)(wm * 16 * 1024 + lane_hi);
        a_lane[1] = (unsigned)(wm * 16 * 1024 + lane_lo);
        const int slot_nxt = slot_cur + A_STAGE == NA * A_STAGE ? 0 : slot_cur + A_STAGE;
        a_addr[0][0] = a_lane[0] + (unsigned)slot_cur; a_addr[0][1] = a_lane[1] + (unsigned)slot_cur;
        a_addr[1][0] = a_lane[0] + (unsigned)slot_nxt; a_addr[1][1] = a_lane[1] + (unsigned)slot_nxt;
        b_addr[0][0] = (unsigned)(B_BASE + wn * (TN * 2) * 1024 + lane_hi); b_addr[0][1] = (unsigned)(B_BASE + wn * (TN * 2) * 1024 + lane_lo);
        b_addr[1][0] = b_addr[0][0] + B_STAGE; b_addr[1][1] = b_addr[0][1] + B_STAGE;
    };
    setup_frag_lane(lane, 0);
    // The fragment reads are inline asm so that they stay where the schedule puts them -- at the HEAD of the phase before the one
    // that consumes them (left to hipcc they sink to the last use of the registers they reuse, i.e. to the end of the phase, in
    // front of the wait) -- which also makes their completion invisible to the compiler: every phase ends with lgkmcnt(0) and a
    // sched_barrier (cdna_hip_programming.md 5.4 rule 18).
    u32x4 A[2][4][2], B[2][TNH][2];
    auto read_a = [&](auto which_c, auto half_c, auto set_c) {        // which: 0 = the current k-tile's ring slot, 1 = the next one's
        constexpr int which = decltype(which_c)::value, half = decltype(half_c)::value, set = decltype(set_c)::value;
        if (ABL(8)) return;
        ds_read_frags4<half * 4 * 2048>(A[set], a_addr[which][0], a_addr[which][1]);
    };
    auto read_b = [&](auto stage_c, auto half_c, auto set_c) {
        constexpr int stage = decltype(stage_c)::value, half = decltype(half_c)::value, set = decltype(set_c)::value;
        constexpr int j0 = half ? TN0 : 0, nj = half ? TN - TN0 : TN0;
        if (ABL(8)) return;
        ds_read_frags<j0 * 2048, nj, TNH>(B[set], b_addr[stage][0], b_addr[stage][1]);
    };
    f32x4 acc[8][TN];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
    // `hook(i)` runs after the MFMAs of pixel block i of the phase (pinned there): the DMA pieces of the next tiles are spread over
    // the MFMA stream this way instead of being issued in one burst behind the barrier, where both waves of every SIMD would stall
    // on their ~8 x 100-cycle issue at the same time
    auto mfma = [&](auto ahalf_c, auto aset_c, auto bhalf_c, auto bset_c, auto&& hook) {
        constexpr int ahalf = decltype(ahalf_c)::value, aset = decltype(aset_c)::value, bhalf = decltype(bhalf_c)::value, bset = decltype(bset_c)::value;
        constexpr int j0 = bhalf ? TN0 : 0, nj = bhalf ? TN - TN0 : TN0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (!ABL(4)) {
#pragma unroll
                for (int j = 0; j < nj; ++j) {
                    f32x4& c = acc[ahalf * 4 + i][j0 + j];
                    const bf16x8 bh = __builtin_bit_cast(bf16x8, B[bset][j][0]), bl = __builtin_bit_cast(bf16x8, B[bset][j][1]);
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, A[aset][i][0]), al = __builtin_bit_cast(bf16x8, A[aset][i][1]);
                    // weights as the row operand: D[channel 4 fc + e][pixel frow]; same products and k order as conv_gemm.hip
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, al, c, 0, 0, 0);
                    if (!ABL(512)) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, ah, c, 0, 0, 0);     // (512: two matrix instructions per product)
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah, c, 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            hook(i);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // The destinations of the asm reads must stay ALLOCATED until the wait that follows them: hipcc regards an asm's outputs as written at
    // the statement, so where it finds them dead (the look-ahead reads of a workgroup's last k-tile feed no MFMA) it gives them all one
    // register quad and re-uses that quad for its own values while the LDS data is still on its way (tools/isa_audit.py found a
    // v_cndmask + v_cmp pair on such a register in the odd-k-tile tail: the returning data could have changed a branch).  An empty asm that
    // reads the registers behind the phase's wait keeps them live across it (cdna_hip_programming.md 5.7 item 1, form (iii)).
    auto keep_a = [&](auto set_c) {
        constexpr int set = decltype(set_c)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) keep_regs(A[set][i][0], A[set][i][1]);
    };
    auto keep_b = [&](auto set_c) {
        constexpr int set = decltype(set_c)::value;
#pragma unroll
        for (int j = 0; j < TNH; ++j) keep_regs(B[set][j][0], B[set][j][1]);
    };
    auto no_hook = [](int) {};
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    auto phase_end = [&]() {        // (the sched_barrier in FRONT keeps the phase's MFMAs above the wait: they are not ordered against an asm)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    // DMA schedule.  Rings: A (pixels) three slots, B (weights) two.  At the middle of k-tile t (its barrier: every wave has read
    // all of tile t's fragments) the slots of A(t) and B(t) are free: B(t+2) and A(t+3) are issued there, in that order.  The same
    // barrier needs tile t+1 landed: B(t+1) was issued one k-tile ago, A(t+1) two k-tiles ago, and the only younger pieces are the
    // four of A(t+2) => vmcnt(4), never 0 in the steady state: up to 96 KB per CU are in flight across the barrier.
    const int nk = a.nk;
    // Waves 4-7 share their SIMDs with waves 0-3 and would otherwise run every phase in step with them; one of each pair at priority 1
    // for the whole loop lets that wave's MFMA rows go out ahead while its partner's reads and waits fill in behind (cdna_hip_programming.md
    // T5, static form): -1.5 .. -3 % on the three segmentation GEMM shapes, any asymmetric choice (prio 1 / 3, either half) alike.
    // dbg bit 32 switches it off.
    if (!(a.dbg & 32) && __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(1);
    dma_a(0, 0);
    dma_b(0, 0);
    if (nk > 1) { dma_a(1, A_STAGE); dma_b(1, 1); }
    if (nk > 2) dma_a(2, 2 * A_STAGE);
    if (nk > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 + NB) : "memory");
    else if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + NB) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int slot_free = 0;             // byte offset of the A ring slot that holds the current tile (free after its barrier)
#pragma unroll 1
    for (;;) {
    // A tile after the first: its k-tiles 0, 1 (and A of 2) were requested from the previous tile's last k-tiles and the ring
    // positions simply continue (nk is even on this path, so k-tile 0 is B stage 0 again).
    has_next = !RES && logical + G < nwg;
    {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        setup_dma_lane(ln);
        setup_frag_lane(ln, slot_free);
        // every fragment register is (re)defined below before its first use: tell the allocator so (no instruction)
#pragma unroll
        for (int st = 0; st < 2; ++st) {
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" : "=v"(A[st][i][0]), "=v"(A[st][i][1]));
#pragma unroll
            for (int j = 0; j < TNH; ++j) asm volatile("" : "=v"(B[st][j][0]), "=v"(B[st][j][1]));
        }
    }
    __builtin_amdgcn_s_barrier();
    read_a(I0{}, I0{}, I0{});
    read_b(I0{}, I0{}, I0{});
    phase_end();
    keep_a(I0{});
    keep_b(I0{});

    // one k-tile; X = kt & 1 = its B stage = the B register set that holds its B0
    auto ktile = [&](int kt, auto xc) {
        constexpr int X = decltype(xc)::value;
        using S = std::integral_constant<int, X>;
        using T = std::integral_constant<int, X ^ 1>;
        // phase 0: (A0, B0) while B1 of this tile is read into the other B set
        read_b(S{}, I1{}, T{});
        __builtin_amdgcn_sched_barrier(0);
        mfma(I0{}, I0{}, I0{}, S{}, no_hook);
        phase_end();
        keep_b(T{});
        // phase 1: (A0, B1) while A1 of this tile is read
        read_a(I0{}, I1{}, I1{});
        __builtin_amdgcn_sched_barrier(0);
        mfma(I0{}, I0{}, I1{}, T{}, no_hook);
        phase_end();
        keep_a(I1{});
        // every read of tile kt is back; tile kt+1 must have landed (only A(kt+2)'s four pieces may still be in flight)
        if (kt + 2 < nk || has_next) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!ABL(2)) __builtin_amdgcn_s_barrier();
        const bool more_b = (kt + 2 < nk || has_next) && !ABL(1), more_a = (kt + 3 < nk || has_next) && !ABL(1);
        // k-tiles past the end of this tile are the first ones of the next tile
        const bool wrap_b = kt + 2 >= nk, wrap_a = kt + 3 >= nk;
        const int nl = logical + G;
        int nm0, nmend, nimg;
        tile_rows(nl / a.n_tiles, nm0, nmend, nimg);
        const __amdgpu_buffer_rsrc_t rs_b_use = wrap_b ? make_rs_b((nl % a.n_tiles) * BN, nimg) : rs_b, rs_a_use = wrap_a ? make_rs_a(nm0, nmend) : rs_a;
        const int kb = wrap_b ? kt + 2 - nk : kt + 2, ka = wrap_a ? kt + 3 - nk : kt + 3;
        const int slot = slot_free;
        slot_free = slot_free + A_STAGE == NA * A_STAGE ? 0 : slot_free + A_STAGE;
        __builtin_amdgcn_sched_barrier(0);
        // phase 2: (A1, B1) while A0 of the next tile is read; B(kt+2) goes out piece by piece between the MFMA rows
        read_a(I1{}, I0{}, I0{});
        __builtin_amdgcn_sched_barrier(0);
        mfma(I1{}, I1{}, I1{}, T{}, [&](int i) { if (more_b) dma_b_piece(rs_b_use, kb, X, i); });
        phase_end();
        keep_a(I0{});
        // phase 3: (A1, B0) while B0 of the next tile is read into the set B1 just left; A(kt+3) goes out likewise
        read_b(T{}, I0{}, T{});
        __builtin_amdgcn_sched_barrier(0);
        mfma(I1{}, I1{}, I0{}, S{}, [&](int i) { if (more_a) dma_a_piece(rs_a_use, ka, slot, i); });
        // rotate the A slot addresses: next becomes current, the one after it is slot_free + A_STAGE (mod the ring)
        {
            const int nxt = slot_free + A_STAGE == NA * A_STAGE ? 0 : slot_free + A_STAGE;
            a_addr[0][0] = a_addr[1][0]; a_addr[0][1] = a_addr[1][1];
            a_addr[1][0] = a_lane[0] + (unsigned)nxt; a_addr[1][1] = a_lane[1] + (unsigned)nxt;
        }
        phase_end();
        keep_b(T{});
    };
    int kt = 0;
#pragma unroll 1
    for (; kt + 1 < nk; kt += 2) {
        ktile(kt, I0{});
        ktile(kt + 1, I1{});
    }
    if (kt < nk) ktile(kt, I0{});

    gemm_s32_epilogue<BN, RES>(a, acc, m0, m_end, n0, wm, wn, lane);
    if (RES || !has_next) break;
    // ---- on to the next tile: its first k-tiles are in flight or landed ----
    logical += G;
    tile_rows(logical / a.n_tiles, m0, m_end, img);
    n0 = (logical % a.n_tiles) * BN;
    rs_a = make_rs_a(m0, m_end);
    rs_b = make_rs_b(n0, img);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
    // (vmcnt(0) also drains this tile's stores -- issued up to an epilogue ago, mostly acknowledged by now)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#endif
}


// ---- PP ("ping-pong", round 6; cf. conv3x3_halo_s32.hip): the two waves of a SIMD (wave w and w + 4) run a k-tile's matrix segment C (all of
// its MFMAs from ONE fragment set) and its load segment L (the fragment reads of the next k-tile, the DMA pieces) in OPPOSITE order, with one
// barrier per k-tile and wave: waves 0-3 close their interval behind L (C(t) L(t+1) |), waves 4-7 behind C (L(t) C(t) |).  PMC of the lockstep
// form: matrix pipe 0.58 busy.
// LDS protocol (interval X = what lies between the barriers X-1 and X; stream k-tile X is multiplied in interval X): k-tile X's fragments are
// read by waves 0-3 at the END of interval X-1 and by waves 4-7 at the START of interval X.  ALL pieces are issued by waves 4-7, at the start
// of an interval (behind a barrier), and retired by their vmcnt(0) at its end (in front of the next barrier, which publishes them):
//   * the pixel rows of a tile split by reader: rows 0..127 (H_A) are read by waves 0-3 only, rows 128..255 (H_B) by waves 4-7 only.  At the
//     start of interval X:  H_A(X+2) -> ring slot X & 1 (k-tile X's H_A was read before barrier X-1; first read at the end of interval
//     X+1, behind barrier X);  H_B(X+1) -> slot (X+1) & 1 (k-tile X-1's H_B was read at the start of interval X-1; first read at the start of
//     interval X+1).  TWO slots of 32 KB do for the pixels, where the lockstep form needs three;
//   * the weights are read by every wave: W(X+2) -> slot (X+2) % 3 = the slot of k-tile X-1 (last read at the start of interval X-1); first
//     read at the end of interval X+1.  Three slots.   2 x 32 KB + 3 x BN x 128 B = 160 KB at BN = 256.
// Same products, same order per accumulator as the lockstep form: bit-identical outputs.
template <int BN, bool RES>
__device__ __forceinline__ void gemm_s32_pp_body(const GemmS32Args& a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int TN = BN / 64;                  // 16-channel blocks per wave
    constexpr int B_STAGE = BN * 128;
    constexpr int B_BASE = 2 * A_STAGE;
    constexpr int NBP = BN / 32;                 // weight pieces (8 rows) per issuing wave and k-tile
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int nwg = a.m_tiles * a.n_tiles;
    const int G = gridDim.x;
    const int orig = blockIdx.x;
    const int xcd = orig % 8, q = G / 8, r = G % 8;
    int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const bool group_b = wave >= 4;
    const int gi = wave & 3;                     // issuing index among waves 4-7
    const int nk = a.nk;

    auto tile_rows = [&](int mt, int& m0_, int& mend_, int& img_) {
        if (a.img_tiles > 0) {
            img_ = mt / a.img_tiles;
            m0_ = img_ * a.rows_per_image + (mt - img_ * a.img_tiles) * BM;
            mend_ = (img_ + 1) * a.rows_per_image;
        } else {
            img_ = 0;
            m0_ = mt * BM;
            mend_ = a.M;
        }
    };
    auto make_rs_a = [&](int m0_, int mend_) {
        const long a_bytes = ((long)(mend_ - m0_) * a.ldx - a.xoff) * 4;
        return __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + ((long)m0_ * a.ldx + a.xoff) * 4), 0,
                                                 (int)(a_bytes > 0xFFFFFFFFL ? 0xFFFFFFFFu : (unsigned)a_bytes), 0x00020000);
    };
    auto make_rs_b = [&](int n0_, int img_) {
        const long b_bytes = (long)(a.Cout - n0_) * a.K * 4;
        return __builtin_amdgcn_make_buffer_rsrc((void*)(a.w + (long)img_ * a.w_img_stride + (long)n0_ * a.K * 4), 0,
                                                 (int)(b_bytes > 0xFFFFFFFFL ? 0xFFFFFFFFu : (unsigned)b_bytes), 0x00020000);
    };
    int m0, m_end, img, n0 = (logical % a.n_tiles) * BN;
    tile_rows(logical / a.n_tiles, m0, m_end, img);
    __amdgpu_buffer_rsrc_t rs_a = make_rs_a(m0, m_end), rs_b = make_rs_b(n0, img);
    __amdgpu_buffer_rsrc_t rs_a_n = rs_a, rs_b_n = rs_b;      // the NEXT tile's descriptors (valid while has_next)
    bool has_next = false;
    auto next_tile_descriptors = [&]() {
        has_next = !RES && logical + G < nwg;
        if (has_next) {
            const int nl = logical + G;
            int nm0, nmend, nimg;
            tile_rows(nl / a.n_tiles, nm0, nmend, nimg);
            rs_a_n = make_rs_a(nm0, nmend);
            rs_b_n = make_rs_b((nl % a.n_tiles) * BN, nimg);
        }
    };
    next_tile_descriptors();

    // ---- DMA (waves 4-7): a piece = 8 rows x 128 B; lanes 8 r .. 8 r + 7 fetch ONE row, its 16-B chunks permuted by the row's XOR swizzle
    // ((row >> 1) & 7).  Wave gi owns the pieces gi * 4 + i (i = 0..3) of a pixel half and gi * NBP + i of the weights; a piece's 8-row offset
    // goes through the instruction's SCALAR offset (it is part of the range check on gfx950: tools/attic/probes/soffset_range_probe.hip), so two
    // lane offsets per operand (the swizzle depends on the piece's parity) serve all pieces.
    unsigned va_par[2], vb_par[2];
    auto setup_dma_lane = [&](int ln) {
        const int r8 = ln >> 3, c8 = ln & 7;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int sw = (4 * p + (r8 >> 1)) & 7;
            va_par[p] = (unsigned)((gi * 32 + r8) * a.ldx * 4 + ((c8 ^ sw) * 16));
            vb_par[p] = (unsigned)((gi * NBP * 8 + r8) * a.K * 4 + ((c8 ^ sw) * 16));
        }
    };
    setup_dma_lane(lane);
    const unsigned a_row8 = (unsigned)(8 * a.ldx * 4), b_row8 = (unsigned)(8 * a.K * 4);
    // stream k-tile `idx` counted from this tile's k-tile 0 (idx >= nk: the next tile's idx - nk)
    auto dma_half = [&](int idx, int slot, int half) {      // pixel rows half * 128 + [0, 128) into A ring slot `slot`
        const bool nxt = idx >= nk;
        if (nxt && !has_next) return;
        const __amdgpu_buffer_rsrc_t rs = nxt ? rs_a_n : rs_a;
        const unsigned so = (unsigned)((nxt ? idx - nk : idx) * 128) + (unsigned)half * 16u * a_row8;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + slot * A_STAGE + (half * 16 + gi * 4 + i) * 1024), 16, va_par[i & 1],
                                                     so + (unsigned)i * a_row8, 0, 0);
    };
    auto dma_w = [&](int idx, int slot) {
        const bool nxt = idx >= nk;
        if (nxt && !has_next) return;
        const __amdgpu_buffer_rsrc_t rs = nxt ? rs_b_n : rs_b;
        const unsigned so = (unsigned)((nxt ? idx - nk : idx) * 128);
#pragma unroll
        for (int i = 0; i < NBP; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + B_BASE + slot * B_STAGE + (gi * NBP + i) * 1024), 16, vb_par[i & 1],
                                                     so + (unsigned)i * b_row8, 0, 0);
    };

    // ---- fragment addresses (cf. the lockstep form) ----------------------------------------------------------------------------------------
    unsigned a_lane[2], b_lane[2];
    auto setup_frag_lane = [&](int ln) {
        const int frow = ln & 15, fc = ln >> 4;
        const int swz = (frow >> 1) & 7;
        const int lane_hi = frow * 128 + ((fc ^ swz) * 16), lane_lo = frow * 128 + (((4 + fc) ^ swz) * 16);
        a_lane[0] = (unsigned)(wm * 16 * 1024 + lane_hi);
        a_lane[1] = (unsigned)(wm * 16 * 1024 + lane_lo);
        b_lane[0] = (unsigned)(B_BASE + wn * (TN * 2) * 1024 + lane_hi);
        b_lane[1] = (unsigned)(B_BASE + wn * (TN * 2) * 1024 + lane_lo);
    };
    setup_frag_lane(lane);
    u32x4 Af[8][2], Bfr[TN][2];
    auto read_all = [&](int sa, int sb) {
        const unsigned ao = (unsigned)(sa * A_STAGE), bo = (unsigned)(sb * B_STAGE);
        ds_read_blocks<TN>(Bfr, b_lane[0] + bo, b_lane[1] + bo);
        ds_read_blocks<8>(Af, a_lane[0] + ao, a_lane[1] + ao);
    };
    auto keep_all = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i) keep_regs(Af[i][0], Af[i][1]);
#pragma unroll
        for (int j = 0; j < TN; ++j) keep_regs(Bfr[j][0], Bfr[j][1]);
    };
    f32x4 acc[8][TN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
    };
    zero_acc();
    auto mfma_all = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                f32x4& c = acc[i][j];
                const bf16x8 bh = __builtin_bit_cast(bf16x8, Bfr[j][0]), bl = __builtin_bit_cast(bf16x8, Bfr[j][1]);
                const bf16x8 ah = __builtin_bit_cast(bf16x8, Af[i][0]), al = __builtin_bit_cast(bf16x8, Af[i][1]);
                // weights as the row operand: D[channel 4 fc + e][pixel frow]; same products and k order as conv_gemm.hip
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, al, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, ah, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah, c, 0, 0, 0);
            }
    };
    auto phase_end = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    // the duties at the start of interval X (waves 4-7): `kt` = X counted from the current tile's k-tile 0; sa / sb = the ring slots of k-tile X
    auto duties = [&](int x, int sa, int sb) {
        dma_half(x + 1, sa ^ 1, 1);                      // H_B(X+1)
        dma_half(x + 2, sa, 0);                          // H_A(X+2)
        dma_w(x + 2, sb == 0 ? 2 : sb - 1);              // W(X+2) -> slot (X+2) % 3 = (X-1) % 3
    };

    // ---- prologue: k-tile 0 whole, H_A and the weights of k-tile 1; then every wave reads k-tile 0's fragments ---------------------------------
    if (group_b) {
        dma_half(0, 0, 0);
        dma_half(0, 0, 1);
        dma_w(0, 0);
        dma_half(1, 1, 0);
        dma_w(1, 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int sa = 0, sb = 0;                                  // ring slots of the k-tile whose fragments are in the registers
    read_all(0, 0);
    phase_end();
    keep_all();
    __builtin_amdgcn_s_barrier();                        // (waves 0-3 have read k-tile 0's H_A: its slot may take H_A(2))
    if (group_b) duties(0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);

    // One flat loop over the stream of k-tiles.  A tile's epilogue sits BETWEEN its last matrix segment and whatever follows it (waves 4-7: their
    // barrier; waves 0-3: the reads of the next tile's first k-tile): the fragment registers are dead there, so the accumulators' 128 and
    // the epilogue's own do not meet the 96 of a fragment set (with the epilogue behind L: 28 spilled registers at BN = 256).
    int kt = 0;
    bool done = false;
#pragma unroll 1
    for (;;) {
        // C(kt)
        mfma_all();
        __builtin_amdgcn_sched_barrier(0);
        if (kt == nk - 1) {
            // (RES, one tile per workgroup: its epilogue -- 64 more registers for the residual tile -- runs behind the loop, where nothing
            // of the loop is live any more)
            if constexpr (!RES) gemm_s32_epilogue<BN, RES>(a, acc, m0, m_end, n0, wm, wn, lane);
            if (RES || !has_next) {
                done = true;
            } else {        // on to the next tile: the stream simply continues (its first k-tiles are in flight or landed)
                logical += G;
                tile_rows(logical / a.n_tiles, m0, m_end, img);
                n0 = (logical % a.n_tiles) * BN;
                rs_a = rs_a_n;
                rs_b = rs_b_n;
                next_tile_descriptors();
                zero_acc();
                kt = -1;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (group_b) {          // waves 4-7 close their interval here: what they issued at its start has landed, the barrier publishes it
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_sched_barrier(0);
        if (done) break;
        // L(kt + 1): the next k-tile of the stream
        ++kt;
        sa ^= 1;
        sb = sb == 2 ? 0 : sb + 1;
        read_all(sa, sb);
        __builtin_amdgcn_sched_barrier(0);
        if (group_b) duties(kt, sa, sb);
        phase_end();
        keep_all();
        if (!group_b) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    if (!group_b) {             // (waves 0-3 close the interval in which waves 4-7 multiplied their last k-tile)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if constexpr (RES) gemm_s32_epilogue<BN, RES>(a, acc, m0, m_end, n0, wm, wn, lane);
#endif
}

// PP: the ping-pong schedule (gemm_s32_pp_body, round 6) / the lockstep schedule of rounds 3-5, which stays the product: same-process A/B over
// the ten 1x1 shapes of the step (tools/mb_gemm_pp.py), bit-identical outputs, PP 8.6 % SLOWER (up_1 mix 3.91 vs 3.67 ms).  Unlike halo_s32
// (24 KB per tap and CU), this kernel moves 64 KB per k-tile and CU from L2, and the LDS budget forces ALL of it onto waves 4-7 at the start
// of their interval (a piece issued by waves 0-3 would need a fourth ring slot): 16 pieces per wave in one burst in front of their own MFMAs,
// where the lockstep form spreads 8 per wave between its MFMA rows.  ape_conv_gemm_s32_debug bit 8192 launches the PP kernels (A/B, bitwise test).
template <int BN, bool PP>
__global__ __launch_bounds__(512, 2) void gemm_s32_kernel(const GemmS32Args a)
{
    if constexpr (PP) gemm_s32_pp_body<BN, false>(a); else gemm_s32_body<BN, false>(a);
}
template <int BN, bool PP>
__global__ __launch_bounds__(512, 2) void gemm_s32_res_kernel(const GemmS32Args a)
{
    if constexpr (PP) gemm_s32_pp_body<BN, true>(a); else gemm_s32_body<BN, true>(a);
}

template <int BN, bool RES, bool PP>
int launch_s32_form(GemmS32Args& a, hipStream_t st)
{
    // lockstep: three pixel slots + two weight slots; ping-pong: two + three
    constexpr size_t lds = PP ? 2 * (size_t)A_STAGE + 3 * (size_t)BN * 128 : 3 * (size_t)A_STAGE + 2 * (size_t)BN * 128;
    auto kern = RES ? gemm_s32_res_kernel<BN, PP> : gemm_s32_kernel<BN, PP>;
    static ape::DeviceOnce once;       // (one per instantiation of this function, i.e. per kernel)
    int ncu_dev = 0;
    if (int rc = ape::device_once(once, reinterpret_cast<const void*>(kern), (int)lds, &ncu_dev)) return rc;
    a.m_tiles = a.img_tiles > 0 ? (a.M / a.rows_per_image) * a.img_tiles : ape::ceil_div(a.M, BM);
    a.n_tiles = ape::ceil_div(a.Cout, BN);
    // persistent walk (one workgroup per CU) where the k-tile stream can run through the tile boundary: at least four k-tiles (the look-ahead
    // of three stays inside one tile) and, in the lockstep form, an even number of them (its weight stage parity repeats); dbg bit 64: one
    // tile per workgroup
    const int ncu = ncu_dev > 8 ? ncu_dev / 8 * 8 : 8;
    const int tiles = a.m_tiles * a.n_tiles;
    const bool walk = !RES && !(a.dbg & 64) && a.nk >= 4 && (PP || a.nk % 2 == 0) && tiles > ncu;
    hipLaunchKernelGGL(kern, dim3(walk ? ncu : tiles), dim3(512), lds, st, a);
    return ape::check_launch("ape_conv_gemm_s32");
}

template <int BN, bool RES>
int launch_s32_res(GemmS32Args& a, hipStream_t st)
{
    return (a.dbg & 8192) ? launch_s32_form<BN, RES, true>(a, st) : launch_s32_form<BN, RES, false>(a, st);
}

template <int BN>
int launch_s32(GemmS32Args& a, hipStream_t st)
{
    return a.res ? launch_s32_res<BN, true>(a, st) : launch_s32_res<BN, false>(a, st);
}

bool supported_s32(const ape_conv_params& p)
{
    if (p.KH != 1 || p.KW != 1 || p.stride != 1 || p.pad != 0 || p.ups != 0) return false;
    if (p.B < 0 || p.H < 1 || p.W < 1 || p.Ho != p.H || p.Wo != p.W) return false;
    if (p.Cin < 32 || p.Cin % 32 || p.ldx % 32 || p.xoff % 32 || p.xoff + p.Cin > p.ldx) return false;
    if (p.Cout < 128 || p.Cout % 4 || p.yoff + p.Cout > p.ldy || p.ldy % 4 || p.yoff % 4) return false;
    if (p.act < APE_ACT_NONE || p.act > APE_ACT_SIGMOID) return false;
    if ((long)p.B * p.H * p.W > (1L << 30) || 256L * p.ldx * 4 >= (1L << 31) || 256L * p.Cin * 4 >= (1L << 31)) return false;
    return true;
}

}  // namespace

/* w[cout][K] f32 (K % 32 == 0) -> S32K: [cout][K/32][hi 32 | lo 32] bf16 */
__global__ void pack_weights_s32k_kernel(const float* __restrict__ w, __bf16* __restrict__ out, long total, int K)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long n = i / K;
        const int k = (int)(i - n * K);
        const float v = w[i];
        const __bf16 h = (__bf16)v;
        __bf16* g = out + (n * (K / 32) + k / 32) * 64;
        g[k % 32] = h;
        g[32 + k % 32] = (__bf16)(v - (float)h);
    }
}

extern "C" int ape_pack_weights_s32k(const float* w, void* out, int cout, int K, void* stream)
{
    if (!w || !out || cout < 1 || K < 32 || K % 32) return APE_EINVAL;
    const long total = (long)cout * K;
    long g = (total + 255) / 256;
    g = g > 4096 ? 4096 : g;
    hipLaunchKernelGGL(pack_weights_s32k_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, w, (__bf16*)out, total, K);
    return ape::check_launch("ape_pack_weights_s32k");
}

/* fp32 [rows][C] <-> S32 [rows][C] (C % 32 == 0); one thread per 4 channels */
__global__ void f32_to_s32_kernel(const float4* __restrict__ x, char* __restrict__ y, long n4, int C)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const long row = i / (C / 4);
        const int c = (int)(i - row * (C / 4)) * 4;
        const float4 v = x[i];
        bf16x4 h, l;
        h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
        l[0] = (__bf16)(v.x - (float)h[0]); l[1] = (__bf16)(v.y - (float)h[1]); l[2] = (__bf16)(v.z - (float)h[2]); l[3] = (__bf16)(v.w - (float)h[3]);
        char* p = y + row * C * 4 + (c >> 5) * 128 + (c & 31) * 2;
        *reinterpret_cast<bf16x4*>(p) = h;
        *reinterpret_cast<bf16x4*>(p + 64) = l;
    }
}
__global__ void s32_to_f32_kernel(const char* __restrict__ x, float4* __restrict__ y, long n4, int C)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const long row = i / (C / 4);
        const int c = (int)(i - row * (C / 4)) * 4;
        const char* p = x + row * C * 4 + (c >> 5) * 128 + (c & 31) * 2;
        const bf16x4 h = *reinterpret_cast<const bf16x4*>(p), l = *reinterpret_cast<const bf16x4*>(p + 64);
        y[i] = make_float4((float)h[0] + (float)l[0], (float)h[1] + (float)l[1], (float)h[2] + (float)l[2], (float)h[3] + (float)l[3]);
    }
}
extern "C" int ape_convert_s32(const void* x, void* y, long rows, int C, int to_s32, void* stream)
{
    if (!x || !y || rows < 0 || C < 32 || C % 32) return APE_EINVAL;
    if (rows == 0) return APE_OK;
    const long n4 = rows * (C / 4);
    long g = (n4 + 255) / 256;
    g = g > 65536 ? 65536 : g;
    if (to_s32) hipLaunchKernelGGL(f32_to_s32_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, (const float4*)x, (char*)y, n4, C);
    else hipLaunchKernelGGL(s32_to_f32_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, (const char*)x, (float4*)y, n4, C);
    return ape::check_launch("ape_convert_s32");
}

static int g_s32_dbg = 0;
/* timing ablations of ape_conv_gemm_s32 (development aid; any non-zero value makes the results wrong) */
extern "C" int ape_conv_gemm_s32_debug(int bits) { g_s32_dbg = bits; return APE_OK; }

extern "C" int ape_conv_gemm_s32_supported(const ape_conv_params* params) { return params && supported_s32(*params) ? 1 : 0; }

static int conv_gemm_s32_run(const void* x_s32, const void* w_s32k, long w_image_stride_bytes, const float* bias, const void* residual,
                             int res_fmt, void* y, int out_fmt, const ape_conv_params* params, void* stream)
{
    if (!x_s32 || !w_s32k || !y || !params) return APE_EINVAL;
    const ape_conv_params& p = *params;
    if (!supported_s32(p)) return APE_EINVAL;
    if ((out_fmt != APE_FMT_F32 && out_fmt != APE_FMT_S32) || (residual && res_fmt != APE_FMT_F32 && res_fmt != APE_FMT_S32)) return APE_EINVAL;
    if (out_fmt == APE_FMT_S32 && (p.ldy % 32 || p.yoff % 4)) return APE_EINVAL;
    if (residual && (p.roff + p.Cout > p.ldr || p.ldr % 4 || p.roff % 4 || (res_fmt == APE_FMT_S32 && p.ldr % 32))) return APE_EINVAL;
    const long M = (long)p.B * p.H * p.W;
    if (M == 0) return APE_OK;
    GemmS32Args a;
    a.x = (const char*)x_s32; a.w = (const char*)w_s32k; a.bias = bias; a.res = (const char*)residual; a.y = (char*)y;
    a.M = (int)M; a.K = p.Cin; a.Cout = p.Cout;
    a.ldx = p.ldx; a.xoff = p.xoff; a.ldy = p.ldy; a.yoff = p.yoff; a.ldr = p.ldr; a.roff = p.roff;
    a.act = p.act; a.alpha = p.alpha; a.bias_bstride = p.bias_bstride; a.rows_per_image = p.H * p.W;
    a.out_fmt = out_fmt; a.res_fmt = res_fmt;
    a.nk = p.Cin / 32;
    a.dbg = g_s32_dbg;
    a.img_tiles = 0;
    a.w_img_stride = 0;
    if (w_image_stride_bytes) {
        a.img_tiles = ape::ceil_div(a.rows_per_image, BM);
        a.w_img_stride = w_image_stride_bytes;
    }
    hipStream_t st = (hipStream_t)stream;
    const int waste256 = ape::ceil_div(p.Cout, 256) * 256 - p.Cout;
    const int waste192 = ape::ceil_div(p.Cout, 192) * 192 - p.Cout;
    if (p.Cout <= 128) return launch_s32<128>(a, st);
    if (waste192 < waste256) return launch_s32<192>(a, st);
    return launch_s32<256>(a, st);
}

extern "C" int ape_conv_gemm_s32(const void* x_s32, const void* w_s32k, const float* bias, const void* residual, int res_fmt, void* y,
                                 int out_fmt, const ape_conv_params* params, void* stream)
{
    return conv_gemm_s32_run(x_s32, w_s32k, 0, bias, residual, res_fmt, y, out_fmt, params, stream);
}

extern "C" int ape_conv_gemm_s32_per_image(const void* x_s32, const void* w_s32k, long w_image_stride_bytes, const float* bias, void* y,
                                           int out_fmt, const ape_conv_params* params, void* stream)
{
    if (w_image_stride_bytes <= 0 || w_image_stride_bytes % 16) return APE_EINVAL;
    return conv_gemm_s32_run(x_s32, w_s32k, w_image_stride_bytes, bias, nullptr, APE_FMT_F32, y, out_fmt, params, stream);
}

// ---- the PSP bottleneck with its prior sum folded into the contraction (pspnet.py:12-24) ---------------------------------------------------
// bottleneck(cat(up(stage_s(pool_s(f))), f)) = W_f . f + sum_s up(Z_s) + b with Z_s = (W_b,s W_s) . pool_s(f) ([s x s x Cout] per frame,
// network.py _PSPPlan.prior) and `up` the bilinear (align_corners = False) resize to the map.  The resize is linear with coefficients that
// depend on the pixel alone: up(Z_s)[p] = sum_cells c_s[p][cell] Z_s[cell], at most four non-zero per scale.  So the prior sum is 50 more
// K-columns of the same contraction -- c[p][0..49] appended to the pixel's channels, Z_f[0..49][co] appended to frame f's weight rows --
// instead of a 1.26 GB fp32 tensor written by one kernel and read back by the GEMM's epilogue (ape_psp_prior_sum_f32 + the residual
// operand of ape_conv_gemm_s32: 1.68 ms per 64 frames; folded: 1.09 ms + these two small kernels).  Cells in the order 1x1 | 2x2 | 3x3 | 6x6,
// row-major, padded to 64 columns with zeros.
namespace {
constexpr int PSP_CELLS = 50, PSP_KPAD = 64;
__device__ __forceinline__ float psp_src_index(int dst, float scale)        // ATen area_pixel_compute_source_index, align_corners = False (= ops.hip src_index)
{
    const float s = scale * ((float)dst + 0.5f) - 0.5f;
    return s < 0.f ? 0.f : s;
}
// bilinear weight of prior cell `i` (0 .. S-1 along one axis) for output index `o` of `n`: ape_psp_prior_sum_f32's own expressions
__device__ __forceinline__ float psp_axis_weight(int o, int n, int S, int i)
{
    const float f = psp_src_index(o, (float)S / (float)n);
    const int i0 = (int)f;
    const int i1 = i0 + (i0 < S - 1 ? 1 : 0);
    const float l1 = f - (float)i0, l0 = 1.f - l1;
    return (i == i0 ? l0 : 0.f) + (i == i1 ? l1 : 0.f);
}
// channels [coff, coff + 64) of every pixel of x (S32, ld channels per pixel) <- the pixel's 50 coefficients (hi | lo); one thread per
// (pixel, 4 cells)
__global__ void psp_fill_coeffs_kernel(char* __restrict__ x, int B, int h, int w, int ld, int coff)
{
    // the coefficients depend on the pixel's position alone: formed once per (oy, ox, channel quad) and written to every frame
    const long total = (long)h * w * (PSP_KPAD / 4);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int q = (int)(i % (PSP_KPAD / 4));
        const long pix0 = i / (PSP_KPAD / 4);
        const int ox = (int)(pix0 % w), oy = (int)(pix0 / w);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = q * 4 + e;
            int S, cell;
            if (j < 1) { S = 1; cell = j; }
            else if (j < 5) { S = 2; cell = j - 1; }
            else if (j < 14) { S = 3; cell = j - 5; }
            else { S = 6; cell = j - 14; }
            v[e] = j < PSP_CELLS ? psp_axis_weight(oy, h, S, cell / S) * psp_axis_weight(ox, w, S, cell % S) : 0.f;
        }
        bf16x4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) { hi[e] = (__bf16)v[e]; lo[e] = (__bf16)(v[e] - (float)hi[e]); }
        const int c = coff + q * 4;
        char* p = x + pix0 * ld * 4 + (c >> 5) * 128 + (c & 31) * 2;
        const long frame = (long)h * w * ld * 4;
        for (int b = 0; b < B; ++b, p += frame) {
            *reinterpret_cast<bf16x4*>(p) = hi;
            *reinterpret_cast<bf16x4*>(p + 64) = lo;
        }
    }
}
// out[b][co][g] (128-B groups, G = K/32 + 2 per row): g < K/32: the shared weights' group; the last two: frame b's Z values of row co
__global__ void psp_pack_weights_kernel(const char* __restrict__ wf, const float* __restrict__ z1, const float* __restrict__ z2,
                                        const float* __restrict__ z3, const float* __restrict__ z6, char* __restrict__ out, int B, int Cout, int KG)
{
    const int G = KG + 2;
    const long total = (long)B * Cout * G * 8;          // 16-byte pieces
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int piece = (int)(i & 7);
        const long grp = i >> 3;
        const int g = (int)(grp % G);
        const long row = grp / G;
        const int co = (int)(row % Cout), b = (int)(row / Cout);
        uint4 val;
        if (g < KG) {
            val = *reinterpret_cast<const uint4*>(wf + ((long)co * KG + g) * 128 + piece * 16);
        } else {
            // piece 0..3: hi of cells 8 (piece) .. + 7 of this group; 4..7: their lo
            const int j0 = (g - KG) * 32 + (piece & 3) * 8;
            __bf16 h8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int j = j0 + e;
                float v = 0.f;
                if (j < 1) v = z1[((long)b * 1 + j) * Cout + co];
                else if (j < 5) v = z2[((long)b * 4 + (j - 1)) * Cout + co];
                else if (j < 14) v = z3[((long)b * 9 + (j - 5)) * Cout + co];
                else if (j < PSP_CELLS) v = z6[((long)b * 36 + (j - 14)) * Cout + co];
                const __bf16 hh = (__bf16)v;
                h8[e] = piece < 4 ? hh : (__bf16)(v - (float)hh);
            }
            val = *reinterpret_cast<const uint4*>(h8);
        }
        *reinterpret_cast<uint4*>(out + grp * 128 + piece * 16) = val;
    }
}
}  // namespace

extern "C" int ape_psp_fold_operands(const void* wf_s32k, const float* z1, const float* z2, const float* z3, const float* z6, void* w_out,
                                     void* x_s32, int B, int h, int w, int ld, int Cin, int Cout, void* stream)
{
    if (!wf_s32k || !z1 || !z2 || !z3 || !z6 || !w_out || !x_s32 || B < 0 || h < 1 || w < 1) return APE_EINVAL;
    if (Cin < 32 || Cin % 32 || ld < Cin + PSP_KPAD || ld % 32 || Cout < 1) return APE_EINVAL;
    if (B == 0) return APE_OK;
    hipStream_t st = (hipStream_t)stream;
    {
        const long total = (long)h * w * (PSP_KPAD / 4);
        long g = (total + 255) / 256;
        g = g > 65536 ? 65536 : g;
        hipLaunchKernelGGL(psp_fill_coeffs_kernel, dim3((int)g), dim3(256), 0, st, (char*)x_s32, B, h, w, ld, Cin);
    }
    {
        const long total = (long)B * Cout * (Cin / 32 + 2) * 8;
        long g = (total + 255) / 256;
        g = g > 65536 ? 65536 : g;
        hipLaunchKernelGGL(psp_pack_weights_kernel, dim3((int)g), dim3(256), 0, st, (const char*)wf_s32k, z1, z2, z3, z6, (char*)w_out, B, Cout, Cin / 32);
    }
    return ape::check_launch("ape_psp_fold_operands");
}
