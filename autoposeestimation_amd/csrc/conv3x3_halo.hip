// 3x3 stride-1 convolution (pad = dilation, dilation 1/2/4, Cin % 32 == 0) on the bf16 matrix cores with the input HALO tile
// staged ONCE per 32-channel chunk in LDS and re-used by all nine filter taps.
//
// Why: the generic implicit-GEMM kernel (conv_bf16.hip) re-fetches the A operand for every tap and re-fetches the weights
// for every 128-pixel tile; profiling showed it capped at ~9 TB/s of L2->CU operand traffic (30 KB per algorithmic MFLOP),
// i.e. bandwidth-bound far below the MFMA rate.  Here one workgroup owns a 16x16-pixel output tile (BM = 256) x 128 output
// channels:
//   * MFMA shape v_mfma_f32_16x16x32_bf16 (one 32-channel chunk per instruction): A/B'd 5-6 % faster than 32x32x16 here at
//     equal LDS traffic (the chip sustains a higher clock on this shape, MI355X_MICROARCH.md "DVFS give-back" item 7);
//   * A: the (16+2d)^2-pixel halo of the current 32-channel chunk is split to bf16 hi/lo while it is staged
//        HBM -> registers -> LDS, then every tap reads its MFMA fragments from the SAME LDS image at a shifted pixel row
//        (ds_read_b128 per fragment; 64-B rows with XOR-swizzled 16-B chunks => conflict-free reads and writes except 2-way at tile-row seams);
//   * B: one [128 cout][32 ci] weight tile per tap, double-buffered in LDS, prefetched through registers under the MFMAs;
//   * one s_barrier per tap; the next chunk's halo is fetched during taps 4..8 into the other A buffer (d = 1).
// Operand traffic drops to ~9.8 KB per MFLOP (3x less), which moves the kernel from the L2 roof to the MFMA roof.
// Numerics are identical to conv_bf16.hip (same split-bf16 x3 / plain bf16 products, fp32 accumulate), only the K order is
// (channel chunk, tap) instead of (tap, channel) -- covered by the same conv tests.
#include <type_traits>
#include "common.h"
#include "seg_head.h"

#ifdef APE_UPS_DEBUG
// Diagnostic build only (make dbgups -> libape_hip_dbgups.so, tools/dbg_ups.py): ups_lerp re-reads its table entry at the end of
// the tap -- the form that round 1 saw fail on grids larger than the chip -- and records every disagreement with the
// register-carried copy.  word 0 = mismatch count, then 16-word records.
__device__ unsigned ape_ups_dbg[1 + 16 * 64];
extern "C" int ape_ups_debug_read(unsigned* host_out, int reset)
{
    if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(ape_ups_dbg), sizeof(unsigned) * (1 + 16 * 64)) != hipSuccess) return -2;
    if (reset) {
        static unsigned zeros[1 + 16 * 64];
        if (hipMemcpyToSymbol(HIP_SYMBOL(ape_ups_dbg), zeros, sizeof(zeros)) != hipSuccess) return -2;
    }
    return 0;
}
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int TS = 16;          // output tile is TS x TS pixels
constexpr int CK = 32;          // channels per chunk
constexpr int LDH = CK;         // bf16 per LDS row of a WEIGHT tile: 64 B, unpadded, 16-B chunks XOR-swizzled (swz below)
// Halo image: ONE row per pixel holding the hi plane (64 B), the lo plane (64 B) and 32 B of padding: a 160-B pitch (96 B
// = hi + padding for plain bf16).  With these pitches the 16 lanes of every ds_read_b128 lane group (rows r..r+3 / r+12..r+15
// of one chunk, rows r+4..r+11 of the next) fall on 16 distinct 16-B bank slots (10 r + k, resp. 6 r + k, mod 16), exactly
// like the XOR swizzle used before, but the address is LINEAR in the pixel: a tap shift is a compile-time byte offset folded
// into the ds_read immediate instead of ~7 VALU operations per fragment read (the kernel spent 2 VALU instructions per MFMA).
constexpr int halo_lda(int npl) { return npl == 2 ? 2 * CK + 16 : CK + 16; }     // bf16 elements per halo pixel
// 16x16x32 fragments: lane l reads chunk l>>4 of row l&15; the ds_read_b128 lane groups then mix two chunks, and the chunk
// permutation that keeps all 16 lanes on distinct bank slots is chunk ^ ((-(row>>2)) & 3)
__device__ __forceinline__ int swz(int row, int chunk16) { return row * LDH + ((chunk16 ^ ((0 - (row >> 2)) & 3)) << 3); }
typedef float f32x4 __attribute__((ext_vector_type(4)));
// BN = 128: 8 waves = 4 (pixels) x 2 (channels), each 64 px x 64 cout, one workgroup per CU, halo double-buffered.
// BN = 64:  4 waves along the pixel axis, each 64 px x 64 cout, halo single-buffered so that TWO workgroups share a CU
//           (57 KB LDS each): one's prologue / chunk refill / epilogue is covered by the other's MFMAs.
constexpr int halo_threads(int bnh) { return bnh == 64 ? 256 : 512; }
constexpr bool halo_a_double(int npl, int hp, int bnh) { return bnh != 64 && (2 * hp * halo_lda(npl) * 2 + 2 * npl * bnh * LDH * 2) <= 160 * 1024; }

struct HaloArgs {
    const float* x;
    const __bf16* w;
    const float* bias;
    const float* res;
    float* y;
    ape_conv_params p;
    int Kp, tiles_x, tiles_y, n_tiles;
    long plane_stride;
    // fused segmentation head (HEAD kernels only): y is not written, label / score are
    const float* head_w;
    const float* head_b;
    uint8_t* label;
    float* score;
    int head_c, head_dsm;
};

// NONE / RELU / PRELU as selects on loop-invariant scalars (a `switch` per element compiled to a cascade of scalar compares and branches
// per element: ~1.8 k scalar instructions in a 256-element epilogue); same values bit for bit (1 * v == v, also for -0 and NaN)
__device__ __forceinline__ float activate_h(float v, int act, float alpha)
{
    if (act == APE_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
    const float neg = act == APE_ACT_RELU ? 0.f : (act == APE_ACT_PRELU ? alpha : 1.f) * v;
    return v > 0.f ? v : neg;
}

template <int NSPLIT, int D, int BNH, bool UPS, bool HEAD>
__global__ __launch_bounds__(halo_threads(BNH), 2) void conv3x3_halo_kernel(const HaloArgs a)
{
    constexpr int NTH = halo_threads(BNH);
    constexpr int NPL = NSPLIT == 3 ? 2 : 1;
    constexpr int WNW = BNH / 64;               // waves along the channel axis (2 / 1)
    constexpr int WMW = 4;                      // waves along the pixel axis
    constexpr int TMW = 256 / WMW / 32;         // 32-pixel MFMA tiles per wave (2)
    static_assert(WNW * WMW * 64 == NTH, "wave grid");
    constexpr int B_PASSES = (BNH * 4 + NTH - 1) / NTH;   // 16-B weight chunks per thread per plane (1)
    constexpr int HW_ = TS + 2 * D;            // halo width
    constexpr int HP = HW_ * HW_;              // halo pixels
    constexpr int A_ITEMS = (HP * (CK / 4) + NTH - 1) / NTH;   // float4 loads per thread per chunk
    constexpr bool A_DOUBLE = halo_a_double(NPL, HP, BNH);
    constexpr int NA = A_DOUBLE ? 2 : 1;

    extern __shared__ __attribute__((aligned(16))) __bf16 smem[];
    constexpr int LDA = halo_lda(NPL);
    __bf16* As = smem;                                         // [NA][HP][LDA]: hi at +0, lo at +CK of every pixel row
    __bf16* Bs = smem + (size_t)NA * HP * LDA;                 // [2][NPL][BNH*LDH]

    const ape_conv_params& p = a.p;
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const int nwg = p.B * tiles_per_img * a.n_tiles;
    const int orig = blockIdx.x;
    const int xcd = orig % 8, q = nwg / 8, r = nwg % 8;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
    const int n_tile = logical % a.n_tiles;
    const int mt = logical / a.n_tiles;
    const int b = mt / tiles_per_img, trem = mt - b * tiles_per_img;
    const int y0 = (trem / a.tiles_x) * TS, x0 = (trem % a.tiles_x) * TS;
    const int n0 = n_tile * BNH;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WNW, wn = wave % WNW;

    // ---- staging helpers -----------------------------------------------------------------------------------------
    float4 areg[A_ITEMS];
    uint4 breg[NPL];
    unsigned a_okmask = 0;      // bit j: item j of the pending halo lies inside the image (zero-filled at store time otherwise)
    // UPS: x is the half-resolution tensor; the halo pixel is its bilinear x2 (align_corners=True) sample, computed with the
    // operation order of ops.hip:bilinear_kernel so the fused and unfused paths agree bit for bit.
    const int hl = p.H / 2, wl = p.W / 2;
    const float ups_sh = (UPS && p.H > 1) ? (float)(hl - 1) / (float)(p.H - 1) : 0.f;
    const float ups_sw = (UPS && p.W > 1) ? (float)(wl - 1) / (float)(p.W - 1) : 0.f;
    // UPS halo items are fetched in two steps so that the four corner loads of an item can fly under a tap's MFMAs:
    // ups_fetch issues them (unconditional, clamped coordinates), ups_lerp blends them into areg[j] later.
    auto item_coords = [&](int j, int& c4, int& gy, int& gx) -> bool {
        int t = tid;
        asm volatile("" : "+v"(t));     // opaque: keeps the per-item address math next to its use instead of hoisted (and spilled)
        const int e = t + NTH * j;
        const int px = e >> 3;
        c4 = e & 7;
        const int hy = px / HW_, hx = px - hy * HW_;
        gy = y0 - D + hy; gx = x0 - D + hx;
        return e < HP * 8 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    };
    // UPS: the per-pixel part of the bilinear sample -- corner offset, +1 steps, blend weights, in-image bit -- is the same for the
    // eight channel items of a halo pixel and for every chunk: it is computed ONCE per tile into a 16-B LDS table entry per halo
    // pixel (the per-item form spent ~100 VALU operations per item, fetch and blend together, 2.2 k per wave and tile in a kernel
    // that is VALU-issue-bound).  Same expressions, so the blend weights and the result are bit-identical.
#if defined(APE_UPS_DEBUG) && APE_UPS_DEBUG == 2
    uint4* const ups_tbl = reinterpret_cast<uint4*>(Bs + (size_t)2 * NPL * BNH * LDH);      // round-1 placement: dynamic LDS, behind the weight tiles
#else
    __shared__ uint4 ups_tbl[UPS ? HP : 1];
#endif
    if (UPS) {
        for (int px = tid; px < HP; px += NTH) {
            const int hy = px / HW_, hx = px - hy * HW_;
            const int gy = y0 - D + hy, gx = x0 - D + hx;
            const bool ok = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
            const int cy = gy < 0 ? 0 : (gy >= p.H ? p.H - 1 : gy), cx = gx < 0 ? 0 : (gx >= p.W ? p.W - 1 : gx);
            const float fy = ups_sh * (float)cy, fx = ups_sw * (float)cx;
            const int iy0 = (int)fy, ix0 = (int)fx;
            const float ly1 = fy - (float)iy0, lx1 = fx - (float)ix0;
            uint4 ent;
            ent.x = (unsigned)(((b * hl + iy0) * wl + ix0) * p.ldx + p.xoff);
            ent.y = (ix0 < wl - 1 ? 1u : 0u) | (iy0 < hl - 1 ? 2u : 0u) | (ok ? 4u : 0u);
            ent.z = __float_as_uint(lx1);
            ent.w = __float_as_uint(ly1);
            ups_tbl[px] = ent;
        }
        __syncthreads();
    }
    // (the blend weights and the in-image bit travel from ups_fetch to ups_lerp in registers: a second read of the table entry at
    // the end of the tap returned wrong weights under load -- large grids only, never found why -- the first read does not)
    auto ups_fetch = [&](int j, int ci0, float4 (&r)[4], float (&wgt)[3]) {
        int t_op = tid;
        asm volatile("" : "+v"(t_op));      // opaque: the eleven items' table addresses are formed here, not hoisted in front of the chunk loop and spilled
        const int e = t_op + NTH * j;
        const int px = e < HP * 8 ? e >> 3 : HP - 1;
        const uint4 ent = ups_tbl[px];
        const unsigned base = ent.x + (unsigned)(ci0 + (e & 7) * 4);
        const unsigned dxo = (ent.y & 1u) ? (unsigned)p.ldx : 0u, dyo = (ent.y & 2u) ? (unsigned)(wl * p.ldx) : 0u;
        r[0] = *reinterpret_cast<const float4*>(a.x + base);
        r[1] = *reinterpret_cast<const float4*>(a.x + base + dxo);
        r[2] = *reinterpret_cast<const float4*>(a.x + base + dyo);
        r[3] = *reinterpret_cast<const float4*>(a.x + base + dyo + dxo);
        wgt[0] = __uint_as_float(ent.z);
        wgt[1] = __uint_as_float(ent.w);
        wgt[2] = (e < HP * 8 && (ent.y & 4u)) ? 1.f : 0.f;
    };
    auto ups_lerp = [&](int j, const float4 (&r)[4], const float (&wgt)[3]) {
#ifdef APE_UPS_DEBUG
        {
            const int e = tid + NTH * j;
            int px = e < HP * 8 ? e >> 3 : HP - 1;
            asm volatile("" : "+v"(px));                 // opaque: a real second ds_read, not the first one's registers
            const uint4 ent = ups_tbl[px];
            const float w2 = (e < HP * 8 && (ent.y & 4u)) ? 1.f : 0.f;
            if (__float_as_uint(wgt[0]) != ent.z || __float_as_uint(wgt[1]) != ent.w || w2 != wgt[2]) {
                const unsigned slot = atomicAdd(&ape_ups_dbg[0], 1u);
                if (slot < 64) {
                    unsigned* d = &ape_ups_dbg[1 + 16 * slot];
                    d[0] = blockIdx.x; d[1] = tid; d[2] = j; d[3] = px; d[4] = ent.x; d[5] = ent.y; d[6] = ent.z; d[7] = ent.w;
                    d[8] = __float_as_uint(wgt[0]); d[9] = __float_as_uint(wgt[1]); d[10] = __float_as_uint(wgt[2]);
                    d[11] = (unsigned)y0; d[12] = (unsigned)x0; d[13] = (unsigned)b;
                    d[14] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));    // HW_ID
                    d[15] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));   // XCC_ID
                }
            }
        }
#endif
        const bool ok = wgt[2] != 0.f;
        a_okmask = ok ? (a_okmask | (1u << j)) : (a_okmask & ~(1u << j));
        const float lx1 = wgt[0], ly1 = wgt[1], ly0 = 1.f - ly1, lx0 = 1.f - lx1;
        float4 o;
        o.x = ly0 * (lx0 * r[0].x + lx1 * r[1].x) + ly1 * (lx0 * r[2].x + lx1 * r[3].x);
        o.y = ly0 * (lx0 * r[0].y + lx1 * r[1].y) + ly1 * (lx0 * r[2].y + lx1 * r[3].y);
        o.z = ly0 * (lx0 * r[0].z + lx1 * r[1].z) + ly1 * (lx0 * r[2].z + lx1 * r[3].z);
        o.w = ly0 * (lx0 * r[0].w + lx1 * r[1].w) + ly1 * (lx0 * r[2].w + lx1 * r[3].w);
        areg[j] = o;
    };
    auto load_a_item = [&](int j, int ci0) {          // !UPS: one float4 per item
        int c4, gy, gx;
        const bool ok = item_coords(j, c4, gy, gx);
        const int cy = gy < 0 ? 0 : (gy >= p.H ? p.H - 1 : gy), cx = gx < 0 ? 0 : (gx >= p.W ? p.W - 1 : gx);
        const unsigned off = (unsigned)(((b * p.H + cy) * p.W + cx) * p.ldx + p.xoff + ci0 + c4 * 4);
        areg[j] = *reinterpret_cast<const float4*>(a.x + off);       // unconditional: see the note at load_b
        a_okmask = ok ? (a_okmask | (1u << j)) : (a_okmask & ~(1u << j));
    };
    auto load_a = [&](int ci0) {
        if (!UPS) {
#pragma unroll
            for (int j = 0; j < A_ITEMS; ++j) load_a_item(j, ci0);
        } else {
            // prologue only: four items (16 float4) in flight at a time keeps the register footprint bounded
#pragma unroll
            for (int j0 = 0; j0 < A_ITEMS; j0 += 4) {
                float4 r[4][4];
                float wg[4][3];
#pragma unroll
                for (int q = 0; q < 4; ++q) if (j0 + q < A_ITEMS) ups_fetch(j0 + q, ci0, r[q], wg[q]);
#pragma unroll
                for (int q = 0; q < 4; ++q) if (j0 + q < A_ITEMS) ups_lerp(j0 + q, r[q], wg[q]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto store_a = [&](int buf) {
        int t_op = tid;
        asm volatile("" : "+v"(t_op));      // (as in ups_fetch: keeps the per-item LDS offsets out of long-lived registers)
#pragma unroll
        for (int j = 0; j < A_ITEMS; ++j) {
            const int e = t_op + NTH * j;
            if (e >= HP * 8) continue;
            const int px = e >> 3, c4 = e & 7;
            const float4 v = (a_okmask >> j) & 1u ? areg[j] : make_float4(0.f, 0.f, 0.f, 0.f);
            bf16x4 hi, lo;
            hi[0] = (__bf16)v.x; hi[1] = (__bf16)v.y; hi[2] = (__bf16)v.z; hi[3] = (__bf16)v.w;
            __bf16* dst = As + (size_t)buf * HP * LDA + px * LDA + c4 * 4;
            *reinterpret_cast<bf16x4*>(dst) = hi;
            if (NPL == 2) {
                lo[0] = (__bf16)(v.x - (float)hi[0]); lo[1] = (__bf16)(v.y - (float)hi[1]);
                lo[2] = (__bf16)(v.z - (float)hi[2]); lo[3] = (__bf16)(v.w - (float)hi[3]);
                *reinterpret_cast<bf16x4*>(dst + CK) = lo;
            }
        }
    };
    // B tile: BNH rows x 32 k = 4 chunks of 8 per row: one 16-B chunk per thread per plane
    static_assert(B_PASSES == 1, "B staging");
    const int b_row = tid >> 2, b_k8 = tid & 3;
    const bool b_active = b_row < BNH;
    // Loads are UNCONDITIONAL (row clamped into range) and the zero-fill happens at LDS-store time: a `cond ? load : 0`
    // makes hipcc branch around the load and wait vmcnt(0) right behind it, which serialises the prefetch with the MFMAs
    // (cdna_hip_programming.md section 5, ".s-level traps", item (c)).
    const int b_n = n0 + b_row;
    const bool b_ok = b_active && b_n < p.Cout;
    const unsigned b_rowoff = (unsigned)((b_n < p.Cout ? b_n : p.Cout - 1) * a.Kp);
    auto load_b = [&](int tap, int ci0) {
        const unsigned col = (unsigned)(tap * p.Cin + ci0 + b_k8 * 8);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
            breg[pl] = *reinterpret_cast<const uint4*>(a.w + pl * a.plane_stride + b_rowoff + col);
    };
    auto store_b = [&](int buf) {
        if (!b_active) return;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
            *reinterpret_cast<uint4*>(Bs + ((size_t)buf * NPL + pl) * BNH * LDH + swz(b_row, b_k8)) =
                b_ok ? breg[pl] : make_uint4(0u, 0u, 0u, 0u);
    };

    constexpr int TI = 2 * TMW;          // 16-pixel row tiles per wave
    f32x4 acc[TI][4];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
    const int r16 = lane & 15, kq = lane >> 4;
    int a_pix16[TI];
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int pidx = wm * (32 * TMW) + i * 16 + r16;
        a_pix16[i] = ((pidx >> 4) * HW_ + (pidx & 15)) * LDA + kq * 8;     // element offset of this lane's fragment chunk at tap (0, 0)
    }

    const int nchunks = p.Cin / CK;
    // prologue: halo of chunk 0 and weights of (chunk 0, tap 0)
    load_a(0);
    load_b(0, 0);
    store_a(0);
    store_b(0);
    __syncthreads();

    // The nine taps are unrolled (tap, ky, kx, the halo shift and the prefetch schedule are compile-time constants).  APF: the A
    // fragments of tap t + 1 are read from the (unchanged) halo image while tap t's MFMAs run, into the other of two register
    // sets -- after the per-tap barrier only the eight B fragments stand between a wave and its MFMAs.  (With all 16 fragment
    // reads behind the barrier the eight waves queued ~128 KB on the LDS port before any MFMA could start: ~1/3 of a tap.)
    constexpr bool APF = !UPS && !HEAD;
    int bbuf = 0;
    bf16x8 afh[2][TI], afl[2][TI];
    auto read_a = [&](const __bf16* Ah, int shift, bf16x8 (&h)[TI], bf16x8 (&l)[TI]) {
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const int ao = a_pix16[i] + shift * LDA;        // shift is a compile-time constant: folded into the ds_read offset
            h[i] = *reinterpret_cast<const bf16x8*>(Ah + ao);
            if (NPL == 2) l[i] = *reinterpret_cast<const bf16x8*>(Ah + CK + ao);
        }
    };
    auto tap_body = [&](auto tapc, const int c, const __bf16* Ah) {
        constexpr int tap = decltype(tapc)::value;
        constexpr bool last = tap == 8;
        constexpr int ky = tap / 3, kx = tap % 3;
        constexpr int shift = (ky * D) * HW_ + kx * D;
        constexpr int nshift = (((tap + 1) / 3) * D) * HW_ + ((tap + 1) % 3) * D;
        constexpr int S = APF ? (tap & 1) : 0;
        // prefetch the next weight tile (next tap, or tap 0 of the next chunk) and, mid-chunk, the next halo
        const bool more = !(last && c + 1 == nchunks);
        if (more) load_b(last ? 0 : tap + 1, last ? (c + 1) * CK : c * CK);
        constexpr int IPT = (A_ITEMS + 8) / 9;          // UPS: halo items fetched per tap
        float4 raw[IPT][4];
        float rwg[IPT][3];
        const bool nxt = c + 1 < nchunks;
        if (nxt) {   // single-buffered: the registers hold the next halo until the chunk ends
            if (!UPS) {
                if (tap == 4) load_a((c + 1) * CK);
            } else {
#pragma unroll
                for (int q = 0; q < IPT; ++q)
#pragma unroll
                    for (int j = q; j < A_ITEMS; j += IPT)
                        if (tap == j / IPT) ups_fetch(j, (c + 1) * CK, raw[q], rwg[q]);
            }
        }
        // keep the global prefetches at the top of the tap: with the taps unrolled (no branch around them) hipcc sinks the loads
        // to their first use below the MFMAs, which exposes the full L2 round trip in front of every barrier
        __builtin_amdgcn_sched_barrier(0);
        const __bf16* Bh = Bs + ((size_t)bbuf * NPL) * BNH * LDH;
        bf16x8 bh[4], bl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int bo = swz(wn * 64 + j * 16 + r16, kq);
            bh[j] = *reinterpret_cast<const bf16x8*>(Bh + bo);
            if (NPL == 2) bl[j] = *reinterpret_cast<const bf16x8*>(Bh + BNH * LDH + bo);
        }
        if (!APF) read_a(Ah, shift, afh[0], afl[0]);
        auto mfma_rows = [&](int i0, int i1) {
#pragma unroll
            for (int i = i0; i < i1; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // WEIGHTS as the row operand: D[row = channel 4 kq + e][col = pixel r16] -- a lane ends up with four CONSECUTIVE
                    // output channels of one pixel (same products, same k order as with the operands the other way round)
                    if (NPL == 2) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], afl[S][i], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], afh[S][i], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], afh[S][i], acc[i][j], 0, 0, 0);
                }
        };
        if (APF && !last && BNH == 64) {
            read_a(Ah, nshift, afh[1 - S], afl[1 - S]);     // 4-wave kernels: left to hipcc's scheduler (end of the burst)
            mfma_rows(0, TI);
        } else if (APF && !last) {
            // look-ahead reads in the MIDDLE of the MFMA burst: left alone hipcc puts them at its end (the lgkmcnt(0) in front of
            // the barrier then waits for them), pinned in front of the burst they cost 30 more live registers and spill
            mfma_rows(0, TI / 2);
            __builtin_amdgcn_sched_barrier(0);
            read_a(Ah, nshift, afh[1 - S], afl[1 - S]);
            __builtin_amdgcn_sched_barrier(0);
            mfma_rows(TI / 2, TI);
        } else {
            mfma_rows(0, TI);
        }
        if (more) store_b(bbuf ^ 1);
        if (UPS && nxt) {
#pragma unroll
            for (int q = 0; q < IPT; ++q)
#pragma unroll
                for (int j = q; j < A_ITEMS; j += IPT)
                    if (tap == j / IPT) ups_lerp(j, raw[q], rwg[q]);
        }
        if (A_DOUBLE && last && nxt) store_a((c + 1) & 1);
        __syncthreads();
        bbuf ^= 1;
    };
#pragma unroll 1
    for (int c = 0; c < nchunks; ++c) {
        const int abuf = A_DOUBLE ? (c & 1) : 0;
        const __bf16* Ah = As + (size_t)abuf * HP * LDA;
        if (APF) read_a(Ah, 0, afh[0], afl[0]);
        tap_body(std::integral_constant<int, 0>{}, c, Ah);
        tap_body(std::integral_constant<int, 1>{}, c, Ah);
        tap_body(std::integral_constant<int, 2>{}, c, Ah);
        tap_body(std::integral_constant<int, 3>{}, c, Ah);
        tap_body(std::integral_constant<int, 4>{}, c, Ah);
        tap_body(std::integral_constant<int, 5>{}, c, Ah);
        tap_body(std::integral_constant<int, 6>{}, c, Ah);
        tap_body(std::integral_constant<int, 7>{}, c, Ah);
        tap_body(std::integral_constant<int, 8>{}, c, Ah);
        if (!A_DOUBLE && c + 1 < nchunks) {   // single A buffer: refill between chunks (fetched under the taps above)
            store_a(0);
            __syncthreads();
        }
    }

    // C/D map with the weights as the row operand: lane (r16, kq) holds, for every (i, j), channels 16 j + 4 kq .. + 3 of pixel
    // 16 i + r16 of its wave's 64 pixels.
    if constexpr (HEAD) {
        // ---- fused head epilogue (Cout == 64, one channel tile).  That register layout IS the input layout of seg_head_group
        // (lane = (pixel, channel quad), x[j] = channels 16 j + 4 kq .. + 3), so bias + activation are applied in place and every
        // 16-pixel row block goes straight into the 64 -> C head + softmax(+softmax) + arg-max: no LDS pass, no barrier, and the
        // 64-channel activation never reaches HBM.
        static_assert(BNH == 64, "the fused head needs the whole 64-channel pixel in one workgroup");
        // opaque copies of the lane coordinates: the epilogue's address parts are formed here instead of being hoisted in front of the
        // chunk loop and spilled around it
        int r16_e = r16, kq_e = kq;
        asm volatile("" : "+v"(r16_e), "+v"(kq_e));
        float wreg[16], hbias[4];
        ape_seg::seg_head_load_weights(a.head_w, a.head_b, a.head_c, lane, wreg, hbias);
        const ape::ActFast af = ape::act_fast_make(p.act, p.alpha);
        float4 cb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = j * 16 + kq_e * 4;
            const float* bp = a.bias ? a.bias + (p.bias_bstride ? (size_t)b * p.bias_bstride : 0) : nullptr;
            cb[j] = make_float4(bp && n < p.Cout ? bp[n] : 0.f, bp && n + 1 < p.Cout ? bp[n + 1] : 0.f, bp && n + 2 < p.Cout ? bp[n + 2] : 0.f,
                                bp && n + 3 < p.Cout ? bp[n + 3] : 0.f);
        }
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const int pidx = wm * (32 * TMW) + i * 16 + r16_e;
            const int gy = y0 + (pidx >> 4), gx = x0 + (pidx & 15);
            float4 xv[4];
            if (p.act == APE_ACT_SIGMOID) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    xv[j] = make_float4(activate_h(acc[i][j][0] + cb[j].x, p.act, p.alpha), activate_h(acc[i][j][1] + cb[j].y, p.act, p.alpha),
                                        activate_h(acc[i][j][2] + cb[j].z, p.act, p.alpha), activate_h(acc[i][j][3] + cb[j].w, p.act, p.alpha));
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    xv[j] = make_float4(ape::act_fast(acc[i][j][0] + cb[j].x, af), ape::act_fast(acc[i][j][1] + cb[j].y, af),
                                        ape::act_fast(acc[i][j][2] + cb[j].z, af), ape::act_fast(acc[i][j][3] + cb[j].w, af));
            }
            int am;
            float pm;
            ape_seg::seg_head_group(xv, wreg, hbias, a.head_c, lane, a.head_dsm, am, pm);
            if (kq_e == 0 && gy < p.Ho && gx < p.Wo) {
                const size_t m = ((size_t)b * p.Ho + gy) * p.Wo + gx;
                a.label[m] = (uint8_t)am;
                a.score[m] = pm;
            }
        }
        return;
    }
    // ---- epilogue straight from the registers: the four lanes of a pixel write 64 contiguous bytes per (i, j), neighbouring j
    // blocks complete the 128-B lines in L2; no LDS pass, no barrier (the staged form cost two passes of stage -> barrier -> store
    // -> barrier per tile, which with K = 576 was a large part of a tile)
    const bool vec_ok = (p.ldy % 4 == 0) && (p.yoff % 4 == 0) && (!a.res || (p.ldr % 4 == 0 && p.roff % 4 == 0));
    const int nq = n0 + wn * 64 + kq * 4;               // + 16 j
    const ape::ActFast af2 = ape::act_fast_make(p.act, p.alpha);
    const bool sigm = p.act == APE_ACT_SIGMOID;
    const float* bp = a.bias ? a.bias + (p.bias_bstride ? (size_t)b * p.bias_bstride : 0) : nullptr;
    float4 b4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = nq + j * 16;
        b4[j] = make_float4(bp && n < p.Cout ? bp[n] : 0.f, bp && n + 1 < p.Cout ? bp[n + 1] : 0.f, bp && n + 2 < p.Cout ? bp[n + 2] : 0.f,
                            bp && n + 3 < p.Cout ? bp[n + 3] : 0.f);
    }
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int pidx = wm * (32 * TMW) + i * 16 + r16;
        const int gy = y0 + (pidx >> 4), gx = x0 + (pidx & 15);
        const bool pok = gy < p.Ho && gx < p.Wo;
        const size_t m = ((size_t)b * p.Ho + (pok ? gy : 0)) * p.Wo + (pok ? gx : 0);
        float4 rr[4];
        if (a.res && vec_ok) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (nq + j * 16 + 4 <= p.Cout) rr[j] = *reinterpret_cast<const float4*>(a.res + m * p.ldr + p.roff + nq + j * 16);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = nq + j * 16;
            if (!pok || n >= p.Cout) continue;
            float vv[4] = {acc[i][j][0] + b4[j].x, acc[i][j][1] + b4[j].y, acc[i][j][2] + b4[j].z, acc[i][j][3] + b4[j].w};
            const int nvalid = min(4, p.Cout - n);
            if (vec_ok && nvalid == 4) {
                if (a.res) { vv[0] += rr[j].x; vv[1] += rr[j].y; vv[2] += rr[j].z; vv[3] += rr[j].w; }
                *reinterpret_cast<float4*>(a.y + m * p.ldy + p.yoff + n) =
                    sigm ? make_float4(activate_h(vv[0], p.act, p.alpha), activate_h(vv[1], p.act, p.alpha), activate_h(vv[2], p.act, p.alpha),
                                       activate_h(vv[3], p.act, p.alpha))
                         : make_float4(ape::act_fast(vv[0], af2), ape::act_fast(vv[1], af2), ape::act_fast(vv[2], af2), ape::act_fast(vv[3], af2));
            } else {
                for (int k = 0; k < nvalid; ++k) {
                    float t = vv[k];
                    if (a.res) t += a.res[m * p.ldr + p.roff + n + k];
                    a.y[m * p.ldy + p.yoff + n + k] = activate_h(t, p.act, p.alpha);
                }
            }
        }
    }
}

template <int NSPLIT, int D, int BNH, bool UPS, bool HEAD = false>
int launch_halo(const HaloArgs& a, hipStream_t st)
{
    constexpr int NPL = NSPLIT == 3 ? 2 : 1;
    constexpr int HP = (TS + 2 * D) * (TS + 2 * D);
    constexpr bool A_DOUBLE = halo_a_double(NPL, HP, BNH);
#if defined(APE_UPS_DEBUG) && APE_UPS_DEBUG == 2
    constexpr size_t lds_ops = ((A_DOUBLE ? 2 : 1) * HP * halo_lda(NPL) + 2 * NPL * BNH * LDH) * 2 + (UPS ? HP * 16 : 0);
#else
    constexpr size_t lds_ops = ((A_DOUBLE ? 2 : 1) * HP * halo_lda(NPL) + 2 * NPL * BNH * LDH) * 2;
#endif
    constexpr size_t lds = lds_ops;      // (the epilogues work from the registers: no staging rows)
    static_assert(lds <= 160 * 1024, "LDS budget");
    static_assert(!UPS || D == 1, "fused up-sampling is built for the d = 1 kernel");
    auto kern = conv3x3_halo_kernel<NSPLIT, D, BNH, UPS, HEAD>;
    static ape::DeviceOnce once;       // (per kernel instantiation)
    if (int rc = ape::device_once(once, reinterpret_cast<const void*>(kern), (int)lds, nullptr)) return rc;
    const int grid = a.p.B * a.tiles_x * a.tiles_y * a.n_tiles;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(halo_threads(BNH)), lds, st, a);
    return ape::check_launch("ape_conv3x3_halo_bf16");
}

}  // namespace

/* 1 if ape_conv3x3_halo_bf16 supports this geometry (3x3, stride 1, pad == dil in {1,2,4}, Cin % 32 == 0), else 0 */
extern "C" int ape_conv3x3_halo_supported(const ape_conv_params* params)
{
    if (!params) return 0;
    const ape_conv_params& p = *params;
    return (p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == p.dil && (p.dil == 1 || p.dil == 2 || p.dil == 4) &&
            p.Cin % CK == 0 && p.Cin >= CK && p.H == p.Ho && p.W == p.Wo) ? 1 : 0;
}

extern "C" int ape_conv3x3_halo_bf16(const float* x, const void* w_packed, const float* bias, const float* residual, float* y,
                                     const ape_conv_params* params, int nsplit, void* stream)
{
    if (!x || !w_packed || !y || !params || (nsplit != 1 && nsplit != 3)) return APE_EINVAL;
    if (!ape_conv3x3_halo_supported(params)) return APE_EINVAL;
    const ape_conv_params& p = *params;
    if (p.B < 0 || p.H < 1 || p.W < 1 || p.Cout < 1) return APE_EINVAL;
    if (p.ldx % 4 || p.xoff % 4 || p.xoff + p.Cin > p.ldx || p.yoff + p.Cout > p.ldy) return APE_EINVAL;
    if (residual && p.roff + p.Cout > p.ldr) return APE_EINVAL;
    if (p.act < APE_ACT_NONE || p.act > APE_ACT_SIGMOID) return APE_EINVAL;
    if (p.B == 0) return APE_OK;
    const long K = 9L * p.Cin, Kp = (K + 7) / 8 * 8;
    if (p.ups != 0 && p.ups != 1) return APE_EINVAL;
    if ((long)p.B * p.H * p.W * p.ldx >= (1L << 31) || (long)p.Cout * Kp >= (1L << 31)) return APE_EINVAL;
    HaloArgs a;
    a.x = x; a.w = (const __bf16*)w_packed; a.bias = bias; a.res = residual; a.y = y; a.p = p;
    a.Kp = (int)Kp;
    a.plane_stride = (long)p.Cout * Kp;
    a.tiles_x = ape::ceil_div(p.W, TS);
    a.tiles_y = ape::ceil_div(p.H, TS);
    a.head_w = a.head_b = nullptr; a.label = nullptr; a.score = nullptr; a.head_c = 0; a.head_dsm = 0;
    hipStream_t st = (hipStream_t)stream;
    const bool narrow = p.Cout <= 64;
    a.n_tiles = ape::ceil_div(p.Cout, narrow ? 64 : 128);
#define HALO_DISPATCH(NS, DD) (narrow ? launch_halo<NS, DD, 64, false>(a, st) : launch_halo<NS, DD, 128, false>(a, st))
    if (p.ups) {
        if (p.dil != 1 || (p.H & 1) || (p.W & 1)) return APE_EINVAL;
        if (nsplit == 3) return narrow ? launch_halo<3, 1, 64, true>(a, st) : launch_halo<3, 1, 128, true>(a, st);
        return narrow ? launch_halo<1, 1, 64, true>(a, st) : launch_halo<1, 1, 128, true>(a, st);
    }
    if (nsplit == 3) {
        if (p.dil == 1) return HALO_DISPATCH(3, 1);
        if (p.dil == 2) return HALO_DISPATCH(3, 2);
        return HALO_DISPATCH(3, 4);
    }
    if (p.dil == 1) return HALO_DISPATCH(1, 1);
    if (p.dil == 2) return HALO_DISPATCH(1, 2);
    return HALO_DISPATCH(1, 4);
#undef HALO_DISPATCH
}

/* Same convolution (Cout must be 64, no residual) with the segmentation head fused into its epilogue: see include/ape_hip.h */
extern "C" int ape_conv3x3_halo_seghead_bf16(const float* x, const void* w_packed, const float* bias, const ape_conv_params* params,
                                             int nsplit, const float* head_w, const float* head_b, int C, uint8_t* label, float* score,
                                             int double_softmax, void* stream)
{
    if (!x || !w_packed || !params || !head_w || !label || !score || (nsplit != 1 && nsplit != 3) || C < 1 || C > 16) return APE_EINVAL;
    if (!ape_conv3x3_halo_supported(params)) return APE_EINVAL;
    const ape_conv_params& p = *params;
    if (p.Cout != 64 || p.dil != 1 || p.B < 0 || p.H < 1 || p.W < 1) return APE_EINVAL;
    if (p.ldx % 4 || p.xoff % 4 || p.xoff + p.Cin > p.ldx) return APE_EINVAL;
    if (p.act < APE_ACT_NONE || p.act > APE_ACT_SIGMOID || (p.ups != 0 && p.ups != 1)) return APE_EINVAL;
    if (p.ups && ((p.H & 1) || (p.W & 1))) return APE_EINVAL;
    if (p.B == 0) return APE_OK;
    const long K = 9L * p.Cin, Kp = (K + 7) / 8 * 8;
    if ((long)p.B * p.H * p.W * p.ldx >= (1L << 31) || (long)p.Cout * Kp >= (1L << 31)) return APE_EINVAL;
    HaloArgs a;
    a.x = x; a.w = (const __bf16*)w_packed; a.bias = bias; a.res = nullptr; a.y = nullptr; a.p = p;
    a.Kp = (int)Kp;
    a.plane_stride = (long)p.Cout * Kp;
    a.tiles_x = ape::ceil_div(p.W, TS);
    a.tiles_y = ape::ceil_div(p.H, TS);
    a.n_tiles = 1;
    a.head_w = head_w; a.head_b = head_b; a.label = label; a.score = score; a.head_c = C; a.head_dsm = double_softmax;
    hipStream_t st = (hipStream_t)stream;
    if (nsplit == 3) return p.ups ? launch_halo<3, 1, 64, true, true>(a, st) : launch_halo<3, 1, 64, false, true>(a, st);
    return p.ups ? launch_halo<1, 1, 64, true, true>(a, st) : launch_halo<1, 1, 64, false, true>(a, st);
}
