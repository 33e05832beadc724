// up_3 of the segmentor (pspnet.py:51: bilinear x2, align_corners=True -> 3x3 conv 64 -> 64 -> PReLU) with the classification head
// (final 1x1 conv rows 0..C-1 + softmax (+ softmax) + arg-max, pspnet.py:53-55, pipeline/utils.py:429-435) in its epilogue, as a
// WAVE-SPECIALISED persistent kernel: the successor of conv3x3_halo_kernel<3,1,64,true,true> (conv3x3_halo.hip), which spent 3.1 VALU
// instructions per MFMA on the fused up-sampling / bf16 split / head and ran them in the SAME waves as the MFMAs, phase after phase
// (matrix pipe busy 42 % of the cycles, VALU and MFMA co-executing 6 %).  Here one workgroup of eight waves owns a CU:
//   * waves 0-3 ("matrix waves", one per SIMD) do nothing but ds_read_b128 fragments -> 48 MFMAs per tap, and the head epilogue of
//     their 64 pixels x 64 channels from the accumulators;
//   * waves 4-7 ("producer waves", the SIMD partners of 0-3) build the operands: per halo pixel a 16-B table entry (corner offset,
//     +1 steps, blend weights, in-image bit), four corner loads per 4-channel item, the bilinear blend in the operation order of
//     ops.hip:bilinear_kernel, the split into bf16 hi | lo, and the ds_write into the OTHER of two halo images; and the weights of
//     tap t+2 into a ring of three 8-KB tiles.  Every global load is requested two (weights) or three (halo corners) taps before its
//     data is used: a tap lasts about as long as ONE memory round trip, and a producer that waits for the previous tap's loads at
//     the top of every tap makes every tap last a round trip (measured: 5.6 ms for the layer instead of 5.2 before).
//   An MFMA holds a SIMD's issue port for 8 of its 16 cycles, so the partner's VALU stream (~1.2 k instructions per tile against
//   864 MFMAs) runs in the other half: matrix and vector pipes work at the same time by construction instead of by luck.
//   * one barrier per tap for all eight waves; the producers run ONE CHUNK AHEAD (the halo of chunk g+1 is built during the taps
//     0..7 of chunk g, its first loads go out in tap 8 of chunk g-1) and straight across tile seams (workgroups are persistent:
//     blockIdx, + gridDim, ...), so the matrix waves never see a prologue after their first tile;
//   * every LDS address of the matrix waves is lane base + compile-time offset (18 taps fully unrolled, halo buffer = chunk parity,
//     weight slot = tap % 3, fragment register set = tap parity).
// Same operands (same blend expressions, same hi = bf16(v), lo = bf16(v - hi)), same products in the same order (chunk outer, tap
// inner; hi.lo, lo.hi, hi.hi) as the kernel it replaces, hence the same accumulators, labels and scores bit for bit
// (tests/test_gpu_conv.py compares both with the unfused conv + ape_seg_head_f32 pair).
#include <type_traits>
#include <utility>
#include "common.h"
#include "seg_head.h"

// timing-only ablation switches (results wrong when set): compiled out of the ISA-audit build (tools/isa_audit.py, -DAPE_NO_ABLATIONS)
#ifdef APE_NO_ABLATIONS
#define ABL(bit) 0
#else
#define ABL(bit) (a.dbg & (bit))
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TS = 16;                      // output tile edge
constexpr int HW_ = TS + 2;                 // halo edge (3x3, dilation 1)
constexpr int HP = HW_ * HW_;               // 324 halo pixels
constexpr int PIX_B = 160;                  // bytes per halo pixel of one 32-channel chunk: hi 64 | lo 64 | pad 32 (conflict-free ds_read_b128, conv3x3_halo.hip)
constexpr int HALO_B = HP * PIX_B;          // 51,840
constexpr int WT_PLANE_B = 64 * 64;         // one plane (hi or lo) of one tap's weights: 64 rows x 32 bf16
constexpr int WT_B = 2 * WT_PLANE_B;        // 8 KB
constexpr int TBL_B = HP * 16;
constexpr int OFF_HALO = 0;
constexpr int OFF_WT = OFF_HALO + 2 * HALO_B;
constexpr int OFF_TBL = OFF_WT + 3 * WT_B;
constexpr int LDS_BYTES = OFF_TBL + 2 * TBL_B;      // 138,624
constexpr int NPROD = 256;                  // producer threads
constexpr int A_ITEMS = (HP * 8 + NPROD - 1) / NPROD;      // 4-channel items per producer thread per chunk (11)
static_assert(OFF_WT % 16 == 0 && OFF_TBL % 16 == 0 && LDS_BYTES <= 160 * 1024, "LDS carve");
static_assert(A_ITEMS == 11, "the batch schedule below deals eleven items over the taps 0..7");
// item j is blended and written in tap BATCH_OF[j] of the chunk before the one it belongs to (its corner loads go out one tap earlier)
__device__ constexpr int batch_of(int j) { return j < 6 ? j / 2 : j - 3; }       // {0,0,1,1,2,2,3,4,5,6,7}

struct Up3Args {
    const float* x;         // [B][H/2][W/2][ldx] fp32, channels xoff .. xoff + 63
    const __bf16* w;        // packed weights of ape_pack_weights_bf16: hi plane [64][Kp], lo plane at + plane_stride
    const float* bias;
    const float* head_w;
    const float* head_b;
    uint8_t* label;
    float* score;
    int B, H, W, ldx, xoff, Kp;
    long plane_stride;
    int act;
    float alpha;
    int bias_bstride, head_c, head_dsm;
    int tiles_x, tiles_y;
    unsigned long long* stamps;     // STAMP build only (diagnostic): per workgroup and matrix wave {burst, barrier, head} cycle sums
    int dbg;                // 1: matrix waves at priority 1; timing-only ablations: 4 = no halo building in the steady state, 8 = no MFMAs, 16 = no head
};

// same chunk permutation as conv3x3_halo.hip: the 16 lanes of every ds_read_b128 lane group fall on 16 distinct bank slots
__device__ __forceinline__ int swz_b(int row, int chunk16) { return row * 64 + ((chunk16 ^ ((0 - (row >> 2)) & 3)) << 4); }    // BYTES

// NONE / RELU / PRELU as selects on loop-invariant scalars (a `switch` per element compiled to a cascade of scalar compares and branches
// per element: ~1.8 k scalar instructions in a 256-element epilogue); same values bit for bit (1 * v == v, also for -0 and NaN)
__device__ __forceinline__ float act_u3(float v, int act, float alpha)
{
    if (act == APE_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
    const float neg = act == APE_ACT_RELU ? 0.f : (act == APE_ACT_PRELU ? alpha : 1.f) * v;
    return v > 0.f ? v : neg;
}

// every LDS access of this wave done, then the workgroup barrier.  The global loads a producer has in flight stay in flight
// (__syncthreads() would drain them: it waits vmcnt(0) too).
__device__ __forceinline__ void wg_barrier()
{
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_s_waitcnt(0xC07F);         // lgkmcnt(0) alone, as a builtin: hipcc then knows this wave's LDS reads are back (inside an
                                                // asm it would keep counting them and, 32 reads later, wait for two of the NEW tap's fragments)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#endif
}

template <int... Ts, class F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, Ts...>, F&& f)
{
    (f(std::integral_constant<int, Ts>{}), ...);
}

template <bool STAMP>
__global__ __launch_bounds__(512, 2) void up3_head_ws_kernel(const Up3Args a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const int nwg = a.B * tiles_per_img;
    const int q = nwg / 8, r = nwg % 8;
    const int grid = (int)gridDim.x;
    const int my_tiles = (nwg - 1 - (int)blockIdx.x) / grid + 1;
    // tile number k of this workgroup -> (frame, tile origin); dispatch ids blockIdx + k * gridDim, logical ids dealt to the XCDs in runs
    auto decode = [&](int k, int& tb, int& ty0, int& tx0) __attribute__((always_inline)) {
        const int orig = (int)blockIdx.x + k * grid;
        const int xcd = orig % 8;
        const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
        tb = logical / tiles_per_img;
        const int trem = logical - tb * tiles_per_img;
        ty0 = (trem / a.tiles_x) * TS;
        tx0 = (trem % a.tiles_x) * TS;
    };
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Cin = 64;

    if (wave >= 4) {
        // =================================================== producer waves ===================================================
        if (a.dbg & 32) __builtin_amdgcn_s_setprio(3);
        const int ptid = tid - 256;
        const int hl = a.H / 2, wl = a.W / 2;
        const float ups_sh = a.H > 1 ? (float)(hl - 1) / (float)(a.H - 1) : 0.f;
        const float ups_sw = a.W > 1 ? (float)(wl - 1) / (float)(a.W - 1) : 0.f;
        // the per-pixel part of the bilinear sample of tile k, one 16-B entry per halo pixel (expressions of conv3x3_halo.hip / bilinear_kernel)
        auto build_table = [&](int k) __attribute__((always_inline)) {
            int tb, ty0, tx0;
            decode(k, tb, ty0, tx0);
            uint4* tbl = reinterpret_cast<uint4*>(smem + OFF_TBL + (k & 1) * TBL_B);
            for (int px = ptid; px < HP; px += NPROD) {
                const int hy = px / HW_, hx = px - hy * HW_;
                const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
                const bool ok = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
                const int cy = gy < 0 ? 0 : (gy >= a.H ? a.H - 1 : gy), cx = gx < 0 ? 0 : (gx >= a.W ? a.W - 1 : gx);
                const float fy = ups_sh * (float)cy, fx = ups_sw * (float)cx;
                const int iy0 = (int)fy, ix0 = (int)fx;
                const float ly1 = fy - (float)iy0, lx1 = fx - (float)ix0;
                uint4 ent;
                ent.x = (unsigned)(((tb * hl + iy0) * wl + ix0) * a.ldx + a.xoff);
                ent.y = (ix0 < wl - 1 ? 1u : 0u) | (iy0 < hl - 1 ? 2u : 0u) | (ok ? 4u : 0u);
                ent.z = __float_as_uint(lx1);
                ent.w = __float_as_uint(ly1);
                tbl[px] = ent;
            }
        };
        // item j of this thread: 4 channels (ci0 + 4 (e & 7) ..) of halo pixel e >> 3, e = ptid + 256 j; the four corner loads
        auto fetch = [&](int j, int ci0, int tbl_par, float4 (&rw)[4], float4& wgt) __attribute__((always_inline)) {
            const int e = ptid + NPROD * j;
            const int px = e < HP * 8 ? e >> 3 : HP - 1;
            const uint4 ent = reinterpret_cast<const uint4*>(smem + OFF_TBL + tbl_par * TBL_B)[px];
            const unsigned base = ent.x + (unsigned)(ci0 + (e & 7) * 4);
            const unsigned dxo = (ent.y & 1u) ? (unsigned)a.ldx : 0u, dyo = (ent.y & 2u) ? (unsigned)(wl * a.ldx) : 0u;
            rw[0] = *reinterpret_cast<const float4*>(a.x + base);
            rw[1] = *reinterpret_cast<const float4*>(a.x + base + dxo);
            rw[2] = *reinterpret_cast<const float4*>(a.x + base + dyo);
            rw[3] = *reinterpret_cast<const float4*>(a.x + base + dyo + dxo);
            wgt = make_float4(__uint_as_float(ent.z), __uint_as_float(ent.w), (e < HP * 8 && (ent.y & 4u)) ? 1.f : 0.f, 0.f);
        };
        // blend -> zero outside the image -> split -> the two 8-byte halves of the pixel's LDS row in halo image `buf`
        auto blend_store = [&](int j, int buf, const float4 (&rw)[4], const float4& wgt) __attribute__((always_inline)) {
            const int e = ptid + NPROD * j;
            if (e >= HP * 8) return;
            const bool ok = wgt.z != 0.f;
            const float lx1 = wgt.x, ly1 = wgt.y, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
            float4 o;
            o.x = ly0 * (lx0 * rw[0].x + lx1 * rw[1].x) + ly1 * (lx0 * rw[2].x + lx1 * rw[3].x);
            o.y = ly0 * (lx0 * rw[0].y + lx1 * rw[1].y) + ly1 * (lx0 * rw[2].y + lx1 * rw[3].y);
            o.z = ly0 * (lx0 * rw[0].z + lx1 * rw[1].z) + ly1 * (lx0 * rw[2].z + lx1 * rw[3].z);
            o.w = ly0 * (lx0 * rw[0].w + lx1 * rw[1].w) + ly1 * (lx0 * rw[2].w + lx1 * rw[3].w);
            const float4 v = ok ? o : make_float4(0.f, 0.f, 0.f, 0.f);
            bf16x4 hi, lo;
            hi[0] = (__bf16)v.x; hi[1] = (__bf16)v.y; hi[2] = (__bf16)v.z; hi[3] = (__bf16)v.w;
            lo[0] = (__bf16)(v.x - (float)hi[0]); lo[1] = (__bf16)(v.y - (float)hi[1]);
            lo[2] = (__bf16)(v.z - (float)hi[2]); lo[3] = (__bf16)(v.w - (float)hi[3]);
            char* dst = smem + OFF_HALO + buf * HALO_B + (e >> 3) * PIX_B + (e & 7) * 8;
            *reinterpret_cast<bf16x4*>(dst) = hi;
            *reinterpret_cast<bf16x4*>(dst + 64) = lo;
        };
        // weights of (chunk c, tap): 64 rows x 32 k x 2 planes = 512 16-B pieces, two per producer thread (one per plane)
        const int b_row = ptid >> 2, b_k8 = ptid & 3;
        const unsigned b_rowoff = (unsigned)(b_row * a.Kp);
        // two register sets (a tile is requested TWO taps before it is written: with one tap of cover every tap lasted a memory round
        // trip); indexed by compile-time constants only -- an array indexed by a loop variable stays in scratch here
        uint4 bw0_hi, bw0_lo, bw1_hi, bw1_lo;
        auto load_w = [&](auto set_c, int c, int tap) __attribute__((always_inline)) {
            constexpr int set = decltype(set_c)::value;
            const unsigned col = (unsigned)(tap * Cin + c * 32 + b_k8 * 8);
            const uint4 h = *reinterpret_cast<const uint4*>(a.w + b_rowoff + col);
            const uint4 l = *reinterpret_cast<const uint4*>(a.w + a.plane_stride + b_rowoff + col);
            if constexpr (set == 0) { bw0_hi = h; bw0_lo = l; } else { bw1_hi = h; bw1_lo = l; }
        };
        auto store_w = [&](auto set_c, int slot) __attribute__((always_inline)) {
            constexpr int set = decltype(set_c)::value;
            char* dst = smem + OFF_WT + slot * WT_B + swz_b(b_row, b_k8);
            if constexpr (set == 0) {
                *reinterpret_cast<uint4*>(dst) = bw0_hi;
                *reinterpret_cast<uint4*>(dst + WT_PLANE_B) = bw0_lo;
            } else {
                *reinterpret_cast<uint4*>(dst) = bw1_hi;
                *reinterpret_cast<uint4*>(dst + WT_PLANE_B) = bw1_lo;
            }
        };

        // ---- prologue: table and whole halo of (tile 0, chunk 0), weights of its taps 0 and 1 (2 and 3 requested), the first three
        // batches of chunk 1 requested -------------------------------------------------------------------------------------------
        using C0 = std::integral_constant<int, 0>;
        using C1 = std::integral_constant<int, 1>;
        float4 raw[3][2][4];        // [register set = batch % 3][item of the batch][corner]
        float4 rwg[3][2];
        build_table(0);
        wg_barrier();
#pragma unroll
        for (int j0 = 0; j0 < A_ITEMS; j0 += 4) {
            float4 pr[4][4];
            float4 pw[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) if (j0 + u < A_ITEMS) fetch(j0 + u, 0, 0, pr[u], pw[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) if (j0 + u < A_ITEMS) blend_store(j0 + u, 0, pr[u], pw[u]);
        }
        load_w(C0{}, 0, 0);
        store_w(C0{}, 0);
        load_w(C1{}, 0, 1);
        store_w(C1{}, 1);
        load_w(C0{}, 0, 2);                               // written in tap 0
        load_w(C1{}, 0, 3);                               // written in tap 1
        if (!(ABL(4))) {
            static_for(std::make_integer_sequence<int, 6>{}, [&](auto jc) __attribute__((always_inline)) {      // batches 0..2 of (tile 0, chunk 1)
                constexpr int j = decltype(jc)::value;
                fetch(j, 32, 0, raw[j / 2][j & 1], rwg[j / 2][j & 1]);
            });
        }
        wg_barrier();

        // ---- steady state: tile k, tap T = 9 c + tap of its 18, fully unrolled (every register set is a compile-time constant).
        //   halo under construction = the NEXT chunk (chunk 1 of this tile during c = 0, chunk 0 of the next tile during c = 1), image
        //   1 - c: tap t <= 7 blends and writes batch t (requested three taps earlier: a batch's corner loads have three taps, about
        //   two memory round trips under load, to land) and requests batch t + 3 -- past 7: batch t - 6 of the chunk after, i.e. of
        //   chunk c of the NEXT tile, whose table is built in tap 4 of chunk 0.  Batches 0..2 hold two items, 3..7 one.
#pragma unroll 1
        for (int k = 0; k < my_tiles; ++k) {
            const bool has_next = k + 1 < my_tiles;
            const bool halo_on = !(ABL(4));
            static_for(std::make_integer_sequence<int, 18>{}, [&](auto Tc) __attribute__((always_inline)) {
                constexpr int T = decltype(Tc)::value;
                constexpr int c = T / 9, tap = T % 9;
                using WS = std::integral_constant<int, (T & 1)>;
                // weights: write the tile of tap T + 2 (requested during tap T - 2) into ring slot (T + 2) % 3 -- the matrix waves read it
                // from tap T + 1 on -- then request the tile of tap T + 4 into the same registers
                if (T + 2 < 18 || has_next) store_w(WS{}, (T + 2) % 3);
                if (T + 4 < 18 || has_next) load_w(WS{}, ((T + 4) % 18) / 9, (T + 4) % 9);
                const bool build_ok = (c == 0 || has_next) && halo_on;
                constexpr int nbuf = 1 - c, nci0 = (1 - c) * 32;
                const int ntbl = (k + c) & 1;                                   // table of the tile the chunk under construction belongs to
                if (tap <= 7 && build_ok) {
                    static_for(std::make_integer_sequence<int, A_ITEMS>{}, [&](auto jc) __attribute__((always_inline)) {
                        constexpr int j = decltype(jc)::value, u = j < 6 ? (j & 1) : 0;
                        if constexpr (batch_of(j) == tap) blend_store(j, nbuf, raw[tap % 3][u], rwg[tap % 3][u]);
                    });
                }
                if constexpr (tap + 3 <= 7) {
                    if (build_ok) {
                        static_for(std::make_integer_sequence<int, A_ITEMS>{}, [&](auto jc) __attribute__((always_inline)) {
                            constexpr int j = decltype(jc)::value, u = j < 6 ? (j & 1) : 0;
                            if constexpr (batch_of(j) == tap + 3) fetch(j, nci0, ntbl, raw[tap % 3][u], rwg[tap % 3][u]);
                        });
                    }
                } else if constexpr (tap >= 6) {
                    // taps 6, 7, 8: batches 0, 1, 2 (items 2 nb, 2 nb + 1) of the chunk after the one under construction = chunk c of the next tile
                    constexpr int nb = tap - 6;
                    static_assert(nb >= 0 && nb <= 2, "batch");
                    if (has_next && halo_on) {
                        fetch(2 * nb, c * 32, (k + 1) & 1, raw[tap % 3][0], rwg[tap % 3][0]);
                        fetch(2 * nb + 1, c * 32, (k + 1) & 1, raw[tap % 3][1], rwg[tap % 3][1]);
                    }
                }                                           // (tap 5 requests nothing)
                if (c == 0 && tap == 4 && has_next) build_table(k + 1);          // read from tap 6 on, behind this tap's barrier
                wg_barrier();
            });
        }
        return;
    }

    // ======================================================= matrix waves =======================================================
    if (a.dbg & 1) __builtin_amdgcn_s_setprio(1);
    const int wm = wave;
    const int r16 = lane & 15, kq = lane >> 4;
    unsigned a_base[4], b_base[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a_base[i] = (unsigned)(OFF_HALO + ((wm * 4 + i) * HW_ + r16) * PIX_B + kq * 16);
#pragma unroll
    for (int j = 0; j < 4; ++j) b_base[j] = (unsigned)(OFF_WT + swz_b(j * 16 + r16, kq));
    float wreg[16], hbias[4];
    ape_seg::seg_head_load_weights(a.head_w, a.head_b, a.head_c, lane, wreg, hbias);
    const ape::ActFast af = ape::act_fast_make(a.act, a.alpha);
    f32x4 acc[4][4];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
    };
    zero_acc();
    bf16x8 afh[2][4], afl[2][4], bfh[2][4], bfl[2][4];
    // fragments of tap T (of a tile) into register set T & 1: halo image T / 9 shifted by the tap, weight slot T % 3
    auto read_frags = [&](auto Tc) __attribute__((always_inline)) {
        constexpr int T = decltype(Tc)::value;
        constexpr int S = T & 1, c = T / 9, tap = T % 9;
        constexpr int shift = ((tap / 3) * HW_ + tap % 3) * PIX_B + c * HALO_B;
        constexpr int slot = (T % 3) * WT_B;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bfh[S][j] = *reinterpret_cast<const bf16x8*>(smem + b_base[j] + slot);
            bfl[S][j] = *reinterpret_cast<const bf16x8*>(smem + b_base[j] + slot + WT_PLANE_B);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            afh[S][i] = *reinterpret_cast<const bf16x8*>(smem + a_base[i] + shift);
            afl[S][i] = *reinterpret_cast<const bf16x8*>(smem + a_base[i] + shift + 64);
        }
    };
    wg_barrier();       // (the producers' table)
    wg_barrier();       // halo of (tile 0, chunk 0), weights of taps 0 and 1
    read_frags(std::integral_constant<int, 0>{});
    int tb, ty0, tx0;
    // diagnostic build: s_memtime around the tap's burst (fragment reads + MFMAs) and around its wait + barrier, summed per wave
    unsigned long long st_burst = 0, st_bar = 0, st_head = 0;
    auto stamp = [&]() __attribute__((always_inline)) -> unsigned long long {
        unsigned long long t = 0;
        if (STAMP) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        return t;
    };
#pragma unroll 1
    for (int k = 0; k < my_tiles; ++k) {
        static_for(std::make_integer_sequence<int, 18>{}, [&](auto Tc) __attribute__((always_inline)) {
            constexpr int T = decltype(Tc)::value;
            constexpr int S = T & 1;
            const unsigned long long t0 = stamp();
            // the next tap's fragments (across the tile seam: tap 0 of the next tile, whose halo and weights the producers have ready;
            // after the last tile the read returns stale bytes nobody uses), pinned in front of the MFMAs
            read_frags(std::integral_constant<int, (T + 1) % 18>{});
            __builtin_amdgcn_sched_barrier(0);
            if (!(ABL(8)))
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // weights as the row operand: D[channel 4 kq + e][pixel r16]; product order of conv3x3_halo.hip
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfh[S][j], afl[S][i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfl[S][j], afh[S][i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfh[S][j], afh[S][i], acc[i][j], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long t1 = stamp();
            wg_barrier();
            const unsigned long long t2 = stamp();
            if (STAMP) { st_burst += t1 - t0; st_bar += t2 - t1; }
        });
        const unsigned long long th0 = stamp();
        // ---- head epilogue of tile k from the accumulators (lane (r16, kq): channels 16 j + 4 kq .. + 3 of pixel (4 wm + i, r16)) ----
        decode(k, tb, ty0, tx0);
        const float* bp = a.bias ? a.bias + (a.bias_bstride ? (size_t)tb * a.bias_bstride : 0) : nullptr;
        float4 cb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) cb[j] = bp ? *reinterpret_cast<const float4*>(bp + j * 16 + kq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 xv[4][4];        // [pixel row i = group][channel block j]: the head's input layout (the accumulators, activated in place)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (a.act == APE_ACT_SIGMOID) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    xv[i][j] = make_float4(act_u3(acc[i][j][0] + cb[j].x, a.act, a.alpha), act_u3(acc[i][j][1] + cb[j].y, a.act, a.alpha),
                                           act_u3(acc[i][j][2] + cb[j].z, a.act, a.alpha), act_u3(acc[i][j][3] + cb[j].w, a.act, a.alpha));
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    xv[i][j] = make_float4(ape::act_fast(acc[i][j][0] + cb[j].x, af), ape::act_fast(acc[i][j][1] + cb[j].y, af),
                                           ape::act_fast(acc[i][j][2] + cb[j].z, af), ape::act_fast(acc[i][j][3] + cb[j].w, af));
            }
        }
        int am[4];
        float pm[4];
        if (!(ABL(16))) {
            ape_seg::seg_head_groups<4>(xv, wreg, hbias, a.head_c, lane, a.head_dsm, am, pm);     // the four rows' chains interleaved
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) { am[i] = 0; pm[i] = xv[i][0].x + xv[i][1].y + xv[i][2].z + xv[i][3].w; }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gy = ty0 + wm * 4 + i, gx = tx0 + r16;
            if (kq == 0 && gy < a.H && gx < a.W) {
                const size_t m = ((size_t)tb * a.H + gy) * a.W + gx;
                a.label[m] = (uint8_t)am[i];
                a.score[m] = pm[i];
            }
        }
        zero_acc();
        if (STAMP) st_head += stamp() - th0;
    }
    if (STAMP && a.stamps && lane == 0) {
        unsigned long long* o = a.stamps + ((size_t)blockIdx.x * 4 + wave) * 4;
        o[0] = st_burst; o[1] = st_bar; o[2] = st_head; o[3] = (unsigned long long)my_tiles;
    }
#endif
}

}  // namespace

static int g_up3_dbg = 0;
static unsigned long long* g_up3_stamps = nullptr;
/* diagnostic: a device buffer of 256 x 4 x 4 u64 -> the next launches run the STAMP build (in-kernel s_memtime sums per matrix wave:
 * burst, wait + barrier, head, tiles); nullptr switches back.  The stamped build is slower: read its SHARES, not its length. */
extern "C" int ape_up3_seghead_stamps(void* device_buffer) { g_up3_stamps = (unsigned long long*)device_buffer; return APE_OK; }
/* bits: 1 = matrix waves at static priority 1, 32 = producer waves at priority 3 (results unchanged); 2 = ape_conv3x3_halo_seghead_bf16
 * dispatches to this kernel; 4 / 8 / 16 = timing-only ablations (wrong results) */
extern "C" int ape_up3_seghead_debug(int bits) { g_up3_dbg = bits; return APE_OK; }
extern "C" int ape_up3_seghead_debug_get(void) { return g_up3_dbg; }

extern "C" int ape_up3_seghead_ws_supported(const ape_conv_params* params, int nsplit)
{
    if (!params) return 0;
    const ape_conv_params& p = *params;
    if (nsplit != 3 || p.ups != 1 || p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != 1 || p.dil != 1) return 0;
    if (p.Cin != 64 || p.Cout != 64 || p.H != p.Ho || p.W != p.Wo || (p.H & 1) || (p.W & 1) || p.H < 2 || p.W < 2 || p.B < 0) return 0;
    if (p.ldx % 4 || p.xoff % 4 || p.xoff + p.Cin > p.ldx) return 0;
    if (p.act < APE_ACT_NONE || p.act > APE_ACT_SIGMOID) return 0;
    if ((long)p.B * (p.H / 2) * (p.W / 2) * p.ldx >= (1L << 31)) return 0;
    return 1;
}

extern "C" int ape_up3_seghead_ws_bf16(const float* x, const void* w_packed, const float* bias, const ape_conv_params* params, int nsplit,
                                       const float* head_w, const float* head_b, int C, uint8_t* label, float* score, int double_softmax,
                                       void* stream)
{
    if (!x || !w_packed || !params || !head_w || !label || !score || C < 1 || C > 16) return APE_EINVAL;
    if (!ape_up3_seghead_ws_supported(params, nsplit)) return APE_EINVAL;
    const ape_conv_params& p = *params;
    if (p.B == 0) return APE_OK;
    if (bias && ((size_t)bias % 16 || (p.bias_bstride % 4))) return APE_EINVAL;        // the epilogue reads the bias as float4
    Up3Args a;
    a.x = x; a.w = (const __bf16*)w_packed; a.bias = bias; a.head_w = head_w; a.head_b = head_b; a.label = label; a.score = score;
    a.B = p.B; a.H = p.H; a.W = p.W; a.ldx = p.ldx; a.xoff = p.xoff;
    const long K = 9L * p.Cin, Kp = (K + 7) / 8 * 8;
    a.Kp = (int)Kp;
    a.plane_stride = (long)p.Cout * Kp;
    a.act = p.act; a.alpha = p.alpha; a.bias_bstride = p.bias_bstride; a.head_c = C; a.head_dsm = double_softmax;
    a.tiles_x = ape::ceil_div(p.W, TS); a.tiles_y = ape::ceil_div(p.H, TS);
    a.dbg = g_up3_dbg;
    a.stamps = g_up3_stamps;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(up3_head_ws_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(up3_head_ws_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) {
            ape::set_last_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
            return APE_ELAUNCH;
        }
        attr_set = true;
    }
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu < 8) ncu = 256;
    }
    const int nwg = p.B * a.tiles_x * a.tiles_y;
    const int grid = nwg < ncu ? nwg : ncu;       // one persistent workgroup per CU (138 KB of LDS, eight waves)
    if (a.stamps) hipLaunchKernelGGL(up3_head_ws_kernel<true>, dim3(grid), dim3(512), LDS_BYTES, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(up3_head_ws_kernel<false>, dim3(grid), dim3(512), LDS_BYTES, (hipStream_t)stream, a);
    return ape::check_launch("ape_up3_seghead_ws_bf16");
}
