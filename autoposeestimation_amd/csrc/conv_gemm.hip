// GEMM-shaped convolutions (every 1x1 conv / Conv1d(k=1) / Linear of the path, and k x k convs the LDS-halo kernel does not
// take) on the gfx950 bf16 matrix cores with split-bf16 operands: the successor of conv_bf16_kernel for Cin % 32 == 0.
//
// Same contract as ape_conv2d_nhwc_bf16 (NHWC fp32 activations in HBM, packed hi/lo bf16 weight planes, fp32 accumulate,
// bias / residual / activation fused, XCD-aware bijective tile map).  What differs is the main loop:
//   * v_mfma_f32_16x16x32_bf16 (one 32-deep k-tile per instruction; the chip holds a higher clock on this shape than on
//     32x32x16 at equal LDS traffic), wave tile 128x64 (256x256 block, 8 waves) or 64x64 (128x128 / 256x64 block, 4 waves);
//   * two LDS stages of unpadded 64-B rows, 16-B chunks XOR-swizzled with -(row >> 2) & 3: the 16 rows x 4 chunks of one
//     ds_read_b128 fragment and the 4 rows x 4 chunks of one ds_write_b128 lane group each cover 16 distinct bank slots;
//   * register staging in the "write after the barrier, re-issue at once" order: tile t+1 sits in registers while tile t
//     is multiplied; at the top of iteration t it is split into hi/lo planes and written to the other stage (that stage
//     was last read in iteration t-1, before the barrier), and the global loads of tile t+2 are issued immediately, so a
//     load has a whole iteration of MFMAs (~3000 cycles) to land and the split + ds_write pass runs under the MFMAs
//     instead of in front of the barrier; ONE barrier per k-tile;
//   * the weights are the MFMA's ROW operand (D[channel][pixel]): a lane ends up with four consecutive output channels of
//     one pixel and the epilogue stores float4s straight from the registers -- no LDS staging pass, no barrier after the k-loop.
// Loads are unconditional (coordinates clamped into the tensor); out-of-image taps are zeroed when the registers are
// written to LDS, rows >= M and columns >= Cout are computed on clamped data and dropped by the epilogue.
#include "common.h"
#include "s32.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct GemmArgs {
    const float* x;
    const __bf16* w;   // plane 0 (hi) at w, plane 1 (lo) at w + plane_stride
    const float* bias;
    const float* res;
    float* y;
    ape_conv_params p;
    int M, Kp, m_tiles, n_tiles, nk;
    int out_fmt;       // APE_FMT_F32 | APE_FMT_S32 (pre-split output for the S32 consumers; needs ldy % 32 == 0 and the vector path)
    int dbg;           // ablation bits (timing experiments only, results are wrong): 1 no in-loop global loads, 2 no in-loop LDS restage, 4 no MFMAs
    long plane_stride;
    int nk_per;        // split-K (training tape, small M): blockIdx.y takes k-tiles [y * nk_per, ...) and writes RAW sums to y + blockIdx.y * split_stride; 0 = off
    long split_stride;
};

// NONE / RELU / PRELU as selects on loop-invariant scalars (a `switch` per element compiled to a cascade of scalar compares and branches
// per element: ~1.8 k scalar instructions in a 256-element epilogue); same values bit for bit (1 * v == v, also for -0 and NaN)
__device__ __forceinline__ float activate(float v, int act, float alpha)
{
    if (act == APE_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
    const float neg = act == APE_ACT_RELU ? 0.f : (act == APE_ACT_PRELU ? alpha : 1.f) * v;
    return v > 0.f ? v : neg;
}

__device__ __forceinline__ void split8(const f32x4 v0, const f32x4 v1, bf16x8& hi, bf16x8& lo)
{
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        hi[e] = (__bf16)v0[e];
        lo[e] = (__bf16)(v0[e] - (float)hi[e]);
        hi[4 + e] = (__bf16)v1[e];
        lo[4 + e] = (__bf16)(v1[e] - (float)hi[4 + e]);
    }
}

constexpr int BK = 32;

// element offset of 16-B chunk c of tile row `row` (rows of 32 bf16)
// (ds_read_b128 services the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...: the permutation that keeps the 16 lanes of
// a group on 16 distinct 16-B bank slots for the fragment map lane -> (row = lane & 15, chunk = lane >> 4) is chunk ^ (-(row >> 2) & 3))
__device__ __forceinline__ int lds_off(int row, int c) { return row * BK + ((c ^ ((0 - (row >> 2)) & 3)) << 3); }

// PURE: 1x1 / stride 1 / pad 0 (a row-major GEMM: row m of A starts at x + m * ldx + xoff)
//
// PP ("ping-pong", 8-wave blocks): the k-tile is cut into four segments -- read fragments + restage | MFMAs of the first row
// half | read the second half's A fragments | its MFMAs -- each closed by a barrier, and the waves of the second row half
// (wm = 1, the second wave of every SIMD) run ONE segment behind the first: while one wave of a SIMD issues its 48 MFMAs the
// other one does its LDS reads / split / ds_writes / global loads, instead of all eight waves queueing on the LDS port and
// then all eight on the matrix pipes.  The LDS stage protocol is unchanged (tile t+1 is written >= 2 barriers after the last
// read of tile t-1 and >= 2 barriers before the first read of tile t+1 by either group).
// SK: the split-K build (training tape only; the inference instantiations do not carry its code or registers)
template <int NSPLIT, int BM, int BN, int WM, int WN, bool PURE, bool PP, bool SK>
__device__ __forceinline__ void conv_gemm_body(const GemmArgs& a, const int orig)
{
    static_assert(!PP || WM == 2, "the ping-pong schedule pairs the two row halves of an 8-wave block");
    constexpr int NT = WM * WN * 64;
    constexpr int NPL = NSPLIT == 3 ? 2 : 1;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int A_IT = BM * 4 / NT;            // 8-float chunks per thread
    constexpr int B_IT = (BN * 4 + NT - 1) / NT; // 16-B chunks per thread per plane (the last one predicated when BN*4 % NT != 0)
    constexpr bool B_RAGGED = (BN * 4) % NT != 0;
    static_assert(BM * 4 % NT == 0 && A_IT >= 1 && B_IT >= 1, "staging shape");
    static_assert(WTM % 32 == 0 && WTN % 16 == 0 && BM % 64 == 0, "wave tile");

    constexpr size_t kStageElems = (size_t)NPL * (BM + BN) * BK;             // bf16 per pipeline stage
    __shared__ __attribute__((aligned(16))) char smem_raw[2 * kStageElems * 2];
    __bf16* const S0 = reinterpret_cast<__bf16*>(smem_raw);
    auto a_tile = [&](int stage, int pl) { return S0 + stage * kStageElems + (size_t)pl * BM * BK; };
    auto b_tile = [&](int stage, int pl) { return S0 + stage * kStageElems + (size_t)NPL * BM * BK + (size_t)pl * BN * BK; };

    const ape_conv_params& p = a.p;
    const int nwg = a.m_tiles * a.n_tiles;
    const int xcd = orig % 8, q = nwg / 8, r = nwg % 8;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
    const int n_tile = logical % a.n_tiles;
    const int m_tile = logical / a.n_tiles;
    const int m0 = m_tile * BM, n0 = n_tile * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int HoWo = p.Ho * p.Wo;

    // ---- staging coordinates -------------------------------------------------------------------------------------
    const int sc = tid & 3;          // 16-B chunk (8 elements) of the 32-deep k-tile
    const int srow = tid >> 2;       // + i * (NT / 4)
    unsigned a_off[A_IT];            // PURE: element offset of the row start;  else: b * H
    int a_iy0[PURE ? 1 : A_IT], a_ix0[PURE ? 1 : A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        int m = m0 + srow + i * (NT / 4);
        m = m < a.M ? m : a.M - 1;
        if (PURE) {
            a_off[i] = (unsigned)m * (unsigned)p.ldx + (unsigned)p.xoff + sc * 8;
        } else {
            const int b = m / HoWo, rem = m - b * HoWo;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            a_iy0[i] = oy * p.stride - p.pad;
            a_ix0[i] = ox * p.stride - p.pad;
            a_off[i] = (unsigned)(b * p.H);
        }
    }
    unsigned b_off[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        int n = n0 + srow + i * (NT / 4);
        n = n < p.Cout ? n : p.Cout - 1;      // (also keeps the predicated-off rows of a ragged last item in bounds)
        b_off[i] = (unsigned)n * (unsigned)a.Kp + sc * 8;
    }

    f32x4 areg[A_IT][2];
    u32x4 breg[NPL][B_IT];
    unsigned a_ok = 0;                       // bit i: item i of the tile held in registers lies inside the image
    int t_ci0 = 0, t_ky = 0, t_kx = 0;       // (chunk, tap) of the NEXT tile to load; K order = chunk outer, taps inner
    const int kt0 = SK ? (int)blockIdx.y * a.nk_per : 0;
    if (SK) {
        if (PURE) {
            t_ci0 = kt0 * BK;
        } else {
            const int taps = p.KH * p.KW, chunk = kt0 / taps, tap = kt0 - chunk * taps;
            t_ci0 = chunk * BK;
            t_ky = tap / p.KW;
            t_kx = tap - t_ky * p.KW;
        }
    }
    float* const yb = SK ? a.y + (size_t)blockIdx.y * a.split_stride : a.y;
    auto clampi = [](int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); };
    auto load_tiles = [&]() {
        unsigned kb;
        if (PURE) {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const f32x4* src = reinterpret_cast<const f32x4*>(a.x + a_off[i] + (unsigned)t_ci0);
                areg[i][0] = src[0];
                areg[i][1] = src[1];
            }
            kb = (unsigned)t_ci0;
            t_ci0 = t_ci0 + BK < p.Cin ? t_ci0 + BK : t_ci0;      // the loads run two tiles ahead of the loop: stay inside K
        } else {
            const int dy = t_ky * p.dil, dx = t_kx * p.dil;
            a_ok = 0;
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const int iy = a_iy0[i] + dy, ix = a_ix0[i] + dx;
                const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                const unsigned off = (unsigned)((((int)a_off[i] + clampi(iy, p.H - 1)) * p.W + clampi(ix, p.W - 1)) * p.ldx + p.xoff + t_ci0 + sc * 8);
                const f32x4* src = reinterpret_cast<const f32x4*>(a.x + off);
                areg[i][0] = src[0];
                areg[i][1] = src[1];
                a_ok |= ok ? (1u << i) : 0u;
            }
            kb = (unsigned)((t_ky * p.KW + t_kx) * p.Cin + t_ci0);
            if (++t_kx == p.KW) { t_kx = 0; if (++t_ky == p.KH) { t_ky = 0; t_ci0 = t_ci0 + BK < p.Cin ? t_ci0 + BK : t_ci0; } }
        }
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int i = 0; i < B_IT; ++i)
                breg[pl][i] = *reinterpret_cast<const u32x4*>(a.w + pl * a.plane_stride + b_off[i] + kb);
    };
    auto store_tiles = [&](int stage) {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const bool ok = PURE || ((a_ok >> i) & 1u);
            bf16x8 hi, lo;
            split8(ok ? areg[i][0] : z4, ok ? areg[i][1] : z4, hi, lo);
            const int o = lds_off(srow + i * (NT / 4), sc);
            *reinterpret_cast<bf16x8*>(a_tile(stage, 0) + o) = hi;
            if (NPL == 2) *reinterpret_cast<bf16x8*>(a_tile(stage, NPL - 1) + o) = lo;
        }
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int i = 0; i < B_IT; ++i)
                if (!B_RAGGED || i + 1 < B_IT || srow + i * (NT / 4) < BN)
                    *reinterpret_cast<u32x4*>(b_tile(stage, pl) + lds_off(srow + i * (NT / 4), sc)) = breg[pl][i];
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

    const int frow = lane & 15, fc = lane >> 4;
    const int a_row0 = wm * WTM + frow, b_row0 = wn * WTN + frow;
    bf16x8 bh[TN], bl[TN];
    auto read_b = [&](int stage) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int o = lds_off(b_row0 + j * 16, fc);
            bh[j] = *reinterpret_cast<const bf16x8*>(b_tile(stage, 0) + o);
            if (NPL == 2) bl[j] = *reinterpret_cast<const bf16x8*>(b_tile(stage, NPL - 1) + o);
        }
    };
    constexpr int TMH = TM / 2;
    bf16x8 ah[TMH], al[TMH];
    // fragment rows [i0, i0 + n) of the tile -> registers ah / al [r0, r0 + n)
    auto read_a_rows = [&](int stage, int i0, int r0, int n) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < n; ++i) {
            const int o = lds_off(a_row0 + (i0 + i) * 16, fc);
            ah[r0 + i] = *reinterpret_cast<const bf16x8*>(a_tile(stage, 0) + o);
            if (NPL == 2) al[r0 + i] = *reinterpret_cast<const bf16x8*>(a_tile(stage, NPL - 1) + o);
        }
    };
    auto read_a = [&](int stage, int i0) { read_a_rows(stage, i0, 0, TMH); };
    // the matrix instructions of accumulator rows [i0 + r0, i0 + r0 + n) from registers ah / al [r0, r0 + n)
    auto mfma_rows = [&](int i0, int r0, int n) __attribute__((always_inline)) {
#pragma unroll
        for (int i = r0; i < r0 + n; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (NPL == 2) {
                    acc[i0 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], al[i], acc[i0 + i][j], 0, 0, 0);
                    acc[i0 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], ah[i], acc[i0 + i][j], 0, 0, 0);
                }
                acc[i0 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah[i], acc[i0 + i][j], 0, 0, 0);
            }
    };
    auto mfma_half = [&](int i0) {
#pragma unroll
        for (int i = 0; i < TMH; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                // WEIGHTS as the row operand: D[row = channel 4 fc + e][col = pixel frow], i.e. a lane ends up with four CONSECUTIVE
                // output channels of one pixel -- a float4 it can store straight from its registers (same products, same k order)
                if (NPL == 2) {
                    acc[i0 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], al[i], acc[i0 + i][j], 0, 0, 0);
                    acc[i0 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], ah[i], acc[i0 + i][j], 0, 0, 0);
                }
                acc[i0 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah[i], acc[i0 + i][j], 0, 0, 0);

            }
    };

    const int nk = SK ? (a.nk - kt0 < a.nk_per ? a.nk - kt0 : a.nk_per) : a.nk;
    // 8-wave blocks: dbg bit 8 = static priority 1 for waves 4-7 (cf. conv_gemm_s32.hip)
    if (NT == 512 && (a.dbg & 8) && __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(1);
    load_tiles();
    store_tiles(0);
    load_tiles();
    __syncthreads();
    const bool behind = PP && __builtin_amdgcn_readfirstlane(wm) == 1;
    auto seg = [&]() {
        if (PP) {
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if (behind) __syncthreads();
#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        read_b(cur);
        read_a(cur, 0);
        // branch-free body (one basic block, so the scheduler can place the restage / loads between the MFMAs): the last two
        // iterations restage and load a clamped, never-read tile
        store_tiles(cur ^ 1);      // tile kt+1: registers -> the stage last read in iteration kt-1
        load_tiles();              // tile kt+2: lands during this iteration's MFMAs
        if constexpr (!PP && TMH % 2 == 0) {
            // the second half's fragment reads in two pieces, each issued BEHIND a quarter's matrix instructions into the registers that quarter has
            // just read: left as "all MFMAs of the first half, then all reads of the second" every k-tile stood still for one LDS round trip
            constexpr int Q = TMH / 2;
            mfma_rows(0, 0, Q);
            __builtin_amdgcn_sched_barrier(0);
            read_a_rows(cur, TMH, 0, Q);
            __builtin_amdgcn_sched_barrier(0);
            mfma_rows(0, Q, Q);
            __builtin_amdgcn_sched_barrier(0);
            read_a_rows(cur, TMH + Q, Q, Q);
            __builtin_amdgcn_sched_barrier(0);
            mfma_rows(TMH, 0, Q);
            mfma_rows(TMH, Q, Q);
        } else {
            seg();
            mfma_half(0);
            seg();
            read_a(cur, TMH);
            seg();
            mfma_half(TMH);
        }
        __syncthreads();
    }
    if (PP && !behind) __syncthreads();

    const bool vec_ok = (p.ldy % 4 == 0) && (p.yoff % 4 == 0) && (!a.res || (p.ldr % 4 == 0 && p.roff % 4 == 0));
    // ---- epilogue straight from the registers: lane (frow, fc) holds channels 16 j + 4 fc .. + 3 of pixel 16 i + frow for every
    // (i, j); the four lanes of a pixel write 64 contiguous bytes, two neighbouring j blocks complete the 128-B line in L2.  No
    // LDS pass, no barrier: the stores start behind the last MFMA (the staged form spent 8 barriers and ~10 us per 256 x 192 tile,
    // as long as the whole k-loop of a K = 256 layer).
    {
        const int nq = n0 + wn * WTN + fc * 4;          // + 16 j
        const ape::ActFast af = ape::act_fast_make(p.act, p.alpha);
        const bool sigm = p.act == APE_ACT_SIGMOID;
        float4 b4[TN];
        if (a.bias && !p.bias_bstride) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = nq + j * 16;
                b4[j] = make_float4(n < p.Cout ? a.bias[n] : 0.f, n + 1 < p.Cout ? a.bias[n + 1] : 0.f, n + 2 < p.Cout ? a.bias[n + 2] : 0.f,
                                    n + 3 < p.Cout ? a.bias[n + 3] : 0.f);
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * WTM + i * 16 + frow;
            const bool mok = m < a.M;
            const int mc = mok ? m : a.M - 1;
            float4 rr[TN];
            if (a.res && vec_ok) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    if (nq + j * 16 + 4 <= p.Cout) rr[j] = *reinterpret_cast<const float4*>(a.res + (size_t)mc * p.ldr + p.roff + nq + j * 16);
            }
            const float* brow = (a.bias && p.bias_bstride) ? a.bias + (size_t)(mc / HoWo) * p.bias_bstride : nullptr;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = nq + j * 16;
                if (!mok || n >= p.Cout) continue;
                float vv[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                const int nvalid = min(4, p.Cout - n);
                if (vec_ok && nvalid == 4) {
                    if (brow) { vv[0] += brow[n]; vv[1] += brow[n + 1]; vv[2] += brow[n + 2]; vv[3] += brow[n + 3]; }
                    else if (a.bias) { vv[0] += b4[j].x; vv[1] += b4[j].y; vv[2] += b4[j].z; vv[3] += b4[j].w; }
                    if (a.res) { vv[0] += rr[j].x; vv[1] += rr[j].y; vv[2] += rr[j].z; vv[3] += rr[j].w; }
                    const float4 o4 = sigm ? make_float4(activate(vv[0], p.act, p.alpha), activate(vv[1], p.act, p.alpha), activate(vv[2], p.act, p.alpha),
                                                         activate(vv[3], p.act, p.alpha))
                                           : make_float4(ape::act_fast(vv[0], af), ape::act_fast(vv[1], af), ape::act_fast(vv[2], af), ape::act_fast(vv[3], af));
                    if (a.out_fmt == APE_FMT_S32) ape::s32_store4(yb, (long)m, p.ldy / 4, (p.yoff + n) / 4, o4);
                    else *reinterpret_cast<float4*>(yb + (size_t)m * p.ldy + p.yoff + n) = o4;
                } else {
                    for (int k = 0; k < nvalid; ++k) {
                        float t = vv[k];
                        if (a.bias) t += (brow ? brow : a.bias)[n + k];
                        if (a.res) t += a.res[(size_t)m * p.ldr + p.roff + n + k];
                        yb[(size_t)m * p.ldy + p.yoff + n + k] = activate(t, p.act, p.alpha);
                    }
                }
            }
        }
    }
}

template <int NSPLIT, int BM, int BN, int WM, int WN, bool PURE, bool PP>
__global__ __launch_bounds__(WM * WN * 64, 2) void conv_gemm_kernel(const GemmArgs a)
{
    conv_gemm_body<NSPLIT, BM, BN, WM, WN, PURE, PP, false>(a, (int)blockIdx.x);
}

template <int NSPLIT, bool PURE>
__global__ __launch_bounds__(256, 2) void conv_gemm_splitk_kernel(const GemmArgs a)
{
    conv_gemm_body<NSPLIT, 128, 128, 2, 2, PURE, false, true>(a, (int)blockIdx.x);
}

// Up to four independent 1x1 problems in ONE launch (ape_conv_gemm_bf16_multi): workgroups first[i] .. first[i + 1] - 1 are problem i's tiles.
// For problems of a few tiles each, whose launches would run one after the other with most of the chip idle (the four PSP stage
// convolutions: 1, 2, 5 and 18 tiles at 64 frames, ~25 us each -- a k-loop's latency chain, not work).
constexpr int kMultiMax = 4;
struct GemmMultiArgs {
    GemmArgs a[kMultiMax];
    int first[kMultiMax + 1];       // (entries past the last problem hold the total)
};
template <int NSPLIT>
__global__ __launch_bounds__(256, 2) void conv_gemm_multi_kernel(const GemmMultiArgs m)
{
    const int blk = (int)blockIdx.x;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < kMultiMax; ++i) pi = blk >= m.first[i] ? i : pi;
    conv_gemm_body<NSPLIT, 128, 128, 2, 2, true, false, false>(m.a[pi], blk - m.first[pi]);
}

template <int NSPLIT, int BM, int BN, int WM, int WN, bool PP = false>
void launch(GemmArgs& a, bool pure, hipStream_t st)
{
    a.m_tiles = ape::ceil_div(a.M, BM);
    a.n_tiles = ape::ceil_div(a.p.Cout, BN);
    if (pure)
        hipLaunchKernelGGL((conv_gemm_kernel<NSPLIT, BM, BN, WM, WN, true, PP>), dim3(a.m_tiles * a.n_tiles), dim3(WM * WN * 64), 0, st, a);
    else
        hipLaunchKernelGGL((conv_gemm_kernel<NSPLIT, BM, BN, WM, WN, false, PP>), dim3(a.m_tiles * a.n_tiles), dim3(WM * WN * 64), 0, st, a);
}

template <int NSPLIT>
void launch_splitk(GemmArgs& a, bool pure, hipStream_t st)
{
    a.m_tiles = ape::ceil_div(a.M, 128);
    a.n_tiles = ape::ceil_div(a.p.Cout, 128);
    const int gy = ape::ceil_div(a.nk, a.nk_per);
    if (pure)
        hipLaunchKernelGGL((conv_gemm_splitk_kernel<NSPLIT, true>), dim3(a.m_tiles * a.n_tiles, gy), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((conv_gemm_splitk_kernel<NSPLIT, false>), dim3(a.m_tiles * a.n_tiles, gy), dim3(256), 0, st, a);
}

// split-K second pass: y = act(sum over the splits (fixed order) + bias + residual)
__global__ void splitk_finish_kernel(const float* __restrict__ ws, int splits, long M, int Cout, const float* __restrict__ bias, int bias_bstride,
                                     int HoWo, const float* __restrict__ res, int ldr, int roff, float* __restrict__ y, int ldy, int yoff, int act,
                                     float alpha)
{
    const long n = M * Cout;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long m = i / Cout;
        const int c = (int)(i - m * Cout);
        float v = ws[i];
        for (int k = 1; k < splits; ++k) v += ws[(long)k * n + i];
        if (bias) v += bias[(bias_bstride ? (m / HoWo) * bias_bstride : 0) + c];
        if (res) v += res[m * ldr + roff + c];
        y[m * ldy + yoff + c] = activate(v, act, alpha);
    }
}

// k-tiles per split for a small-M problem on the 128 x 128 block (0: do not split)
int splitk_tiles(const ape_conv_params& p, long M)
{
    const int nk = p.KH * p.KW * p.Cin / BK;
    const long tiles = (long)ape::ceil_div(M, 128) * ape::ceil_div(p.Cout, 128);
    if (p.Cout <= 64 || tiles >= 96 || nk < 8) return 0;
    long splits = ape::ceil_div(256L, tiles);
    splits = splits > nk / 4 ? nk / 4 : splits;
    splits = splits > 32 ? 32 : splits;
    if (splits < 2) return 0;
    return ape::ceil_div(nk, (int)splits);
}

bool supported(const ape_conv_params& p)
{
    if (p.B < 0 || p.H < 1 || p.W < 1 || p.Cin < 32 || p.Cin % 32 || p.Cout < 1 || p.KH < 1 || p.KW < 1 || p.stride < 1 ||
        p.dil < 1 || p.pad < 0 || p.Ho < 1 || p.Wo < 1 || p.ups != 0)
        return false;
    if (p.ldx % 4 || p.xoff % 4 || p.xoff + p.Cin > p.ldx || p.yoff + p.Cout > p.ldy) return false;
    if (p.act < APE_ACT_NONE || p.act > APE_ACT_SIGMOID) return false;
    const int ho = (p.H + 2 * p.pad - p.dil * (p.KH - 1) - 1) / p.stride + 1;
    const int wo = (p.W + 2 * p.pad - p.dil * (p.KW - 1) - 1) / p.stride + 1;
    if (ho != p.Ho || wo != p.Wo) return false;
    const long M = (long)p.B * p.Ho * p.Wo;
    // 32-bit element offsets into x and w
    if (M > (1L << 30) || (long)p.B * p.H * p.W * p.ldx >= (1L << 31) || (long)p.Cout * p.KH * p.KW * p.Cin >= (1L << 31)) return false;
    return true;
}

}  // namespace

extern "C" int ape_conv_gemm_supported(const ape_conv_params* params) { return params && supported(*params) ? 1 : 0; }

/* variant: 0 = chosen from the shape; 1 = 256x256 block (8 waves); 2 = 128x128 block (4 waves); 3 = 256x64 block (4 waves);
 * 4 = 256x192 block (8 waves; Cout = 576 of the up_2 channel mix is 3 x 192) */
extern "C" int ape_conv_gemm_bf16(const float* x, const void* w_packed, const float* bias, const float* residual, float* y,
                                  const ape_conv_params* params, int nsplit, int variant, void* stream)
{
    return ape_conv_gemm_bf16_fmt(x, w_packed, bias, residual, y, APE_FMT_F32, params, nsplit, variant, stream);
}

/* the same with the OUTPUT in either activation format (APE_FMT_S32: ldy % 32 == 0, Cout % 4 == 0, yoff % 4 == 0) */
extern "C" int ape_conv_gemm_bf16_fmt(const float* x, const void* w_packed, const float* bias, const float* residual, void* y_, int out_fmt,
                                      const ape_conv_params* params, int nsplit, int variant, void* stream)
{
    float* y = (float*)y_;
    if (out_fmt != APE_FMT_F32 && out_fmt != APE_FMT_S32) return APE_EINVAL;
    if (out_fmt == APE_FMT_S32 && params && (params->ldy % 32 || params->Cout % 4 || params->yoff % 4 || (residual && (params->ldr % 4 || params->roff % 4))))
        return APE_EINVAL;
    if (!x || !w_packed || !y || !params || (nsplit != 1 && nsplit != 3) || variant < 0 || (variant & 15) > 6) return APE_EINVAL;
    const ape_conv_params& p = *params;
    if (!supported(p)) return APE_EINVAL;
    if (residual && p.roff + p.Cout > p.ldr) return APE_EINVAL;
    const long M = (long)p.B * p.Ho * p.Wo;
    if (M == 0) return APE_OK;
    GemmArgs a;
    a.x = x; a.w = (const __bf16*)w_packed; a.bias = bias; a.res = residual; a.y = y; a.p = p;
    a.M = (int)M;
    const int K = p.KH * p.KW * p.Cin;
    a.Kp = (K + 7) / 8 * 8;
    a.nk = K / BK;
    a.plane_stride = (long)p.Cout * a.Kp;
    a.dbg = variant >> 4;
    a.out_fmt = out_fmt;
    a.nk_per = 0;
    a.split_stride = 0;
    variant &= 15;
    const bool pure = p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0;
    hipStream_t st = (hipStream_t)stream;
    if (variant == 0) {
        const int waste256 = ape::ceil_div(p.Cout, 256) * 256 - p.Cout;
        const int waste192 = ape::ceil_div(p.Cout, 192) * 192 - p.Cout;
        if (p.Cout <= 64) variant = 3;
        // the 256-row blocks need about a chip's worth of workgroups (256 CUs); below that the 128x128 block fills it better
        else if (p.Cout >= 256 && waste256 * 8 <= p.Cout && K >= 256 && (long)ape::ceil_div(M, 256) * ape::ceil_div(p.Cout, 256) >= 192) variant = 1;
        else if (p.Cout >= 192 && waste192 * 8 <= p.Cout && K >= 256 && (long)ape::ceil_div(M, 256) * ape::ceil_div(p.Cout, 192) >= 192) variant = 4;
        else variant = 2;
    }
    if (variant == 1) {
        if (nsplit == 3) launch<3, 256, 256, 2, 4>(a, pure, st); else launch<1, 256, 256, 2, 4>(a, pure, st);
    } else if (variant == 2) {
        if (nsplit == 3) launch<3, 128, 128, 2, 2>(a, pure, st); else launch<1, 128, 128, 2, 2>(a, pure, st);
    } else if (variant == 3) {
        if (nsplit == 3) launch<3, 256, 64, 4, 1>(a, pure, st); else launch<1, 256, 64, 4, 1>(a, pure, st);
    } else if (variant == 4) {
        if (nsplit == 3) launch<3, 256, 192, 2, 4>(a, pure, st); else launch<1, 256, 192, 2, 4>(a, pure, st);
    } else if (variant == 5) {
        if (nsplit == 3) launch<3, 256, 256, 2, 4, true>(a, pure, st); else launch<1, 256, 256, 2, 4, true>(a, pure, st);
    } else {
        if (nsplit == 3) launch<3, 128, 192, 2, 2>(a, pure, st); else launch<1, 128, 192, 2, 2>(a, pure, st);
    }
    return ape::check_launch("ape_conv_gemm_bf16");
}

/* n <= 4 independent 1x1 / stride-1 convolutions (fp32 in, fp32 out, no residual) in ONE launch of the 128 x 128 block: problem i multiplies
 * x[i] by w_packed[i] under params[i].  Every output element is the very sum ape_conv_gemm_bf16 forms (same k order in every block shape). */
extern "C" int ape_conv_gemm_bf16_multi(int n, const float* const* x, const void* const* w_packed, const float* const* bias, float* const* y,
                                        const ape_conv_params* params, int nsplit, void* stream)
{
    if (n < 1 || n > kMultiMax || !x || !w_packed || !y || !params || (nsplit != 1 && nsplit != 3)) return APE_EINVAL;
    GemmMultiArgs m;
    int total = 0;
    for (int i = 0; i < kMultiMax; ++i) {
        m.first[i] = total;
        if (i >= n) { m.a[i] = m.a[0]; continue; }
        const ape_conv_params& p = params[i];
        if (!x[i] || !w_packed[i] || !y[i] || !supported(p) || p.KH != 1 || p.KW != 1 || p.stride != 1 || p.pad != 0) return APE_EINVAL;
        const long M = (long)p.B * p.Ho * p.Wo;
        GemmArgs& a = m.a[i];
        a.x = x[i]; a.w = (const __bf16*)w_packed[i]; a.bias = bias ? bias[i] : nullptr; a.res = nullptr; a.y = y[i]; a.p = p;
        a.M = (int)M;
        a.Kp = (p.Cin + 7) / 8 * 8;
        a.nk = p.Cin / BK;
        a.plane_stride = (long)p.Cout * a.Kp;
        a.dbg = 0;
        a.out_fmt = APE_FMT_F32;
        a.nk_per = 0;
        a.split_stride = 0;
        a.m_tiles = ape::ceil_div(a.M, 128);
        a.n_tiles = ape::ceil_div(p.Cout, 128);
        total += a.m_tiles * a.n_tiles;       // (an empty problem, M = 0, has no tiles)
    }
    m.first[kMultiMax] = total;
    if (total == 0) return APE_OK;
    hipStream_t st = (hipStream_t)stream;
    if (nsplit == 3) hipLaunchKernelGGL((conv_gemm_multi_kernel<3>), dim3(total), dim3(256), 0, st, m);
    else hipLaunchKernelGGL((conv_gemm_multi_kernel<1>), dim3(total), dim3(256), 0, st, m);
    return ape::check_launch("ape_conv_gemm_bf16_multi");
}

/* Split-K form for the training tape's batch-1 layers (a 20 x 20 map is 4 row tiles: 4..16 workgroups walking K = 4608 alone took 89 us):
 * the k-tiles are dealt to `splits` workgroups per output tile, raw sums go to the workspace and a second pass adds them in a fixed
 * order with bias / residual / activation.  Same products, another summation order than ape_conv_gemm_bf16 -- the inference path
 * never takes it.  ape_conv_gemm_splitk_workspace_bytes returns 0 when the shape is not worth splitting (use ape_conv_gemm_bf16).
 * (A single-launch form -- the last split of a tile to arrive reduces -- was measured and dropped: its device-scope fences write back
 * and invalidate whole XCD L2s per workgroup, 7.8 -> 18.8 ms per training step.) */
extern "C" size_t ape_conv_gemm_splitk_workspace_bytes(const ape_conv_params* params)
{
    if (!params || !supported(*params)) return 0;
    const ape_conv_params& p = *params;
    const long M = (long)p.B * p.Ho * p.Wo;
    const int per = splitk_tiles(p, M);
    if (!per) return 0;
    const int nk = p.KH * p.KW * p.Cin / BK;
    return (size_t)ape::ceil_div(nk, per) * M * p.Cout * sizeof(float);
}

extern "C" int ape_conv_gemm_bf16_splitk(const float* x, const void* w_packed, const float* bias, const float* residual, float* y,
                                         const ape_conv_params* params, int nsplit, void* workspace, size_t workspace_bytes, void* stream)
{
    if (!x || !w_packed || !y || !params || !workspace || (nsplit != 1 && nsplit != 3)) return APE_EINVAL;
    const ape_conv_params& p = *params;
    if (!supported(p)) return APE_EINVAL;
    if (residual && p.roff + p.Cout > p.ldr) return APE_EINVAL;
    const long M = (long)p.B * p.Ho * p.Wo;
    const size_t need = ape_conv_gemm_splitk_workspace_bytes(params);
    if (!need) return APE_EINVAL;
    if (workspace_bytes < need) return APE_EWORKSPACE;
    GemmArgs a;
    a.x = x; a.w = (const __bf16*)w_packed; a.bias = nullptr; a.res = nullptr; a.y = (float*)workspace; a.p = p;
    a.p.ldy = p.Cout; a.p.yoff = 0; a.p.act = APE_ACT_NONE; a.p.bias_bstride = 0; a.p.ldr = 0; a.p.roff = 0;
    a.M = (int)M;
    const int K = p.KH * p.KW * p.Cin;
    a.Kp = (K + 7) / 8 * 8;
    a.nk = K / BK;
    a.plane_stride = (long)p.Cout * a.Kp;
    a.dbg = 0;
    a.out_fmt = APE_FMT_F32;
    a.nk_per = splitk_tiles(p, M);
    a.split_stride = M * p.Cout;
    const int splits = ape::ceil_div(a.nk, a.nk_per);
    const bool pure = p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0;
    hipStream_t st = (hipStream_t)stream;
    if (nsplit == 3) launch_splitk<3>(a, pure, st); else launch_splitk<1>(a, pure, st);
    long g = (M * p.Cout + 255) / 256;
    g = g > 4096 ? 4096 : g;
    hipLaunchKernelGGL(splitk_finish_kernel, dim3((int)g), dim3(256), 0, st, (const float*)workspace, splits, M, p.Cout, bias, p.bias_bstride,
                       p.Ho * p.Wo, residual, p.ldr, p.roff, y, p.ldy, p.yoff, p.act, p.alpha);
    return ape::check_launch("ape_conv_gemm_bf16_splitk");
}
