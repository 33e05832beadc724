// Implicit-GEMM convolution / 1x1 / Linear on the gfx950 bf16 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulate),
// same GEMM view, NHWC fp32 activations in HBM, fused epilogue and XCD-aware tile map as conv_f32.hip, with the operands
// handed to the MFMA as bf16 in one of two ways:
//
//   NSPLIT = 3  "split-bf16": x = x_hi + x_lo with x_hi = bf16(x), x_lo = bf16(x - x_hi) (16 significand bits), and
//               acc += a_lo*b_hi + a_hi*b_lo + a_hi*b_hi   -- 3 MFMAs per product, relative error ~2^-16 per product
//               instead of 2^-8; measured end-to-end pose error vs the fp32 reference ~1e-5 (tolerance 1e-4), at an
//               effective dense peak of 2.5 PF / 3 = 833 TFLOP/s versus 157 TFLOP/s for the exact-fp32 MFMA.
//   NSPLIT = 1  plain bf16 operands (segmentation only: its consumer is an arg-max).
//
// Activations stay fp32 in HBM (the pooling / resize / gather kernels are shared with the fp32 path); the A tile is split
// into hi/lo planes on the fly while it is staged  HBM -> registers -> LDS  (3 VALU ops per element, hidden under the
// 24 MFMAs of a k-tile).  Weights are split once at load time (ape_pack_weights_bf16) into two bf16 planes [Cout][Kp].
//
// LDS: four bf16 tiles (A_hi, A_lo, B_hi, B_lo), K-contiguous rows of 64 elements padded to 72 (144 B = 9 x 16-B slots, odd
// => the 16 rows of a ds_read_b128 lane group hit 16 distinct bank slots); one ds_read_b128 is exactly one MFMA fragment
// (lane l: row l&31, k = 8*(l>>5) + 0..7 of the 16-deep k-step).
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// tile shapes: <BM=128, BN<=128, BK=64, 4 waves> for everything, <BM=256, BN=256, BK=32, 8 waves> for big GEMM-like layers
// (Cout >= 256, M large): a 256x256 tile fetches 16 KB of operands per algorithmic MFLOP instead of 30 KB -- these layers
// (1x1 convs with K = 256..1024) sit on the ~9 TB/s L2->CU operand roof, not on the MFMA roof.

struct ConvArgsB {
    const float* x;
    const __bf16* w;   // plane 0 (hi) at w, plane 1 (lo) at w + plane_stride
    const float* bias;
    const float* res;
    float* y;
    ape_conv_params p;
    int M, K, Kp, m_tiles, n_tiles;
    long plane_stride;
};

// NONE / RELU / PRELU as selects on loop-invariant scalars (a `switch` per element compiled to a cascade of scalar compares and branches
// per element: ~1.8 k scalar instructions in a 256-element epilogue); same values bit for bit (1 * v == v, also for -0 and NaN)
__device__ __forceinline__ float activate(float v, int act, float alpha)
{
    if (act == APE_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
    const float neg = act == APE_ACT_RELU ? 0.f : (act == APE_ACT_PRELU ? alpha : 1.f) * v;
    return v > 0.f ? v : neg;
}

__device__ __forceinline__ void split4(const float4 v, bf16x4& hi, bf16x4& lo)
{
    hi[0] = (__bf16)v.x; hi[1] = (__bf16)v.y; hi[2] = (__bf16)v.z; hi[3] = (__bf16)v.w;
    lo[0] = (__bf16)(v.x - (float)hi[0]); lo[1] = (__bf16)(v.y - (float)hi[1]);
    lo[2] = (__bf16)(v.z - (float)hi[2]); lo[3] = (__bf16)(v.w - (float)hi[3]);
}

// DB = true: BK = 32, two LDS stages of unpadded 64-B rows whose four 16-B chunks are XOR-swizzled with (row >> 1) & 3
// (conflict-free ds_read_b128 fragments and ds_write_b128 staging), ONE barrier per k-tile: the next tile is written into
// the other stage right after this tile's MFMAs are issued, so a wave's split + LDS-store phase overlaps the other waves'
// MFMA phase instead of sitting between two barriers.
template <int NSPLIT, int BM, int BN, int WM, int WN, int BK, bool DB>
__global__ __launch_bounds__(WM * WN * 64) void conv_bf16_kernel(const ConvArgsB a)
{
    static_assert(!DB || BK == 32, "the double-buffered layout is built for 32-deep k-tiles");
    constexpr int NT = WM * WN * 64;
    constexpr int LD = DB ? BK : BK + 8;   // bf16 per LDS row: padded 144 B (BK = 64) / 80 B (BK = 32), or unpadded swizzled 64 B
    constexpr int NSTAGE = DB ? 2 : 1;
    constexpr int TM = BM / WM / 32;
    constexpr int TN = BN / WN / 32;
    constexpr int NPL = NSPLIT == 3 ? 2 : 1;           // operand planes (hi [, lo])
    constexpr int CPR = BK / 8;                        // 16-B (8 x bf16) chunks per tile row
    constexpr int RPP = NT / CPR;                      // rows covered per pass of the 256 threads (32)
    constexpr int A_ROWS = BM / RPP;                   // rows per thread
    static_assert(BM % RPP == 0 && BN % RPP == 0 && BM % 64 == 0, "tile/staging shape");
    constexpr int B_ITEMS = BN / RPP;                  // 16-B chunks per thread per plane
    static_assert(B_ITEMS >= 1, "BN");

    // ONE LDS array: operand tiles during the K loop, fp32 staging rows during the epilogue
    constexpr int ELD = BN + 4;                         // floats per staged epilogue row
    constexpr size_t kOperandBytes = (size_t)NSTAGE * NPL * (BM + BN) * LD * 2, kStageBytes = (size_t)64 * ELD * 4;
    __shared__ __attribute__((aligned(16))) char smem_raw[kOperandBytes > kStageBytes ? kOperandBytes : kStageBytes];
    __bf16* const As0 = reinterpret_cast<__bf16*>(smem_raw);                 // [NSTAGE][NPL][BM * LD]
    __bf16* const Bs0 = As0 + NSTAGE * NPL * BM * LD;                        // [NSTAGE][NPL][BN * LD]
    auto a_tile = [&](int stage, int pl) { return As0 + ((size_t)stage * NPL + pl) * BM * LD; };
    auto b_tile = [&](int stage, int pl) { return Bs0 + ((size_t)stage * NPL + pl) * BN * LD; };
    // element offset of the 16-B chunk `c16` of tile row `row`
    auto lds_off = [](int row, int c16) { return DB ? row * LD + ((c16 ^ ((row >> 1) & 3)) << 3) : row * LD + c16 * 8; };

    const ape_conv_params& p = a.p;
    const int nwg = a.m_tiles * a.n_tiles;
    const int orig = blockIdx.x;
    const int xcd = orig % 8, q = nwg / 8, r = nwg % 8;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
    const int n_tile = logical % a.n_tiles;
    const int m_tile = logical / a.n_tiles;
    const int m0 = m_tile * BM, n0 = n_tile * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    const int k8 = tid % CPR;    // which 8-element chunk of the k-tile
    const int srow = tid / CPR;  // 0..RPP-1
    const int HoWo = p.Ho * p.Wo;
    int a_base[A_ROWS], a_iy0[A_ROWS], a_ix0[A_ROWS];
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
        const int m = m0 + srow + RPP * i;
        if (m < a.M) {
            const int b = m / HoWo, rem = m - b * HoWo;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            a_iy0[i] = oy * p.stride - p.pad;
            a_ix0[i] = ox * p.stride - p.pad;
            a_base[i] = b * p.H;
        } else {
            a_iy0[i] = -(1 << 28);
            a_ix0[i] = 0;
            a_base[i] = 0;
        }
    }

    float4 areg[A_ROWS][2];
    uint4 breg[NPL][B_ITEMS];
    // Every global load below is UNCONDITIONAL (coordinates clamped into the tensor) and the zero-fill of halo / tail elements
    // happens when the registers are written to LDS, from predicate bits saved here.  A `cond ? load : 0` makes hipcc wait
    // vmcnt(0) right behind the load (the select needs the value), which exposed the full A-tile latency on every k-tile
    // instead of hiding it under the MFMAs (cdna_hip_programming.md section 5, ".s-level traps", item (c)).
    unsigned a_okmask = 0, b_okmask = 0;
    const bool tap_uniform = (p.Cin % BK) == 0;
    int t_ci0 = 0, t_kx = 0, t_ky = 0;      // tap state of the NEXT tile to load (wave-uniform), K order = (chunk, ky, kx)
    auto clampi = [](int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); };
    auto load_tiles = [&](int kt) {
        int t_kb = 0;
        a_okmask = 0;
        if (tap_uniform) {
            // Cin is a multiple of BK: a whole k-tile lies inside ONE filter tap and (ky, kx, ci0) advance incrementally in
            // scalar registers -- no per-lane integer division.  The KH*KW consecutive k-tiles of one channel chunk re-read the
            // same 128-B lines of x shifted by one pixel / row, so they hit L1/L2.
            const int dy = t_ky * p.dil, dx = t_kx * p.dil;
            t_kb = (t_ky * p.KW + t_kx) * p.Cin + t_ci0;      // column of w[Cout][KH][KW][Cin] for this (tap, chunk)
#pragma unroll
            for (int i = 0; i < A_ROWS; ++i) {
                const int iy = a_iy0[i] + dy, ix = a_ix0[i] + dx;
                const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                const unsigned off = (unsigned)(((a_base[i] + clampi(iy, p.H - 1)) * p.W + clampi(ix, p.W - 1)) * p.ldx + p.xoff + t_ci0 + k8 * 8);
                const float4* src = reinterpret_cast<const float4*>(a.x + off);
                areg[i][0] = src[0];
                areg[i][1] = src[1];
                a_okmask |= ok ? (3u << (2 * i)) : 0u;
            }
            if (++t_kx == p.KW) { t_kx = 0; if (++t_ky == p.KH) { t_ky = 0; t_ci0 += BK; } }
        } else {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {   // the two float4 halves of the 8-element chunk may sit in different taps (Cin = 4)
                const int k = kt * BK + k8 * 8 + hf * 4;
                const bool kin = k < a.K;
                const int kc = kin ? k : 0;
                const int tap = kc / p.Cin, ci = kc - tap * p.Cin;
                const int ky = tap / p.KW, kx = tap - ky * p.KW;
                const int dy = ky * p.dil, dx = kx * p.dil;
#pragma unroll
                for (int i = 0; i < A_ROWS; ++i) {
                    const int iy = a_iy0[i] + dy, ix = a_ix0[i] + dx;
                    const bool ok = kin && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                    const unsigned off = (unsigned)(((a_base[i] + clampi(iy, p.H - 1)) * p.W + clampi(ix, p.W - 1)) * p.ldx + p.xoff + ci);
                    areg[i][hf] = *reinterpret_cast<const float4*>(a.x + off);
                    a_okmask |= ok ? (1u << (2 * i + hf)) : 0u;
                }
            }
        }
        const int kb = (tap_uniform ? t_kb : kt * BK) + k8 * 8;
        const bool kb_ok = kb < a.Kp;
        const unsigned kbc = (unsigned)(kb_ok ? kb : 0);
        b_okmask = 0;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int i = 0; i < B_ITEMS; ++i) {
                const int n = n0 + srow + RPP * i;
                const bool ok = kb_ok && n < p.Cout;
                breg[pl][i] = *reinterpret_cast<const uint4*>(a.w + pl * a.plane_stride + (unsigned)((n < p.Cout ? n : p.Cout - 1) * a.Kp) + kbc);
                b_okmask |= ok ? (1u << i) : 0u;
            }
    };
    auto store_tiles = [&](int stage) {
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) {
            bf16x4 h0, l0, h1, l1;
            split4((a_okmask >> (2 * i)) & 1u ? areg[i][0] : z4, h0, l0);
            split4((a_okmask >> (2 * i + 1)) & 1u ? areg[i][1] : z4, h1, l1);
            bf16x8 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) { hi[e] = h0[e]; hi[4 + e] = h1[e]; lo[e] = l0[e]; lo[4 + e] = l1[e]; }
            *reinterpret_cast<bf16x8*>(a_tile(stage, 0) + lds_off(srow + RPP * i, k8)) = hi;
            if (NPL == 2) *reinterpret_cast<bf16x8*>(a_tile(stage, NPL - 1) + lds_off(srow + RPP * i, k8)) = lo;
        }
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int i = 0; i < B_ITEMS; ++i)
                *reinterpret_cast<uint4*>(b_tile(stage, pl) + lds_off(srow + RPP * i, k8)) =
                    (b_okmask >> i) & 1u ? breg[pl][i] : make_uint4(0u, 0u, 0u, 0u);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = (a.K + BK - 1) / BK;
    const int frow = lane & 31, fh = lane >> 5;
    const int a_row0 = wm * (BM / WM) + frow, b_row0 = wn * (BN / WN) + frow;
    auto compute = [&](int stage) {
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int o = lds_off(a_row0 + i * 32, 2 * s + fh);
                ah[i] = *reinterpret_cast<const bf16x8*>(a_tile(stage, 0) + o);
                if (NPL == 2) al[i] = *reinterpret_cast<const bf16x8*>(a_tile(stage, NPL - 1) + o);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int o = lds_off(b_row0 + j * 32, 2 * s + fh);
                bh[j] = *reinterpret_cast<const bf16x8*>(b_tile(stage, 0) + o);
                if (NPL == 2) bl[j] = *reinterpret_cast<const bf16x8*>(b_tile(stage, NPL - 1) + o);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (NPL == 2) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
    };

    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    if (DB) {
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) load_tiles(kt + 1);
            compute(kt & 1);
            if (kt + 1 < nk) store_tiles((kt + 1) & 1);
            __syncthreads();
        }
    } else {
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) load_tiles(kt + 1);
            compute(0);
            __syncthreads();
            if (kt + 1 < nk) {
                store_tiles(0);
                __syncthreads();
            }
        }
    }

    // ---- epilogue: accumulators -> LDS (fp32, 64 rows at a time, reusing the operand tiles) -> 16-byte coalesced stores with
    // bias / residual / activation applied on float4s.  (Per-lane dword stores straight from the MFMA layout issued 64 store
    // and up to 64 residual-load instructions per lane and held 1x1 layers with small K at ~1 TB/s of output bandwidth.)
    float* stage = reinterpret_cast<float*>(smem_raw);
    const bool vec_ok = (p.ldy % 4 == 0) && (p.yoff % 4 == 0) && (!a.res || (p.ldr % 4 == 0 && p.roff % 4 == 0));
#pragma unroll 1
    for (int half = 0; half < BM / 64; ++half) {
        __syncthreads();
        // waves whose rows fall in this half dump their tiles
        {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int rbase = wm * (BM / WM) + i * 32;
                if (rbase / 64 != half) continue;
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = (e & 3) + 8 * (e >> 2) + 4 * fh;
                        stage[(rbase - half * 64 + row) * ELD + wn * (BN / WN) + j * 32 + frow] = acc[i][j][e];
                    }
            }
        }
        __syncthreads();
        for (int it = tid; it < 64 * (BN / 4); it += NT) {
            const int row = it / (BN / 4), c4 = it - row * (BN / 4);
            const int m = m0 + half * 64 + row, n = n0 + c4 * 4;
            if (m >= a.M || n >= p.Cout) continue;
            float4 v = *reinterpret_cast<const float4*>(&stage[row * ELD + c4 * 4]);
            float vv[4] = {v.x, v.y, v.z, v.w};
            const int nvalid = min(4, p.Cout - n);
            const float* bptr = a.bias ? a.bias + (p.bias_bstride ? (size_t)(m / HoWo) * p.bias_bstride : 0) + n : nullptr;
            if (vec_ok && nvalid == 4) {
                if (bptr) { vv[0] += bptr[0]; vv[1] += bptr[1]; vv[2] += bptr[2]; vv[3] += bptr[3]; }
                if (a.res) {
                    const float4 r = *reinterpret_cast<const float4*>(a.res + (size_t)m * p.ldr + p.roff + n);
                    vv[0] += r.x; vv[1] += r.y; vv[2] += r.z; vv[3] += r.w;
                }
                float4 o = make_float4(activate(vv[0], p.act, p.alpha), activate(vv[1], p.act, p.alpha),
                                       activate(vv[2], p.act, p.alpha), activate(vv[3], p.act, p.alpha));
                *reinterpret_cast<float4*>(a.y + (size_t)m * p.ldy + p.yoff + n) = o;
            } else {
                for (int k = 0; k < nvalid; ++k) {
                    float t = vv[k];
                    if (bptr) t += bptr[k];
                    if (a.res) t += a.res[(size_t)m * p.ldr + p.roff + n + k];
                    a.y[(size_t)m * p.ldy + p.yoff + n + k] = activate(t, p.act, p.alpha);
                }
            }
        }
    }
}

template <int NSPLIT, int BM, int BN, int WM, int WN, int BK, bool DB = false>
void launch(const ConvArgsB& a, hipStream_t st)
{
    hipLaunchKernelGGL((conv_bf16_kernel<NSPLIT, BM, BN, WM, WN, BK, DB>), dim3(a.m_tiles * a.n_tiles), dim3(WM * WN * 64), 0, st, a);
}

__global__ void pack_weights_kernel(const float* __restrict__ w, __bf16* __restrict__ out, int cout, int K, int Kp)
{
    const long total = (long)cout * Kp;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = i % Kp;
        const long n = i / Kp;
        const float v = k < K ? w[n * K + k] : 0.f;
        const __bf16 hi = (__bf16)v;
        out[i] = hi;
        out[total + i] = (__bf16)(v - (float)hi);
    }
}

}  // namespace

extern "C" long ape_packed_weights_bf16_elems(int cout, int K)
{
    const long Kp = (K + 7) / 8 * 8;
    return 2L * cout * Kp;
}

/* w[cout][K] f32 -> out: bf16 hi plane [cout][Kp] followed by lo plane [cout][Kp], Kp = K rounded up to 8, zero padded */
extern "C" int ape_pack_weights_bf16(const float* w, void* out, int cout, int K, void* stream)
{
    if (!w || !out || cout < 1 || K < 1) return APE_EINVAL;
    const int Kp = (K + 7) / 8 * 8;
    const long total = (long)cout * Kp;
    long g = (total + 255) / 256;
    g = g > 4096 ? 4096 : g;
    hipLaunchKernelGGL(pack_weights_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, w, (__bf16*)out, cout, K, Kp);
    return ape::check_launch("ape_pack_weights_bf16");
}

extern "C" int ape_conv2d_nhwc_bf16(const float* x, const void* w_packed, const float* bias, const float* residual, float* y,
                                    const ape_conv_params* params, int nsplit, void* stream)
{
    if (!x || !w_packed || !y || !params || (nsplit != 1 && nsplit != 3)) return APE_EINVAL;
    const ape_conv_params& p = *params;
    if (p.B < 0 || p.H < 1 || p.W < 1 || p.Cin < 4 || p.Cout < 1 || p.KH < 1 || p.KW < 1 || p.stride < 1 || p.dil < 1 ||
        p.pad < 0 || p.Ho < 1 || p.Wo < 1)
        return APE_EINVAL;
    if (p.Cin % 4 || p.ldx % 4 || p.xoff % 4 || p.xoff + p.Cin > p.ldx || p.yoff + p.Cout > p.ldy) return APE_EINVAL;
    if (residual && p.roff + p.Cout > p.ldr) return APE_EINVAL;
    if (p.act < APE_ACT_NONE || p.act > APE_ACT_SIGMOID || p.ups != 0) return APE_EINVAL;
    const int ho = (p.H + 2 * p.pad - p.dil * (p.KH - 1) - 1) / p.stride + 1;
    const int wo = (p.W + 2 * p.pad - p.dil * (p.KW - 1) - 1) / p.stride + 1;
    if (ho != p.Ho || wo != p.Wo) return APE_EINVAL;
    const long M = (long)p.B * p.Ho * p.Wo;
    if (M == 0) return APE_OK;
    // the kernel addresses x and w with 32-bit ELEMENT offsets
    if (M > (1L << 30) || (long)p.B * p.H * p.W * p.ldx >= (1L << 31) || (long)p.Cout * ((p.KH * p.KW * p.Cin + 7) / 8 * 8) >= (1L << 31))
        return APE_EINVAL;

    ConvArgsB a;
    a.x = x; a.w = (const __bf16*)w_packed; a.bias = bias; a.res = residual; a.y = y; a.p = p;
    a.M = (int)M;
    a.K = p.KH * p.KW * p.Cin;
    a.Kp = (a.K + 7) / 8 * 8;
    a.plane_stride = (long)p.Cout * a.Kp;
    hipStream_t st = (hipStream_t)stream;
    // A/B on MI355X (tools/microbench_generic.py): the double-buffered layout pays for the 256x256 tile (+1..8 %) and for the
    // Cout <= 64 tile (stem: +10 %); the 128x128 tile is bound by L2 -> CU operand traffic (30 KB per algorithmic MFLOP,
    // ~6.7 TB/s at 220 TFLOP/s) either way and keeps the 64-deep single-buffered k-tile (fewer barriers per flop).
    // 256x256 tiles when they are mostly full: Cout a (near) multiple of 256 and enough rows to fill the chip
    const int waste256 = ape::ceil_div(p.Cout, 256) * 256 - p.Cout;
    if (p.Cout >= 256 && waste256 * 8 <= p.Cout && M >= 256L * 256) {
        a.m_tiles = ape::ceil_div(M, 256);
        a.n_tiles = ape::ceil_div(p.Cout, 256);
        if (nsplit == 3) launch<3, 256, 256, 4, 2, 32, true>(a, st); else launch<1, 256, 256, 4, 2, 32, true>(a, st);
    } else if (p.Cout > 64) {
        a.m_tiles = ape::ceil_div(M, 128);
        a.n_tiles = ape::ceil_div(p.Cout, 128);
        if (nsplit == 3) launch<3, 128, 128, 2, 2, 64>(a, st); else launch<1, 128, 128, 2, 2, 64>(a, st);
    } else {
        a.m_tiles = ape::ceil_div(M, 128);
        a.n_tiles = 1;
        if (nsplit == 3) launch<3, 128, 64, 4, 1, 32, true>(a, st); else launch<1, 128, 64, 4, 1, 32, true>(a, st);
    }
    return ape::check_launch("ape_conv2d_nhwc_bf16");
}
