// Implicit-GEMM convolution / 1x1 / Linear on the gfx950 matrix cores in exact fp32
// (v_mfma_f32_32x32x2_f32: bit-for-bit a k-ordered fmaf chain, 157 TFLOP/s dense peak).
//
// Covers every dense contraction of the DenseFusion slice with ONE kernel family:
//   - PSPNet/ResNet encoder-decoder convs   DenseFusion/lib/extractors.py:14-43,82-89 (3x3 s1/s2, dilation 1/2/4,
//     7x7 s2 stem, 1x1 s2 downsample), pspnet.py:12-17 (1x1 stage + bottleneck), :30-33 (3x3 + PReLU), :53-55 (1x1)
//   - PointNet 1x1 Conv1d chains            DenseFusion/lib/network.py:42-49,76-92,139-146 (points are the "pixels")
//   - PoseRefineNet Linear stacks           network.py:175-182
// with the pointwise tail fused into the epilogue: bias (shared or per-image), residual add (extractors.py:40),
// ReLU / PReLU(single slope) / sigmoid.
//
// GEMM view: Y[M = B*Ho*Wo][N = Cout] = A[M][K = KH*KW*Cin] * Wt[N][K]^T, activations NHWC so that the Cin run of
// each filter tap is contiguous in HBM (coalesced 16-B loads, 128 B per pixel per k-tile), weights [Cout][KH][KW][Cin].
// Channel stride (ld) and channel offset on X, Y and the residual let a layer read or write a channel SLICE of a
// wider buffer, so torch.cat (network.py:56,60,68,155-161) never materialises.
//
// Tiling (wave64, 4 waves / workgroup):
//   workgroup tile BM=128 pixels x BN in {128,64,32} channels, BK=32; each wave owns (128/WM) x (BN/WN) outputs as
//   32x32 MFMA tiles with the accumulators in AGPR/VGPRs.
//   A and B tiles are staged  HBM -> registers -> LDS  (next tile's loads in flight under the current tile's MFMAs),
//   LDS rows are K-contiguous with a 4-float pad (row stride 36 floats): one ds_read_b128 per lane brings the
//   operands of FOUR consecutive MFMAs and its 16-lane groups touch 16 distinct 16-B bank slots (conflict-free).
//   Because a 32x32x2 MFMA sums k over the two lane halves, lane half h supplies k = 4h+e for MFMA e of an 8-deep
//   chunk -- a permutation of k applied identically to A and B, which the sum does not see.
//   blockIdx is remapped so that the N-tiles of one M-tile (which re-read the same A pixels) and neighbouring M-tiles
//   (overlapping 3x3 halos) run on the same XCD and share its 4 MiB L2.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int LDS_LD = BK + 4;  // floats per LDS row
constexpr int NTHREADS = 256;

struct ConvArgs {
    const float* x;
    const float* w;
    const float* bias;
    const float* res;
    float* y;
    ape_conv_params p;
    int M, K, m_tiles, n_tiles;
};

// NONE / RELU / PRELU as selects on loop-invariant scalars (a `switch` per element compiled to a cascade of scalar compares and branches
// per element: ~1.8 k scalar instructions in a 256-element epilogue); same values bit for bit (1 * v == v, also for -0 and NaN)
__device__ __forceinline__ float activate(float v, int act, float alpha)
{
    if (act == APE_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
    const float neg = act == APE_ACT_RELU ? 0.f : (act == APE_ACT_PRELU ? alpha : 1.f) * v;
    return v > 0.f ? v : neg;
}

// BN = workgroup tile width; WM x WN = wave grid (WM*WN == 4)
template <int BN, int WM, int WN>
__global__ __launch_bounds__(NTHREADS) void conv_f32_kernel(const ConvArgs a)
{
    constexpr int TM = BM / WM / 32;  // 32x32 tiles per wave along M
    constexpr int TN = BN / WN / 32;
    constexpr int A_LOADS = BM * BK / 4 / NTHREADS;  // float4 loads per thread per k-tile (4)
    constexpr int B_LOADS = BN * BK / 4 / NTHREADS;  // 4 / 2 / 1
    static_assert(TM >= 1 && TN >= 1 && B_LOADS >= 1, "tile shape");

    __shared__ __attribute__((aligned(16))) float As[BM * LDS_LD];
    __shared__ __attribute__((aligned(16))) float Bs[BN * LDS_LD];

    const ape_conv_params& p = a.p;
    // ---- XCD-aware tile mapping (bijective; guide section 5 "XCD swizzle must be bijective") -----------------
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

    // ---- per-thread staging coordinates: k4 = which float4 of the 32-deep k-tile, rows tid/8 + 32*i ---------
    const int k4 = tid & 7;
    const int srow = tid >> 3;  // 0..31
    const int HoWo = p.Ho * p.Wo;
    int a_base[A_LOADS];        // element offset of pixel (b, iy0, ix0) channel 0; only used when in range
    int a_iy0[A_LOADS], a_ix0[A_LOADS];
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
        const int m = m0 + srow + 32 * i;
        if (m < a.M) {
            const int b = m / HoWo, rem = m - b * HoWo;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            a_iy0[i] = oy * p.stride - p.pad;
            a_ix0[i] = ox * p.stride - p.pad;
            a_base[i] = b * p.H;
        } else {
            a_iy0[i] = -(1 << 28);  // forces the bounds test to fail for every tap
            a_ix0[i] = 0;
            a_base[i] = 0;
        }
    }

    float4 areg[A_LOADS], breg[B_LOADS];
    auto load_tiles = [&](int kt) {
        const int k = kt * BK + k4 * 4;
        const bool kin = k < a.K;
        const int tap = k / p.Cin, ci = k - tap * p.Cin;
        const int ky = tap / p.KW, kx = tap - ky * p.KW;
        const int dy = ky * p.dil, dx = kx * p.dil;
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            const int iy = a_iy0[i] + dy, ix = a_ix0[i] + dx;
            const bool ok = kin && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const size_t off = ((size_t)(a_base[i] + iy) * p.W + ix) * p.ldx + p.xoff + ci;
            areg[i] = ok ? *reinterpret_cast<const float4*>(a.x + off) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) {
            const int n = n0 + srow + 32 * i;
            const bool ok = kin && n < p.Cout;
            breg[i] = ok ? *reinterpret_cast<const float4*>(a.w + (size_t)n * a.K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i)
            *reinterpret_cast<float4*>(&As[(srow + 32 * i) * LDS_LD + k4 * 4]) = areg[i];
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i)
            *reinterpret_cast<float4*>(&Bs[(srow + 32 * i) * LDS_LD + k4 * 4]) = breg[i];
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
    const float* a_frag = &As[(wm * (BM / WM) + frow) * LDS_LD + 4 * fh];
    const float* b_frag = &Bs[(wn * (BN / WN) + frow) * LDS_LD + 4 * fh];

    load_tiles(0);
    store_tiles();
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) load_tiles(kt + 1);  // global loads fly under this tile's MFMAs
#pragma unroll
        for (int kc = 0; kc < BK / 8; ++kc) {
            float4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const float4*>(a_frag + i * 32 * LDS_LD + kc * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const float4*>(b_frag + j * 32 * LDS_LD + kc * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const float av = e == 0 ? af[i].x : e == 1 ? af[i].y : e == 2 ? af[i].z : af[i].w;
                        const float bv = e == 0 ? bf[j].x : e == 1 ? bf[j].y : e == 2 ? bf[j].z : bf[j].w;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                    }
        }
        __syncthreads();
        if (kt + 1 < nk) {
            store_tiles();
            __syncthreads();
        }
    }

    // ---- epilogue: C/D map of the 32x32 tile: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) ---------
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BN / WN) + j * 32 + frow;
        if (n >= p.Cout) continue;
        const float bshared = (a.bias && p.bias_bstride == 0) ? a.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * fh;
                const int m = m0 + wm * (BM / WM) + i * 32 + row;
                if (m >= a.M) continue;
                float v = acc[i][j][e] + bshared;
                if (a.bias && p.bias_bstride != 0) v += a.bias[(size_t)(m / HoWo) * p.bias_bstride + n];
                if (a.res) v += a.res[(size_t)m * p.ldr + p.roff + n];
                a.y[(size_t)m * p.ldy + p.yoff + n] = activate(v, p.act, p.alpha);
            }
        }
    }
}

template <int BN, int WM, int WN>
void launch(const ConvArgs& a, hipStream_t st)
{
    hipLaunchKernelGGL((conv_f32_kernel<BN, WM, WN>), dim3(a.m_tiles * a.n_tiles), dim3(NTHREADS), 0, st, a);
}

}  // namespace

extern "C" int ape_conv2d_nhwc_f32(const float* x, const float* w, const float* bias, const float* residual, float* y,
                                   const ape_conv_params* params, void* stream)
{
    if (!x || !w || !y || !params) return APE_EINVAL;
    const ape_conv_params& p = *params;
    if (p.B < 0 || p.H < 1 || p.W < 1 || p.Cin < 4 || p.Cout < 1 || p.KH < 1 || p.KW < 1 || p.stride < 1 || p.dil < 1 ||
        p.pad < 0 || p.Ho < 1 || p.Wo < 1)
        return APE_EINVAL;
    // 16-byte vector loads along the channel run
    if (p.Cin % 4 || p.ldx % 4 || p.xoff % 4 || p.xoff + p.Cin > p.ldx || p.yoff + p.Cout > p.ldy) return APE_EINVAL;
    if (residual && p.roff + p.Cout > p.ldr) return APE_EINVAL;
    if (p.act < APE_ACT_NONE || p.act > APE_ACT_SIGMOID || p.ups != 0) return APE_EINVAL;
    // output extent must be what the geometry yields (guards against a host-side shape slip -> OOB stores)
    const int ho = (p.H + 2 * p.pad - p.dil * (p.KH - 1) - 1) / p.stride + 1;
    const int wo = (p.W + 2 * p.pad - p.dil * (p.KW - 1) - 1) / p.stride + 1;
    if (ho != p.Ho || wo != p.Wo) return APE_EINVAL;
    const long M = (long)p.B * p.Ho * p.Wo;
    if (M == 0) return APE_OK;
    if (M > (1L << 30) || (long)p.B * p.H * p.W * p.ldx > (1L << 40)) return APE_EINVAL;

    ConvArgs a;
    a.x = x; a.w = w; a.bias = bias; a.res = residual; a.y = y; a.p = p;
    a.M = (int)M;
    a.K = p.KH * p.KW * p.Cin;
    a.m_tiles = ape::ceil_div(M, BM);
    hipStream_t st = (hipStream_t)stream;
    if (p.Cout > 64) {
        a.n_tiles = ape::ceil_div(p.Cout, 128);
        launch<128, 2, 2>(a, st);
    } else if (p.Cout > 32) {
        a.n_tiles = 1;
        launch<64, 4, 1>(a, st);
    } else {
        a.n_tiles = 1;
        launch<32, 4, 1>(a, st);
    }
    return ape::check_launch("ape_conv2d_nhwc_f32");
}
