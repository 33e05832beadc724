// HBM-bound glue kernels of the DenseFusion slice (NHWC fp32, 16-byte accesses along the channel run):
//   max-pool 3x3/s2/p1          extractors.py:85,117
//   adaptive average pool       pspnet.py:15 (nn.AdaptiveAvgPool2d(size), sizes 1,2,3,6)
//   bilinear resize             pspnet.py:22 (F.upsample, align_corners=False) and :31 (nn.Upsample x2, align_corners=True)
//   row gather                  network.py:100-102 (torch.gather of the embedding at `choose`)
//   log-softmax over channels   pspnet.py:55 (nn.LogSoftmax, implicit dim=1)
//   mean over points            network.py:51,65 (AvgPool1d(num_points))
//   channel padding 3->4        (x[B,N,3] -> [B,N,4] so the first 1x1 conv can use 16-byte loads)
//   head output layer           network.py:115-126 (conv4_{r,t,c} evaluated only for the selected object + sigmoid)
// Each is one pass over its input (algorithmic bytes = bytes read + bytes written once); grids are sized to
// >= 2048 workgroups-worth of work and grid-stride the rest.
#include "common.h"
#include <map>
#include <mutex>
#include "s32.h"

namespace {

constexpr int kThreads = 256;
inline int grid_for(long work) { long g = (work + kThreads - 1) / kThreads; return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g)); }

__global__ void maxpool3x3s2_kernel(const float4* __restrict__ x, float4* __restrict__ y, int B, int H, int W, int C4,
                                    int Ho, int Wo)
{
    const long total = (long)B * Ho * Wo * C4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = i % C4;
        long t = i / C4;
        const int ox = t % Wo; t /= Wo;
        const int oy = t % Ho;
        const int b = t / Ho;
        float4 m = make_float4(-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff());
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 - 1 + ky;
            if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 - 1 + kx;
                if ((unsigned)ix >= (unsigned)W) continue;
                const float4 v = x[((long)(b * H + iy) * W + ix) * C4 + c];
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
        y[i] = m;
    }
}

// grid (bins, channel chunks of 64 float4); the 4 waves of a workgroup split the bin's pixels and combine through LDS in a
// fixed order.  (A whole-bin-per-workgroup version left 64 workgroups streaming 4800 pixels each for S = 1.)
__global__ __launch_bounds__(256) void adaptive_avgpool_kernel(const float4* __restrict__ x, float4* __restrict__ y, int H, int W, int C4, int S)
{
    __shared__ float4 part[4][64];
    const int bin = blockIdx.x;
    const int ox = bin % S, oy = (bin / S) % S, b = bin / (S * S);
    const int y0 = (oy * H) / S, y1 = ((oy + 1) * H + S - 1) / S;
    const int x0 = (ox * W) / S, x1 = ((ox + 1) * W + S - 1) / S;
    const int bw = x1 - x0, npx = (y1 - y0) * bw;
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + lane;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < C4)
        for (int p = g; p < npx; p += 4) {
            const int iy = y0 + p / bw, ix = x0 + p % bw;
            const float4 v = x[((long)(b * H + iy) * W + ix) * C4 + c];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    part[g][lane] = s;
    __syncthreads();
    if (g == 0 && c < C4) {
        const float inv = 1.f / (float)npx;
        float4 t = part[0][lane];
        for (int k = 1; k < 4; ++k) { t.x += part[k][lane].x; t.y += part[k][lane].y; t.z += part[k][lane].z; t.w += part[k][lane].w; }
        y[(long)bin * C4 + c] = make_float4(t.x * inv, t.y * inv, t.z * inv, t.w * inv);
    }
}

// Several AdaptiveAvgPool2d sizes of ONE map in one pass over it (the PSP module pools the same 512-channel map to 2x2, 3x3 and
// 6x6, pspnet.py:15: three passes over 0.63 GB as separate launches).  The bin edges of all sizes cut each axis into "atoms"
// (60 x 80 with sizes 2, 3, 6: rows 0|10|20|..|60, columns 0|13|14|26|27|40|53|54|66|67|80); pass 1 sums every atom once,
// pass 2 adds up the atoms of each bin and divides by its area.
constexpr int kPoolMaxAtoms = 12, kPoolMaxSizes = 4, kPoolMaxS = 8;
struct PoolPlan {
    int ny, nx;
    int ye[kPoolMaxAtoms + 1], xe[kPoolMaxAtoms + 1];          // atom edges
    int nsizes, S[kPoolMaxSizes];
    int ya[kPoolMaxSizes][kPoolMaxS + 1], xa[kPoolMaxSizes][kPoolMaxS + 1];   // bin o of size i covers atoms [a[i][2o'] ..): see host
    int yb[kPoolMaxSizes][kPoolMaxS], xb[kPoolMaxSizes][kPoolMaxS];           // first / one-past-last atom of bin o: ya = first, yb = end
    float* out[kPoolMaxSizes];
};

// grid (B * ny, C4 / 64): one atom ROW per workgroup; the 4 waves split its pixel rows, a lane owns one float4 of channels and
// keeps one accumulator per atom column
template <bool S32IN>
__global__ __launch_bounds__(256) void avgpool_atoms_kernel(const float4* __restrict__ x, float4* __restrict__ atoms, const PoolPlan pl,
                                                            int H, int W, int C4, int ld4)
{
    // C4 = float4 groups of channels pooled, ld4 = float4 groups per pixel of x (>= C4: trailing channels are not read)
    __shared__ float4 part[3][kPoolMaxAtoms][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int ay = blockIdx.x % pl.ny, b = blockIdx.x / pl.ny;
    const int c = blockIdx.y * 64 + lane;
    float4 acc[kPoolMaxAtoms];
#pragma unroll
    for (int a = 0; a < kPoolMaxAtoms; ++a) acc[a] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < C4) {
        auto load = [&](long pix) { return S32IN ? ape::s32_load4(x, pix, ld4, c) : x[pix * ld4 + c]; };
        for (int y = pl.ye[ay] + g; y < pl.ye[ay + 1]; y += 4) {
            const long row = (long)(b * H + y) * W;
#pragma unroll
            for (int a = 0; a < kPoolMaxAtoms; ++a) {
                if (a >= pl.nx) break;
                // four pixels' loads in flight before the first add (the sum keeps its pixel order: same result as one at a time)
                int xx = pl.xe[a];
                const int xend = pl.xe[a + 1];
                for (; xx + 4 <= xend; xx += 4) {
                    const float4 v0 = load(row + xx), v1 = load(row + xx + 1), v2 = load(row + xx + 2), v3 = load(row + xx + 3);
                    acc[a].x += v0.x; acc[a].y += v0.y; acc[a].z += v0.z; acc[a].w += v0.w;
                    acc[a].x += v1.x; acc[a].y += v1.y; acc[a].z += v1.z; acc[a].w += v1.w;
                    acc[a].x += v2.x; acc[a].y += v2.y; acc[a].z += v2.z; acc[a].w += v2.w;
                    acc[a].x += v3.x; acc[a].y += v3.y; acc[a].z += v3.z; acc[a].w += v3.w;
                }
                for (; xx < xend; ++xx) {
                    const float4 v = load(row + xx);
                    acc[a].x += v.x; acc[a].y += v.y; acc[a].z += v.z; acc[a].w += v.w;
                }
            }
        }
    }
    if (g > 0) {
#pragma unroll
        for (int a = 0; a < kPoolMaxAtoms; ++a) part[g - 1][a][lane] = acc[a];
    }
    __syncthreads();
    if (g == 0 && c < C4) {
#pragma unroll
        for (int a = 0; a < kPoolMaxAtoms; ++a) {
            if (a >= pl.nx) break;
            float4 t = acc[a];
            for (int k = 0; k < 3; ++k) { t.x += part[k][a][lane].x; t.y += part[k][a][lane].y; t.z += part[k][a][lane].z; t.w += part[k][a][lane].w; }
            atoms[(((long)b * pl.ny + ay) * pl.nx + a) * C4 + c] = t;
        }
    }
}

// one thread per (image, size, bin, float4 of channels)
__global__ void avgpool_bins_kernel(const float4* __restrict__ atoms, const PoolPlan pl, int B, int H, int W, int C4, int bins_total)
{
    const long total = (long)B * bins_total * C4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = i % C4;
        long t = i / C4;
        int bin = t % bins_total;
        const int b = t / bins_total;
        int si = 0;
        while (bin >= pl.S[si] * pl.S[si]) { bin -= pl.S[si] * pl.S[si]; ++si; }
        const int S = pl.S[si], oy = bin / S, ox = bin - oy * S;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int ay = pl.ya[si][oy]; ay < pl.yb[si][oy]; ++ay)
            for (int ax = pl.xa[si][ox]; ax < pl.xb[si][ox]; ++ax) {
                const float4 v = atoms[(((long)b * pl.ny + ay) * pl.nx + ax) * C4 + c];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        const float inv = 1.f / (float)((pl.ye[pl.yb[si][oy]] - pl.ye[pl.ya[si][oy]]) * (pl.xe[pl.xb[si][ox]] - pl.xe[pl.xa[si][ox]]));
        reinterpret_cast<float4*>(pl.out[si])[((long)(b * S + oy) * S + ox) * C4 + c] = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
    }
}

__device__ __forceinline__ float src_index(int dst, float scale, bool align_corners)
{
    if (align_corners) return scale * (float)dst;
    const float s = scale * ((float)dst + 0.5f) - 0.5f;  // ATen area_pixel_compute_source_index
    return s < 0.f ? 0.f : s;
}

__global__ void bilinear_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C4, int ldx4,
                                int Ho, int Wo, int ldy4, int yoff4, float sh, float sw, int align_corners, int accumulate)
{
    const float4* x4 = reinterpret_cast<const float4*>(x);
    float4* y4 = reinterpret_cast<float4*>(y);
    const long total = (long)B * Ho * Wo * C4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = i % C4;
        long t = i / C4;
        const int ox = t % Wo; t /= Wo;
        const int oy = t % Ho;
        const int b = t / Ho;
        const float fy = src_index(oy, sh, align_corners), fx = src_index(ox, sw, align_corners);
        const int iy0 = (int)fy, ix0 = (int)fx;
        const int iy1 = iy0 + (iy0 < H - 1 ? 1 : 0), ix1 = ix0 + (ix0 < W - 1 ? 1 : 0);
        const float ly1 = fy - (float)iy0, lx1 = fx - (float)ix0;
        const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
        const float4 v00 = x4[((long)(b * H + iy0) * W + ix0) * ldx4 + c];
        const float4 v01 = x4[((long)(b * H + iy0) * W + ix1) * ldx4 + c];
        const float4 v10 = x4[((long)(b * H + iy1) * W + ix0) * ldx4 + c];
        const float4 v11 = x4[((long)(b * H + iy1) * W + ix1) * ldx4 + c];
        float4 o;
        o.x = ly0 * (lx0 * v00.x + lx1 * v01.x) + ly1 * (lx0 * v10.x + lx1 * v11.x);
        o.y = ly0 * (lx0 * v00.y + lx1 * v01.y) + ly1 * (lx0 * v10.y + lx1 * v11.y);
        o.z = ly0 * (lx0 * v00.z + lx1 * v01.z) + ly1 * (lx0 * v10.z + lx1 * v11.z);
        o.w = ly0 * (lx0 * v00.w + lx1 * v01.w) + ly1 * (lx0 * v10.w + lx1 * v11.w);
        const long yo = ((long)(b * Ho + oy) * Wo + ox) * ldy4 + yoff4 + c;
        if (accumulate) {
            const float4 p = y4[yo];
            o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
        }
        y4[yo] = o;
    }
}

__global__ void gather_rows_kernel(const float4* __restrict__ x, const int64_t* __restrict__ index, float4* __restrict__ y,
                                   int B, int rows_in, int n, int C4)
{
    const long total = (long)B * n * C4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = i % C4;
        const long bn = i / C4;
        const int b = bn / n;
        long src = index[bn];
        src = src < 0 ? 0 : (src >= rows_in ? rows_in - 1 : src);  // never read out of bounds on a bad index
        y[i] = x[((long)b * rows_in + src) * C4 + c];
    }
}

// 3x3 patches of the bilinear x2 (align_corners=True) up-sampling of x[B][h][w][C], gathered ONLY at chosen pixels:
// out[(b*n + i)][tap*C + c] = up(x)[b][y + ky - 1][x + kx - 1][c] (0 outside the 2h x 2w image, the conv's zero padding) for the
// pixel index[b][i] = y * 2w + x of the up-sampled image.  PoseNet keeps 1000 of a crop's 25 600 embedding pixels
// (network.py:100-102), so its last 3x3 conv (up_3, pspnet.py:30-33) only has to be evaluated there: this gather + one
// [B*n, 9C] x [9C, Cout] GEMM replaces the full-resolution convolution.  Interpolation op order as bilinear_kernel.
__global__ void ups_patch_gather_kernel(const float4* __restrict__ x, const int64_t* __restrict__ index, float4* __restrict__ out,
                                        int B, int h, int w, int C4, int n, float sh, float sw)
{
    const int Ho = 2 * h, Wo = 2 * w;
    const long total = (long)B * n * 9 * C4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = i % C4;
        long t = i / C4;
        const int tap = t % 9;
        const long bn = t / 9;
        const int b = bn / n;
        long pix = index[bn];
        pix = pix < 0 ? 0 : (pix >= (long)Ho * Wo ? (long)Ho * Wo - 1 : pix);      // never read out of bounds on a bad index
        const int qy = (int)(pix / Wo) + tap / 3 - 1, qx = (int)(pix % Wo) + tap % 3 - 1;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)qy < (unsigned)Ho && (unsigned)qx < (unsigned)Wo) {
            const float fy = sh * (float)qy, fx = sw * (float)qx;
            const int iy0 = (int)fy, ix0 = (int)fx;
            const int iy1 = iy0 + (iy0 < h - 1 ? 1 : 0), ix1 = ix0 + (ix0 < w - 1 ? 1 : 0);
            const float ly1 = fy - (float)iy0, lx1 = fx - (float)ix0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
            const float4 v00 = x[((long)(b * h + iy0) * w + ix0) * C4 + c], v01 = x[((long)(b * h + iy0) * w + ix1) * C4 + c];
            const float4 v10 = x[((long)(b * h + iy1) * w + ix0) * C4 + c], v11 = x[((long)(b * h + iy1) * w + ix1) * C4 + c];
            o.x = ly0 * (lx0 * v00.x + lx1 * v01.x) + ly1 * (lx0 * v10.x + lx1 * v11.x);
            o.y = ly0 * (lx0 * v00.y + lx1 * v01.y) + ly1 * (lx0 * v10.y + lx1 * v11.y);
            o.z = ly0 * (lx0 * v00.z + lx1 * v01.z) + ly1 * (lx0 * v10.z + lx1 * v11.z);
            o.w = ly0 * (lx0 * v00.w + lx1 * v01.w) + ly1 * (lx0 * v10.w + lx1 * v11.w);
        }
        out[i] = o;
    }
}

// log-softmax over C <= 64 channels of each row; one lane per row (rows are short: 32 channels on this path)
__global__ void log_softmax_rows_kernel(const float* __restrict__ x, float* __restrict__ y, long rows, int C)
{
    for (long r = blockIdx.x * (long)blockDim.x + threadIdx.x; r < rows; r += (long)gridDim.x * blockDim.x) {
        const float* xr = x + r * C;
        float m = xr[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, xr[c]);
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += expf(xr[c] - m);
        const float l = logf(s);
        for (int c = 0; c < C; ++c) y[r * C + c] = (xr[c] - m) - l;
    }
}

// mean over the n rows of each image: grid (C/64, B), block 256 = 4 row-groups x 64 channels
__global__ void mean_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int n, int C)
{
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int g = threadIdx.x >> 6;
    const int b = blockIdx.y;
    float s = 0.f;
    if (c < C)
        for (int r = g; r < n; r += 4) s += x[((long)b * n + r) * C + c];
    part[g][threadIdx.x & 63] = s;
    __syncthreads();
    if (g == 0 && c < C) y[(long)b * C + c] = (((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x]) / (float)n;
}

// C % 4 == 0 and many rows: grid (C/256, B), 16 waves; a wave owns 256 channels (one float4 per lane) of every 16th row, four
// rows in flight per lane; the 16 partial sums meet in LDS and are added in wave order (fixed order: deterministic).
// (The 4-group dword version above kept 1 024 workgroups at 2 TB/s: 64 crops x 1000 points x 1024 channels took 131 us.)
__global__ __launch_bounds__(1024) void mean_rows_wide_kernel(const float4* __restrict__ x, float4* __restrict__ y, int n, int C4)
{
    __shared__ float4 part[16][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int b = blockIdx.y;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
    if (c < C4) {
        const float4* xb = x + (long)b * n * C4 + c;
        int r = g;
        for (; r + 48 < n; r += 64) {
            const float4 v0 = xb[(long)r * C4], v1 = xb[(long)(r + 16) * C4], v2 = xb[(long)(r + 32) * C4], v3 = xb[(long)(r + 48) * C4];
            s0.x += v0.x; s0.y += v0.y; s0.z += v0.z; s0.w += v0.w;
            s1.x += v1.x; s1.y += v1.y; s1.z += v1.z; s1.w += v1.w;
            s2.x += v2.x; s2.y += v2.y; s2.z += v2.z; s2.w += v2.w;
            s3.x += v3.x; s3.y += v3.y; s3.z += v3.z; s3.w += v3.w;
        }
        for (; r < n; r += 16) {
            const float4 v0 = xb[(long)r * C4];
            s0.x += v0.x; s0.y += v0.y; s0.z += v0.z; s0.w += v0.w;
        }
    }
    part[g][lane] = make_float4((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z),
                                (s0.w + s1.w) + (s2.w + s3.w));
    __syncthreads();
    if (g == 0 && c < C4) {
        float4 t = part[0][lane];
        for (int k = 1; k < 16; ++k) { t.x += part[k][lane].x; t.y += part[k][lane].y; t.z += part[k][lane].z; t.w += part[k][lane].w; }
        const float inv = (float)n;
        y[(long)b * C4 + c] = make_float4(t.x / inv, t.y / inv, t.z / inv, t.w / inv);
    }
}

__global__ void pad3to4_kernel(const float* __restrict__ x, float4* __restrict__ y, long rows)
{
    for (long r = blockIdx.x * (long)blockDim.x + threadIdx.x; r < rows; r += (long)gridDim.x * blockDim.x)
        y[r] = make_float4(x[r * 3], x[r * 3 + 1], x[r * 3 + 2], 0.f);
}

// out[b][n][j], j<8: 0..3 quaternion (W_r rows obj*4+j), 4..6 translation (W_t rows obj*3+j-4), 7 sigmoid(confidence)
// h[b*n][ldh] holds the three 128-wide layer-3 activations at channel offsets off_r, off_t, off_c.
__global__ void head_select_kernel(const float* __restrict__ h, int ldh, int off_r, int off_t, int off_c,
                                   const float* __restrict__ wr, const float* __restrict__ br,
                                   const float* __restrict__ wt, const float* __restrict__ bt,
                                   const float* __restrict__ wc, const float* __restrict__ bc,
                                   const int64_t* __restrict__ obj, float* __restrict__ out, int n, int K)
{
    extern __shared__ float wsel[];  // 8 rows x K
    __shared__ float bsel[8];
    const int b = blockIdx.y;
    const int o = (int)obj[b];
    for (int i = threadIdx.x; i < 8 * K; i += blockDim.x) {
        const int j = i / K, k = i - j * K;
        wsel[i] = j < 4 ? wr[(long)(o * 4 + j) * K + k] : j < 7 ? wt[(long)(o * 3 + j - 4) * K + k] : (wc ? wc[(long)o * K + k] : 0.f);
    }
    if (threadIdx.x < 8) {
        const int j = threadIdx.x;
        bsel[j] = j < 4 ? br[o * 4 + j] : j < 7 ? bt[o * 3 + j - 4] : (bc ? bc[o] : 0.f);
    }
    __syncthreads();
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n * 8; t += gridDim.x * blockDim.x) {
        const int j = t & 7, p = t >> 3;
        const float* hp = h + ((long)b * n + p) * ldh + (j < 4 ? off_r : j < 7 ? off_t : off_c);
        const float* w = wsel + j * K;
        float s = 0.f;
        for (int k = 0; k < K; ++k) s = fmaf(hp[k], w[k], s);
        s += bsel[j];
        if (j == 7) s = wc ? 1.f / (1.f + expf(-s)) : 0.f;
        out[((long)b * n + p) * 8 + j] = s;
    }
}

// conv3x3(pad 1) o bilinear-x2(align_corners=True)  ==  sum over the 9 taps of  bilinear-x2( W_tap . x )  shifted by the tap:
// the channel mixing W_tap commutes with the (linear, per-channel) resize, so it is applied at LOW resolution by a 1x1 conv
// producing z[B][h][w][9*C] (channel = tap*C + c), and this kernel does the resize + tap shift + sum + bias + PReLU:
//   out(Y, X, c) = act( bias[c] + sum_{ky,kx} U_tap(Y + ky - 1, X + kx - 1)[c] ),   U_tap(q) = 0 outside the 2h x 2w image
// (the reference zero-pads the UPSAMPLED image, pspnet.py:30-32).  4x fewer MFMA flops than convolving at high resolution
// and the upsampled 2h x 2w x Cin tensor is never written.
constexpr int kUpCols = 12;     // low-resolution columns under 16 + 2 output columns at scale ~1/2 (at most 11)

// FMA: the two interpolation steps as chained fused multiply-adds, acc = fma(l1, v1, fma(l0, v0, acc)), instead of the separately rounded
// acc + (l0 * v0 + l1 * v1): half the vector instructions.  It is the arithmetic of the fused kernel's fast form (upconv_fused.hip), whose
// unfused twin this kernel is; up_1 / up_2 keep the separately rounded form.
template <bool S32OUT, bool FMA = false>
__global__ __launch_bounds__(256) void upconv_gather_kernel(const float4* __restrict__ z, const float* __restrict__ bias,
                                                            float4* __restrict__ out, int B, int h, int w, int C4, float sh, float sw,
                                                            int act, float alpha)
{
    // one workgroup = 16 consecutive output pixels of one row, all channels; consecutive workgroups (same / adjacent rows, which
    // read the same 2-3 low-resolution rows of z) are kept on ONE XCD so that its L2 serves the ~9x re-use of every z element
    const int Ho = 2 * h, Wo = 2 * w;
    const int xt_n = (Wo + 15) / 16;
    const int nwg = gridDim.x;
    const int orig = blockIdx.x;
    const int xcd = orig % 8, q = nwg / 8, r = nwg % 8;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
    const int xt = logical % xt_n;
    const int Y = (logical / xt_n) % Ho;
    const int b = logical / (xt_n * Ho);
    float ly0[3], ly1[3];
    int iy0[3], iy1[3];
    bool yok[3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int qy = Y + ky - 1;
        yok[ky] = (unsigned)qy < (unsigned)Ho;
        const float fy = sh * (float)(yok[ky] ? qy : 0);
        iy0[ky] = (int)fy;
        iy1[ky] = iy0[ky] + (iy0[ky] < h - 1 ? 1 : 0);
        ly1[ky] = fy - (float)iy0[ky];
        ly0[ky] = 1.f - ly1[ky];
    }
    // The resize is separable and its x weights do not depend on the tap row, so for this output row
    //   out(X) = sum_kx lerp_x( S_kx, X + kx - 1 ),   S_kx(ix) = sum_ky lerp_y( z_{ky,kx}(., ix), Y + ky - 1 )
    // phase 1 builds S_kx for the ~10 low-resolution columns under the 16 + 2 output columns in LDS (6 loads per value instead
    // of 36 per output: the kernel was bound by L1 request rate, not by HBM), phase 2 does the x interpolation from LDS.
    extern __shared__ float4 S[];                       // [3][kMaxCols][C4]
    const int X0 = xt * 16;
    const int qx_lo = X0 - 1 < 0 ? 0 : X0 - 1, qx_hi = X0 + 16 > Wo - 1 ? Wo - 1 : X0 + 16;
    const int ix_lo = (int)(sw * (float)qx_lo);
    int ix_hi = (int)(sw * (float)qx_hi);
    ix_hi = ix_hi + (ix_hi < w - 1 ? 1 : 0);
    const int ni = ix_hi - ix_lo + 1;                   // <= kUpCols (host checks)
    // two items per thread and round, all twelve loads issued before the first use (four per round was slower) (rows outside the image are clamped and
    // carry zero weights): the kernel is latency x occupancy bound, not bandwidth bound
    const int nitems = 3 * ni * C4;
    constexpr int U = 2;
    for (int item0 = threadIdx.x; item0 < nitems; item0 += 256 * U) {
        float4 v0[U][3], v1[U][3];
        int sidx[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int item = item0 + 256 * u;
            sidx[u] = -1;
            if (item < nitems) {
                const int c = item % C4;
                const int ii = (item / C4) % ni;
                const int kx = item / (C4 * ni);
                sidx[u] = (kx * kUpCols + ii) * C4 + c;
            } else {
                item = item0;
            }
            const int c = item % C4;
            const int ii = (item / C4) % ni;
            const int kx = item / (C4 * ni);
            const int ix = ix_lo + ii;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int tc = (ky * 3 + kx) * C4 + c;
                v0[u][ky] = z[((long)(b * h + iy0[ky]) * w + ix) * (9 * C4) + tc];
                v1[u][ky] = z[((long)(b * h + iy1[ky]) * w + ix) * (9 * C4) + tc];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                if (!yok[ky]) continue;
                if (FMA) {
                    acc.x = fmaf(ly1[ky], v1[u][ky].x, fmaf(ly0[ky], v0[u][ky].x, acc.x));
                    acc.y = fmaf(ly1[ky], v1[u][ky].y, fmaf(ly0[ky], v0[u][ky].y, acc.y));
                    acc.z = fmaf(ly1[ky], v1[u][ky].z, fmaf(ly0[ky], v0[u][ky].z, acc.z));
                    acc.w = fmaf(ly1[ky], v1[u][ky].w, fmaf(ly0[ky], v0[u][ky].w, acc.w));
                    continue;
                }
                acc.x += ly0[ky] * v0[u][ky].x + ly1[ky] * v1[u][ky].x;
                acc.y += ly0[ky] * v0[u][ky].y + ly1[ky] * v1[u][ky].y;
                acc.z += ly0[ky] * v0[u][ky].z + ly1[ky] * v1[u][ky].z;
                acc.w += ly0[ky] * v0[u][ky].w + ly1[ky] * v1[u][ky].w;
            }
            if (sidx[u] >= 0) S[sidx[u]] = acc;
        }
    }
    __syncthreads();
    for (int item = threadIdx.x; item < 16 * C4; item += 256) {
        const int c = item % C4;
        const int X = X0 + item / C4;
        if (X >= Wo) continue;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int qx = X + kx - 1;
            if ((unsigned)qx >= (unsigned)Wo) continue;
            const float fx = sw * (float)qx;
            const int ix0 = (int)fx, ix1 = ix0 + (ix0 < w - 1 ? 1 : 0);
            const float lx1 = fx - (float)ix0, lx0 = 1.f - lx1;
            const float4 s0 = S[(kx * kUpCols + ix0 - ix_lo) * C4 + c], s1 = S[(kx * kUpCols + ix1 - ix_lo) * C4 + c];
            if (FMA) {
                acc.x = fmaf(lx1, s1.x, fmaf(lx0, s0.x, acc.x));
                acc.y = fmaf(lx1, s1.y, fmaf(lx0, s0.y, acc.y));
                acc.z = fmaf(lx1, s1.z, fmaf(lx0, s0.z, acc.z));
                acc.w = fmaf(lx1, s1.w, fmaf(lx0, s0.w, acc.w));
                continue;
            }
            acc.x += lx0 * s0.x + lx1 * s1.x;
            acc.y += lx0 * s0.y + lx1 * s1.y;
            acc.z += lx0 * s0.z + lx1 * s1.z;
            acc.w += lx0 * s0.w + lx1 * s1.w;
        }
        if (bias) { acc.x += bias[c * 4]; acc.y += bias[c * 4 + 1]; acc.z += bias[c * 4 + 2]; acc.w += bias[c * 4 + 3]; }
        if (act == APE_ACT_RELU) { acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f); }
        else if (act == APE_ACT_PRELU) {
            acc.x = acc.x > 0.f ? acc.x : alpha * acc.x; acc.y = acc.y > 0.f ? acc.y : alpha * acc.y;
            acc.z = acc.z > 0.f ? acc.z : alpha * acc.z; acc.w = acc.w > 0.f ? acc.w : alpha * acc.w;
        }
        if (S32OUT) ape::s32_store4(out, (long)(b * Ho + Y) * Wo + X, C4, c, acc);
        else out[((long)(b * Ho + Y) * Wo + X) * C4 + c] = acc;
    }
}

// The same operator for channel counts that are multiples of 64, as a walk DOWN the image: one workgroup = a strip of `rows` output rows x
// 16 output columns x 64 channels (256-byte runs of z and of the output).  The row kernel above re-reads z for every output row -- each z
// element is wanted by ~4 output rows (2 as the upper, 2 as the lower row of the y interpolation), 6 row loads per output row and S value --
// and is bound by that L2 -> L1 traffic (~5x the tensor per launch), not by HBM.  Here a thread keeps the two z rows of each of its 9 taps in
// registers while the strip moves down: going from output row Y to Y + 1 a tap's row pair (iy0, iy1) either stays or becomes (iy1, iy1 + 1)
// -- one new row load, on average 1.5 per output row instead of 6 -- and the loads for row Y + 1 are issued before row Y's x interpolation,
// so they fly during it.  The arithmetic per output is the row kernel's, operation for operation (same S_kx, same x interpolation, same
// epilogue): bit-identical outputs.
typedef float ups_f32x2 __attribute__((ext_vector_type(2)));
template <bool S32OUT, bool FMA>
__global__ __launch_bounds__(256) void upconv_gather_strip_kernel(const float4* __restrict__ z, const float* __restrict__ bias,
                                                                  float4* __restrict__ out, int B, int h, int w, int C4, float sh, float sw,
                                                                  int act, float alpha, int rows)
{
    const int Ho = 2 * h, Wo = 2 * w;
    const int xt_n = (Wo + 15) / 16, ns = (Ho + rows - 1) / rows, nslab = C4 / 16;
    const int nwg = gridDim.x;
    const int orig = blockIdx.x;
    const int xcd = orig % 8, q = nwg / 8, r = nwg % 8;
    int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;      // neighbours in x (shared halo columns) on one XCD
    const int slab = logical % nslab; logical /= nslab;
    const int xt = logical % xt_n; logical /= xt_n;
    const int strip = logical % ns;
    const int b = logical / ns;
    const int Ys = strip * rows, Ye = Ys + rows < Ho ? Ys + rows : Ho;
    __shared__ float4 S[2][3 * kUpCols * 16];
    const int X0 = xt * 16;
    const int qx_lo = X0 - 1 < 0 ? 0 : X0 - 1, qx_hi = X0 + 16 > Wo - 1 ? Wo - 1 : X0 + 16;
    const int ix_lo = (int)(sw * (float)qx_lo);
    int ix_hi = (int)(sw * (float)qx_hi);
    ix_hi = ix_hi + (ix_hi < w - 1 ? 1 : 0);
    const int ni = ix_hi - ix_lo + 1;                   // <= kUpCols (host checks)
    const int c4 = threadIdx.x & 15, hi4 = threadIdx.x >> 4;
    const int c = slab * 16 + c4;
    // phase 1 role: the 3 * ni (<= 33) pairs (low-resolution column, kx) are dealt to the 16 column slots of the workgroup, slot hi4 takes
    // items hi4, hi4 + 16 (and hi4 + 32: slot 0 of an 11-column tile only), item = 3 * column + kx.  (One column with its three kx per
    // slot left 6 of the 16 slots idle -- and loading: a third of the tap loads and of the y interpolation were thrown away.)
    constexpr int NIT = 3;
    bool p1[NIT];
    int it_s[NIT];                                      // the item's S slot: (kx * kUpCols + column) * 16 + c4
    const long rs = (long)w * 9 * C4;                   // float4 per low-resolution row of z
    const float4* zi[NIT];                              // tap row ky of the item's (column, kx) at zi + iy * rs + ky * 3 * C4
#pragma unroll
    for (int r = 0; r < NIT; ++r) {
        const int item = hi4 + 16 * r;
        p1[r] = item < 3 * ni;
        const int col = p1[r] ? item / 3 : 0, kx = p1[r] ? item - 3 * col : 0;
        it_s[r] = (kx * kUpCols + col) * 16 + c4;
        zi[r] = z + (long)b * h * rs + (long)(ix_lo + col) * 9 * C4 + kx * C4 + c;
    }
    // (wave-uniform: does any lane of this wave hold a third item?  lanes of a wave: hi4 = 4 wave .. 4 wave + 3)
    const bool third = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * 4 + 32 < 3 * ni;
    float4 v0[3][NIT], v1[3][NIT];                      // [ky][item]: the held row pair of every tap
    int hy0[3];
    // phase 2 role: output column X0 + hi4
    const int X = X0 + hi4;
    const bool p2 = X < Wo;
    int so0[3], so1[3];
    float lx0[3], lx1[3];
    bool xok[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int qx = X + kx - 1;
        xok[kx] = p2 && (unsigned)qx < (unsigned)Wo;
        const float fx = sw * (float)(xok[kx] ? qx : 0);
        const int ix0 = (int)fx, ix1 = ix0 + (ix0 < w - 1 ? 1 : 0);
        lx1[kx] = fx - (float)ix0;
        lx0[kx] = 1.f - lx1[kx];
        so0[kx] = xok[kx] ? (kx * kUpCols + ix0 - ix_lo) * 16 + c4 : c4;
        so1[kx] = xok[kx] ? (kx * kUpCols + ix1 - ix_lo) * 16 + c4 : c4;
    }
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) bv = make_float4(bias[c * 4], bias[c * 4 + 1], bias[c * 4 + 2], bias[c * 4 + 3]);

    // ---- the window of z rows ------------------------------------------------------------------------------------------------------
    // hy0[ky]: the upper row of the pair tap row ky holds.  ADVANCE(ky, YY): make the pair fit output row YY, assuming (host-checked floor
    // pattern, ape_upconv3x3_gather_ex) it either stays or moves down by one: v0 <- keep ? v0 : v1 (a select on a scalar condition, no
    // branch), v1 <- the pair's lower row, loaded unconditionally (the same row again when the pair stays: first rows of the image, last
    // row's clamp).  Which tap row advances when is static: q = Y + ky - 1 and the upper row changes when q turns odd, so ky = 1 advances
    // on the way to an odd Y, ky = 0 and 2 on the way to an even one -- strips start on even rows.
#define APE_UPS_ROWS(YY, ky, iy0, iy1)                                                                                                     \
    const int qy_##ky = (YY) + (ky) - 1;                                                                                                   \
    const int iy0 = __builtin_amdgcn_readfirstlane((int)(sh * (float)((unsigned)qy_##ky < (unsigned)Ho ? qy_##ky : 0)));                   \
    const int iy1 = iy0 + (iy0 < h - 1 ? 1 : 0);
#define APE_UPS_ADVANCE(YY, ky)                                                                                                            \
    {                                                                                                                                      \
        APE_UPS_ROWS(YY, ky, iy0, iy1)                                                                                                     \
        const bool keep = iy0 == hy0[ky];                                                                                                  \
        const long o1 = (long)iy1 * rs + (ky) * 3 * C4;                                                                                    \
        _Pragma("unroll") for (int r = 0; r < NIT; ++r) {                                                                                  \
            if (r == 2 && !third) break;                                                                                                   \
            v0[ky][r].x = keep ? v0[ky][r].x : v1[ky][r].x; v0[ky][r].y = keep ? v0[ky][r].y : v1[ky][r].y;                                \
            v0[ky][r].z = keep ? v0[ky][r].z : v1[ky][r].z; v0[ky][r].w = keep ? v0[ky][r].w : v1[ky][r].w;                                \
            v1[ky][r] = zi[r][o1];                                                                                                         \
        }                                                                                                                                  \
        hy0[ky] = iy0;                                                                                                                     \
    }
#define APE_UPS_INIT(YY, ky)                                                                                                               \
    {                                                                                                                                      \
        APE_UPS_ROWS(YY, ky, iy0, iy1)                                                                                                     \
        const long o0 = (long)iy0 * rs + (ky) * 3 * C4, o1 = (long)iy1 * rs + (ky) * 3 * C4;                                               \
        _Pragma("unroll") for (int r = 0; r < NIT; ++r) { v0[ky][r] = zi[r][o0]; v1[ky][r] = zi[r][o1]; }                                  \
        hy0[ky] = iy0;                                                                                                                     \
    }
    // S_kx of output row YY from the window -> LDS buffer Sb
#define APE_UPS_PHASE1(YY, Sb)                                                                                                             \
    {                                                                                                                                      \
        float ly0[3], ly1[3];                                                                                                              \
        bool yok[3];                                                                                                                       \
        _Pragma("unroll") for (int ky = 0; ky < 3; ++ky) {                                                                                 \
            const int qy = (YY) + ky - 1;                                                                                                  \
            yok[ky] = (unsigned)qy < (unsigned)Ho;                                                                                         \
            const float fy = sh * (float)(yok[ky] ? qy : 0);                                                                               \
            ly1[ky] = fy - (float)(int)fy;                                                                                                 \
            ly0[ky] = 1.f - ly1[ky];                                                                                                       \
        }                                                                                                                                  \
        _Pragma("unroll") for (int kx = 0; kx < NIT; ++kx) {             /* (kx: the thread's item index here) */                        \
            if (kx == 2 && !third) break;                                                                                                  \
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);                                                                                  \
            _Pragma("unroll") for (int ky = 0; ky < 3; ++ky) {                                                                             \
                if (!yok[ky]) continue;                                                                                                    \
                const float4 a0 = v0[ky][kx], a1 = v1[ky][kx];                                                                             \
                if (FMA) {                                                                                                                 \
                    acc.x = fmaf(ly1[ky], a1.x, fmaf(ly0[ky], a0.x, acc.x));                                                               \
                    acc.y = fmaf(ly1[ky], a1.y, fmaf(ly0[ky], a0.y, acc.y));                                                               \
                    acc.z = fmaf(ly1[ky], a1.z, fmaf(ly0[ky], a0.z, acc.z));                                                               \
                    acc.w = fmaf(ly1[ky], a1.w, fmaf(ly0[ky], a0.w, acc.w));                                                               \
                    continue;                                                                                                              \
                }                                                                                                                          \
                /* two channels per instruction (v_pk_mul_f32 / v_pk_add_f32): the same separately rounded products and sums */          \
                const ups_f32x2 w0 = {ly0[ky], ly0[ky]}, w1 = {ly1[ky], ly1[ky]};                                                          \
                const ups_f32x2 tl = w0 * ups_f32x2{a0.x, a0.y} + w1 * ups_f32x2{a1.x, a1.y};                                              \
                const ups_f32x2 th = w0 * ups_f32x2{a0.z, a0.w} + w1 * ups_f32x2{a1.z, a1.w};                                              \
                const ups_f32x2 sl = ups_f32x2{acc.x, acc.y} + tl, sh2 = ups_f32x2{acc.z, acc.w} + th;                                     \
                acc = make_float4(sl.x, sl.y, sh2.x, sh2.y);                                                                               \
            }                                                                                                                              \
            if (p1[kx]) (Sb)[it_s[kx]] = acc;                                                                                              \
        }                                                                                                                                  \
    }
    // the x interpolation, epilogue and store of output row YY from LDS buffer Sb
#define APE_UPS_PHASE2(YY, Sb)                                                                                                             \
    if (p2) {                                                                                                                              \
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);                                                                                      \
        _Pragma("unroll") for (int kx = 0; kx < 3; ++kx) {                                                                                 \
            if (!xok[kx]) continue;                                                                                                        \
            const float4 s0 = (Sb)[so0[kx]], s1 = (Sb)[so1[kx]];                                                                           \
            if (FMA) {                                                                                                                     \
                acc.x = fmaf(lx1[kx], s1.x, fmaf(lx0[kx], s0.x, acc.x));                                                                   \
                acc.y = fmaf(lx1[kx], s1.y, fmaf(lx0[kx], s0.y, acc.y));                                                                   \
                acc.z = fmaf(lx1[kx], s1.z, fmaf(lx0[kx], s0.z, acc.z));                                                                   \
                acc.w = fmaf(lx1[kx], s1.w, fmaf(lx0[kx], s0.w, acc.w));                                                                   \
                continue;                                                                                                                  \
            }                                                                                                                              \
            {                                                                                                                              \
                const ups_f32x2 w0 = {lx0[kx], lx0[kx]}, w1 = {lx1[kx], lx1[kx]};                                                          \
                const ups_f32x2 tl = w0 * ups_f32x2{s0.x, s0.y} + w1 * ups_f32x2{s1.x, s1.y};                                              \
                const ups_f32x2 th = w0 * ups_f32x2{s0.z, s0.w} + w1 * ups_f32x2{s1.z, s1.w};                                              \
                const ups_f32x2 sl = ups_f32x2{acc.x, acc.y} + tl, sh2 = ups_f32x2{acc.z, acc.w} + th;                                     \
                acc = make_float4(sl.x, sl.y, sh2.x, sh2.y);                                                                               \
            }                                                                                                                              \
        }                                                                                                                                  \
        if (bias) { acc.x += bv.x; acc.y += bv.y; acc.z += bv.z; acc.w += bv.w; }                                                          \
        if (act == APE_ACT_RELU) { acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f); } \
        else if (act == APE_ACT_PRELU) {                                                                                                   \
            acc.x = acc.x > 0.f ? acc.x : alpha * acc.x; acc.y = acc.y > 0.f ? acc.y : alpha * acc.y;                                      \
            acc.z = acc.z > 0.f ? acc.z : alpha * acc.z; acc.w = acc.w > 0.f ? acc.w : alpha * acc.w;                                      \
        }                                                                                                                                  \
        if (S32OUT) ape::s32_store4_pair(out, (long)(b * Ho + (YY)) * Wo + X, C4, c, acc);    /* (p2 is the same for both lanes of a pair) */ \
        else out[((long)(b * Ho + (YY)) * Wo + X) * C4 + c] = acc;                                                                         \
    }

    APE_UPS_INIT(Ys, 0) APE_UPS_INIT(Ys, 1) APE_UPS_INIT(Ys, 2)
    for (int Y = Ys; Y < Ye; Y += 2) {                  // Ys even, rows even, Ho even: whole pairs
        APE_UPS_PHASE1(Y, S[0])
        APE_UPS_ADVANCE(Y + 1, 1)                       // the odd row's new z row: in flight during the x interpolation below
        __syncthreads();
        APE_UPS_PHASE2(Y, S[0])
        APE_UPS_PHASE1(Y + 1, S[1])
        APE_UPS_ADVANCE(Y + 2, 0)
        APE_UPS_ADVANCE(Y + 2, 2)
        __syncthreads();
        APE_UPS_PHASE2(Y + 1, S[1])
    }
#undef APE_UPS_ROWS
#undef APE_UPS_ADVANCE
#undef APE_UPS_INIT
#undef APE_UPS_PHASE1
#undef APE_UPS_PHASE2
}

// sum of the four PSP priors, each up-sampled bilinearly (align_corners=False) from its s x s map (s = 1,2,3,6) to h x w:
// one pass writing the result once, instead of four read-modify-write passes over the [B,h,w,C] accumulator (pspnet.py:22).
// One thread owns a column (b, ox, 4 channels) and walks down the h rows: the x-interpolated values  tx = lx0 * z[iy][ix0] +
// lx1 * z[iy][ix1]  of all 1 + 2 + 3 + 6 prior rows are formed ONCE (24 loads per thread) and every output is
// sum_k ly0 * tx_k[iy0] + ly1 * tx_k[iy1] -- the same products and the same summation order as the element-wise form
// ly0 * (lx0 v00 + lx1 v01) + ly1 * (lx0 v10 + lx1 v11) accumulated over k = 1, 2, 3, 6, so the result is bit-identical,
// but without 16 loads and ~100 index operations per output float4 (0.81 -> HBM-bound write of the 1.26 GB result).
__global__ void psp_prior_sum_kernel(const float4* __restrict__ z1, const float4* __restrict__ z2, const float4* __restrict__ z3,
                                     const float4* __restrict__ z6, float4* __restrict__ out, int B, int h, int w, int C4)
{
    const long total = (long)B * w * C4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = i % C4;
        const long t = i / C4;
        const int ox = t % w;
        const int b = t / w;
        float4 tx[12];                              // rows of the 1x1, 2x2, 3x3, 6x6 priors at this column
        const float4* zs[4] = {z1, z2, z3, z6};
        const int ss[4] = {1, 2, 3, 6};
        const int base[4] = {0, 1, 3, 6};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int S = ss[k];
            const float fx = src_index(ox, (float)S / (float)w, false);
            const int ix0 = (int)fx;
            const int ix1 = ix0 + (ix0 < S - 1 ? 1 : 0);
            const float lx1 = fx - (float)ix0, lx0 = 1.f - lx1;
            const float4* z = zs[k] + (long)b * S * S * C4 + c;
#pragma unroll
            for (int iy = 0; iy < S; ++iy) {
                const float4 v0 = z[(iy * S + ix0) * C4], v1 = z[(iy * S + ix1) * C4];
                tx[base[k] + iy] = make_float4(lx0 * v0.x + lx1 * v1.x, lx0 * v0.y + lx1 * v1.y, lx0 * v0.z + lx1 * v1.z,
                                               lx0 * v0.w + lx1 * v1.w);
            }
        }
        float4* o = out + ((long)b * h * w + ox) * C4 + c;
        for (int oy = 0; oy < h; ++oy) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int S = ss[k];
                const float fy = src_index(oy, (float)S / (float)h, false);
                const int iy0 = (int)fy;
                const int iy1 = iy0 + (iy0 < S - 1 ? 1 : 0);
                const float ly1 = fy - (float)iy0, ly0 = 1.f - ly1;
                float4 t0 = tx[base[k]], t1 = tx[base[k]];
#pragma unroll
                for (int r = 0; r < S; ++r) {       // register select instead of a runtime-indexed array (scratch)
                    if (r == iy0) t0 = tx[base[k] + r];
                    if (r == iy1) t1 = tx[base[k] + r];
                }
                acc.x += ly0 * t0.x + ly1 * t1.x;
                acc.y += ly0 * t0.y + ly1 * t1.y;
                acc.z += ly0 * t0.z + ly1 * t1.z;
                acc.w += ly0 * t0.w + ly1 * t1.w;
            }
            o[(long)oy * w * C4] = acc;
        }
    }
}

}  // namespace

extern "C" int ape_maxpool3x3s2_nhwc_f32(const float* x, float* y, int B, int H, int W, int C, void* stream)
{
    if (!x || !y || B < 0 || H < 1 || W < 1 || C < 4 || C % 4) return APE_EINVAL;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long total = (long)B * Ho * Wo * (C / 4);
    if (total == 0) return APE_OK;
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream,
                       (const float4*)x, (float4*)y, B, H, W, C / 4, Ho, Wo);
    return ape::check_launch("ape_maxpool3x3s2_nhwc_f32");
}

extern "C" int ape_adaptive_avgpool_nhwc_f32(const float* x, float* y, int B, int H, int W, int C, int S, void* stream)
{
    if (!x || !y || B < 0 || H < 1 || W < 1 || C < 4 || C % 4 || S < 1) return APE_EINVAL;
    if (B == 0) return APE_OK;
    hipLaunchKernelGGL(adaptive_avgpool_kernel, dim3(B * S * S, ape::ceil_div(C / 4, 64)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)x, (float4*)y, H, W, C / 4, S);
    return ape::check_launch("ape_adaptive_avgpool_nhwc_f32");
}

// atom edges of one axis for the given pool sizes; returns the atom count or -1 when it exceeds kPoolMaxAtoms
static int pool_axis_plan(int L, const int* sizes, int nsizes, int* edges, int (*first)[kPoolMaxS + 1], int (*end)[kPoolMaxS])
{
    int cuts[2 * kPoolMaxSizes * kPoolMaxS + 2], n = 0;
    for (int i = 0; i < nsizes; ++i)
        for (int o = 0; o < sizes[i]; ++o) {
            cuts[n++] = (o * L) / sizes[i];
            cuts[n++] = ((o + 1) * L + sizes[i] - 1) / sizes[i];
        }
    for (int i = 1; i < n; ++i)                         // insertion sort + unique
        for (int j = i; j > 0 && cuts[j - 1] > cuts[j]; --j) { const int t = cuts[j]; cuts[j] = cuts[j - 1]; cuts[j - 1] = t; }
    int m = 0;
    for (int i = 0; i < n; ++i)
        if (m == 0 || cuts[i] != edges[m - 1]) {
            if (m > kPoolMaxAtoms) return -1;
            edges[m++] = cuts[i];
        }
    const int atoms = m - 1;
    if (atoms < 1 || atoms > kPoolMaxAtoms) return -1;
    for (int i = 0; i < nsizes; ++i)
        for (int o = 0; o < sizes[i]; ++o) {
            const int lo = (o * L) / sizes[i], hi = ((o + 1) * L + sizes[i] - 1) / sizes[i];
            int a0 = 0, a1 = 0;
            while (edges[a0] != lo) ++a0;
            while (edges[a1] != hi) ++a1;
            first[i][o] = a0;
            end[i][o] = a1;
        }
    return atoms;
}

extern "C" size_t ape_adaptive_avgpool_multi_workspace_bytes(int B, int C)
{
    return (size_t)(B < 0 ? 0 : B) * kPoolMaxAtoms * kPoolMaxAtoms * (size_t)(C < 0 ? 0 : C) * sizeof(float);
}

/* nn.AdaptiveAvgPool2d((S_i, S_i)) for nsizes <= 4 sizes S_i <= 8 of the same map in one pass: x[B][H][W][C] -> ys[i][B][S_i][S_i][C].
 * Returns APE_EINVAL when the bin edges give more than 12 atoms per axis (call ape_adaptive_avgpool_nhwc_f32 per size then). */
extern "C" int ape_adaptive_avgpool_multi_nhwc_f32(const float* x, float* const* ys_host, const int* sizes_host, int nsizes, int B, int H,
                                                   int W, int C, void* workspace, size_t workspace_bytes, void* stream)
{
    return ape_adaptive_avgpool_multi_nhwc_fmt(x, APE_FMT_F32, ys_host, sizes_host, nsizes, B, H, W, C, workspace, workspace_bytes, stream);
}

/* the same with x in either activation format (APE_FMT_S32: C % 32 == 0); the pooled outputs are fp32 */
extern "C" int ape_adaptive_avgpool_multi_nhwc_fmt(const void* x, int in_fmt, float* const* ys_host, const int* sizes_host, int nsizes, int B, int H,
                                                   int W, int C, void* workspace, size_t workspace_bytes, void* stream)
{
    return ape_adaptive_avgpool_multi_nhwc_ld(x, in_fmt, ys_host, sizes_host, nsizes, B, H, W, C, C, workspace, workspace_bytes, stream);
}

/* ... of the first C channels of a map with ldx channels per pixel (ldx >= C; APE_FMT_S32: both % 32 == 0); outputs [B,s,s,C] */
extern "C" int ape_adaptive_avgpool_multi_nhwc_ld(const void* x, int in_fmt, float* const* ys_host, const int* sizes_host, int nsizes, int B, int H,
                                                  int W, int C, int ldx, void* workspace, size_t workspace_bytes, void* stream)
{
    if ((in_fmt != APE_FMT_F32 && in_fmt != APE_FMT_S32) || (in_fmt == APE_FMT_S32 && (C % 32 || ldx % 32))) return APE_EINVAL;
    if (!x || !ys_host || !sizes_host || !workspace || nsizes < 1 || nsizes > kPoolMaxSizes || B < 0 || H < 1 || W < 1 || C < 4 || C % 4 ||
        ldx < C || ldx % 4)
        return APE_EINVAL;
    if (B == 0) return APE_OK;
    PoolPlan pl;
    pl.nsizes = nsizes;
    int bins_total = 0;
    for (int i = 0; i < nsizes; ++i) {
        if (sizes_host[i] < 1 || sizes_host[i] > kPoolMaxS || !ys_host[i]) return APE_EINVAL;
        pl.S[i] = sizes_host[i];
        pl.out[i] = ys_host[i];
        bins_total += sizes_host[i] * sizes_host[i];
    }
    for (int i = nsizes; i < kPoolMaxSizes; ++i) { pl.S[i] = 1 << 20; pl.out[i] = nullptr; }
    pl.ny = pool_axis_plan(H, sizes_host, nsizes, pl.ye, pl.ya, pl.yb);
    pl.nx = pool_axis_plan(W, sizes_host, nsizes, pl.xe, pl.xa, pl.xb);
    if (pl.ny < 0 || pl.nx < 0) return APE_EINVAL;
    if (workspace_bytes < (size_t)B * pl.ny * pl.nx * C * sizeof(float)) return APE_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (in_fmt == APE_FMT_S32)
        hipLaunchKernelGGL(avgpool_atoms_kernel<true>, dim3(B * pl.ny, ape::ceil_div(C / 4, 64)), dim3(256), 0, st, (const float4*)x,
                           (float4*)workspace, pl, H, W, C / 4, ldx / 4);
    else
        hipLaunchKernelGGL(avgpool_atoms_kernel<false>, dim3(B * pl.ny, ape::ceil_div(C / 4, 64)), dim3(256), 0, st, (const float4*)x,
                           (float4*)workspace, pl, H, W, C / 4, ldx / 4);
    const long total = (long)B * bins_total * (C / 4);
    hipLaunchKernelGGL(avgpool_bins_kernel, dim3(grid_for(total)), dim3(kThreads), 0, st, (const float4*)workspace, pl, B, H, W, C / 4,
                       bins_total);
    return ape::check_launch("ape_adaptive_avgpool_multi_nhwc_f32");
}

extern "C" int ape_bilinear_nhwc_f32(const float* x, float* y, int B, int H, int W, int C, int ldx, int Ho, int Wo, int ldy,
                                     int yoff, int align_corners, int accumulate, void* stream)
{
    if (!x || !y || B < 0 || H < 1 || W < 1 || Ho < 1 || Wo < 1 || C < 4 || C % 4 || ldx % 4 || ldy % 4 || yoff % 4 ||
        C > ldx || yoff + C > ldy)
        return APE_EINVAL;
    float sh, sw;
    if (align_corners) {
        sh = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
        sw = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    } else {
        sh = (float)H / (float)Ho;
        sw = (float)W / (float)Wo;
    }
    const long total = (long)B * Ho * Wo * (C / 4);
    if (total == 0) return APE_OK;
    hipLaunchKernelGGL(bilinear_kernel, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream, x, y, B, H, W, C / 4,
                       ldx / 4, Ho, Wo, ldy / 4, yoff / 4, sh, sw, align_corners, accumulate);
    return ape::check_launch("ape_bilinear_nhwc_f32");
}

extern "C" int ape_gather_rows_f32(const float* x, const int64_t* index, float* y, int B, int rows_in, int n, int C, void* stream)
{
    if (!x || !index || !y || B < 0 || rows_in < 1 || n < 0 || C < 4 || C % 4) return APE_EINVAL;
    const long total = (long)B * n * (C / 4);
    if (total == 0) return APE_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream, (const float4*)x,
                       index, (float4*)y, B, rows_in, n, C / 4);
    return ape::check_launch("ape_gather_rows_f32");
}

extern "C" int ape_ups_patch_gather_f32(const float* x, const int64_t* index, float* out, int B, int h, int w, int C, int n, void* stream)
{
    if (!x || !index || !out || B < 0 || h < 1 || w < 1 || C < 4 || C % 4 || n < 0) return APE_EINVAL;
    const long total = (long)B * n * 9 * (C / 4);
    if (total == 0) return APE_OK;
    const float sh = 2 * h > 1 ? (float)(h - 1) / (float)(2 * h - 1) : 0.f;
    const float sw = 2 * w > 1 ? (float)(w - 1) / (float)(2 * w - 1) : 0.f;
    hipLaunchKernelGGL(ups_patch_gather_kernel, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream, (const float4*)x, index,
                       (float4*)out, B, h, w, C / 4, n, sh, sw);
    return ape::check_launch("ape_ups_patch_gather_f32");
}

extern "C" int ape_log_softmax_rows_f32(const float* x, float* y, long rows, int C, void* stream)
{
    if (!x || !y || rows < 0 || C < 1) return APE_EINVAL;
    if (rows == 0) return APE_OK;
    hipLaunchKernelGGL(log_softmax_rows_kernel, dim3(grid_for(rows)), dim3(kThreads), 0, (hipStream_t)stream, x, y, rows, C);
    return ape::check_launch("ape_log_softmax_rows_f32");
}

extern "C" int ape_mean_rows_f32(const float* x, float* y, int B, int n, int C, void* stream)
{
    if (!x || !y || B < 0 || n < 1 || C < 1) return APE_EINVAL;
    if (B == 0) return APE_OK;
    if (C % 4 == 0 && n >= 64)
        hipLaunchKernelGGL(mean_rows_wide_kernel, dim3(ape::ceil_div(C / 4, 64), B), dim3(1024), 0, (hipStream_t)stream,
                           (const float4*)x, (float4*)y, n, C / 4);
    else
        hipLaunchKernelGGL(mean_rows_kernel, dim3(ape::ceil_div(C, 64), B), dim3(256), 0, (hipStream_t)stream, x, y, n, C);
    return ape::check_launch("ape_mean_rows_f32");
}

extern "C" int ape_pad3to4_f32(const float* x, float* y, long rows, void* stream)
{
    if (!x || !y || rows < 0) return APE_EINVAL;
    if (rows == 0) return APE_OK;
    hipLaunchKernelGGL(pad3to4_kernel, dim3(grid_for(rows)), dim3(kThreads), 0, (hipStream_t)stream, x, (float4*)y, rows);
    return ape::check_launch("ape_pad3to4_f32");
}

extern "C" int ape_head_select_f32(const float* h, int ldh, int off_r, int off_t, int off_c, const float* wr, const float* br,
                                   const float* wt, const float* bt, const float* wc, const float* bc, const int64_t* obj,
                                   float* out, int B, int n, int K, void* stream)
{
    if (!h || !wr || !br || !wt || !bt || (wc && !bc) || !obj || !out || B < 0 || n < 0 || K < 1 || K > 1024) return APE_EINVAL;
    if (B == 0 || n == 0) return APE_OK;
    dim3 grid(ape::ceil_div((long)n * 8, kThreads), B);
    hipLaunchKernelGGL(head_select_kernel, grid, dim3(kThreads), 8 * K * sizeof(float), (hipStream_t)stream, h, ldh, off_r,
                       off_t, off_c, wr, br, wt, bt, wc, bc, obj, out, n, K);
    return ape::check_launch("ape_head_select_f32");
}

extern "C" int ape_upconv3x3_gather_f32(const float* z, const float* bias, float* out, int B, int h, int w, int C, int act, float alpha,
                                        void* stream)
{
    return ape_upconv3x3_gather_fmt(z, bias, out, APE_FMT_F32, B, h, w, C, act, alpha, stream);
}

/* the same with the OUTPUT in either activation format (APE_FMT_S32: C % 32 == 0) */
extern "C" int ape_upconv3x3_gather_fmt(const float* z, const float* bias, void* out, int out_fmt, int B, int h, int w, int C, int act,
                                        float alpha, void* stream)
{
    return ape_upconv3x3_gather_ex(z, bias, out, out_fmt, B, h, w, C, act, alpha, 0, stream);
}

// The strip kernel advances a tap's z row pair on a fixed schedule: the upper source row floor(sh * q) of up-sampled row q may change
// only when q turns odd, and then by one.  True for the x2 align_corners scale in exact arithmetic; checked here in the kernel's own
// float expression (cached per height).
static bool strip_schedule_ok(int h, float sh)
{
    static std::mutex mu;
    static std::map<int, bool> seen;
    std::lock_guard<std::mutex> lock(mu);
    auto it = seen.find(h);
    if (it != seen.end()) return it->second;
    bool ok = true;
    int prev = (int)(sh * 0.f);
    for (int q = 1; q < 2 * h && ok; ++q) {
        const int cur = (int)(sh * (float)q);
        ok = (q & 1) ? (cur == prev || cur == prev + 1) : cur == prev;
        prev = cur;
    }
    seen[h] = ok;
    return ok;
}

static int g_upg_strip_rows = 30;       // output rows per workgroup of the strip walk; 0 = the row kernel for every channel count
/* tuning / test hook: rows per strip (>= 1), 0 switches the strip walk off; returns the previous value */
extern "C" int ape_upconv3x3_gather_strip_rows(int rows)
{
    const int old = g_upg_strip_rows;
    if (rows >= 0) g_upg_strip_rows = rows;
    return old;
}

/* ... and with the choice of the interpolation arithmetic: fma = 0 the separately rounded products of ape_bilinear_nhwc_f32, fma = 1 chained
 * fused multiply-adds (the unfused twin of ape_upconv3x3_fused_* built with fma = 1) */
extern "C" int ape_upconv3x3_gather_ex(const float* z, const float* bias, void* out, int out_fmt, int B, int h, int w, int C, int act,
                                       float alpha, int fma, void* stream)
{
    if ((out_fmt != APE_FMT_F32 && out_fmt != APE_FMT_S32) || (out_fmt == APE_FMT_S32 && C % 32) || (fma != 0 && fma != 1)) return APE_EINVAL;
    if (!z || !out || B < 0 || h < 1 || w < 1 || C < 4 || C % 4 || act < APE_ACT_NONE || act > APE_ACT_PRELU) return APE_EINVAL;
    const long total = (long)B * 4 * h * w * (C / 4);
    if (total == 0) return APE_OK;
    const float sh = 2 * h > 1 ? (float)(h - 1) / (float)(2 * h - 1) : 0.f;
    const float sw = 2 * w > 1 ? (float)(w - 1) / (float)(2 * w - 1) : 0.f;
    if (C % 64 == 0 && g_upg_strip_rows > 0 && strip_schedule_ok(h, sh)) {          // the strip walk (256-byte channel slabs)
        const int rows = (g_upg_strip_rows + 1) & ~1;   // whole row pairs
        const long gs = (long)B * ((2 * h + rows - 1) / rows) * ((2 * w + 15) / 16) * (C / 64);
        if (gs >= (1L << 31)) return APE_EINVAL;
#define APE_UPS_LAUNCH(S32O, FM)                                                                                                           \
    hipLaunchKernelGGL((upconv_gather_strip_kernel<S32O, FM>), dim3((unsigned)gs), dim3(256), 0, (hipStream_t)stream, (const float4*)z,    \
                       bias, (float4*)out, B, h, w, C / 4, sh, sw, act, alpha, rows)
        if (out_fmt == APE_FMT_S32) { if (fma) APE_UPS_LAUNCH(true, true); else APE_UPS_LAUNCH(true, false); }
        else { if (fma) APE_UPS_LAUNCH(false, true); else APE_UPS_LAUNCH(false, false); }
#undef APE_UPS_LAUNCH
        return ape::check_launch("ape_upconv3x3_gather_f32");
    }
    const long g = (long)B * 2 * h * ((2 * w + 15) / 16);
    if (g >= (1L << 31)) return APE_EINVAL;
    const size_t lds = (size_t)3 * kUpCols * (C / 4) * sizeof(float4);
    if (lds > 64 * 1024) return APE_EINVAL;             // C <= 1364
#define APE_UPG_LAUNCH(S32O, FM)                                                                                                          \
    hipLaunchKernelGGL((upconv_gather_kernel<S32O, FM>), dim3((unsigned)g), dim3(256), lds, (hipStream_t)stream, (const float4*)z, bias, \
                       (float4*)out, B, h, w, C / 4, sh, sw, act, alpha)
    if (out_fmt == APE_FMT_S32) { if (fma) APE_UPG_LAUNCH(true, true); else APE_UPG_LAUNCH(true, false); }
    else { if (fma) APE_UPG_LAUNCH(false, true); else APE_UPG_LAUNCH(false, false); }
#undef APE_UPG_LAUNCH
    return ape::check_launch("ape_upconv3x3_gather_f32");
}

extern "C" int ape_psp_prior_sum_f32(const float* z1, const float* z2, const float* z3, const float* z6, float* out, int B, int h,
                                     int w, int C, void* stream)
{
    if (!z1 || !z2 || !z3 || !z6 || !out || B < 0 || h < 1 || w < 1 || C < 4 || C % 4) return APE_EINVAL;
    const long total = (long)B * w * (C / 4);      // one thread per (image, column, 4 channels)
    if (total == 0) return APE_OK;
    hipLaunchKernelGGL(psp_prior_sum_kernel, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream, (const float4*)z1,
                       (const float4*)z2, (const float4*)z3, (const float4*)z6, (float4*)out, B, h, w, C / 4);
    return ape::check_launch("ape_psp_prior_sum_f32");
}
