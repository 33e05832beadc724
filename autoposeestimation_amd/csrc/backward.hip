// Backward kernels of the DenseFusion training step (DenseFusion/tools/train.py:205-238:
// estimator / refiner forward -> Loss / Loss_refine -> .backward() -> optimizer.step()).  The reference gets these from
// torch autograd over cuDNN; here every gradient is an explicit kernel behind the C ABI and the host side only keeps the tape.
//
//   weight gradient of a convolution     implicit GEMM dW[co][ky][kx][ci] = sum_p dY[p][co] * X[pix(p) + tap][ci] on the exact-fp32
//                                        matrix cores (v_mfma_f32_32x32x2_f32), split over pixel ranges into a workspace and
//                                        reduced in a fixed order (deterministic, no float atomics)
//   input gradient of a convolution      the forward kernel with the flipped / transposed weights (host side), nothing here
//   activation, bias, pooling, resize, log-softmax, gather, mean: one small streaming kernel each
//   Adam                                 torch.optim.Adam's update rule on flat fp32 buffers
// All tensors are NHWC fp32 as in the forward path.
#include "common.h"

namespace {

constexpr int kT = 256;
static inline int grid_for(long n) { long g = (n + kT - 1) / kT; return (int)(g < 1 ? 1 : (g > 65535 ? 65535 : g)); }

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- convolution weight gradient ---------------------------------------------------------------------------------------
// workgroup tile: 64 output channels x 64 weight columns (column n = tap * Cin + ci), 16 pixels per step, 4 waves each a
// 32 x 32 MFMA tile.  grid (column tiles, channel tiles, pixel splits).
constexpr int WG_M = 64, WG_N = 64, WG_P = 16;

__global__ __launch_bounds__(kT) void conv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                        float* __restrict__ ws, ape_conv_params p, int ncols, long npix,
                                                        long pix_per_split)
{
    __shared__ __attribute__((aligned(16))) float As[WG_P][WG_M + 4];   // dY^T chunk: [pixel][cout]
    __shared__ __attribute__((aligned(16))) float Bs[WG_P][WG_N + 4];   // im2col chunk: [pixel][column]
    const int n0 = blockIdx.x * WG_N, m0 = blockIdx.y * WG_M;
    const long pbeg = (long)blockIdx.z * pix_per_split;
    long pend = pbeg + pix_per_split;
    pend = pend > npix ? npix : pend;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lpx = tid >> 4, l4 = (tid & 15) * 4;          // this thread stages pixel lpx, 4 consecutive channels / columns
    // column -> (tap, ci): Cin is a multiple of 4, so the 4 columns of a thread share their tap
    const int ncol = n0 + l4;
    const bool col_ok = ncol < ncols;
    const int tap = col_ok ? ncol / p.Cin : 0, ci = col_ok ? ncol - tap * p.Cin : 0;
    const int ky = tap / p.KW, kx = tap - ky * p.KW;
    const bool m_ok = m0 + l4 < p.Cout;                      // Cout may be any value: the tail is guarded per element below
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int HoWo = p.Ho * p.Wo;
    for (long p0 = pbeg; p0 < pend; p0 += WG_P) {
        const long pp = p0 + lpx;
        float4 av = make_float4(0.f, 0.f, 0.f, 0.f), bv = av;
        if (pp < pend) {
            if (m_ok) {
                const float* src = dy + pp * p.ldy + p.yoff + m0 + l4;
                const int left = p.Cout - (m0 + l4);
                av.x = src[0];
                if (left > 1) av.y = src[1];
                if (left > 2) av.z = src[2];
                if (left > 3) av.w = src[3];
            }
            if (col_ok) {
                const int b = (int)(pp / HoWo), rem = (int)(pp - (long)b * HoWo);
                const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
                const int iy = oy * p.stride - p.pad + ky * p.dil, ix = ox * p.stride - p.pad + kx * p.dil;
                if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
                    bv = *reinterpret_cast<const float4*>(x + (((long)b * p.H + iy) * p.W + ix) * p.ldx + p.xoff + ci);
            }
        }
        __syncthreads();
        *reinterpret_cast<float4*>(&As[lpx][l4]) = av;
        *reinterpret_cast<float4*>(&Bs[lpx][l4]) = bv;
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < WG_P / 2; ++kk) {
            const float a = As[kk * 2 + (lane >> 5)][wm * 32 + (lane & 31)];
            const float b = Bs[kk * 2 + (lane >> 5)][wn * 32 + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }
    // C/D layout of 32x32x2: element i of lane l is row (i/4)*8 + (l/32)*4 + i%4, column l%32
    float* out = ws + (long)blockIdx.z * p.Cout * ncols;
    const int col = n0 + wn * 32 + (lane & 31);
    if (col < ncols) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = m0 + wm * 32 + (i >> 2) * 8 + (lane >> 5) * 4 + (i & 3);
            if (row < p.Cout) out[(long)row * ncols + col] = acc[i];
        }
    }
}

__global__ void split_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out, long n, int splits)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float s = ws[i];
        for (int k = 1; k < splits; ++k) s += ws[(long)k * n + i];
        out[i] = s;
    }
}

// the same fixed-order sum, written in the PARAMETER layout [Cout][cin][taps] (ws: [split][Cout][taps][cx])
__global__ void split_reduce_param_kernel(const float* __restrict__ ws, float* __restrict__ out, int cout, int cin, int taps, int cx, int splits)
{
    const long n = (long)cout * cin * taps, nws = (long)cout * taps * cx;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int t = (int)(i % taps);
        const long r = i / taps;
        const int ci = (int)(r % cin);
        const long co = r / cin;
        const long j = (co * taps + t) * cx + ci;
        float s = ws[j];
        for (int k = 1; k < splits; ++k) s += ws[(long)k * nws + j];
        out[i] = s;
    }
}

static int wgrad_splits(const ape_conv_params& p, int ncols, long npix)
{
    const long tiles = (long)ape::ceil_div(ncols, WG_N) * ape::ceil_div(p.Cout, WG_M);
    long s = (1024 + tiles - 1) / tiles;
    const long smax = (npix + 63) / 64;
    s = s > smax ? smax : s;
    s = s > 64 ? 64 : s;
    return (int)(s < 1 ? 1 : s);
}

// ---- element-wise / small kernels ----------------------------------------------------------------------------------------
// dx = dy * act'(.) ; ReLU and sigmoid use the OUTPUT y, PReLU the INPUT x (ref holds whichever applies)
__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ ref, float* __restrict__ dx, long n, int act,
                               float alpha)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float g = dy[i], r = ref[i];
        float o;
        switch (act) {
            case APE_ACT_RELU: o = r > 0.f ? g : 0.f; break;
            case APE_ACT_PRELU: o = r > 0.f ? g : alpha * g; break;
            case APE_ACT_SIGMOID: o = g * r * (1.f - r); break;
            default: o = g;
        }
        dx[i] = o;
    }
}

__global__ void prelu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n, float alpha)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = x[i];
        y[i] = v > 0.f ? v : alpha * v;
    }
}

// partial[block] = sum over the block's elements of dy * x * [x <= 0]   (d/d alpha of PReLU); fixed-order two-stage sum
__global__ __launch_bounds__(kT) void prelu_dalpha_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                          float* __restrict__ partial, long n)
{
    __shared__ float red[4];
    float s = 0.f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = x[i];
        if (!(v > 0.f)) s += dy[i] * v;
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

__global__ void sum_small_kernel(const float* __restrict__ partial, float* __restrict__ out, int n)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < n; ++i) s += partial[i];
        out[0] = s;
    }
}

// column sums of x[rows][ld] (channels off..off+C): grid (C/64, row groups); part[g][C]; then reduced in order
__global__ __launch_bounds__(kT) void colsum_kernel(const float* __restrict__ x, float* __restrict__ part, long rows, int C, int ld,
                                                    int off, long rows_per_group)
{
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    const long r0 = (long)blockIdx.y * rows_per_group;
    long r1 = r0 + rows_per_group;
    r1 = r1 > rows ? rows : r1;
    float s = 0.f;
    if (c < C)
        for (long r = r0 + g; r < r1; r += 4) s += x[r * ld + off + c];
    red[g][threadIdx.x & 63] = s;
    __syncthreads();
    if (g == 0 && c < C)
        part[(long)blockIdx.y * C + c] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// max-pool 3x3 stride 2 pad 1 backward in gather form: an input pixel receives dy of every window whose FIRST maximum (row-major
// scan of the window, ATen's tie rule: `val > maxval`) is that pixel.
__global__ void maxpool3x3s2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, int B, int H,
                                        int W, int C, int Ho, int Wo)
{
    const long total = (long)B * H * W * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = i % C;
        long t = i / C;
        const int ix = t % W; t /= W;
        const int iy = t % H;
        const int b = t / H;
        float g = 0.f;
        // windows (oy, ox) with oy*2-1 <= iy <= oy*2+1
        for (int oy = (iy + 1) / 2 - ((iy + 1) % 2 == 0 ? 1 : 0); oy <= (iy + 1) / 2; ++oy) {
            if (oy < 0 || oy >= Ho) continue;
            for (int ox = (ix + 1) / 2 - ((ix + 1) % 2 == 0 ? 1 : 0); ox <= (ix + 1) / 2; ++ox) {
                if (ox < 0 || ox >= Wo) continue;
                float best = -__builtin_inff();
                int by = -1, bx = -1;
                for (int ky = 0; ky < 3; ++ky) {
                    const int yy = oy * 2 - 1 + ky;
                    if ((unsigned)yy >= (unsigned)H) continue;
                    for (int kx = 0; kx < 3; ++kx) {
                        const int xx = ox * 2 - 1 + kx;
                        if ((unsigned)xx >= (unsigned)W) continue;
                        const float v = x[(((long)b * H + yy) * W + xx) * C + c];
                        if (v > best || by < 0) { best = v; by = yy; bx = xx; }
                    }
                }
                if (by == iy && bx == ix) g += dy[(((long)b * Ho + oy) * Wo + ox) * C + c];
            }
        }
        dx[i] = g;
    }
}

// adaptive average pool backward, gather form (bins [floor(o*H/S), ceil((o+1)*H/S)) may overlap by one pixel)
__global__ void adaptive_avgpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int B, int H, int W, int C, int S)
{
    const long total = (long)B * H * W * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = i % C;
        long t = i / C;
        const int ix = t % W; t /= W;
        const int iy = t % H;
        const int b = t / H;
        float g = 0.f;
        const int oy_c = (int)(((long)iy * S) / H), ox_c = (int)(((long)ix * S) / W);
        for (int oy = oy_c - 1; oy <= oy_c + 1; ++oy) {
            if (oy < 0 || oy >= S) continue;
            const int y0 = (oy * H) / S, y1 = ((oy + 1) * H + S - 1) / S;
            if (iy < y0 || iy >= y1) continue;
            for (int ox = ox_c - 1; ox <= ox_c + 1; ++ox) {
                if (ox < 0 || ox >= S) continue;
                const int x0 = (ox * W) / S, x1 = ((ox + 1) * W + S - 1) / S;
                if (ix < x0 || ix >= x1) continue;
                g += dy[(((long)b * S + oy) * S + ox) * C + c] / (float)((y1 - y0) * (x1 - x0));
            }
        }
        dx[i] = g;
    }
}

__device__ __forceinline__ float src_index_b(int dst, float scale, bool align_corners)
{
    if (align_corners) return scale * (float)dst;
    const float s = scale * ((float)dst + 0.5f) - 0.5f;
    return s < 0.f ? 0.f : s;
}

// bilinear resize backward: scatter of every output gradient to its four sources (dx zeroed by the caller; fp32 atomics, the
// summation order is not fixed -> last-bit differences between runs)
__global__ void bilinear_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int B, int H, int W, int C, int Ho, int Wo,
                                    float sh, float sw, int align_corners)
{
    const long total = (long)B * Ho * Wo * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = i % C;
        long t = i / C;
        const int ox = t % Wo; t /= Wo;
        const int oy = t % Ho;
        const int b = t / Ho;
        const float fy = src_index_b(oy, sh, align_corners), fx = src_index_b(ox, sw, align_corners);
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
        const float ly1 = fy - (float)y0, lx1 = fx - (float)x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
        const float g = dy[i];
        float* base = dx + (long)b * H * W * C + c;
        atomicAdd(base + ((long)y0 * W + x0) * C, g * ly0 * lx0);
        atomicAdd(base + ((long)y0 * W + x1) * C, g * ly0 * lx1);
        atomicAdd(base + ((long)y1 * W + x0) * C, g * ly1 * lx0);
        atomicAdd(base + ((long)y1 * W + x1) * C, g * ly1 * lx1);
    }
}

// log-softmax backward per row: dx = dy - exp(y) * sum(dy)
__global__ void log_softmax_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx, long rows, int C)
{
    for (long r = blockIdx.x * (long)blockDim.x + threadIdx.x; r < rows; r += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += dy[r * C + c];
        for (int c = 0; c < C; ++c) dx[r * C + c] = dy[r * C + c] - expf(y[r * C + c]) * s;
    }
}

// gather_rows backward: dx[b][index[b][j]][:] += dy[b][j][:]  (dx zeroed by the caller; indices may repeat -> atomics)
__global__ void scatter_add_rows_kernel(const float* __restrict__ dy, const int64_t* __restrict__ index, float* __restrict__ dx, int B,
                                        int rows_in, int n, int C)
{
    const long total = (long)B * n * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = i % C;
        const long r = i / C;
        const int b = r / n;
        const long src = index[r];
        if (src < 0 || src >= rows_in) continue;
        atomicAdd(dx + ((long)b * rows_in + src) * C + c, dy[i]);
    }
}

// mean over rows backward: dx[b][r][c] = dy[b][c] / n
__global__ void mean_rows_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int B, int n, int C)
{
    const long total = (long)B * n * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = i % C;
        const int b = (i / C) / n;
        dx[i] = dy[(long)b * C + c] / (float)n;
    }
}

// torch.optim.Adam (no amsgrad, no weight decay unless wd != 0: grad += wd * p):
//   m = b1 m + (1 - b1) g ; v = b2 v + (1 - b2) g^2 ; p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
                            float lr, float b1, float b2, float eps, float bc1, float bc2_sqrt, float wd)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float gi = g[i];
        if (wd != 0.f) gi += wd * p[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] -= (lr / bc1) * (mi / denom);
    }
}

// the same update for up to kAdamJobs parameter buffers in ONE launch (blockIdx.y = buffer)
constexpr int kAdamJobs = 64;
struct AdamBatch { ape_adam_job j[kAdamJobs]; };
__global__ void adam_multi_kernel(const AdamBatch b, float lr, float b1, float b2, float eps, float wd)
{
    const ape_adam_job& a = b.j[blockIdx.y];
    float* __restrict__ p = a.param;
    const float* __restrict__ g = a.grad;
    float* __restrict__ m = a.exp_avg;
    float* __restrict__ v = a.exp_avg_sq;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < a.n; i += (long)gridDim.x * blockDim.x) {
        float gi = g[i];
        if (wd != 0.f) gi += wd * p[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / a.bc2_sqrt + eps;
        p[i] -= (lr / a.bc1) * (mi / denom);
    }
}

// parameter layout [Cout][Cin][taps] (rows src_ld elements apart: a column block of a wider parameter packs in place) -> the conv kernels' operand layouts, many parameters per launch (blockIdx.y = job):
// f32 [N][taps][C4] and the split-bf16 planes hi [N][Kp] | lo [N][Kp] (K = taps * C4, Kp = K rounded up to 8, zero padded).
// transpose = 0: N = Cout, C = Cin (the forward operand); 1: N = Cin, C = Cout with the taps reversed = the flipped, transposed
// weights whose forward conv is the input gradient.
__global__ void pack_train_weights_kernel(const ape_pack_job* __restrict__ jobs)
{
    const ape_pack_job a = jobs[blockIdx.y];
    const int N = a.transpose ? a.cin : a.cout, C = a.transpose ? a.cout : a.cin;
    const int C4 = (C + 3) / 4 * 4, K = a.taps * C4, Kp = (K + 7) / 8 * 8;
    const long total = (long)N * Kp;
    __bf16* hi = (__bf16*)a.dst_bf16;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % Kp);
        const long n = i / Kp;
        float v = 0.f;
        if (k < K) {
            const int t = k / C4, c = k - t * C4;
            if (c < C) v = a.transpose ? a.src[(long)c * a.src_ld + n * a.taps + (a.taps - 1 - t)] : a.src[n * a.src_ld + (long)c * a.taps + t];
            a.dst_f32[n * K + k] = v;
        }
        if (hi) {
            const __bf16 h = (__bf16)v;
            hi[i] = h;
            hi[total + i] = (__bf16)(v - (float)h);
        }
    }
}

}  // namespace

extern "C" size_t ape_conv2d_wgrad_workspace_bytes(const ape_conv_params* params)
{
    if (!params) return 0;
    const ape_conv_params& p = *params;
    const int ncols = p.KH * p.KW * p.Cin;
    const long npix = (long)p.B * p.Ho * p.Wo;
    return (size_t)wgrad_splits(p, ncols, npix) * p.Cout * ncols * sizeof(float) + 256;
}

static int wgrad_run(const float* x, const float* dy, float* dw, const ape_conv_params* params, int cin_param, void* workspace,
                     size_t workspace_bytes, void* stream, const char* what)
{
    if (!x || !dy || !dw || !params || !workspace) return APE_EINVAL;
    const ape_conv_params& p = *params;
    if (p.B < 0 || p.H < 1 || p.W < 1 || p.Ho < 1 || p.Wo < 1 || p.Cin < 4 || p.Cin % 4 || p.Cout < 1 || p.KH < 1 || p.KW < 1 ||
        p.stride < 1 || p.dil < 1 || p.pad < 0 || p.ldx % 4 || p.xoff % 4 || p.xoff + p.Cin > p.ldx || p.yoff + p.Cout > p.ldy)
        return APE_EINVAL;
    if (cin_param > p.Cin) return APE_EINVAL;
    if ((p.H + 2 * p.pad - p.dil * (p.KH - 1) - 1) / p.stride + 1 != p.Ho || (p.W + 2 * p.pad - p.dil * (p.KW - 1) - 1) / p.stride + 1 != p.Wo)
        return APE_EINVAL;
    if (workspace_bytes < ape_conv2d_wgrad_workspace_bytes(params)) return APE_EWORKSPACE;
    const int ncols = p.KH * p.KW * p.Cin;
    const long npix = (long)p.B * p.Ho * p.Wo;
    const long nw = cin_param > 0 ? (long)p.Cout * cin_param * p.KH * p.KW : (long)p.Cout * ncols;
    hipStream_t st = (hipStream_t)stream;
    if (npix == 0) {
        if (hipMemsetAsync(dw, 0, nw * sizeof(float), st) != hipSuccess) { ape::set_last_error("hipMemsetAsync"); return APE_ELAUNCH; }
        return APE_OK;
    }
    const int splits = wgrad_splits(p, ncols, npix);
    long pps = (npix + splits - 1) / splits;
    pps = (pps + WG_P - 1) / WG_P * WG_P;
    float* ws = (float*)workspace;
    hipLaunchKernelGGL(conv_wgrad_kernel, dim3(ape::ceil_div(ncols, WG_N), ape::ceil_div(p.Cout, WG_M), splits), dim3(kT), 0, st, x, dy,
                       ws, p, ncols, npix, pps);
    if (cin_param > 0)
        hipLaunchKernelGGL(split_reduce_param_kernel, dim3(grid_for(nw)), dim3(kT), 0, st, ws, dw, p.Cout, cin_param, p.KH * p.KW, p.Cin, splits);
    else
        hipLaunchKernelGGL(split_reduce_kernel, dim3(grid_for(nw)), dim3(kT), 0, st, ws, dw, nw, splits);
    return ape::check_launch(what);
}

/* dw[Cout][KH][KW][Cin] (Cin % 4 == 0, the packed forward layout) = sum over the B*Ho*Wo output pixels */
extern "C" int ape_conv2d_wgrad_nhwc_f32(const float* x, const float* dy, float* dw, const ape_conv_params* params, void* workspace,
                                         size_t workspace_bytes, void* stream)
{
    return wgrad_run(x, dy, dw, params, 0, workspace, workspace_bytes, stream, "ape_conv2d_wgrad_nhwc_f32");
}

/* the same sums written as the reference's PARAMETER: dw[Cout][cin_param][KH][KW], cin_param <= Cin (the zero channels that pad x drop out) */
extern "C" int ape_conv2d_wgrad_param_f32(const float* x, const float* dy, float* dw, const ape_conv_params* params, int cin_param,
                                          void* workspace, size_t workspace_bytes, void* stream)
{
    if (cin_param < 1) return APE_EINVAL;
    return wgrad_run(x, dy, dw, params, cin_param, workspace, workspace_bytes, stream, "ape_conv2d_wgrad_param_f32");
}

extern "C" int ape_act_bwd_f32(const float* dy, const float* ref, float* dx, long n, int act, float alpha, void* stream)
{
    if (!dy || !dx || n < 0 || act < APE_ACT_NONE || act > APE_ACT_SIGMOID || (act != APE_ACT_NONE && !ref)) return APE_EINVAL;
    if (n == 0) return APE_OK;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n)), dim3(kT), 0, (hipStream_t)stream, dy, ref ? ref : dy, dx, n, act, alpha);
    return ape::check_launch("ape_act_bwd_f32");
}

extern "C" int ape_prelu_f32(const float* x, float* y, long n, float alpha, void* stream)
{
    if (!x || !y || n < 0) return APE_EINVAL;
    if (n == 0) return APE_OK;
    hipLaunchKernelGGL(prelu_fwd_kernel, dim3(grid_for(n)), dim3(kT), 0, (hipStream_t)stream, x, y, n, alpha);
    return ape::check_launch("ape_prelu_f32");
}

/* dalpha[0] = sum dy * x * [x <= 0]; scratch: >= 1024 floats */
extern "C" int ape_prelu_dalpha_f32(const float* dy, const float* x, float* dalpha, long n, float* scratch1024, void* stream)
{
    if (!dy || !x || !dalpha || !scratch1024 || n < 0) return APE_EINVAL;
    int g = grid_for(n);
    g = g > 1024 ? 1024 : g;
    hipLaunchKernelGGL(prelu_dalpha_kernel, dim3(g), dim3(kT), 0, (hipStream_t)stream, dy, x, scratch1024, n);
    hipLaunchKernelGGL(sum_small_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, scratch1024, dalpha, g);
    return ape::check_launch("ape_prelu_dalpha_f32");
}

/* out[C] = column sums of x[rows][ld] at channel offset off; scratch: >= 64 * C floats */
extern "C" int ape_colsum_f32(const float* x, float* out, long rows, int C, int ld, int off, float* scratch, void* stream)
{
    if (!x || !out || !scratch || rows < 0 || C < 1 || ld < C || off < 0 || off + C > ld) return APE_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (rows == 0) {
        if (hipMemsetAsync(out, 0, (size_t)C * sizeof(float), st) != hipSuccess) { ape::set_last_error("hipMemsetAsync"); return APE_ELAUNCH; }
        return APE_OK;
    }
    int groups = (int)((rows + 255) / 256);
    groups = groups > 64 ? 64 : groups;
    const long rpg = (rows + groups - 1) / groups;
    hipLaunchKernelGGL(colsum_kernel, dim3(ape::ceil_div(C, 64), groups), dim3(kT), 0, st, x, scratch, rows, C, ld, off, rpg);
    hipLaunchKernelGGL(split_reduce_kernel, dim3(grid_for(C)), dim3(kT), 0, st, scratch, out, (long)C, groups);
    return ape::check_launch("ape_colsum_f32");
}

extern "C" int ape_maxpool3x3s2_bwd_nhwc_f32(const float* x, const float* dy, float* dx, int B, int H, int W, int C, void* stream)
{
    if (!x || !dy || !dx || B < 0 || H < 1 || W < 1 || C < 1) return APE_EINVAL;
    const long total = (long)B * H * W * C;
    if (total == 0) return APE_OK;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(maxpool3x3s2_bwd_kernel, dim3(grid_for(total)), dim3(kT), 0, (hipStream_t)stream, x, dy, dx, B, H, W, C, Ho, Wo);
    return ape::check_launch("ape_maxpool3x3s2_bwd_nhwc_f32");
}

extern "C" int ape_adaptive_avgpool_bwd_nhwc_f32(const float* dy, float* dx, int B, int H, int W, int C, int S, void* stream)
{
    if (!dy || !dx || B < 0 || H < 1 || W < 1 || C < 1 || S < 1) return APE_EINVAL;
    const long total = (long)B * H * W * C;
    if (total == 0) return APE_OK;
    hipLaunchKernelGGL(adaptive_avgpool_bwd_kernel, dim3(grid_for(total)), dim3(kT), 0, (hipStream_t)stream, dy, dx, B, H, W, C, S);
    return ape::check_launch("ape_adaptive_avgpool_bwd_nhwc_f32");
}

/* dy[B][Ho][Wo][C] -> dx[B][H][W][C] (overwritten) for y = bilinear resize of x */
extern "C" int ape_bilinear_bwd_nhwc_f32(const float* dy, float* dx, int B, int H, int W, int C, int Ho, int Wo, int align_corners,
                                         void* stream)
{
    if (!dy || !dx || B < 0 || H < 1 || W < 1 || Ho < 1 || Wo < 1 || C < 1) return APE_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const long nin = (long)B * H * W * C, total = (long)B * Ho * Wo * C;
    if (nin == 0) return APE_OK;
    if (hipMemsetAsync(dx, 0, nin * sizeof(float), st) != hipSuccess) { ape::set_last_error("hipMemsetAsync"); return APE_ELAUNCH; }
    float sh, sw;
    if (align_corners) {
        sh = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
        sw = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    } else {
        sh = (float)H / (float)Ho;
        sw = (float)W / (float)Wo;
    }
    hipLaunchKernelGGL(bilinear_bwd_kernel, dim3(grid_for(total)), dim3(kT), 0, st, dy, dx, B, H, W, C, Ho, Wo, sh, sw, align_corners);
    return ape::check_launch("ape_bilinear_bwd_nhwc_f32");
}

extern "C" int ape_log_softmax_bwd_rows_f32(const float* dy, const float* y, float* dx, long rows, int C, void* stream)
{
    if (!dy || !y || !dx || rows < 0 || C < 1) return APE_EINVAL;
    if (rows == 0) return APE_OK;
    hipLaunchKernelGGL(log_softmax_bwd_kernel, dim3(grid_for(rows)), dim3(kT), 0, (hipStream_t)stream, dy, y, dx, rows, C);
    return ape::check_launch("ape_log_softmax_bwd_rows_f32");
}

/* dx[B][rows_in][C] (overwritten) = scatter-add of dy[B][n][C] at index[B][n] */
extern "C" int ape_scatter_add_rows_f32(const float* dy, const int64_t* index, float* dx, int B, int rows_in, int n, int C, void* stream)
{
    if (!dy || !index || !dx || B < 0 || rows_in < 1 || n < 0 || C < 1) return APE_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const long nin = (long)B * rows_in * C;
    if (nin == 0) return APE_OK;
    if (hipMemsetAsync(dx, 0, nin * sizeof(float), st) != hipSuccess) { ape::set_last_error("hipMemsetAsync"); return APE_ELAUNCH; }
    const long total = (long)B * n * C;
    if (total == 0) return APE_OK;
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(grid_for(total)), dim3(kT), 0, st, dy, index, dx, B, rows_in, n, C);
    return ape::check_launch("ape_scatter_add_rows_f32");
}

extern "C" int ape_mean_rows_bwd_f32(const float* dy, float* dx, int B, int n, int C, void* stream)
{
    if (!dy || !dx || B < 0 || n < 1 || C < 1) return APE_EINVAL;
    const long total = (long)B * n * C;
    if (total == 0) return APE_OK;
    hipLaunchKernelGGL(mean_rows_bwd_kernel, dim3(grid_for(total)), dim3(kT), 0, (hipStream_t)stream, dy, dx, B, n, C);
    return ape::check_launch("ape_mean_rows_bwd_f32");
}

/* one Adam update of a flat fp32 parameter buffer; step = 1, 2, ... (bias correction) */
extern "C" int ape_adam_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1,
                                 float beta2, float eps, int step, float weight_decay, void* stream)
{
    if (!param || !grad || !exp_avg || !exp_avg_sq || n < 0 || step < 1 || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f))
        return APE_EINVAL;
    if (n == 0) return APE_OK;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(kT), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, lr, beta1,
                       beta2, eps, (float)bc1, (float)sqrt(bc2), weight_decay);
    return ape::check_launch("ape_adam_step_f32");
}

/* ape_adam_step_f32 for n parameter buffers, 64 per launch; bc1 = 1 - beta1^step, bc2_sqrt = sqrt(1 - beta2^step) per buffer (its own step) */
extern "C" int ape_adam_step_multi_f32(int n, const ape_adam_job* jobs, float lr, float beta1, float beta2, float eps, float weight_decay,
                                       void* stream)
{
    if (n < 0 || (n && !jobs) || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f)) return APE_EINVAL;
    for (int i = 0; i < n; ++i)
        if (!jobs[i].param || !jobs[i].grad || !jobs[i].exp_avg || !jobs[i].exp_avg_sq || jobs[i].n < 0 || !(jobs[i].bc1 > 0.f) || !(jobs[i].bc2_sqrt > 0.f))
            return APE_EINVAL;
    for (int i0 = 0; i0 < n; i0 += kAdamJobs) {
        AdamBatch b{};
        const int nb = n - i0 < kAdamJobs ? n - i0 : kAdamJobs;
        long nmax = 1;
        for (int i = 0; i < nb; ++i) { b.j[i] = jobs[i0 + i]; nmax = jobs[i0 + i].n > nmax ? jobs[i0 + i].n : nmax; }
        int gx = grid_for(nmax);
        gx = gx > 256 ? 256 : gx;
        hipLaunchKernelGGL(adam_multi_kernel, dim3(gx, nb), dim3(kT), 0, (hipStream_t)stream, b, lr, beta1, beta2, eps, weight_decay);
    }
    return ape::check_launch("ape_adam_step_multi_f32");
}

/* n jobs (DEVICE array) of parameter -> conv operand repacking in one launch; max_elems = the largest N * Kp among them */
extern "C" int ape_pack_train_weights(int n, const ape_pack_job* jobs_device, long max_elems, void* stream)
{
    if (n < 0 || (n && !jobs_device) || max_elems < 0) return APE_EINVAL;
    if (n == 0 || max_elems == 0) return APE_OK;
    int gx = grid_for(max_elems);
    gx = gx > 128 ? 128 : gx;
    for (int i0 = 0; i0 < n; i0 += 65535)
        hipLaunchKernelGGL(pack_train_weights_kernel, dim3(gx, n - i0 < 65535 ? n - i0 : 65535), dim3(kT), 0, (hipStream_t)stream, jobs_device + i0);
    return ape::check_launch("ape_pack_train_weights");
}
