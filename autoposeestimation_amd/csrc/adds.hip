// ADD / ADD-S distance and the DenseFusion loss forward on the device.
//   DenseFusion/lib/loss.py:12-73 (loss_calculation), loss_refiner.py:12-64, tools/eval_linemod.py:118-130.
//
// The reference materialises model_points and target N times (`repeat`, loss.py:32-33: two N x M x 3 tensors, 24 MB at
// 1000 x 1000), runs one bmm, and for symmetric objects sends all N*M predicted points through the k-NN extension
// (10^6 queries x 1000 refs, through a 4 GB distance matrix on its CUDA path).  Here one workgroup owns one predicted
// pose n: it rotates the M model points in registers, keeps the M targets in LDS as float4, and for symmetric objects
// scans them with the SAME float32 distance arithmetic and tie rule as the k-NN kernel (knn.hip) -- no distance matrix,
// no repeated tensors; `pred` is written only if the caller asks for it.
//
// Work: N*M point transforms + (symmetric ? N*M*M : N*M) pair evaluations; bytes: 28 N + 24 M in, 8 N out (+12 N M if
// `pred` is requested) => fp32-VALU bound for symmetric objects, launch-latency bound otherwise.
#include "common.h"

namespace {

constexpr int kT = 256;

__device__ __forceinline__ void quat_base(float w, float x, float y, float z, float R[9])
{   // loss.py:20-28 -- the nine terms as the reference spells them (ori_base, row major)
    R[0] = 1.0f - 2.0f * (y * y + z * z);
    R[1] = 2.0f * x * y - 2.0f * w * z;
    R[2] = 2.0f * w * y + 2.0f * x * z;
    R[3] = 2.0f * x * y + 2.0f * z * w;
    R[4] = 1.0f - 2.0f * (x * x + z * z);
    R[5] = -2.0f * w * x + 2.0f * y * z;
    R[6] = -2.0f * w * y + 2.0f * x * z;
    R[7] = 2.0f * w * x + 2.0f * y * z;
    R[8] = 1.0f - 2.0f * (x * x + y * y);
}

__device__ __forceinline__ void normalise4(const float* r, float q[4])
{
    const float nrm = sqrtf(((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]) + r[3] * r[3]);
    q[0] = r[0] / nrm; q[1] = r[1] / nrm; q[2] = r[2] / nrm; q[3] = r[3] / nrm;
}

__device__ __forceinline__ float block_sum(float v, float* red)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((red[0] + red[1]) + red[2]) + red[3];
}

// one workgroup per pose n
__global__ __launch_bounds__(kT) void adds_dis_kernel(const float* __restrict__ pred_r, const float* __restrict__ pred_t,
                                                      const float* __restrict__ points, const float* __restrict__ model,
                                                      const float* __restrict__ target, int M, int symmetric,
                                                      float* __restrict__ pred_out, float* __restrict__ dis,
                                                      float* __restrict__ stdv)
{
    extern __shared__ float4 tgt[];   // M targets
    __shared__ float red[4];
    const int n = blockIdx.x;
    for (int m = threadIdx.x; m < M; m += kT) tgt[m] = make_float4(target[m * 3], target[m * 3 + 1], target[m * 3 + 2], 0.f);
    float q[4], R[9];
    normalise4(pred_r + (size_t)n * 4, q);
    quat_base(q[0], q[1], q[2], q[3], R);
    float tx = pred_t[n * 3], ty = pred_t[n * 3 + 1], tz = pred_t[n * 3 + 2];
    if (points) { tx += points[n * 3]; ty += points[n * 3 + 1]; tz += points[n * 3 + 2]; }   // points + pred_t (loss.py:38)
    __syncthreads();

    // pass 1: distances (kept in registers when M <= 4*kT, recomputed otherwise) and their sum
    float local[4];
    float s = 0.f;
    for (int it = 0, m = threadIdx.x; m < M; m += kT, ++it) {
        const float mx = model[m * 3], my = model[m * 3 + 1], mz = model[m * 3 + 2];
        // bmm(model_points, base) with base = ori_base^T: pred_j = sum_i model_i * ori_base[j][i]
        const float px = ((mx * R[0] + my * R[1]) + mz * R[2]) + tx;
        const float py = ((mx * R[3] + my * R[4]) + mz * R[5]) + ty;
        const float pz = ((mx * R[6] + my * R[7]) + mz * R[8]) + tz;
        if (pred_out) {
            float* po = pred_out + ((size_t)n * M + m) * 3;
            po[0] = px; po[1] = py; po[2] = pz;
        }
        float4 t = tgt[m];
        if (symmetric) {
            float best = __builtin_inff();
            int bi = 0;
            for (int r = 0; r < M; ++r) {   // same arithmetic and tie rule as knn1_d3 (ref - query, lowest index wins)
                const float4 c = tgt[r];
                const float dx = c.x - px, dy = c.y - py, dz = c.z - pz;
                const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
                if (d < best) { best = d; bi = r; }
            }
            t = tgt[bi];
        }
        const float dx = px - t.x, dy = py - t.y, dz = pz - t.z;
        const float d = sqrtf((dx * dx + dy * dy) + dz * dz);
        if (it < 4) local[it] = d;
        s += d;
    }
    const float mean = block_sum(s, red) / (float)M;
    float v = 0.f;
    if (M <= 4 * kT) {
        for (int it = 0, m = threadIdx.x; m < M; m += kT, ++it) { const float e = local[it] - mean; v += e * e; }
    } else {
        for (int m = threadIdx.x; m < M; m += kT) {   // rare large-M path: recompute
            const float mx = model[m * 3], my = model[m * 3 + 1], mz = model[m * 3 + 2];
            const float px = ((mx * R[0] + my * R[1]) + mz * R[2]) + tx;
            const float py = ((mx * R[3] + my * R[4]) + mz * R[5]) + ty;
            const float pz = ((mx * R[6] + my * R[7]) + mz * R[8]) + tz;
            float4 t = tgt[m];
            if (symmetric) {
                float best = __builtin_inff();
                int bi = 0;
                for (int r = 0; r < M; ++r) {
                    const float4 c = tgt[r];
                    const float dx = c.x - px, dy = c.y - py, dz = c.z - pz;
                    const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
                    if (d < best) { best = d; bi = r; }
                }
                t = tgt[bi];
            }
            const float dx = px - t.x, dy = py - t.y, dz = pz - t.z;
            const float e = sqrtf((dx * dx + dy * dy) + dz * dz) - mean;
            v += e * e;
        }
    }
    const float var = block_sum(v, red) / (float)(M > 1 ? M - 1 : 1);   // torch.std: unbiased
    if (threadIdx.x == 0) {
        dis[n] = mean;
        if (stdv) stdv[n] = sqrtf(var);
    }
}

// single workgroup: which = argmax c (first), loss = mean((dis + 2 std) c - w log c), dis[which], q = r[which], t = t[which](+p[which])
__global__ __launch_bounds__(kT) void adds_select_kernel(const float* __restrict__ dis, const float* __restrict__ stdv,
                                                         const float* __restrict__ pred_c, const float* __restrict__ pred_r,
                                                         const float* __restrict__ pred_t, const float* __restrict__ points,
                                                         int N, float w, float* __restrict__ out /* loss, dis, q[4], t[3] */,
                                                         int* __restrict__ which_out)
{
    __shared__ float red[4];
    __shared__ float sc[kT];
    __shared__ int si[kT];
    float s = 0.f, best = -__builtin_inff();
    int bi = 0x7fffffff;
    for (int n = threadIdx.x; n < N; n += kT) {
        const float c = pred_c[n];
        s += (dis[n] + stdv[n] * 2.f) * c - w * logf(c);
        if (c > best || bi == 0x7fffffff) { best = c; bi = n; }
    }
    const float loss = block_sum(s, red) / (float)N;
    sc[threadIdx.x] = best;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int off = kT / 2; off > 0; off >>= 1) {
        if (threadIdx.x < off) {
            const float oc = sc[threadIdx.x + off];
            const int oi = si[threadIdx.x + off];
            if (oi != 0x7fffffff && (si[threadIdx.x] == 0x7fffffff || oc > sc[threadIdx.x] || (oc == sc[threadIdx.x] && oi < si[threadIdx.x]))) {
                sc[threadIdx.x] = oc;
                si[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int wi = si[0];
        out[0] = loss;
        out[1] = dis[wi];
        for (int i = 0; i < 4; ++i) out[2 + i] = pred_r[(size_t)wi * 4 + i];
        for (int i = 0; i < 3; ++i) out[6 + i] = pred_t[(size_t)wi * 3 + i] + (points ? points[(size_t)wi * 3 + i] : 0.f);
        if (which_out) *which_out = wi;
    }
}

// out[i] = (pts[i] - t) . ori_base(q / |q|)       (loss.py:61-69, loss_refiner.py:51-60)
__global__ void recentre_qt_kernel(const float* __restrict__ pts, const float* __restrict__ qt, float* __restrict__ out, int n)
{
    float q[4], R[9];
    normalise4(qt, q);
    quat_base(q[0], q[1], q[2], q[3], R);
    const float tx = qt[4], ty = qt[5], tz = qt[6];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float dx = pts[i * 3] - tx, dy = pts[i * 3 + 1] - ty, dz = pts[i * 3 + 2] - tz;
        out[i * 3 + 0] = (dx * R[0] + dy * R[3]) + dz * R[6];
        out[i * 3 + 1] = (dx * R[1] + dy * R[4]) + dz * R[7];
        out[i * 3 + 2] = (dx * R[2] + dy * R[5]) + dz * R[8];
    }
}

// Gradient of the DenseFusion loss (loss.py:50-53: mean((dis + 2 std) c - w log c)) or, full == 0, of Loss_refine's `dis`
// (loss_refiner.py:47) with respect to pred_r, pred_t (and pred_c).  One workgroup per pose n:
//   dL/d nrm_nm = coef_n * (1/M + 2 (nrm_nm - dis_n) / ((M-1) std_n))     coef_n = g c_n / N   (full)   |   g / M-weighted mean only (refine)
//   dL/d pred_nm = dL/d nrm_nm * (pred_nm - tgt_nm) / nrm_nm ; tgt is the matched target (recomputed 1-NN for symmetric objects)
//   dL/d t_n = sum_m dL/d pred_nm ;  dL/d R_n = sum_m dL/d pred_nm (x) model_m ;  R(q/|q|) chain rule to pred_r
__global__ __launch_bounds__(kT) void adds_grad_kernel(const float* __restrict__ pred_r, const float* __restrict__ pred_t,
                                                       const float* __restrict__ points, const float* __restrict__ model,
                                                       const float* __restrict__ target, const float* __restrict__ pred_c,
                                                       const float* __restrict__ dis, const float* __restrict__ stdv,
                                                       const float* __restrict__ gscale, int N, int M, int symmetric, int full, float w,
                                                       float* __restrict__ d_r, float* __restrict__ d_t, float* __restrict__ d_c)
{
    extern __shared__ float4 tgt[];
    __shared__ float red[4];
    const int n = blockIdx.x;
    for (int m = threadIdx.x; m < M; m += kT) tgt[m] = make_float4(target[m * 3], target[m * 3 + 1], target[m * 3 + 2], 0.f);
    const float* rp = pred_r + (size_t)n * 4;
    const float rn = sqrtf(((rp[0] * rp[0] + rp[1] * rp[1]) + rp[2] * rp[2]) + rp[3] * rp[3]);
    float q[4], R[9];
    normalise4(rp, q);
    quat_base(q[0], q[1], q[2], q[3], R);
    float tx = pred_t[n * 3], ty = pred_t[n * 3 + 1], tz = pred_t[n * 3 + 2];
    if (points) { tx += points[n * 3]; ty += points[n * 3 + 1]; tz += points[n * 3 + 2]; }
    const float g = gscale[0];
    const float mean = dis[n];
    const float sd = full ? stdv[n] : 0.f;
    const float coef = full ? g * pred_c[n] / (float)N : g;
    const float a = 1.f / (float)M;
    const float bcoef = (full && M > 1 && sd > 0.f) ? 2.f / ((float)(M - 1) * sd) : 0.f;
    __syncthreads();
    float acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = 0.f;
    for (int m = threadIdx.x; m < M; m += kT) {
        const float mx = model[m * 3], my = model[m * 3 + 1], mz = model[m * 3 + 2];
        const float px = ((mx * R[0] + my * R[1]) + mz * R[2]) + tx;
        const float py = ((mx * R[3] + my * R[4]) + mz * R[5]) + ty;
        const float pz = ((mx * R[6] + my * R[7]) + mz * R[8]) + tz;
        float4 t = tgt[m];
        if (symmetric) {
            float best = __builtin_inff();
            int bi = 0;
            for (int r = 0; r < M; ++r) {
                const float4 c = tgt[r];
                const float dx = c.x - px, dy = c.y - py, dz = c.z - pz;
                const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
                if (d < best) { best = d; bi = r; }
            }
            t = tgt[bi];
        }
        const float dx = px - t.x, dy = py - t.y, dz = pz - t.z;
        const float nrm = sqrtf((dx * dx + dy * dy) + dz * dz);
        if (nrm > 0.f) {
            const float k = coef * (a + bcoef * (nrm - mean)) / nrm;
            const float gx = k * dx, gy = k * dy, gz = k * dz;
            acc[0] += gx * mx; acc[1] += gx * my; acc[2] += gx * mz;
            acc[3] += gy * mx; acc[4] += gy * my; acc[5] += gy * mz;
            acc[6] += gz * mx; acc[7] += gz * my; acc[8] += gz * mz;
            acc[9] += gx; acc[10] += gy; acc[11] += gz;
        }
    }
    float G[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) G[i] = block_sum(acc[i], red);
    if (threadIdx.x == 0) {
        const float qw = q[0], qx = q[1], qy = q[2], qz = q[3];
        // d R_k / d (w, x, y, z) of quat_base, contracted with G
        const float dw = -2.f * qz * G[1] + 2.f * qy * G[2] + 2.f * qz * G[3] - 2.f * qx * G[5] - 2.f * qy * G[6] + 2.f * qx * G[7];
        const float dxq = 2.f * qy * G[1] + 2.f * qz * G[2] + 2.f * qy * G[3] - 4.f * qx * G[4] - 2.f * qw * G[5] + 2.f * qz * G[6] +
                          2.f * qw * G[7] - 4.f * qx * G[8];
        const float dyq = -4.f * qy * G[0] + 2.f * qx * G[1] + 2.f * qw * G[2] + 2.f * qx * G[3] + 2.f * qz * G[5] - 2.f * qw * G[6] +
                          2.f * qz * G[7] - 4.f * qy * G[8];
        const float dzq = -4.f * qz * G[0] - 2.f * qw * G[1] + 2.f * qx * G[2] + 2.f * qw * G[3] - 4.f * qz * G[4] + 2.f * qy * G[5] +
                          2.f * qx * G[6] + 2.f * qy * G[7];
        const float dot = ((qw * dw + qx * dxq) + qy * dyq) + qz * dzq;
        d_r[(size_t)n * 4 + 0] = (dw - qw * dot) / rn;
        d_r[(size_t)n * 4 + 1] = (dxq - qx * dot) / rn;
        d_r[(size_t)n * 4 + 2] = (dyq - qy * dot) / rn;
        d_r[(size_t)n * 4 + 3] = (dzq - qz * dot) / rn;
        d_t[n * 3 + 0] = G[9]; d_t[n * 3 + 1] = G[10]; d_t[n * 3 + 2] = G[11];
        if (full && d_c) d_c[n] = g * ((mean + 2.f * sd) - w / pred_c[n]) / (float)N;
    }
}

// ---- the evaluation of a BATCH of objects (ape_adds_dis_batched_f32): one workgroup per object leaves 7/8 of the chip idle at 32 objects, so
// the pair evaluations are spread like the k-NN kernel spreads them -- S lanes per predicted point, (distance, index) merged by wave
// shuffles with "lowest index wins" -- and the per-object mean is a second, tiny launch that adds in adds_dis_kernel's own order (thread t
// sums points t, t + 256, ...; block_sum), so dis[b] is bit for bit what ape_adds_dis_f32 gives for that object alone.
template <int S>
__global__ __launch_bounds__(kT) void adds_points_kernel(const float* __restrict__ pred_r, const float* __restrict__ pred_t,
                                                         const float* __restrict__ model, const float* __restrict__ target, int M,
                                                         int symmetric, float* __restrict__ dist_out)
{
    extern __shared__ float4 tgt[];   // this object's M targets
    const int b = blockIdx.y;
    model += (size_t)b * M * 3;
    target += (size_t)b * M * 3;
    for (int m = threadIdx.x; m < M; m += kT) tgt[m] = make_float4(target[m * 3], target[m * 3 + 1], target[m * 3 + 2], 0.f);
    float q[4], R[9];
    normalise4(pred_r + (size_t)b * 4, q);
    quat_base(q[0], q[1], q[2], q[3], R);
    const float tx = pred_t[b * 3], ty = pred_t[b * 3 + 1], tz = pred_t[b * 3 + 2];
    __syncthreads();
    const int s = threadIdx.x % S;
    const int m = blockIdx.x * (kT / S) + threadIdx.x / S;
    const int mc = m < M ? m : M - 1;
    const float mx = model[mc * 3], my = model[mc * 3 + 1], mz = model[mc * 3 + 2];
    const float px = ((mx * R[0] + my * R[1]) + mz * R[2]) + tx;
    const float py = ((mx * R[3] + my * R[4]) + mz * R[5]) + ty;
    const float pz = ((mx * R[6] + my * R[7]) + mz * R[8]) + tz;
    int bi = mc;
    if (symmetric) {
        float best = __builtin_inff();
        bi = s;
        for (int r = s; r < M; r += S) {
            const float4 c = tgt[r];
            const float dx = c.x - px, dy = c.y - py, dz = c.z - pz;
            const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
            if (d < best) { best = d; bi = r; }
        }
#pragma unroll
        for (int off = S / 2; off > 0; off >>= 1) {
            const float od = __shfl_xor(best, off);
            const int oi = __shfl_xor(bi, off);
            if (od < best || (od == best && oi < bi)) { best = od; bi = oi; }
        }
        if (bi >= M) bi = M - 1;        // (only when M < S: lanes without a ref keep their start index)
    }
    if (s == 0 && m < M) {
        const float4 t = tgt[bi];
        const float dx = px - t.x, dy = py - t.y, dz = pz - t.z;
        dist_out[(size_t)b * M + m] = sqrtf((dx * dx + dy * dy) + dz * dz);
    }
}

__global__ __launch_bounds__(kT) void adds_mean_kernel(const float* __restrict__ dist, int M, float* __restrict__ dis)
{
    __shared__ float red[4];
    const int b = blockIdx.x;
    float s = 0.f;
    for (int m = threadIdx.x; m < M; m += kT) s += dist[(size_t)b * M + m];
    const float mean = block_sum(s, red) / (float)M;
    if (threadIdx.x == 0) dis[b] = mean;
}

}  // namespace

extern "C" int ape_adds_dis_f32(const float* pred_r, const float* pred_t, const float* points, const float* model,
                                const float* target, int N, int M, int symmetric, float* pred_out, float* dis, float* stdv,
                                void* stream)
{
    if (!pred_r || !pred_t || !model || !target || !dis || N < 0 || M < 1 || M > 8192) return APE_EINVAL;   // 16 B * M of LDS
    if (N == 0) return APE_OK;
    hipLaunchKernelGGL(adds_dis_kernel, dim3(N), dim3(kT), (size_t)M * sizeof(float4), (hipStream_t)stream, pred_r, pred_t, points,
                       model, target, M, symmetric, pred_out, dis, stdv);
    return ape::check_launch("ape_adds_dis_f32");
}

extern "C" int ape_adds_dis_batched_f32(const float* pred_r, const float* pred_t, const float* model, const float* target, int B, int M,
                                        int symmetric, float* workspace, float* dis, void* stream)
{
    if (!pred_r || !pred_t || !model || !target || !workspace || !dis || B < 0 || M < 1 || M > 8192) return APE_EINVAL;   // 16 B * M of LDS
    if (B == 0) return APE_OK;
    hipStream_t st = (hipStream_t)stream;
    // lanes per predicted point: enough wavefronts to fill 256 CUs (as ape_knn_f32 chooses S)
    int S = 1;
    while (S < 64 && (long)B * M * S < 256L * 8 * 64 && S * 4 <= M) S *= 4;
    const size_t lds = (size_t)M * sizeof(float4);
    dim3 grid(ape::ceil_div(M, kT / S), B);
    switch (S) {
        case 1: hipLaunchKernelGGL(adds_points_kernel<1>, grid, dim3(kT), lds, st, pred_r, pred_t, model, target, M, symmetric, workspace); break;
        case 4: hipLaunchKernelGGL(adds_points_kernel<4>, grid, dim3(kT), lds, st, pred_r, pred_t, model, target, M, symmetric, workspace); break;
        case 16: hipLaunchKernelGGL(adds_points_kernel<16>, grid, dim3(kT), lds, st, pred_r, pred_t, model, target, M, symmetric, workspace); break;
        default: hipLaunchKernelGGL(adds_points_kernel<64>, grid, dim3(kT), lds, st, pred_r, pred_t, model, target, M, symmetric, workspace); break;
    }
    hipLaunchKernelGGL(adds_mean_kernel, dim3(B), dim3(kT), 0, st, workspace, M, dis);
    return ape::check_launch("ape_adds_dis_batched_f32");
}

extern "C" int ape_adds_select_f32(const float* dis, const float* stdv, const float* pred_c, const float* pred_r,
                                   const float* pred_t, const float* points, int N, float w, float* out9, int* which,
                                   void* stream)
{
    if (!dis || !stdv || !pred_c || !pred_r || !pred_t || !out9 || N < 1) return APE_EINVAL;
    hipLaunchKernelGGL(adds_select_kernel, dim3(1), dim3(kT), 0, (hipStream_t)stream, dis, stdv, pred_c, pred_r, pred_t, points, N,
                       w, out9, which);
    return ape::check_launch("ape_adds_select_f32");
}

extern "C" int ape_recentre_qt_f32(const float* pts, const float* qt7, float* out, int n, void* stream)
{
    if (!pts || !qt7 || !out || n < 0) return APE_EINVAL;
    if (n == 0) return APE_OK;
    int g = ape::ceil_div(n, kT);
    g = g > 1024 ? 1024 : g;
    hipLaunchKernelGGL(recentre_qt_kernel, dim3(g), dim3(kT), 0, (hipStream_t)stream, pts, qt7, out, n);
    return ape::check_launch("ape_recentre_qt_f32");
}

/* Backward of ape_adds_dis_f32 + ape_adds_select_f32 (loss.py:50-53) when full != 0, of Loss_refine's dis (loss_refiner.py:47)
 * when full == 0 (N = 1, pred_c / stdv / d_c unused).  gscale: DEVICE scalar = upstream gradient. */
extern "C" int ape_adds_grad_f32(const float* pred_r, const float* pred_t, const float* points, const float* model,
                                 const float* target, const float* pred_c, const float* dis, const float* stdv, const float* gscale,
                                 int N, int M, int symmetric, int full, float w, float* d_r, float* d_t, float* d_c, void* stream)
{
    if (!pred_r || !pred_t || !model || !target || !dis || !gscale || !d_r || !d_t || N < 0 || M < 1 || M > 8192) return APE_EINVAL;
    if (full && (!pred_c || !stdv || !d_c)) return APE_EINVAL;
    if (N == 0) return APE_OK;
    hipLaunchKernelGGL(adds_grad_kernel, dim3(N), dim3(kT), (size_t)M * sizeof(float4), (hipStream_t)stream, pred_r, pred_t, points,
                       model, target, pred_c, dis, stdv, gscale, N, M, symmetric, full, w, d_r, d_t, d_c);
    return ape::check_launch("ape_adds_grad_f32");
}
