// Pose extraction / re-centring / refinement composition on the device (no host round trips):
//   pose_select   DenseFusion/tools/utils.py:7-18 (my_estimator_prediction) + :43-86 (get_new_points):
//                 which = argmax_n c (first maximum), q = r[which]/|r[which]|, t = points[which] + t[which],
//                 new_points = (points - t) . R(q)   with R's nine terms exactly as written at utils.py:46-67
//   pose_compose  tools/utils.py:20-40 (my_refined_prediction) with lib/transformations.py:1254-1278
//                 (quaternion_matrix) and :1320-1341,1361-1363 (quaternion_from_matrix, isprecise=True): float64
//   pose_recentre DenseFusion/tools/eval_ycb.py:205-210 (iterative refinement: cloud re-centred with the CURRENT pose,
//                 R and T rounded to float32 as `.astype(np.float32)` does there)
// The reference pulls 7 floats to the host after the estimator and after the refiner (two device syncs per object);
// here the pose stays in HBM as [B][7] float64 (w,x,y,z, tx,ty,tz) until the caller asks for it.
#include "common.h"

namespace {

struct Mat4 { double m[4][4]; };

__device__ Mat4 quaternion_matrix(const double qin[4])
{
    Mat4 M;
    double q[4] = {qin[0], qin[1], qin[2], qin[3]};
    const double n = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) M.m[i][j] = i == j ? 1.0 : 0.0;
    if (n < 2.220446049250313e-16 * 4.0) return M;
    const double s = sqrt(2.0 / n);
    for (int i = 0; i < 4; ++i) q[i] *= s;
    double o[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) o[i][j] = q[i] * q[j];
    M.m[0][0] = 1.0 - o[2][2] - o[3][3]; M.m[0][1] = o[1][2] - o[3][0]; M.m[0][2] = o[1][3] + o[2][0];
    M.m[1][0] = o[1][2] + o[3][0]; M.m[1][1] = 1.0 - o[1][1] - o[3][3]; M.m[1][2] = o[2][3] - o[1][0];
    M.m[2][0] = o[1][3] - o[2][0]; M.m[2][1] = o[2][3] + o[1][0]; M.m[2][2] = 1.0 - o[1][1] - o[2][2];
    return M;
}

__device__ void quaternion_from_matrix_precise(const Mat4& M, double q[4])
{
    const double t0 = M.m[0][0] + M.m[1][1] + M.m[2][2] + M.m[3][3];
    double t;
    if (t0 > M.m[3][3]) {
        t = t0;
        q[0] = t;
        q[3] = M.m[1][0] - M.m[0][1];
        q[2] = M.m[0][2] - M.m[2][0];
        q[1] = M.m[2][1] - M.m[1][2];
    } else {
        int i = 0, j = 1, k = 2;
        if (M.m[1][1] > M.m[0][0]) { i = 1; j = 2; k = 0; }
        if (M.m[2][2] > M.m[i][i]) { i = 2; j = 0; k = 1; }
        t = M.m[i][i] - (M.m[j][j] + M.m[k][k]) + M.m[3][3];
        double p[4];
        p[i] = t;
        p[j] = M.m[i][j] + M.m[j][i];
        p[k] = M.m[k][i] + M.m[i][k];
        p[3] = M.m[k][j] - M.m[j][k];
        q[0] = p[3]; q[1] = p[0]; q[2] = p[1]; q[3] = p[2];
    }
    const double s = 0.5 / sqrt(t * M.m[3][3]);
    for (int a = 0; a < 4; ++a) q[a] *= s;
    if (q[0] < 0.0)
        for (int a = 0; a < 4; ++a) q[a] = -q[a];
}

// nine terms of the rotation "base" exactly as the reference spells them (float32)
__device__ __forceinline__ void quat_base(float w, float x, float y, float z, float R[9])
{
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

// one workgroup per crop.  heads[b][n][8] = (r0..r3, t0..t2, c); points[b][n][4] (xyz,-)
__global__ __launch_bounds__(256) void pose_select_kernel(const float* __restrict__ heads, const float4* __restrict__ points,
                                                          double* __restrict__ pose, int* __restrict__ which_out,
                                                          float4* __restrict__ new_points, int n)
{
    __shared__ float sc[256];
    __shared__ int si[256];
    __shared__ float sR[9];
    __shared__ float st[3];
    const int b = blockIdx.x;
    heads += (size_t)b * n * 8;
    points += (size_t)b * n;
    float best = -__builtin_inff();
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float c = heads[(size_t)i * 8 + 7];
        if (c > best || bi == 0x7fffffff) { best = c; bi = i; }  // ascending i per thread: first maximum kept
    }
    sc[threadIdx.x] = best;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
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
        const int w = si[0];
        const float* h = heads + (size_t)w * 8;
        const float nrm = sqrtf(((h[0] * h[0] + h[1] * h[1]) + h[2] * h[2]) + h[3] * h[3]);
        const float q0 = h[0] / nrm, q1 = h[1] / nrm, q2 = h[2] / nrm, q3 = h[3] / nrm;
        const float4 p = points[w];
        st[0] = p.x + h[4]; st[1] = p.y + h[5]; st[2] = p.z + h[6];   // (points + pred_t)[which]
        quat_base(q0, q1, q2, q3, sR);
        double* po = pose + (size_t)b * 7;
        po[0] = q0; po[1] = q1; po[2] = q2; po[3] = q3;
        po[4] = st[0]; po[5] = st[1]; po[6] = st[2];
        if (which_out) which_out[b] = w;
    }
    __syncthreads();
    if (new_points) {
        const float tx = st[0], ty = st[1], tz = st[2];
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const float4 p = points[i];
            const float dx = p.x - tx, dy = p.y - ty, dz = p.z - tz;
            float4 o;
            o.x = (dx * sR[0] + dy * sR[3]) + dz * sR[6];
            o.y = (dx * sR[1] + dy * sR[4]) + dz * sR[7];
            o.z = (dx * sR[2] + dy * sR[5]) + dz * sR[8];
            o.w = 0.f;
            new_points[(size_t)b * n + i] = o;
        }
    }
}

__global__ void pose_compose_kernel(double* __restrict__ pose, const float* __restrict__ ref_r, int ldr,
                                    const float* __restrict__ ref_t, int ldt, int B)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double* po = pose + (size_t)b * 7;
    Mat4 A = quaternion_matrix(po);
    A.m[0][3] = po[4]; A.m[1][3] = po[5]; A.m[2][3] = po[6];
    const float* r = ref_r + (size_t)b * ldr;
    const float* t = ref_t + (size_t)b * ldt;
    const float nrm = sqrtf(((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]) + r[3] * r[3]);  // torch.norm in float32
    const double q2[4] = {(double)(r[0] / nrm), (double)(r[1] / nrm), (double)(r[2] / nrm), (double)(r[3] / nrm)};
    Mat4 Bm = quaternion_matrix(q2);
    Bm.m[0][3] = (double)t[0]; Bm.m[1][3] = (double)t[1]; Bm.m[2][3] = (double)t[2];
    Mat4 C;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += A.m[i][k] * Bm.m[k][j];
            C.m[i][j] = s;
        }
    const double tx = C.m[0][3], ty = C.m[1][3], tz = C.m[2][3];
    C.m[0][3] = C.m[1][3] = C.m[2][3] = 0.0;
    double q[4];
    quaternion_from_matrix_precise(C, q);
    po[0] = q[0]; po[1] = q[1]; po[2] = q[2]; po[3] = q[3];
    po[4] = tx; po[5] = ty; po[6] = tz;
}

__global__ void pose_recentre_kernel(const float4* __restrict__ points, const double* __restrict__ pose,
                                     float4* __restrict__ new_points, int n)
{
    const int b = blockIdx.y;
    const double* po = pose + (size_t)b * 7;
    const Mat4 M = quaternion_matrix(po);
    float R[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R[i * 3 + j] = (float)M.m[i][j];
    const float tx = (float)po[4], ty = (float)po[5], tz = (float)po[6];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 p = points[(size_t)b * n + i];
        const float dx = p.x - tx, dy = p.y - ty, dz = p.z - tz;
        float4 o;
        o.x = (dx * R[0] + dy * R[3]) + dz * R[6];
        o.y = (dx * R[1] + dy * R[4]) + dz * R[7];
        o.z = (dx * R[2] + dy * R[5]) + dz * R[8];
        o.w = 0.f;
        new_points[(size_t)b * n + i] = o;
    }
}

}  // namespace

extern "C" int ape_pose_select_f32(const float* heads, const float* points4, double* pose, int* which, float* new_points4,
                                   int B, int n, void* stream)
{
    if (!heads || !points4 || !pose || B < 0 || n < 1) return APE_EINVAL;
    if (B == 0) return APE_OK;
    hipLaunchKernelGGL(pose_select_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, heads, (const float4*)points4, pose,
                       which, (float4*)new_points4, n);
    return ape::check_launch("ape_pose_select_f32");
}

extern "C" int ape_pose_compose_f64(double* pose, const float* ref_r, int ldr, const float* ref_t, int ldt, int B, void* stream)
{
    if (!pose || !ref_r || !ref_t || B < 0 || ldr < 4 || ldt < 3) return APE_EINVAL;
    if (B == 0) return APE_OK;
    hipLaunchKernelGGL(pose_compose_kernel, dim3(ape::ceil_div(B, 64)), dim3(64), 0, (hipStream_t)stream, pose, ref_r, ldr,
                       ref_t, ldt, B);
    return ape::check_launch("ape_pose_compose_f64");
}

extern "C" int ape_pose_recentre_f32(const float* points4, const double* pose, float* new_points4, int B, int n, void* stream)
{
    if (!points4 || !pose || !new_points4 || B < 0 || n < 1) return APE_EINVAL;
    if (B == 0) return APE_OK;
    dim3 grid(ape::ceil_div(n, 256), B);
    hipLaunchKernelGGL(pose_recentre_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const float4*)points4, pose,
                       (float4*)new_points4, n);
    return ape::check_launch("ape_pose_recentre_f32");
}
