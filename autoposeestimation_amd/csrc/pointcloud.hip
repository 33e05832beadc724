// Point-cloud kernels of the pose-label path (float64 coordinates, like open3d / numpy in the reference):
//   surface_points     pc_reconstruction/open3d_utils.py:171-192 (get_surface): label & depth pixels -> camera frame (mm)
//                      -> robot frame via a 4x4, in raster order.  The reference does this in a per-pixel Python loop
//                      with a 4x4 np.dot per pixel.
//   voxel_down_sample  open3d 0.9 PointCloud::VoxelDownSample (open3d_utils.py:21,198): mean of the points of every
//                      voxel floor((p - (min_bound - voxel/2)) / voxel); output ordered by voxel key (open3d's hash-map
//                      order is unspecified => UNPINNED, we pick a deterministic one).
//   grid_*             uniform-grid neighbour search over a cloud sorted by cell key (cell = search radius, 27-cell
//                      scan): radius count (RemoveRadiusOutliers, :203), hybrid radius/max_nn normals (EstimateNormals,
//                      :25-27), nearest neighbour within max_correspondence_distance (registration_icp, :98-117).
//   knn_mean_dist      brute-force k-NN mean distance, LDS-tiled (RemoveStatisticalOutliers, :208-211, unbounded radius).
//   icp_*_sums         one-pass reductions for the two ICP estimators: Umeyama sums (point-to-point) and the 6x6 normal
//                      equations J^T J, J^T r with r = (s - t).n_t, J = [s x n_t, n_t] (point-to-plane).
// All of it is HBM/latency-bound index work; sorting and scans come from hipCUB (rocPRIM), everything else is hand written.
// Reductions use fixed-order two-stage trees => bitwise reproducible.
#include "common.h"
#include <hipcub/hipcub.hpp>

namespace {

constexpr int kT = 256;
typedef unsigned long long u64;
inline int grid_for(long work, int cap = 4096) { long g = (work + kT - 1) / kT; return (int)(g < 1 ? 1 : (g > cap ? cap : g)); }

struct Mat4 { double m[16]; };

// ---------------------------------------------------------------------------------------------------------------
__global__ void surface_flags_kernel(const uint8_t* __restrict__ label, const uint16_t* __restrict__ depth, uint8_t* __restrict__ flag, int n)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        flag[i] = (label[i] != 0 && depth[i] != 0) ? 1 : 0;
}

__global__ void surface_points_kernel(const int* __restrict__ pix, const int* __restrict__ n_sel, const uint16_t* __restrict__ depth,
                                      int W, double fx, double fy, double ppx, double ppy, Mat4 T, double* __restrict__ out)
{
    const int n = *n_sel;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int p = pix[i];
        const int py = p / W, px = p - py * W;
        const double p2 = (double)depth[p];
        const double p0 = ((double)px - ppx) * p2 / fx;      // open3d_utils.py:184-188
        const double p1 = ((double)py - ppy) * p2 / fy;
        for (int r = 0; r < 3; ++r)                          // robot2obj = robot2Cam . [I | p] -> column 3
            out[(size_t)i * 3 + r] = ((T.m[r * 4 + 0] * p0 + T.m[r * 4 + 1] * p1) + T.m[r * 4 + 2] * p2) + T.m[r * 4 + 3];
    }
}

__global__ void transform_kernel(double* __restrict__ pts, int n, Mat4 T, double* __restrict__ normals)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double x = pts[i * 3], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
        for (int r = 0; r < 3; ++r) pts[i * 3 + r] = ((T.m[r * 4] * x + T.m[r * 4 + 1] * y) + T.m[r * 4 + 2] * z) + T.m[r * 4 + 3];
        if (normals) {
            const double a = normals[i * 3], b = normals[i * 3 + 1], c = normals[i * 3 + 2];
            for (int r = 0; r < 3; ++r) normals[i * 3 + r] = (T.m[r * 4] * a + T.m[r * 4 + 1] * b) + T.m[r * 4 + 2] * c;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// fixed-order min/max bound: stage 1 per block into part[g][6], stage 2 single block
__global__ void bounds_stage1(const double* __restrict__ pts, int n, double* __restrict__ part)
{
    __shared__ double s[6][kT];
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        for (int d = 0; d < 3; ++d) { const double v = pts[(size_t)i * 3 + d]; lo[d] = fmin(lo[d], v); hi[d] = fmax(hi[d], v); }
    for (int d = 0; d < 3; ++d) { s[d][threadIdx.x] = lo[d]; s[3 + d][threadIdx.x] = hi[d]; }
    __syncthreads();
    for (int off = kT / 2; off > 0; off >>= 1) {
        if (threadIdx.x < off)
            for (int d = 0; d < 3; ++d) {
                s[d][threadIdx.x] = fmin(s[d][threadIdx.x], s[d][threadIdx.x + off]);
                s[3 + d][threadIdx.x] = fmax(s[3 + d][threadIdx.x], s[3 + d][threadIdx.x + off]);
            }
        __syncthreads();
    }
    if (threadIdx.x < 6) part[blockIdx.x * 6 + threadIdx.x] = s[threadIdx.x][0];
}

__global__ void bounds_stage2(const double* __restrict__ part, int g, double* __restrict__ out6)
{
    if (threadIdx.x < 6) {
        double v = part[threadIdx.x];
        for (int b = 1; b < g; ++b) v = threadIdx.x < 3 ? fmin(v, part[b * 6 + threadIdx.x]) : fmax(v, part[b * 6 + threadIdx.x]);
        out6[threadIdx.x] = v;
    }
}

__device__ __forceinline__ u64 pack_key(long cx, long cy, long cz) { return ((u64)cx << 42) | ((u64)cy << 21) | (u64)cz; }

// key of the cell of p for a grid with origin o and cell size h (coordinates clamped into [0, 2^21))
__device__ __forceinline__ void cell_of(const double* p, const double* o, double h, long c[3])
{
    for (int d = 0; d < 3; ++d) {
        long v = (long)floor((p[d] - o[d]) / h);   // the same division keys_kernel uses => identical cell borders
        c[d] = v < 0 ? 0 : (v > 2097151 ? 2097151 : v);
    }
}

__global__ void keys_kernel(const double* __restrict__ pts, int n, const double* __restrict__ bounds6, double h, double shift,
                            u64* __restrict__ keys, unsigned* __restrict__ idx, double* __restrict__ origin_out)
{
    double o[3] = {bounds6[0] - shift, bounds6[1] - shift, bounds6[2] - shift};
    if (origin_out && blockIdx.x == 0 && threadIdx.x < 3) origin_out[threadIdx.x] = o[threadIdx.x];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        long c[3];
        // open3d: floor((p - voxel_min_bound) / voxel_size) -- a division, kept as one for identical cell borders
        for (int d = 0; d < 3; ++d) {
            long v = (long)floor((pts[(size_t)i * 3 + d] - o[d]) / h);
            c[d] = v < 0 ? 0 : (v > 2097151 ? 2097151 : v);
        }
        keys[i] = pack_key(c[0], c[1], c[2]);
        idx[i] = (unsigned)i;
    }
}

__global__ void heads_kernel(const u64* __restrict__ keys, int n, int* __restrict__ head)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}

// seg[i] = exclusive-scan(head)[i] + head[i] - 1 = segment id; one thread per segment start averages its run in order
__global__ void voxel_mean_kernel(const double* __restrict__ pts, const u64* __restrict__ keys, const unsigned* __restrict__ order,
                                  const int* __restrict__ head, const int* __restrict__ scan, int n, double* __restrict__ out,
                                  int* __restrict__ n_out)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (!head[i]) continue;
        double s[3] = {0, 0, 0};
        int c = 0;
        for (int j = i; j < n && keys[j] == keys[i]; ++j, ++c)
            for (int d = 0; d < 3; ++d) s[d] += pts[(size_t)order[j] * 3 + d];
        const int seg = scan[i];
        for (int d = 0; d < 3; ++d) out[(size_t)seg * 3 + d] = s[d] / (double)c;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *n_out = scan[n - 1] + head[n - 1];
}

__global__ void gather_sorted_kernel(const double* __restrict__ pts, const unsigned* __restrict__ order, int n, double* __restrict__ out)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        for (int d = 0; d < 3; ++d) out[(size_t)i * 3 + d] = pts[(size_t)order[i] * 3 + d];
}

// ---------------------------------------------------------------------------------------------------------------
struct Grid {
    const double* sorted;   // [n][3] points in key order
    const u64* keys;        // [n] sorted
    const unsigned* order;  // [n] original index of sorted position
    const double* origin;   // [3]
    int n;
    double h;
};

__device__ __forceinline__ int lower_bound(const u64* keys, int n, u64 k)
{
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (keys[mid] < k) lo = mid + 1; else hi = mid; }
    return lo;
}

// visit every grid point in the 27 cells around q: f(sorted_position, squared distance)
template <class F>
__device__ __forceinline__ void for_neighbours(const Grid& g, const double* q, F f)
{
    long c[3];
    cell_of(q, g.origin, g.h, c);
    for (long dx = -1; dx <= 1; ++dx) {
        const long cx = c[0] + dx;
        if (cx < 0 || cx > 2097151) continue;
        for (long dy = -1; dy <= 1; ++dy) {
            const long cy = c[1] + dy;
            if (cy < 0 || cy > 2097151) continue;
            const long z0 = c[2] > 0 ? c[2] - 1 : 0, z1 = c[2] < 2097151 ? c[2] + 1 : 2097151;
            const u64 k0 = pack_key(cx, cy, z0), k1 = pack_key(cx, cy, z1);   // the three z-cells are contiguous in key order
            for (int j = lower_bound(g.keys, g.n, k0); j < g.n && g.keys[j] <= k1; ++j) {
                const double ex = g.sorted[(size_t)j * 3] - q[0], ey = g.sorted[(size_t)j * 3 + 1] - q[1], ez = g.sorted[(size_t)j * 3 + 2] - q[2];
                f(j, (ex * ex + ey * ey) + ez * ez);
            }
        }
    }
}

__global__ void radius_count_kernel(Grid g, const double* __restrict__ q, int nq, double r2, int* __restrict__ count)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += gridDim.x * blockDim.x) {
        int c = 0;
        for_neighbours(g, q + (size_t)i * 3, [&](int, double d2) { c += d2 < r2 ? 1 : 0; });   // FLANN radius search: d^2 < r^2
        count[i] = c;
    }
}

__global__ void nn1_kernel(Grid g, const double* __restrict__ q, int nq, double r2, int* __restrict__ idx, double* __restrict__ dist2)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += gridDim.x * blockDim.x) {
        double best = r2;
        unsigned bi = 0xffffffffu;
        for_neighbours(g, q + (size_t)i * 3, [&](int j, double d2) {
            const unsigned o = g.order[j];
            if (d2 < best || (d2 == best && bi != 0xffffffffu && o < bi)) { best = d2; bi = o; }   // ties: lowest original index
        });
        idx[i] = bi == 0xffffffffu ? -1 : (int)bi;
        dist2[i] = bi == 0xffffffffu ? 0.0 : best;
    }
}

// smallest-eigenvalue eigenvector of a symmetric 3x3 (cyclic Jacobi, fixed 12 sweeps)
__device__ void smallest_eigvec(double a[3][3], double v[3])
{
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 12; ++sweep)
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (fabs(a[p][q]) < 1e-300) continue;
                const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) { const double akp = a[k][p], akq = a[k][q]; a[k][p] = c * akp - s * akq; a[k][q] = s * akp + c * akq; }
                for (int k = 0; k < 3; ++k) { const double apk = a[p][k], aqk = a[q][k]; a[p][k] = c * apk - s * aqk; a[q][k] = s * apk + c * aqk; }
                for (int k = 0; k < 3; ++k) { const double vkp = V[k][p], vkq = V[k][q]; V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq; }
            }
    int m = 0;
    if (a[1][1] < a[m][m]) m = 1;
    if (a[2][2] < a[m][m]) m = 2;
    for (int k = 0; k < 3; ++k) v[k] = V[k][m];
}

constexpr int kMaxNN = 64;

// hybrid search: neighbours with d < radius, at most max_nn nearest of them; normal = eigenvector of the smallest
// eigenvalue of their covariance, flipped towards +z (open3d's default orientation reference); (0,0,1) if < 3 neighbours
__global__ void normals_kernel(Grid g, const double* __restrict__ q, int nq, double r2, int max_nn, double* __restrict__ normals)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += gridDim.x * blockDim.x) {
        double bd[kMaxNN];
        int bj[kMaxNN];
        int cnt = 0;
        for_neighbours(g, q + (size_t)i * 3, [&](int j, double d2) {
            if (d2 >= r2) return;
            if (cnt < max_nn) {
                int k = cnt++;
                while (k > 0 && bd[k - 1] > d2) { bd[k] = bd[k - 1]; bj[k] = bj[k - 1]; --k; }
                bd[k] = d2; bj[k] = j;
            } else if (bd[max_nn - 1] > d2) {
                int k = max_nn - 1;
                while (k > 0 && bd[k - 1] > d2) { bd[k] = bd[k - 1]; bj[k] = bj[k - 1]; --k; }
                bd[k] = d2; bj[k] = j;
            }
        });
        double nrm[3] = {0, 0, 1};
        if (cnt >= 3) {
            double mu[3] = {0, 0, 0};
            for (int k = 0; k < cnt; ++k) for (int d = 0; d < 3; ++d) mu[d] += g.sorted[(size_t)bj[k] * 3 + d];
            for (int d = 0; d < 3; ++d) mu[d] /= (double)cnt;
            double C[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
            for (int k = 0; k < cnt; ++k) {
                double e[3];
                for (int d = 0; d < 3; ++d) e[d] = g.sorted[(size_t)bj[k] * 3 + d] - mu[d];
                for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) C[a][b] += e[a] * e[b];
            }
            for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) C[a][b] /= (double)cnt;
            smallest_eigvec(C, nrm);
            const double l = sqrt((nrm[0] * nrm[0] + nrm[1] * nrm[1]) + nrm[2] * nrm[2]);
            if (l > 0) for (int d = 0; d < 3; ++d) nrm[d] /= l; else { nrm[0] = 0; nrm[1] = 0; nrm[2] = 1; }
            if (nrm[2] < 0) for (int d = 0; d < 3; ++d) nrm[d] = -nrm[d];
        }
        for (int d = 0; d < 3; ++d) normals[(size_t)i * 3 + d] = nrm[d];
    }
}

// brute-force k-NN (self included, like KDTree SearchKNN on the cloud itself): mean of the k smallest distances
__global__ __launch_bounds__(kT) void knn_mean_kernel(const double* __restrict__ pts, int n, int k, double* __restrict__ mean)
{
    __shared__ double tile[kT][3];
    const int i = blockIdx.x * kT + threadIdx.x;
    const bool valid = i < n;
    double q[3] = {0, 0, 0};
    if (valid) for (int d = 0; d < 3; ++d) q[d] = pts[(size_t)i * 3 + d];
    double bd[kMaxNN];
    int cnt = 0;
    for (int t0 = 0; t0 < n; t0 += kT) {
        __syncthreads();
        if (t0 + threadIdx.x < n) for (int d = 0; d < 3; ++d) tile[threadIdx.x][d] = pts[(size_t)(t0 + threadIdx.x) * 3 + d];
        __syncthreads();
        const int m = min(kT, n - t0);
        if (!valid) continue;
        for (int j = 0; j < m; ++j) {
            const double ex = tile[j][0] - q[0], ey = tile[j][1] - q[1], ez = tile[j][2] - q[2];
            const double d2 = (ex * ex + ey * ey) + ez * ez;
            if (cnt < k) {
                int p = cnt++;
                while (p > 0 && bd[p - 1] > d2) { bd[p] = bd[p - 1]; --p; }
                bd[p] = d2;
            } else if (bd[k - 1] > d2) {
                int p = k - 1;
                while (p > 0 && bd[p - 1] > d2) { bd[p] = bd[p - 1]; --p; }
                bd[p] = d2;
            }
        }
    }
    if (valid) {
        double s = 0;
        for (int p = 0; p < cnt; ++p) s += sqrt(bd[p]);
        mean[i] = cnt ? s / (double)cnt : -1.0;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// generic fixed-order reduction of NV doubles per element: stage 1 -> part[blocks][NV], stage 2 -> out[NV]
template <int NV, class F>
__device__ void reduce_stage1(int n, F value, double* part)
{
    __shared__ double s[kT];
    double acc[NV];
    for (int v = 0; v < NV; ++v) acc[v] = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        double e[NV];
        if (value(i, e)) for (int v = 0; v < NV; ++v) acc[v] += e[v];
    }
    for (int v = 0; v < NV; ++v) {
        __syncthreads();
        s[threadIdx.x] = acc[v];
        __syncthreads();
        for (int off = kT / 2; off > 0; off >>= 1) { if (threadIdx.x < off) s[threadIdx.x] += s[threadIdx.x + off]; __syncthreads(); }
        if (threadIdx.x == 0) part[blockIdx.x * NV + v] = s[0];
    }
}

__global__ void reduce_stage2(const double* __restrict__ part, int g, int nv, double* __restrict__ out)
{
    const int v = threadIdx.x;
    if (v < nv) { double s = 0; for (int b = 0; b < g; ++b) s += part[b * nv + v]; out[v] = s; }
}

// out: [0] count, [1] sum d^2, [2..4] sum s, [5..7] sum t, [8..16] sum s_a t_b (row a, col b)
__global__ __launch_bounds__(kT) void p2p_sums_kernel(const double* __restrict__ src, const double* __restrict__ tgt, const int* __restrict__ corr,
                                                      const double* __restrict__ d2, int n, double* __restrict__ part)
{
    reduce_stage1<17>(n, [&](int i, double* e) {
        const int j = corr[i];
        if (j < 0) return false;
        const double* s = src + (size_t)i * 3;
        const double* t = tgt + (size_t)j * 3;
        e[0] = 1.0; e[1] = d2[i];
        for (int a = 0; a < 3; ++a) { e[2 + a] = s[a]; e[5 + a] = t[a]; for (int b = 0; b < 3; ++b) e[8 + a * 3 + b] = s[a] * t[b]; }
        return true;
    }, part);
}

// out: [0] count, [1] sum d^2, [2..22] upper triangle of J^T J (row major), [23..28] J^T r
__global__ __launch_bounds__(kT) void p2plane_sums_kernel(const double* __restrict__ src, const double* __restrict__ tgt,
                                                          const double* __restrict__ tn, const int* __restrict__ corr,
                                                          const double* __restrict__ d2, int n, double* __restrict__ part)
{
    reduce_stage1<29>(n, [&](int i, double* e) {
        const int j = corr[i];
        if (j < 0) return false;
        const double* s = src + (size_t)i * 3;
        const double* t = tgt + (size_t)j * 3;
        const double* nn = tn + (size_t)j * 3;
        const double r = ((s[0] - t[0]) * nn[0] + (s[1] - t[1]) * nn[1]) + (s[2] - t[2]) * nn[2];
        const double J[6] = {s[1] * nn[2] - s[2] * nn[1], s[2] * nn[0] - s[0] * nn[2], s[0] * nn[1] - s[1] * nn[0], nn[0], nn[1], nn[2]};
        e[0] = 1.0; e[1] = d2[i];
        int k = 2;
        for (int a = 0; a < 6; ++a) for (int b = a; b < 6; ++b) e[k++] = J[a] * J[b];
        for (int a = 0; a < 6; ++a) e[23 + a] = J[a] * r;
        return true;
    }, part);
}

// out: [0..2] sum p, [3..8] sum p_a p_b upper triangle
__global__ __launch_bounds__(kT) void moments_kernel(const double* __restrict__ pts, int n, double* __restrict__ part)
{
    reduce_stage1<9>(n, [&](int i, double* e) {
        const double* p = pts + (size_t)i * 3;
        e[0] = p[0]; e[1] = p[1]; e[2] = p[2];
        e[3] = p[0] * p[0]; e[4] = p[0] * p[1]; e[5] = p[0] * p[2]; e[6] = p[1] * p[1]; e[7] = p[1] * p[2]; e[8] = p[2] * p[2];
        return true;
    }, part);
}

struct Vec12 { double v[12]; };
// sqrt((p - mu)^T Cinv (p - mu));  mc = (mu[3], Cinv[9])
__global__ void mahalanobis_kernel(const double* __restrict__ pts, int n, Vec12 mc, double* __restrict__ out)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        double e[3];
        for (int d = 0; d < 3; ++d) e[d] = pts[(size_t)i * 3 + d] - mc.v[d];
        double s = 0;
        for (int a = 0; a < 3; ++a) s += e[a] * ((mc.v[3 + a * 3] * e[0] + mc.v[3 + a * 3 + 1] * e[1]) + mc.v[3 + a * 3 + 2] * e[2]);
        out[i] = sqrt(s);
    }
}

__global__ void select_rows_kernel(const double* __restrict__ pts, const int* __restrict__ sel, const int* __restrict__ n_sel, double* __restrict__ out)
{
    const int n = *n_sel;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        for (int d = 0; d < 3; ++d) out[(size_t)i * 3 + d] = pts[(size_t)sel[i] * 3 + d];
}

size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

struct Carver {
    char* p; char* end;
    template <class T> T* take(size_t count) { T* r = (T*)p; p += align_up(count * sizeof(T)); return p <= end ? r : nullptr; }
};

size_t sort_temp_bytes(int n)
{
    size_t b = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, b, (u64*)nullptr, (u64*)nullptr, (unsigned*)nullptr, (unsigned*)nullptr, n, 0, 63);
    size_t c = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, c, (int*)nullptr, (int*)nullptr, n);
    size_t d = 0;
    (void)hipcub::DeviceSelect::Flagged(nullptr, d, hipcub::CountingInputIterator<int>(0), (uint8_t*)nullptr, (int*)nullptr, (int*)nullptr, n);
    return align_up(b > c ? (b > d ? b : d) : (c > d ? c : d));
}

Mat4 load_mat(const double* T16_host) { Mat4 m; for (int i = 0; i < 16; ++i) m.m[i] = T16_host[i]; return m; }

}  // namespace

extern "C" size_t ape_pc_workspace_bytes(int n)
{
    if (n < 1) n = 1;
    // keys in/out, order in/out, heads, scan, bounds partials, cub temp
    return sort_temp_bytes(n) + 2 * align_up((size_t)n * 8) + 4 * align_up((size_t)n * 4) + align_up((size_t)n) + align_up(4096 * 32 * 8) + 4096;
}

/* label[H][W] u8, depth[H][W] u16 -> points[n][3] f64 in raster order (capacity H*W), *n_out on the device */
extern "C" int ape_surface_points_f64(const uint8_t* label, const uint16_t* depth, int H, int W, double fx, double fy, double ppx,
                                      double ppy, const double* T16_host, double* points, int* n_out, void* ws, size_t ws_bytes,
                                      void* stream)
{
    if (!label || !depth || !T16_host || !points || !n_out || !ws || H < 1 || W < 1) return APE_EINVAL;
    const int n = H * W;
    if (ws_bytes < ape_pc_workspace_bytes(n)) return APE_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    Carver c{(char*)ws, (char*)ws + ws_bytes};
    uint8_t* flag = c.take<uint8_t>(n);
    int* pix = c.take<int>(n);
    size_t tb = sort_temp_bytes(n);
    void* tmp = c.take<char>(tb);
    if (!tmp) return APE_EWORKSPACE;
    hipLaunchKernelGGL(surface_flags_kernel, dim3(grid_for(n)), dim3(kT), 0, st, label, depth, flag, n);
    if (hipcub::DeviceSelect::Flagged(tmp, tb, hipcub::CountingInputIterator<int>(0), flag, pix, n_out, n, st) != hipSuccess) return APE_ELAUNCH;
    hipLaunchKernelGGL(surface_points_kernel, dim3(grid_for(n)), dim3(kT), 0, st, pix, n_out, depth, W, fx, fy, ppx, ppy, load_mat(T16_host), points);
    return ape::check_launch("ape_surface_points_f64");
}

extern "C" int ape_transform_points_f64(double* pts, double* normals_or_null, int n, const double* T16_host, void* stream)
{
    if (!pts || !T16_host || n < 0) return APE_EINVAL;
    if (n == 0) return APE_OK;
    hipLaunchKernelGGL(transform_kernel, dim3(grid_for(n)), dim3(kT), 0, (hipStream_t)stream, pts, n, load_mat(T16_host), normals_or_null);
    return ape::check_launch("ape_transform_points_f64");
}

/* out capacity n points; *n_out on the device */
extern "C" int ape_voxel_down_sample_f64(const double* pts, int n, double voxel, double* out, int* n_out, void* ws, size_t ws_bytes,
                                         void* stream)
{
    if (!pts || !out || !n_out || !ws || n < 1 || !(voxel > 0)) return APE_EINVAL;
    if (ws_bytes < ape_pc_workspace_bytes(n)) return APE_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    Carver c{(char*)ws, (char*)ws + ws_bytes};
    u64* k0 = c.take<u64>(n); u64* k1 = c.take<u64>(n);
    unsigned* i0 = c.take<unsigned>(n); unsigned* i1 = c.take<unsigned>(n);
    int* head = c.take<int>(n); int* scan = c.take<int>(n);
    double* part = c.take<double>(4096 * 32);
    size_t tb = sort_temp_bytes(n);
    void* tmp = c.take<char>(tb);
    if (!tmp) return APE_EWORKSPACE;
    const int g = grid_for(n, 1024);
    hipLaunchKernelGGL(bounds_stage1, dim3(g), dim3(kT), 0, st, pts, n, part);
    hipLaunchKernelGGL(bounds_stage2, dim3(1), dim3(64), 0, st, part, g, part + 4096 * 16);
    hipLaunchKernelGGL(keys_kernel, dim3(grid_for(n)), dim3(kT), 0, st, pts, n, part + 4096 * 16, voxel, voxel * 0.5, k0, i0, (double*)nullptr);
    if (hipcub::DeviceRadixSort::SortPairs(tmp, tb, k0, k1, i0, i1, n, 0, 63, st) != hipSuccess) return APE_ELAUNCH;
    hipLaunchKernelGGL(heads_kernel, dim3(grid_for(n)), dim3(kT), 0, st, k1, n, head);
    if (hipcub::DeviceScan::ExclusiveSum(tmp, tb, head, scan, n, st) != hipSuccess) return APE_ELAUNCH;
    hipLaunchKernelGGL(voxel_mean_kernel, dim3(grid_for(n)), dim3(kT), 0, st, pts, k1, i1, head, scan, n, out, n_out);
    return ape::check_launch("ape_voxel_down_sample_f64");
}

/* Build the search grid of a cloud: sorted[n][3], keys[n], order[n], origin[3] (all caller-owned device buffers). */
extern "C" int ape_grid_build_f64(const double* pts, int n, double cell, double* sorted, unsigned long long* keys, unsigned* order,
                                  double* origin3, void* ws, size_t ws_bytes, void* stream)
{
    if (!pts || !sorted || !keys || !order || !origin3 || !ws || n < 1 || !(cell > 0)) return APE_EINVAL;
    if (ws_bytes < ape_pc_workspace_bytes(n)) return APE_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    Carver c{(char*)ws, (char*)ws + ws_bytes};
    u64* k0 = c.take<u64>(n); c.take<u64>(n);
    unsigned* i0 = c.take<unsigned>(n); c.take<unsigned>(n);
    c.take<int>(n); c.take<int>(n);
    double* part = c.take<double>(4096 * 32);
    size_t tb = sort_temp_bytes(n);
    void* tmp = c.take<char>(tb);
    if (!tmp) return APE_EWORKSPACE;
    const int g = grid_for(n, 1024);
    hipLaunchKernelGGL(bounds_stage1, dim3(g), dim3(kT), 0, st, pts, n, part);
    hipLaunchKernelGGL(bounds_stage2, dim3(1), dim3(64), 0, st, part, g, part + 4096 * 16);
    hipLaunchKernelGGL(keys_kernel, dim3(grid_for(n)), dim3(kT), 0, st, pts, n, part + 4096 * 16, cell, cell, k0, i0, origin3);
    if (hipcub::DeviceRadixSort::SortPairs(tmp, tb, k0, (u64*)keys, i0, order, n, 0, 63, st) != hipSuccess) return APE_ELAUNCH;
    hipLaunchKernelGGL(gather_sorted_kernel, dim3(grid_for(n)), dim3(kT), 0, st, pts, order, n, sorted);
    return ape::check_launch("ape_grid_build_f64");
}

#define GRID_ARGS const double* sorted, const unsigned long long* keys, const unsigned* order, const double* origin3, int n, double cell
#define MAKE_GRID Grid g{sorted, (const u64*)keys, order, origin3, n, cell}

extern "C" int ape_grid_radius_count_f64(GRID_ARGS, const double* q, int nq, double radius, int* count, void* stream)
{
    if (!sorted || !keys || !order || !origin3 || !q || !count || n < 1 || nq < 0 || radius > cell) return APE_EINVAL;
    if (nq == 0) return APE_OK;
    MAKE_GRID;
    hipLaunchKernelGGL(radius_count_kernel, dim3(grid_for(nq)), dim3(kT), 0, (hipStream_t)stream, g, q, nq, radius * radius, count);
    return ape::check_launch("ape_grid_radius_count_f64");
}

extern "C" int ape_grid_nn1_f64(GRID_ARGS, const double* q, int nq, double max_dist, int* idx, double* dist2, void* stream)
{
    if (!sorted || !keys || !order || !origin3 || !q || !idx || !dist2 || n < 1 || nq < 0 || max_dist > cell) return APE_EINVAL;
    if (nq == 0) return APE_OK;
    MAKE_GRID;
    hipLaunchKernelGGL(nn1_kernel, dim3(grid_for(nq)), dim3(kT), 0, (hipStream_t)stream, g, q, nq, max_dist * max_dist, idx, dist2);
    return ape::check_launch("ape_grid_nn1_f64");
}

extern "C" int ape_grid_normals_f64(GRID_ARGS, const double* q, int nq, double radius, int max_nn, double* normals, void* stream)
{
    if (!sorted || !keys || !order || !origin3 || !q || !normals || n < 1 || nq < 0 || radius > cell || max_nn < 3 || max_nn > kMaxNN)
        return APE_EINVAL;
    if (nq == 0) return APE_OK;
    MAKE_GRID;
    hipLaunchKernelGGL(normals_kernel, dim3(grid_for(nq)), dim3(kT), 0, (hipStream_t)stream, g, q, nq, radius * radius, max_nn, normals);
    return ape::check_launch("ape_grid_normals_f64");
}

extern "C" int ape_knn_mean_dist_f64(const double* pts, int n, int k, double* mean, void* stream)
{
    if (!pts || !mean || n < 1 || k < 1 || k > kMaxNN) return APE_EINVAL;
    hipLaunchKernelGGL(knn_mean_kernel, dim3(ape::ceil_div(n, kT)), dim3(kT), 0, (hipStream_t)stream, pts, n, k, mean);
    return ape::check_launch("ape_knn_mean_dist_f64");
}

/* kind 0: point-to-point sums out[17]; kind 1: point-to-plane out[29] (needs tgt_normals); kind 2: moments of src, out[9] */
extern "C" int ape_icp_sums_f64(int kind, const double* src, const double* tgt, const double* tgt_normals, const int* corr,
                                const double* dist2, int n, double* out, void* ws, size_t ws_bytes, void* stream)
{
    if (!src || !out || !ws || n < 0 || kind < 0 || kind > 2) return APE_EINVAL;
    if (kind < 2 && (!tgt || !corr || !dist2)) return APE_EINVAL;
    if (kind == 1 && !tgt_normals) return APE_EINVAL;
    const int nv = kind == 0 ? 17 : kind == 1 ? 29 : 9;
    const int g = grid_for(n, 512);
    if (ws_bytes < (size_t)g * nv * 8) return APE_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    double* part = (double*)ws;
    if (kind == 0) hipLaunchKernelGGL(p2p_sums_kernel, dim3(g), dim3(kT), 0, st, src, tgt, corr, dist2, n, part);
    else if (kind == 1) hipLaunchKernelGGL(p2plane_sums_kernel, dim3(g), dim3(kT), 0, st, src, tgt, tgt_normals, corr, dist2, n, part);
    else hipLaunchKernelGGL(moments_kernel, dim3(g), dim3(kT), 0, st, src, n, part);
    hipLaunchKernelGGL(reduce_stage2, dim3(1), dim3(64), 0, st, part, g, nv, out);
    return ape::check_launch("ape_icp_sums_f64");
}

extern "C" int ape_mahalanobis_f64(const double* pts, int n, const double* mean_cinv12_host, double* out, void* stream)
{
    if (!pts || !mean_cinv12_host || !out || n < 0) return APE_EINVAL;
    if (n == 0) return APE_OK;
    Vec12 mc;
    for (int i = 0; i < 12; ++i) mc.v[i] = mean_cinv12_host[i];
    hipLaunchKernelGGL(mahalanobis_kernel, dim3(grid_for(n)), dim3(kT), 0, (hipStream_t)stream, pts, n, mc, out);
    return ape::check_launch("ape_mahalanobis_f64");
}

/* out[i] = pts[sel[i]] for the rows with keep[i] != 0, in order (capacity n); *n_out on the device */
extern "C" int ape_select_points_f64(const double* pts, const uint8_t* keep, int n, double* out, int* sel_idx, int* n_out, void* ws,
                                     size_t ws_bytes, void* stream)
{
    if (!pts || !keep || !out || !sel_idx || !n_out || !ws || n < 1) return APE_EINVAL;
    size_t tb = sort_temp_bytes(n);
    if (ws_bytes < tb) return APE_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (hipcub::DeviceSelect::Flagged(ws, tb, hipcub::CountingInputIterator<int>(0), keep, sel_idx, n_out, n, st) != hipSuccess) return APE_ELAUNCH;
    hipLaunchKernelGGL(select_rows_kernel, dim3(grid_for(n)), dim3(kT), 0, st, pts, sel_idx, n_out, out);
    return ape::check_launch("ape_select_points_f64");
}
