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
// All of it is HBM/latency-bound index work, all of it hand written (the sort: seg_sort_batch / seg_merge_batch; the scans and the
// ordered compactions: scan_block / compact_block); the one-cloud entry points are the batched ones called with one cloud.
// Reductions use fixed-order two-stage trees => bitwise reproducible.
#include "common.h"
#include <atomic>

namespace {

constexpr int kT = 256;
typedef unsigned long long u64;
inline int grid_for(long work, int cap = 4096) { long g = (work + kT - 1) / kT; return (int)(g < 1 ? 1 : (g > cap ? cap : g)); }

struct Mat4 { double m[16]; };

// ---------------------------------------------------------------------------------------------------------------

__device__ __forceinline__ void surface_points_kernel_body(const int* __restrict__ pix, const int* __restrict__ n_sel, const uint16_t* __restrict__ depth, int W, double fx, double fy, double ppx, double ppy, Mat4 T, double* __restrict__ out, const int bx, const int gx)
{
    const int n = *n_sel;
    for (int i = bx * blockDim.x + threadIdx.x; i < n; i += gx * blockDim.x) {
        const int p = pix[i];
        const int py = p / W, px = p - py * W;
        const double p2 = (double)depth[p];
        const double p0 = ((double)px - ppx) * p2 / fx;      // open3d_utils.py:184-188
        const double p1 = ((double)py - ppy) * p2 / fy;
        for (int r = 0; r < 3; ++r)                          // robot2obj = robot2Cam . [I | p] -> column 3
            out[(size_t)i * 3 + r] = ((T.m[r * 4 + 0] * p0 + T.m[r * 4 + 1] * p1) + T.m[r * 4 + 2] * p2) + T.m[r * 4 + 3];
    }
}

__device__ __forceinline__ void transform_kernel_body(double* __restrict__ pts, int n, Mat4 T, double* __restrict__ normals, const int bx, const int gx)
{
    for (int i = bx * blockDim.x + threadIdx.x; i < n; i += gx * blockDim.x) {
        const double x = pts[i * 3], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
        for (int r = 0; r < 3; ++r) pts[i * 3 + r] = ((T.m[r * 4] * x + T.m[r * 4 + 1] * y) + T.m[r * 4 + 2] * z) + T.m[r * 4 + 3];
        if (normals) {
            const double a = normals[i * 3], b = normals[i * 3 + 1], c = normals[i * 3 + 2];
            for (int r = 0; r < 3; ++r) normals[i * 3 + r] = (T.m[r * 4] * a + T.m[r * 4 + 1] * b) + T.m[r * 4 + 2] * c;
        }
    }
}

__global__ void transform_kernel(double* __restrict__ pts, int n, Mat4 T, double* __restrict__ normals)
{
    transform_kernel_body(pts, n, T, normals, blockIdx.x, gridDim.x);
}

// ---------------------------------------------------------------------------------------------------------------
// fixed-order min/max bound: stage 1 per block into part[g][6], stage 2 single block
__device__ __forceinline__ void bounds_stage1_body(const double* __restrict__ pts, int n, double* __restrict__ part, const int bx, const int gx)
{
    __shared__ double s[6][kT];
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int i = bx * blockDim.x + threadIdx.x; i < n; i += gx * blockDim.x)
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
    if (threadIdx.x < 6) part[bx * 6 + threadIdx.x] = s[threadIdx.x][0];
}


__device__ __forceinline__ void bounds_stage2_body(const double* __restrict__ part, int g, double* __restrict__ out6, const int bx, const int gx)
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
        long v = (long)floor((p[d] - o[d]) / h);   // the same division keys_kernel_body uses => identical cell borders
        c[d] = v < 0 ? 0 : (v > 2097151 ? 2097151 : v);
    }
}

__device__ __forceinline__ void keys_kernel_body(const double* __restrict__ pts, int n, const double* __restrict__ bounds6, double h, double shift, u64* __restrict__ keys, unsigned* __restrict__ idx, double* __restrict__ origin_out, const int bx, const int gx)
{
    double o[3] = {bounds6[0] - shift, bounds6[1] - shift, bounds6[2] - shift};
    if (origin_out && bx == 0 && threadIdx.x < 3) origin_out[threadIdx.x] = o[threadIdx.x];
    for (int i = bx * blockDim.x + threadIdx.x; i < n; i += gx * blockDim.x) {
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

__device__ __forceinline__ void heads_kernel_body(const u64* __restrict__ keys, int n, int* __restrict__ head, const int bx, const int gx)
{
    for (int i = bx * blockDim.x + threadIdx.x; i < n; i += gx * blockDim.x)
        head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}


// seg[i] = exclusive-scan(head)[i] + head[i] - 1 = segment id; one thread per segment start averages its run in order
__device__ __forceinline__ void voxel_mean_kernel_body(const double* __restrict__ pts, const u64* __restrict__ keys, const unsigned* __restrict__ order, const int* __restrict__ head, const int* __restrict__ scan, int n, double* __restrict__ out, int* __restrict__ n_out, const int bx, const int gx)
{
    for (int i = bx * blockDim.x + threadIdx.x; i < n; i += gx * blockDim.x) {
        if (!head[i]) continue;
        double s[3] = {0, 0, 0};
        int c = 0;
        for (int j = i; j < n && keys[j] == keys[i]; ++j, ++c)
            for (int d = 0; d < 3; ++d) s[d] += pts[(size_t)order[j] * 3 + d];
        const int seg = scan[i];
        for (int d = 0; d < 3; ++d) out[(size_t)seg * 3 + d] = s[d] / (double)c;
    }
    if (bx == 0 && threadIdx.x == 0) *n_out = scan[n - 1] + head[n - 1];
}

__device__ __forceinline__ void gather_sorted_kernel_body(const double* __restrict__ pts, const unsigned* __restrict__ order, int n, double* __restrict__ out, const int bx, const int gx)
{
    for (int i = bx * blockDim.x + threadIdx.x; i < n; i += gx * blockDim.x)
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

// Cooperative form for the small clouds of the label path (10^4 points: one query per lane leaves most of the chip idle and every
// lane walks 27 cells through dependent loads): kG = 32 lanes share a query, lane l < 27 takes cell l of the 3x3x3 block -- one binary
// search and a short run of points -- and the group reduces with shuffles.
constexpr int kG = 32;

template <class F>
__device__ __forceinline__ void for_my_cell(const Grid& g, const double* q, int lane, F f)
{
    long c[3];
    cell_of(q, g.origin, g.h, c);
    if (lane >= 27) return;
    const long cx = c[0] + lane / 9 - 1, cy = c[1] + (lane / 3) % 3 - 1, cz = c[2] + lane % 3 - 1;
    if (cx < 0 || cx > 2097151 || cy < 0 || cy > 2097151 || cz < 0 || cz > 2097151) return;
    const u64 key = pack_key(cx, cy, cz);
    for (int j = lower_bound(g.keys, g.n, key); j < g.n && g.keys[j] == key; ++j) {
        const double ex = g.sorted[(size_t)j * 3] - q[0], ey = g.sorted[(size_t)j * 3 + 1] - q[1], ez = g.sorted[(size_t)j * 3 + 2] - q[2];
        f(j, (ex * ex + ey * ey) + ez * ez);
    }
}

__device__ __forceinline__ void radius_count_group_kernel_body(Grid g, const double* __restrict__ q, int nq, double r2, int* __restrict__ count, const int bx, const int gx)
{
    const int lane = threadIdx.x % kG;
    const int i = (bx * kT + threadIdx.x) / kG;
    int c = 0;
    if (i < nq) for_my_cell(g, q + (size_t)i * 3, lane, [&](int, double d2) { c += d2 < r2 ? 1 : 0; });
    for (int m = kG / 2; m >= 1; m >>= 1) c += __shfl_xor(c, m, kG);
    if (i < nq && lane == 0) count[i] = c;
}

__global__ __launch_bounds__(kT) void radius_count_group_kernel(Grid g, const double* __restrict__ q, int nq, double r2, int* __restrict__ count)
{
    radius_count_group_kernel_body(g, q, nq, r2, count, blockIdx.x, gridDim.x);
}

__device__ __forceinline__ void nn1_group_kernel_body(Grid g, const double* __restrict__ q, int nq, double r2, int* __restrict__ idx, double* __restrict__ dist2, const double* __restrict__ skip, const int bx, const int gx)
{
    if (skip && skip[0] != 0.0) return;
    const int lane = threadIdx.x % kG;
    const int i = (bx * kT + threadIdx.x) / kG;
    double best = r2;
    unsigned bi = 0xffffffffu;
    if (i < nq)
        for_my_cell(g, q + (size_t)i * 3, lane, [&](int j, double d2) {
            const unsigned o = g.order[j];
            if (d2 < best || (d2 == best && bi != 0xffffffffu && o < bi)) { best = d2; bi = o; }   // ties: lowest original index
        });
    for (int m = kG / 2; m >= 1; m >>= 1) {
        const double ob = __shfl_xor(best, m, kG);
        const unsigned oi = __shfl_xor(bi, m, kG);
        if (oi != 0xffffffffu && (bi == 0xffffffffu || ob < best || (ob == best && oi < bi))) { best = ob; bi = oi; }
    }
    if (i < nq && lane == 0) {
        idx[i] = bi == 0xffffffffu ? -1 : (int)bi;
        dist2[i] = bi == 0xffffffffu ? 0.0 : best;
    }
}

__global__ __launch_bounds__(kT) void nn1_group_kernel(Grid g, const double* __restrict__ q, int nq, double r2, int* __restrict__ idx,
                                                       double* __restrict__ dist2, const double* __restrict__ skip)
{
    nn1_group_kernel_body(g, q, nq, r2, idx, dist2, skip, blockIdx.x, gridDim.x);
}

// `skip` (optional, in the kernels of the ICP loop): a device word that, once non-zero, turns the launch into a no-op -- the loop runs
// a fixed number of enqueued iterations and the device decides when it has converged (icp_step)

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
__device__ void normal_from_selection(const Grid& g, const int* bj, int cnt, double nrm[3])
{
    nrm[0] = 0; nrm[1] = 0; nrm[2] = 1;
    if (cnt < 3) return;
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

// kG lanes per query (see nn1_group_kernel): the in-radius candidates of the 27 cells go to the group's LDS list, then min(max_nn,
// candidates) rounds take the smallest (d^2, sorted position) after the last one taken -- the order an insertion sort over the cells
// in key order produces (its ties keep the first visited = lower sorted position) -- and lane 0 runs the covariance / eigenvector
// part on the selection in that order.  More than kNrmCand candidates: the rounds re-walk each lane's cell instead of the list.
constexpr int kNrmCand = 224;

__device__ __forceinline__ void normals_kernel_body(Grid g, const double* __restrict__ q, int nq, double r2, int max_nn, double* __restrict__ normals, const int bx, const int gx)
{
    __shared__ double cand_d[kT / kG][kNrmCand];
    __shared__ int cand_j[kT / kG][kNrmCand];
    __shared__ int sel[kT / kG][kMaxNN];
    __shared__ int ncand[kT / kG];
    const int lane = threadIdx.x % kG, grp = threadIdx.x / kG;
    const int i = (bx * kT + threadIdx.x) / kG;
    if (lane == 0) ncand[grp] = 0;
    __syncthreads();
    double qq[3] = {0, 0, 0};
    if (i < nq) {
        for (int d = 0; d < 3; ++d) qq[d] = q[(size_t)i * 3 + d];
        for_my_cell(g, qq, lane, [&](int j, double d2) {
            if (d2 >= r2) return;
            const int p = atomicAdd(&ncand[grp], 1);
            if (p < kNrmCand) { cand_d[grp][p] = d2; cand_j[grp][p] = j; }
        });
    }
    __syncthreads();
    const int nc = ncand[grp];
    const bool listed = nc <= kNrmCand;
    const int cnt = i < nq ? (nc < max_nn ? nc : max_nn) : 0;
    double last_d = -1.0;
    int last_j = -1;
    for (int r = 0; r < cnt; ++r) {
        double bd = 1e300;
        int bj = 0x7fffffff;
        auto offer = [&](int j, double d) {
            const bool after = d > last_d || (d == last_d && j > last_j);
            if (after && (d < bd || (d == bd && j < bj))) { bd = d; bj = j; }
        };
        if (listed) for (int p = lane; p < nc; p += kG) offer(cand_j[grp][p], cand_d[grp][p]);
        else for_my_cell(g, qq, lane, [&](int j, double d2) { if (d2 < r2) offer(j, d2); });
        for (int m = kG / 2; m >= 1; m >>= 1) {
            const double od = __shfl_xor(bd, m, kG);
            const int oj = __shfl_xor(bj, m, kG);
            if (od < bd || (od == bd && oj < bj)) { bd = od; bj = oj; }
        }
        last_d = bd; last_j = bj;
        if (lane == 0) sel[grp][r] = bj;
    }
    if (i < nq && lane == 0) {
        double nrm[3];
        normal_from_selection(g, sel[grp], cnt, nrm);
        for (int d = 0; d < 3; ++d) normals[(size_t)i * 3 + d] = nrm[d];
    }
}

__global__ __launch_bounds__(kT) void normals_kernel(Grid g, const double* __restrict__ q, int nq, double r2, int max_nn, double* __restrict__ normals)
{
    normals_kernel_body(g, q, nq, r2, max_nn, normals, blockIdx.x, gridDim.x);
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

// The same through the uniform grid, kG lanes per query (see nn1_group_kernel; the queries are the grid's own points in key order).
// For R = 1, 2, 3: the lanes share the cells of the (2R+1)^3 block around the query's cell and drop their squared distances into the
// group's LDS list; if the block holds the k nearest for certain (k-th smallest inside R cell sizes: every point outside the block is
// at least R cells away), k rounds of "smallest entry after the last one taken" (each lane scans its share, the group reduces by
// shuffles) have summed them in ascending order.  Queries that stay unsettled (isolated points, more than kKnnCand candidates) run
// the same k rounds over ALL points, n / kG per lane and round.  No per-lane sort, no scratch, exact for any cell size: the k
// smallest squared distances are the multiset the all-pairs kernel finds, summed in ascending order, hence the same bits.
constexpr int kKnnCand = 448;

__device__ __forceinline__ void knn_round_reduce(double& bd, int& bp)
{
    for (int m = kG / 2; m >= 1; m >>= 1) {
        const double od = __shfl_xor(bd, m, kG);
        const int op = __shfl_xor(bp, m, kG);
        if (od < bd || (od == bd && op < bp)) { bd = od; bp = op; }
    }
}

__device__ __forceinline__ void knn_mean_grid_kernel_body(Grid g, int k, double* __restrict__ mean, const int bx, const int gx)
{
    __shared__ double cand[kT / kG][kKnnCand];
    __shared__ int ncand[kT / kG];
    const int lane = threadIdx.x % kG, grp = threadIdx.x / kG;
    const int i = (bx * kT + threadIdx.x) / kG;
    if (i >= g.n) return;                                    // whole groups leave together; the LDS traffic below is per group (one wave
                                                             // holds two groups: same-wave program order, no block barrier needed)
    const double q[3] = {g.sorted[(size_t)i * 3], g.sorted[(size_t)i * 3 + 1], g.sorted[(size_t)i * 3 + 2]};
    long c[3];
    cell_of(q, g.origin, g.h, c);
    double sum = 0;
    bool settled = false;
    for (int R = 1; R <= 3 && !settled; ++R) {
        if (lane == 0) ncand[grp] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int S = 2 * R + 1;
        for (int cc = lane; cc < S * S * S; cc += kG) {
            const long cx = c[0] + cc / (S * S) - R, cy = c[1] + (cc / S) % S - R, cz = c[2] + cc % S - R;
            if (cx < 0 || cx > 2097151 || cy < 0 || cy > 2097151 || cz < 0 || cz > 2097151) continue;
            const u64 key = pack_key(cx, cy, cz);
            for (int j = lower_bound(g.keys, g.n, key); j < g.n && g.keys[j] == key; ++j) {
                const double ex = g.sorted[(size_t)j * 3] - q[0], ey = g.sorted[(size_t)j * 3 + 1] - q[1], ez = g.sorted[(size_t)j * 3 + 2] - q[2];
                const int p = atomicAdd(&ncand[grp], 1);
                if (p < kKnnCand) cand[grp][p] = (ex * ex + ey * ey) + ez * ez;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int nc = ncand[grp];
        if (nc < k || nc > kKnnCand) { if (nc > kKnnCand) break; continue; }
        double last_d = -1.0;
        int last_p = -1;
        sum = 0;
        for (int r = 0; r < k; ++r) {
            double bd = 1e300;
            int bp = 0x7fffffff;
            for (int p = lane; p < nc; p += kG) {
                const double d = cand[grp][p];
                const bool after = d > last_d || (d == last_d && p > last_p);
                if (after && (d < bd || (d == bd && p < bp))) { bd = d; bp = p; }
            }
            knn_round_reduce(bd, bp);
            last_d = bd; last_p = bp;
            sum += sqrt(bd);
        }
        const double bound = (double)R * g.h;
        settled = last_d < bound * bound * (1.0 - 1e-12);
    }
    if (!settled) {
        double last_d = -1.0;
        int last_p = -1;
        sum = 0;
        for (int r = 0; r < k; ++r) {
            double bd = 1e300;
            int bp = 0x7fffffff;
            auto offer = [&](int p, double x, double y, double z) {
                const double ex = x - q[0], ey = y - q[1], ez = z - q[2];
                const double d = (ex * ex + ey * ey) + ez * ez;
                const bool after = d > last_d || (d == last_d && p > last_p);
                if (after && (d < bd || (d == bd && p < bp))) { bd = d; bp = p; }
            };
            int p = lane;
            for (; p + 3 * kG < g.n; p += 4 * kG) {           // four independent point loads in flight per lane
                double v[4][3];
#pragma unroll
                for (int u = 0; u < 4; ++u) for (int d = 0; d < 3; ++d) v[u][d] = g.sorted[(size_t)(p + u * kG) * 3 + d];
#pragma unroll
                for (int u = 0; u < 4; ++u) offer(p + u * kG, v[u][0], v[u][1], v[u][2]);
            }
            for (; p < g.n; p += kG) offer(p, g.sorted[(size_t)p * 3], g.sorted[(size_t)p * 3 + 1], g.sorted[(size_t)p * 3 + 2]);
            knn_round_reduce(bd, bp);
            last_d = bd; last_p = bp;
            sum += sqrt(bd);
        }
    }
    if (lane == 0) mean[g.order[i]] = sum / (double)k;
}

__global__ __launch_bounds__(kT) void knn_mean_grid_kernel(Grid g, int k, double* __restrict__ mean)
{
    knn_mean_grid_kernel_body(g, k, mean, blockIdx.x, gridDim.x);
}

// ---------------------------------------------------------------------------------------------------------------
// generic fixed-order reduction of NV doubles per element: stage 1 -> part[blocks][NV], stage 2 -> out[NV]
template <int NV, class F>
__device__ void reduce_stage1(int n, F value, double* part, const int bx, const int gx)
{
    __shared__ double s[kT];
    double acc[NV];
    for (int v = 0; v < NV; ++v) acc[v] = 0;
    for (int i = bx * blockDim.x + threadIdx.x; i < n; i += gx * blockDim.x) {
        double e[NV];
        if (value(i, e)) for (int v = 0; v < NV; ++v) acc[v] += e[v];
    }
    for (int v = 0; v < NV; ++v) {
        __syncthreads();
        s[threadIdx.x] = acc[v];
        __syncthreads();
        for (int off = kT / 2; off > 0; off >>= 1) { if (threadIdx.x < off) s[threadIdx.x] += s[threadIdx.x + off]; __syncthreads(); }
        if (threadIdx.x == 0) part[bx * NV + v] = s[0];
    }
}

__device__ __forceinline__ void reduce_stage2_body(const double* __restrict__ part, int g, int nv, double* __restrict__ out, const double* __restrict__ skip, const int bx, const int gx)
{
    if (skip && skip[0] != 0.0) return;
    const int v = threadIdx.x;
    if (v < nv) { double s = 0; for (int b = 0; b < g; ++b) s += part[b * nv + v]; out[v] = s; }
}

__global__ void reduce_stage2(const double* __restrict__ part, int g, int nv, double* __restrict__ out, const double* __restrict__ skip = nullptr)
{
    reduce_stage2_body(part, g, nv, out, skip, blockIdx.x, gridDim.x);
}

// out: [0] count, [1] sum d^2, [2..4] sum s, [5..7] sum t, [8..16] sum s_a t_b (row a, col b)
__device__ __forceinline__ void p2p_sums_kernel_body(const double* __restrict__ src, const double* __restrict__ tgt, const int* __restrict__ corr, const double* __restrict__ d2, int n, double* __restrict__ part, const double* __restrict__ skip, const int bx, const int gx)
{
    if (skip && skip[0] != 0.0) return;
    reduce_stage1<17>(n, [&](int i, double* e) {
        const int j = corr[i];
        if (j < 0) return false;
        const double* s = src + (size_t)i * 3;
        const double* t = tgt + (size_t)j * 3;
        e[0] = 1.0; e[1] = d2[i];
        for (int a = 0; a < 3; ++a) { e[2 + a] = s[a]; e[5 + a] = t[a]; for (int b = 0; b < 3; ++b) e[8 + a * 3 + b] = s[a] * t[b]; }
        return true;
    }, part, bx, gx);
}

__global__ __launch_bounds__(kT) void p2p_sums_kernel(const double* __restrict__ src, const double* __restrict__ tgt, const int* __restrict__ corr,
                                                      const double* __restrict__ d2, int n, double* __restrict__ part,
                                                      const double* __restrict__ skip = nullptr)
{
    p2p_sums_kernel_body(src, tgt, corr, d2, n, part, skip, blockIdx.x, gridDim.x);
}

// out: [0] count, [1] sum d^2, [2..22] upper triangle of J^T J (row major), [23..28] J^T r
__device__ __forceinline__ void p2plane_sums_kernel_body(const double* __restrict__ src, const double* __restrict__ tgt, const double* __restrict__ tn, const int* __restrict__ corr, const double* __restrict__ d2, int n, double* __restrict__ part, const double* __restrict__ skip, const int bx, const int gx)
{
    if (skip && skip[0] != 0.0) return;
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
    }, part, bx, gx);
}

__global__ __launch_bounds__(kT) void p2plane_sums_kernel(const double* __restrict__ src, const double* __restrict__ tgt,
                                                          const double* __restrict__ tn, const int* __restrict__ corr,
                                                          const double* __restrict__ d2, int n, double* __restrict__ part,
                                                          const double* __restrict__ skip = nullptr)
{
    p2plane_sums_kernel_body(src, tgt, tn, corr, d2, n, part, skip, blockIdx.x, gridDim.x);
}

// out: [0..2] sum p, [3..8] sum p_a p_b upper triangle
__device__ __forceinline__ void moments_kernel_body(const double* __restrict__ pts, int n, double* __restrict__ part, const int bx, const int gx)
{
    reduce_stage1<9>(n, [&](int i, double* e) {
        const double* p = pts + (size_t)i * 3;
        e[0] = p[0]; e[1] = p[1]; e[2] = p[2];
        e[3] = p[0] * p[0]; e[4] = p[0] * p[1]; e[5] = p[0] * p[2]; e[6] = p[1] * p[1]; e[7] = p[1] * p[2]; e[8] = p[2] * p[2];
        return true;
    }, part, bx, gx);
}

__global__ __launch_bounds__(kT) void moments_kernel(const double* __restrict__ pts, int n, double* __restrict__ part)
{
    moments_kernel_body(pts, n, part, blockIdx.x, gridDim.x);
}

struct Vec12 { double v[12]; };
// sqrt((p - mu)^T Cinv (p - mu));  mc = (mu[3], Cinv[9])
__device__ __forceinline__ void mahalanobis_kernel_body(const double* __restrict__ pts, int n, Vec12 mc, double* __restrict__ out, const int bx, const int gx)
{
    for (int i = bx * blockDim.x + threadIdx.x; i < n; i += gx * blockDim.x) {
        double e[3];
        for (int d = 0; d < 3; ++d) e[d] = pts[(size_t)i * 3 + d] - mc.v[d];
        double s = 0;
        for (int a = 0; a < 3; ++a) s += e[a] * ((mc.v[3 + a * 3] * e[0] + mc.v[3 + a * 3 + 1] * e[1]) + mc.v[3 + a * 3 + 2] * e[2]);
        out[i] = sqrt(s);
    }
}

__global__ void mahalanobis_kernel(const double* __restrict__ pts, int n, Vec12 mc, double* __restrict__ out)
{
    mahalanobis_kernel_body(pts, n, mc, out, blockIdx.x, gridDim.x);
}

__device__ __forceinline__ void select_rows_kernel_body(const double* __restrict__ pts, const int* __restrict__ sel, const int* __restrict__ n_sel, double* __restrict__ out, const int bx, const int gx)
{
    const int n = *n_sel;
    for (int i = bx * blockDim.x + threadIdx.x; i < n; i += gx * blockDim.x)
        for (int d = 0; d < 3; ++d) out[(size_t)i * 3 + d] = pts[(size_t)sel[i] * 3 + d];
}


size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

struct Carver {
    char* p; char* end;
    template <class T> T* take(size_t count) { T* r = (T*)p; p += align_up(count * sizeof(T)); return p <= end ? r : nullptr; }
};

Mat4 load_mat(const double* T16_host) { Mat4 m; for (int i = 0; i < 16; ++i) m.m[i] = T16_host[i]; return m; }

}  // namespace

extern "C" int ape_transform_points_f64(double* pts, double* normals_or_null, int n, const double* T16_host, void* stream)
{
    if (!pts || !T16_host || n < 0) return APE_EINVAL;
    if (n == 0) return APE_OK;
    hipLaunchKernelGGL(transform_kernel, dim3(grid_for(n)), dim3(kT), 0, (hipStream_t)stream, pts, n, load_mat(T16_host), normals_or_null);
    return ape::check_launch("ape_transform_points_f64");
}

#define GRID_ARGS const double* sorted, const unsigned long long* keys, const unsigned* order, const double* origin3, int n, double cell
#define MAKE_GRID Grid g{sorted, (const u64*)keys, order, origin3, n, cell}

extern "C" int ape_grid_radius_count_f64(GRID_ARGS, const double* q, int nq, double radius, int* count, void* stream)
{
    if (!sorted || !keys || !order || !origin3 || !q || !count || n < 1 || nq < 0 || radius > cell) return APE_EINVAL;
    if (nq == 0) return APE_OK;
    MAKE_GRID;
    hipLaunchKernelGGL(radius_count_group_kernel, dim3(ape::ceil_div((long)nq * kG, (long)kT)), dim3(kT), 0, (hipStream_t)stream, g, q, nq, radius * radius, count);
    return ape::check_launch("ape_grid_radius_count_f64");
}

extern "C" int ape_grid_nn1_f64(GRID_ARGS, const double* q, int nq, double max_dist, int* idx, double* dist2, void* stream)
{
    if (!sorted || !keys || !order || !origin3 || !q || !idx || !dist2 || n < 1 || nq < 0 || max_dist > cell) return APE_EINVAL;
    if (nq == 0) return APE_OK;
    MAKE_GRID;
    hipLaunchKernelGGL(nn1_group_kernel, dim3(ape::ceil_div((long)nq * kG, (long)kT)), dim3(kT), 0, (hipStream_t)stream, g, q, nq, max_dist * max_dist, idx, dist2, (const double*)nullptr);
    return ape::check_launch("ape_grid_nn1_f64");
}

extern "C" int ape_grid_normals_f64(GRID_ARGS, const double* q, int nq, double radius, int max_nn, double* normals, void* stream)
{
    if (!sorted || !keys || !order || !origin3 || !q || !normals || n < 1 || nq < 0 || radius > cell || max_nn < 3 || max_nn > kMaxNN)
        return APE_EINVAL;
    if (nq == 0) return APE_OK;
    MAKE_GRID;
    hipLaunchKernelGGL(normals_kernel, dim3(ape::ceil_div((long)nq * kG, (long)kT)), dim3(kT), 0, (hipStream_t)stream, g, q, nq, radius * radius, max_nn, normals);
    return ape::check_launch("ape_grid_normals_f64");
}

extern "C" int ape_knn_mean_dist_f64(const double* pts, int n, int k, double* mean, void* stream)
{
    if (!pts || !mean || n < 1 || k < 1 || k > kMaxNN) return APE_EINVAL;
    hipLaunchKernelGGL(knn_mean_kernel, dim3(ape::ceil_div(n, kT)), dim3(kT), 0, (hipStream_t)stream, pts, n, k, mean);
    return ape::check_launch("ape_knn_mean_dist_f64");
}

extern "C" int ape_grid_knn_mean_dist_f64(GRID_ARGS, int k, double* mean, void* stream)
{
    if (!sorted || !keys || !order || !origin3 || !mean || n < 1 || k < 1 || k > kMaxNN || k > n || !(cell > 0)) return APE_EINVAL;
    MAKE_GRID;
    hipLaunchKernelGGL(knn_mean_grid_kernel, dim3(ape::ceil_div((long)n * kG, (long)kT)), dim3(kT), 0, (hipStream_t)stream, g, k, mean);
    return ape::check_launch("ape_grid_knn_mean_dist_f64");
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

// ---------------------------------------------------------------------------------------------------------------------------------
// ICP iteration entirely on the device (open3d 0.9 RegistrationICP as called at pc_reconstruction/open3d_utils.py:96-117): the
// correspondence search and the 17 / 29 reduced sums were device work already; here the 3x3 SVD (Umeyama, point-to-point) / the 6x6
// solve (point-to-plane), the composition T <- update . T, the fitness / rmse bookkeeping and the convergence test run on the tail of
// the reduction kernel (icp_reduce_step_kernel), so a registration needs ONE device-to-host copy (its result) instead of one per
// iteration, and an iteration is three launches: [apply the pending update +] correspondence search, partial sums, reduce + step.
// state[40]: [0] done, [1] updates applied, [2] fitness, [3] inlier rmse, [4] correspondences, [5..20] T (row major), [21..36] the
// last update, [37] 1 = converged by the relative criteria, 2 = too few correspondences, 3 = iteration limit.
namespace {

__device__ void svd3_rotation(const double C[3][3], double R[3][3])
{
    // R = U diag(1, 1, det(U) det(V)) V^T for C = U S V^T (Eigen::umeyama without scaling).  One-sided (Hestenes) Jacobi: rotate
    // column pairs of A = C until orthogonal: A V' = U S.
    double A[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) A[i][j] = C[i][j];
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int k = 0; k < 3; ++k) { alpha += A[k][p] * A[k][p]; beta += A[k][q] * A[k][q]; gamma += A[k][p] * A[k][q]; }
                if (gamma == 0.0 || fabs(gamma) <= 1e-17 * sqrt(alpha * beta)) continue;
                off += fabs(gamma);
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                for (int k = 0; k < 3; ++k) {
                    const double ap = A[k][p], aq = A[k][q];
                    A[k][p] = c * ap - sn * aq; A[k][q] = sn * ap + c * aq;
                    const double vp = V[k][p], vq = V[k][q];
                    V[k][p] = c * vp - sn * vq; V[k][q] = sn * vp + c * vq;
                }
            }
        if (off == 0.0) break;
    }
    double sig[3];
    for (int j = 0; j < 3; ++j) sig[j] = sqrt(A[0][j] * A[0][j] + A[1][j] * A[1][j] + A[2][j] * A[2][j]);
    int ord[3] = {0, 1, 2};                               // descending singular values
    for (int a = 0; a < 2; ++a) for (int b2 = a + 1; b2 < 3; ++b2) if (sig[ord[b2]] > sig[ord[a]]) { const int t = ord[a]; ord[a] = ord[b2]; ord[b2] = t; }
    double U[3][3], W[3][3];
    for (int j = 0; j < 3; ++j) {
        const int o = ord[j];
        for (int k = 0; k < 3; ++k) { W[k][j] = V[k][o]; U[k][j] = sig[o] > 0 ? A[k][o] / sig[o] : 0.0; }
    }
    if (!(sig[ord[0]] > 0)) {                             // zero covariance (all pairs coincide with their centroids): no rotation to find
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[i][j] = i == j ? 1.0 : 0.0;
        return;
    }
    const double tiny = 1e-13 * sig[ord[0]];
    if (sig[ord[1]] <= tiny) {                            // rank <= 1: complete U with any orthonormal pair
        double e[3] = {fabs(U[0][0]) < 0.9 ? 1.0 : 0.0, fabs(U[0][0]) < 0.9 ? 0.0 : 1.0, 0.0};
        double d = e[0] * U[0][0] + e[1] * U[1][0] + e[2] * U[2][0];
        double n2 = 0;
        for (int k = 0; k < 3; ++k) { e[k] -= d * U[k][0]; n2 += e[k] * e[k]; }
        for (int k = 0; k < 3; ++k) U[k][1] = e[k] / sqrt(n2);
    }
    if (sig[ord[2]] <= tiny) {                            // rank 2: third left vector = u0 x u1
        U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
        U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
        U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
    }
    auto det3 = [](const double M[3][3]) {
        return M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) + M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
    };
    const double s33 = det3(U) * det3(W) < 0 ? -1.0 : 1.0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R[i][j] = (U[i][0] * W[j][0] + U[i][1] * W[j][1]) + s33 * U[i][2] * W[j][2];
}

__device__ bool solve6(double M[6][7])
{
    // Gaussian elimination with partial pivoting on the augmented system (numpy.linalg.solve = LAPACK gesv); false when exactly singular
    for (int c = 0; c < 6; ++c) {
        int piv = c;
        for (int r = c + 1; r < 6; ++r) if (fabs(M[r][c]) > fabs(M[piv][c])) piv = r;
        if (M[piv][c] == 0.0) return false;
        if (piv != c) for (int k = 0; k < 7; ++k) { const double t = M[c][k]; M[c][k] = M[piv][k]; M[piv][k] = t; }
        for (int r = c + 1; r < 6; ++r) {
            const double f = M[r][c] / M[c][c];
            for (int k = c; k < 7; ++k) M[r][k] -= f * M[c][k];
        }
    }
    for (int r = 5; r >= 0; --r) {
        double v = M[r][6];
        for (int k = r + 1; k < 6; ++k) v -= M[r][k] * M[k][6];
        M[r][6] = v / M[r][r];
    }
    return true;
}

__device__ void icp_step(int kind, const double* s, double* __restrict__ st, int ns, double rel_fitness, double rel_rmse, int max_iteration)
{
    if (st[0] != 0.0) return;
    const double n_corr = floor(s[0] + 0.5);
    const double fitness = n_corr / (double)ns;
    const double rmse = n_corr > 0 ? sqrt(s[1] / n_corr) : 0.0;
    const bool first = st[1] == 0.0 && st[38] == 0.0;      // st[38]: set once the first evaluation has been recorded
    const double pf = st[2], pr = st[3];
    st[2] = fitness; st[3] = rmse; st[4] = n_corr; st[38] = 1.0;
    if (!first && fabs(pf - fitness) < rel_fitness && fabs(pr - rmse) < rel_rmse) { st[0] = 1.0; st[37] = 1.0; return; }
    if (st[1] >= (double)max_iteration) { st[0] = 1.0; st[37] = 3.0; return; }
    if (n_corr < (kind == 0 ? 3.0 : 6.0)) { st[0] = 1.0; st[37] = 2.0; return; }
    double U[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    if (kind == 0) {
        const double n = s[0];
        double mu_s[3], mu_t[3], C[3][3], R[3][3];
        for (int a = 0; a < 3; ++a) { mu_s[a] = s[2 + a] / n; mu_t[a] = s[5 + a] / n; }
        for (int a = 0; a < 3; ++a) for (int b2 = 0; b2 < 3; ++b2) C[a][b2] = s[8 + b2 * 3 + a] / n - mu_t[a] * mu_s[b2];     // (1/n) sum (t - mu_t)(s - mu_s)^T
        svd3_rotation(C, R);
        for (int a = 0; a < 3; ++a) {
            for (int b2 = 0; b2 < 3; ++b2) U[a][b2] = R[a][b2];
            U[a][3] = mu_t[a] - ((R[a][0] * mu_s[0] + R[a][1] * mu_s[1]) + R[a][2] * mu_s[2]);
        }
    } else {
        double M[6][7];
        int k = 2;
        for (int a = 0; a < 6; ++a) for (int b2 = a; b2 < 6; ++b2) { M[a][b2] = M[b2][a] = s[k]; ++k; }
        for (int a = 0; a < 6; ++a) M[a][6] = -s[23 + a];
        if (solve6(M)) {
            const double x[6] = {M[0][6], M[1][6], M[2][6], M[3][6], M[4][6], M[5][6]};
            const double cx = cos(x[0]), sx = sin(x[0]), cy = cos(x[1]), sy = sin(x[1]), cz = cos(x[2]), sz = sin(x[2]);
            // TransformVector6dToMatrix4d: R = Rz(x2) Ry(x1) Rx(x0)
            const double Rx[3][3] = {{1, 0, 0}, {0, cx, -sx}, {0, sx, cx}}, Ry[3][3] = {{cy, 0, sy}, {0, 1, 0}, {-sy, 0, cy}}, Rz[3][3] = {{cz, -sz, 0}, {sz, cz, 0}, {0, 0, 1}};
            double T1[3][3];
            for (int a = 0; a < 3; ++a) for (int b2 = 0; b2 < 3; ++b2) T1[a][b2] = (Rz[a][0] * Ry[0][b2] + Rz[a][1] * Ry[1][b2]) + Rz[a][2] * Ry[2][b2];
            for (int a = 0; a < 3; ++a) {
                for (int b2 = 0; b2 < 3; ++b2) U[a][b2] = (T1[a][0] * Rx[0][b2] + T1[a][1] * Rx[1][b2]) + T1[a][2] * Rx[2][b2];
                U[a][3] = x[3 + a];
            }
        }
    }
    double T[4][4], Tn[4][4];
    for (int a = 0; a < 4; ++a) for (int b2 = 0; b2 < 4; ++b2) T[a][b2] = st[5 + a * 4 + b2];
    for (int a = 0; a < 4; ++a)
        for (int b2 = 0; b2 < 4; ++b2) Tn[a][b2] = ((U[a][0] * T[0][b2] + U[a][1] * T[1][b2]) + U[a][2] * T[2][b2]) + U[a][3] * T[3][b2];
    for (int a = 0; a < 4; ++a) for (int b2 = 0; b2 < 4; ++b2) { st[5 + a * 4 + b2] = Tn[a][b2]; st[21 + a * 4 + b2] = U[a][b2]; }
    st[1] += 1.0;
}

// last stage of the sums reduction (one thread per reduced value, fixed order over the workgroup partials) with the ICP step on its
// tail: thread 0 runs icp_step on the totals it finds in LDS -- one launch instead of two per iteration
__device__ __forceinline__ void icp_reduce_step_kernel_body(const double* __restrict__ part, int g, int nv, double* __restrict__ sums, int kind, double* __restrict__ st, int ns, double rel_fitness, double rel_rmse, int max_iteration, const int bx, const int gx)
{
    if (st[0] != 0.0) return;
    __shared__ double tot[32];
    const int v = threadIdx.x;
    if (v < nv) { double a = 0; for (int b = 0; b < g; ++b) a += part[b * nv + v]; tot[v] = a; sums[v] = a; }
    __syncthreads();
    if (v == 0) icp_step(kind, tot, st, ns, rel_fitness, rel_rmse, max_iteration);
}

__global__ __launch_bounds__(64) void icp_reduce_step_kernel(const double* __restrict__ part, int g, int nv, double* __restrict__ sums, int kind,
                                                             double* __restrict__ st, int ns, double rel_fitness, double rel_rmse, int max_iteration)
{
    icp_reduce_step_kernel_body(part, g, nv, sums, kind, st, ns, rel_fitness, rel_rmse, max_iteration, blockIdx.x, gridDim.x);
}

// correspondence search of the ICP loop with the pending update applied on the way: the query is moved by state[21..36] (the update the
// previous step computed), written back, and searched -- kG lanes per query as in nn1_group_kernel (every lane moves its copy)
__device__ __forceinline__ void icp_move_nn1_kernel_body(Grid g, double* __restrict__ src, int nq, double r2, int* __restrict__ idx, double* __restrict__ dist2, const double* __restrict__ st, int apply, const int bx, const int gx)
{
    if (st[0] != 0.0) return;
    const int lane = threadIdx.x % kG;
    const int i = (bx * kT + threadIdx.x) / kG;
    double q[3] = {0, 0, 0};
    if (i < nq) {
        const double x = src[(size_t)i * 3], y = src[(size_t)i * 3 + 1], z = src[(size_t)i * 3 + 2];
        q[0] = x; q[1] = y; q[2] = z;
        if (apply) {
            const double* T = st + 21;
            for (int r = 0; r < 3; ++r) q[r] = ((T[r * 4] * x + T[r * 4 + 1] * y) + T[r * 4 + 2] * z) + T[r * 4 + 3];
        }
    }
    double best = r2;
    unsigned bi = 0xffffffffu;
    if (i < nq)
        for_my_cell(g, q, lane, [&](int j, double d2) {
            const unsigned o = g.order[j];
            if (d2 < best || (d2 == best && bi != 0xffffffffu && o < bi)) { best = d2; bi = o; }
        });
    for (int m = kG / 2; m >= 1; m >>= 1) {
        const double ob = __shfl_xor(best, m, kG);
        const unsigned oi = __shfl_xor(bi, m, kG);
        if (oi != 0xffffffffu && (bi == 0xffffffffu || ob < best || (ob == best && oi < bi))) { best = ob; bi = oi; }
    }
    if (i < nq && lane == 0) {
        if (apply) { src[(size_t)i * 3] = q[0]; src[(size_t)i * 3 + 1] = q[1]; src[(size_t)i * 3 + 2] = q[2]; }
        idx[i] = bi == 0xffffffffu ? -1 : (int)bi;
        dist2[i] = bi == 0xffffffffu ? 0.0 : best;
    }
}

__global__ __launch_bounds__(kT) void icp_move_nn1_kernel(Grid g, double* __restrict__ src, int nq, double r2, int* __restrict__ idx,
                                                          double* __restrict__ dist2, const double* __restrict__ st, int apply)
{
    icp_move_nn1_kernel_body(g, src, nq, r2, idx, dist2, st, apply, blockIdx.x, gridDim.x);
}

}  // namespace

/* Enqueue `n_iter` ICP iterations (kind 0 point-to-point, 1 point-to-plane) with everything on the device; see the block comment above.
 * `src` [ns][3] is the source ALREADY transformed by the initial guess and is updated in place; `state` [40] doubles on the device: the
 * caller zeroes it and writes the initial T into state[5..20] before the FIRST call of a registration (first_call = 1 also runs the
 * evaluation that precedes open3d's loop and the step that computes the first update), and reads it back (one copy) after each call:
 * state[0] != 0 means finished.  An iteration = apply the pending update, evaluate, step; `max_iteration` of them exhaust the limit.  Grid arguments:
 * the target's search grid from ape_grid_build_f64 (cell >= max_dist).  ws: n-blocks x 29 doubles as for ape_icp_sums_f64. */
extern "C" int ape_icp_run_f64(int kind, GRID_ARGS, double* src, int ns, const double* tgt, const double* tgt_normals, double max_dist,
                               double rel_fitness, double rel_rmse, int max_iteration, int n_iter, int first_call, int* corr, double* dist2,
                               double* sums, double* state, void* ws, size_t ws_bytes, void* stream)
{
    if (kind < 0 || kind > 1 || !sorted || !keys || !order || !origin3 || !src || !tgt || !corr || !dist2 || !sums || !state || !ws) return APE_EINVAL;
    if (kind == 1 && !tgt_normals) return APE_EINVAL;
    if (n < 1 || ns < 1 || max_dist > cell || n_iter < 0 || max_iteration < 0) return APE_EINVAL;
    const int nv = kind == 0 ? 17 : 29;
    const int nb = grid_for(ns, 512);
    if (ws_bytes < (size_t)nb * nv * 8) return APE_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    MAKE_GRID;
    double* part = (double*)ws;
    auto block = [&](int apply) {         // [move +] search, sums, reduce + step: three launches
        hipLaunchKernelGGL(icp_move_nn1_kernel, dim3(ape::ceil_div((long)ns * kG, (long)kT)), dim3(kT), 0, st, g, src, ns, max_dist * max_dist, corr, dist2,
                           (const double*)state, apply);
        if (kind == 0) hipLaunchKernelGGL(p2p_sums_kernel, dim3(nb), dim3(kT), 0, st, (const double*)src, tgt, (const int*)corr, (const double*)dist2, ns, part, (const double*)state);
        else hipLaunchKernelGGL(p2plane_sums_kernel, dim3(nb), dim3(kT), 0, st, (const double*)src, tgt, tgt_normals, (const int*)corr, (const double*)dist2, ns, part, (const double*)state);
        hipLaunchKernelGGL(icp_reduce_step_kernel, dim3(1), dim3(64), 0, st, (const double*)part, nb, nv, sums, kind, state, ns, rel_fitness, rel_rmse, max_iteration);
    };
    if (first_call) block(0);
    for (int it = 0; it < n_iter; ++it) block(1);
    return ape::check_launch("ape_icp_run_f64");
}


// =====================================================================================================================================
// BATCHED forms (`*_batch_f64`): the label path's clouds are tiny (10^3..10^4 points) and every step of a chain depends on the one before,
// so a chain by itself is a string of 5-20 us kernels behind dependent launches -- 38.6 k launches and 13.6 k small copies per 200-view
// step in round 2, three host threads fighting over the GIL to keep the GPU fed.  The (object, direction) chains are independent of each
// other (create_pointcloud.py:276-312: one sequential fusion per rotation directory), so here ONE launch advances up to kMaxBatch chains:
// blockIdx.y selects the chain's argument record (passed by value in the kernel arguments), blockIdx.x walks that chain's own grid -- the
// SAME `*_body` device code with the SAME per-chain grid size as the one-cloud entry points above (the fixed-order two-stage reductions
// keep their partial layout), hence bit-identical results.  hipCUB's sort becomes ONE segmented radix sort over all chains (same stable
// LSD radix per segment); its select / scan calls become a single-workgroup ordered compaction / scan per chain.
namespace {

constexpr int kMaxBatch = 16;
template <class T> struct Batch { T t[kMaxBatch]; };
constexpr int kCT = 1024;           // threads of the single-workgroup compaction / scan kernels

// ordered compaction by one workgroup: out_idx[k] = i for the k-th i in [0, n) with pred(i); returns the count in *n_out.
// Chunks of kCT * 4 consecutive elements; per chunk: every thread's 4-element count -> wave prefix by shuffles -> wave totals through LDS.
template <class P>
__device__ __forceinline__ void compact_block(int n, P pred, int* __restrict__ out_idx, int* __restrict__ n_out)
{
    __shared__ int wtot[kCT / 64];
    __shared__ int base_s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n; c0 += kCT * 4) {
        const int i0 = c0 + tid * 4;
        bool f[4];
        int cnt = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) { f[u] = i0 + u < n && pred(i0 + u); cnt += f[u] ? 1 : 0; }
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off, 64); if (lane >= off) incl += o; }
        if (lane == 63) wtot[wv] = incl;
        __syncthreads();
        int wbase = 0, total = 0;
#pragma unroll
        for (int w = 0; w < kCT / 64; ++w) { const int t = wtot[w]; if (w < wv) wbase += t; total += t; }
        int pos = base_s + wbase + incl - cnt;
#pragma unroll
        for (int u = 0; u < 4; ++u) if (f[u]) out_idx[pos++] = i0 + u;
        __syncthreads();
        if (tid == 0) base_s += total;
        __syncthreads();
    }
    if (tid == 0) *n_out = base_s;
}

// exclusive scan of head[0..n) by one workgroup -> scan[]
__device__ __forceinline__ void scan_block(const int* __restrict__ head, int n, int* __restrict__ scan)
{
    __shared__ int wtot[kCT / 64];
    __shared__ int base_s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n; c0 += kCT * 4) {
        const int i0 = c0 + tid * 4;
        int v[4], cnt = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) { v[u] = i0 + u < n ? head[i0 + u] : 0; cnt += v[u]; }
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off, 64); if (lane >= off) incl += o; }
        if (lane == 63) wtot[wv] = incl;
        __syncthreads();
        int wbase = 0, total = 0;
#pragma unroll
        for (int w = 0; w < kCT / 64; ++w) { const int t = wtot[w]; if (w < wv) wbase += t; total += t; }
        int run = base_s + wbase + incl - cnt;
#pragma unroll
        for (int u = 0; u < 4; ++u) if (i0 + u < n) { scan[i0 + u] = run; run += v[u]; }
        __syncthreads();
        if (tid == 0) base_s += total;
        __syncthreads();
    }
}

// ---- argument records (one per chain) -----------------------------------------------------------------------------------------------
struct BSurf { const uint8_t* label; const uint16_t* depth; int* pix; int* n_out; double* out; Mat4 T; int n, W, gx; double fx, fy, ppx, ppy; };
struct BPts { const double* pts; double* part; double* out6; int n, gx; };                        // bounds
struct BKeys { const double* pts; const double* bounds6; u64* keys; unsigned* idx; double* origin; int n, gx; double h, shift; };
struct BVox { const double* pts; const u64* keys; const unsigned* order; int* head; int* scan; double* out; int* n_out; int n, gx; };
struct BGather { const double* pts; const unsigned* order; double* out; int n, gx; };
struct BGridQ { Grid g; const double* q; int nq, gx; double r2; int* count; double* normals; int max_nn; double* mean; int k; };
struct BSel { const double* pts; const int* count; const double* mean; int* sel; int* n_out; double* out; int n, gx, thr_count; double thr_mean; int mode; const uint8_t* keep; };
struct BMom { const double* pts; double* part; double* out; int n, gx; };
struct BMaha { const double* pts; double* out; int n, gx; Vec12 mc; };
struct BXform { double* pts; double* normals; Mat4 T; int n, gx; };
struct BCopy { const double* a; const double* b; double* out; int na, nb, gx; };
struct BIcp { Grid g; double* src; const double* tgt; const double* tn; int* corr; double* d2; double* part; double* sums; double* st; int ns, gx_nn, gx_sum; };

__global__ __launch_bounds__(kCT) void surface_compact_batch(const Batch<BSurf> b)
{
    const BSurf& a = b.t[blockIdx.y];
    if (a.n <= 0) return;
    compact_block(a.n, [&](int i) { return a.label[i] != 0 && a.depth[i] != 0; }, a.pix, a.n_out);
}
__global__ void surface_points_batch(const Batch<BSurf> b)
{
    const BSurf& a = b.t[blockIdx.y];
    if ((int)blockIdx.x >= a.gx) return;
    surface_points_kernel_body(a.pix, a.n_out, a.depth, a.W, a.fx, a.fy, a.ppx, a.ppy, a.T, a.out, blockIdx.x, a.gx);
}
__global__ void bounds1_batch(const Batch<BPts> b)
{
    const BPts& a = b.t[blockIdx.y];
    if ((int)blockIdx.x >= a.gx) return;
    bounds_stage1_body(a.pts, a.n, a.part, blockIdx.x, a.gx);
}
__global__ void bounds2_batch(const Batch<BPts> b)
{
    const BPts& a = b.t[blockIdx.y];
    if (a.gx <= 0) return;
    bounds_stage2_body(a.part, a.gx, a.out6, 0, 1);
}
__global__ void keys_batch(const Batch<BKeys> b)
{
    const BKeys& a = b.t[blockIdx.y];
    if ((int)blockIdx.x >= a.gx) return;
    keys_kernel_body(a.pts, a.n, a.bounds6, a.h, a.shift, a.keys, a.idx, a.origin, blockIdx.x, a.gx);
}
__global__ void heads_batch(const Batch<BVox> b)
{
    const BVox& a = b.t[blockIdx.y];
    if ((int)blockIdx.x >= a.gx) return;
    heads_kernel_body(a.keys, a.n, a.head, blockIdx.x, a.gx);
}
__global__ __launch_bounds__(kCT) void scan_batch(const Batch<BVox> b)
{
    const BVox& a = b.t[blockIdx.y];
    if (a.n <= 0) return;
    scan_block(a.head, a.n, a.scan);
}
__global__ void voxel_mean_batch(const Batch<BVox> b)
{
    const BVox& a = b.t[blockIdx.y];
    if ((int)blockIdx.x >= a.gx) return;
    voxel_mean_kernel_body(a.pts, a.keys, a.order, a.head, a.scan, a.n, a.out, a.n_out, blockIdx.x, a.gx);
}
__global__ void gather_batch(const Batch<BGather> b)
{
    const BGather& a = b.t[blockIdx.y];
    if ((int)blockIdx.x >= a.gx) return;
    gather_sorted_kernel_body(a.pts, a.order, a.n, a.out, blockIdx.x, a.gx);
}
__global__ __launch_bounds__(kT) void radius_count_batch(const Batch<BGridQ> b)
{
    const BGridQ& a = b.t[blockIdx.y];
    if ((int)blockIdx.x >= a.gx) return;
    radius_count_group_kernel_body(a.g, a.q, a.nq, a.r2, a.count, blockIdx.x, a.gx);
}
__global__ __launch_bounds__(kT) void normals_batch(const Batch<BGridQ> b)
{
    const BGridQ& a = b.t[blockIdx.y];
    if ((int)blockIdx.x >= a.gx) return;
    normals_kernel_body(a.g, a.q, a.nq, a.r2, a.max_nn, a.normals, blockIdx.x, a.gx);
}
__global__ __launch_bounds__(kT) void knn_mean_batch(const Batch<BGridQ> b)
{
    const BGridQ& a = b.t[blockIdx.y];
    if ((int)blockIdx.x >= a.gx) return;
    knn_mean_grid_kernel_body(a.g, a.k, a.mean, blockIdx.x, a.gx);
}
// keep rule evaluated on the device: mode 0: count[i] > thr_count (RemoveRadiusOutliers); mode 1: mean[i] > 0 && mean[i] < thr_mean
// (RemoveStatisticalOutliers; the threshold comes from the host's float64 statistics of the means, as in the one-cloud path)
__global__ __launch_bounds__(kCT) void select_compact_batch(const Batch<BSel> b)
{
    const BSel& a = b.t[blockIdx.y];
    if (a.n <= 0) return;
    if (a.mode == 0) compact_block(a.n, [&](int i) { return a.count[i] > a.thr_count; }, a.sel, a.n_out);
    else if (a.mode == 2) compact_block(a.n, [&](int i) { return a.keep[i] != 0; }, a.sel, a.n_out);
    else compact_block(a.n, [&](int i) { const double m = a.mean[i]; return m > 0.0 && m < a.thr_mean; }, a.sel, a.n_out);
}
__global__ void select_rows_batch(const Batch<BSel> b)
{
    const BSel& a = b.t[blockIdx.y];
    if ((int)blockIdx.x >= a.gx) return;
    select_rows_kernel_body(a.pts, a.sel, a.n_out, a.out, blockIdx.x, a.gx);
}
__global__ __launch_bounds__(kT) void moments1_batch(const Batch<BMom> b)
{
    const BMom& a = b.t[blockIdx.y];
    if ((int)blockIdx.x >= a.gx) return;
    moments_kernel_body(a.pts, a.n, a.part, blockIdx.x, a.gx);
}
__global__ void moments2_batch(const Batch<BMom> b)
{
    const BMom& a = b.t[blockIdx.y];
    if (a.gx <= 0) return;
    reduce_stage2_body(a.part, a.gx, 9, a.out, nullptr, 0, 1);
}
__global__ void mahalanobis_batch(const Batch<BMaha> b)
{
    const BMaha& a = b.t[blockIdx.y];
    if ((int)blockIdx.x >= a.gx) return;
    mahalanobis_kernel_body(a.pts, a.n, a.mc, a.out, blockIdx.x, a.gx);
}
__global__ void transform_batch(const Batch<BXform> b)
{
    const BXform& a = b.t[blockIdx.y];
    if ((int)blockIdx.x >= a.gx) return;
    transform_kernel_body(a.pts, a.n, a.T, a.normals, blockIdx.x, a.gx);
}
// out = [a rows | b rows] (torch.cat of two clouds) or a plain copy (nb = 0)
__global__ void concat_batch(const Batch<BCopy> b)
{
    const BCopy& a = b.t[blockIdx.y];
    if ((int)blockIdx.x >= a.gx) return;
    const long tot = 3L * (a.na + a.nb), na3 = 3L * a.na;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long)a.gx * blockDim.x) a.out[i] = i < na3 ? a.a[i] : a.b[i - na3];
}
__global__ __launch_bounds__(kT) void icp_move_nn1_batch(const Batch<BIcp> b, double r2, int apply)
{
    const BIcp& a = b.t[blockIdx.y];
    if ((int)blockIdx.x >= a.gx_nn) return;
    icp_move_nn1_kernel_body(a.g, a.src, a.ns, r2, a.corr, a.d2, a.st, apply, blockIdx.x, a.gx_nn);
}
__global__ __launch_bounds__(kT) void icp_sums_batch(const Batch<BIcp> b, int kind)
{
    const BIcp& a = b.t[blockIdx.y];
    if ((int)blockIdx.x >= a.gx_sum) return;
    if (kind == 0) p2p_sums_kernel_body(a.src, a.tgt, a.corr, a.d2, a.ns, a.part, a.st, blockIdx.x, a.gx_sum);
    else p2plane_sums_kernel_body(a.src, a.tgt, a.tn, a.corr, a.d2, a.ns, a.part, a.st, blockIdx.x, a.gx_sum);
}
__global__ __launch_bounds__(64) void icp_reduce_step_batch(const Batch<BIcp> b, int kind, double rel_fitness, double rel_rmse, int max_iteration)
{
    const BIcp& a = b.t[blockIdx.y];
    if (a.gx_sum <= 0) return;
    icp_reduce_step_kernel_body(a.part, a.gx_sum, kind == 0 ? 17 : 29, a.sums, kind, a.st, a.ns, rel_fitness, rel_rmse, max_iteration, 0, 1);
}

// ---- the sort of one cloud's (cell key, original index) pairs: stable sort by key = sort by (key, index) --------------------------------
// A library radix sort walks all 64 key bits in 8 passes of several launches (hipCUB's segmented form: 405 us per call for eight 5 k-point
// segments -- it was half the label path's GPU time); the keys are cell coordinates of a compact cloud, so here (cx, cy, cz) is re-coded
// as the lexicographic rank (cx * ny + cy) * nz + cz (ny, nz: 1 + the cloud's largest cell coordinate; same order as the packed
// 3 x 21-bit key), packed with the ceil(log2 n)-bit index into ONE 64-bit word, and runs of 16 k words (128 KB of LDS, one workgroup per
// run) are sorted by a stable LSD radix sort over the rank's bits only (radix_lds: 4 bits per pass, 5 passes for a 10^6-cell grid; a
// first version used a bitonic network in LDS -- ~78 trips of every word through LDS, LDS-bandwidth bound at 88 us per call against 43
// now); a cloud of several runs (a raw 640x480 surface: 20..300 k points = 2..19 runs) is then merged by rank (seg_merge_batch).  Only a
// cloud beyond kSortRuns runs (2^20 points) or whose rank and index do not fit 64 bits together (a grid of more than 2^(64 - index bits)
// cells: stray points far from the object) takes the general form, a bitonic network over the global key / index arrays with the
// (key, index) compare.
struct BSort { const u64* k_in; const unsigned* i_in; u64* k_out; unsigned* i_out; u64* k_scratch; unsigned* i_scratch; int* flag; const double* bounds6; double h, shift; int n, gx; };
constexpr int kSortLdsMax = 16384, kSortRuns = 64;                         // runs of 16 k words (128 KB of LDS), up to 64 x 16 k = 2^20 points
__host__ __device__ __forceinline__ int sort_idx_bits(int n) { int b = 1; while ((1L << b) < (long)n) ++b; return b; }

// positions >= n are +infinity and never touched; all block sizes are powers of two, so the pair -> (i, l) maps are shifts, and i grows
// with p, so a thread stops at its first pair past the data
template <class CE>
__device__ __forceinline__ void bitonic_flip(int n, int lk, int half_pairs, CE cmpex)
{
    const int k = 1 << lk, hk = k >> 1;
    for (int p = threadIdx.x; p < half_pairs; p += blockDim.x) {              // i <-> its mirror inside the block of k
        const int base = (p >> (lk - 1)) << lk, off = p & (hk - 1);
        const int i = base | off, l = base | (k - 1 - off);
        if (i >= n) break;
        if (l < n) cmpex(i, l);
    }
}
template <class CE>
__device__ __forceinline__ void bitonic_disperse(int n, int lj, int half_pairs, CE cmpex)
{
    const int j = 1 << lj;
    for (int p = threadIdx.x; p < half_pairs; p += blockDim.x) {              // i <-> i + j
        const int i = ((p >> lj) << (lj + 1)) | (p & (j - 1)), l = i | j;
        if (i >= n) break;
        if (l < n) cmpex(i, l);
    }
}
template <class CE>
__device__ __forceinline__ void bitonic_network(int n, CE cmpex)
{
    int lpad = 0;
    while ((1 << lpad) < n) ++lpad;
    const int half_pairs = (1 << lpad) >> 1;
    for (int lk = 1; lk <= lpad; ++lk) {
        bitonic_flip(n, lk, half_pairs, cmpex);
        __syncthreads();
        for (int lj = lk - 2; lj >= 0; --lj) {
            bitonic_disperse(n, lj, half_pairs, cmpex);
            __syncthreads();
        }
    }
}

// LDS position of sort word i: two pad words behind every 32, so that the lanes of a wave, each reading ITS run of consecutive words
// (radix_lds: lane stride E words), do not all start in the same banks
__device__ __forceinline__ int sw(int i) { return i + ((i >> 5) << 1); }

// STABLE least-significant-digit radix sort of the n <= 16 k words at s[sw(i)] by the bits [shift0, shift0 + bits) (the cell rank; the
// words start in index order, so stability IS the (key, index) order), 4 bits per pass, in place.  The bitonic network above moves every
// word through LDS ~78 times for 16 k words (LDS-bandwidth bound: 88 us); this moves it twice per pass and a 10^6-cell grid is 5
// passes.  Thread t owns the E = ceil(n / 1024) consecutive words [t E, (t + 1) E): it counts their digits into sixteen 16-bit fields
// packed in four 64-bit registers (fields never exceed n <= 16384, so packed adds do not carry), an inclusive scan of the packed words over
// the lanes (shuffles) and the waves gives each thread its first slot per digit -- digit-major, thread-minor, word order: stable --
// and after a barrier (every thread holds its words in registers) the words are written to their slots.
__device__ __forceinline__ void radix_lds(u64* s, int n, int shift0, int bits)
{
    __shared__ u64 wave_tot[kCT / 64][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int E = (n + kCT - 1) / kCT;                       // <= 16
    const int base = tid * E;
    for (int shift = shift0; shift < shift0 + bits; shift += 4) {
        u64 v[16];
        u64 cnt[4] = {0, 0, 0, 0};
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            if (e < E && base + e < n) {
                v[e] = s[sw(base + e)];
                const int d = (int)(v[e] >> shift) & 15;
                const u64 inc = 1ULL << ((d & 3) * 16);
#pragma unroll
                for (int q = 0; q < 4; ++q) cnt[q] += (d >> 2) == q ? inc : 0ULL;
            } else {
                v[e] = 0;
            }
        }
        u64 inc4[4] = {cnt[0], cnt[1], cnt[2], cnt[3]};
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u64 t = (u64)__shfl_up((unsigned long long)inc4[q], o, 64);
                if (lane >= o) inc4[q] += t;
            }
        }
        if (lane == 63) {
#pragma unroll
            for (int q = 0; q < 4; ++q) wave_tot[wave][q] = inc4[q];
        }
        __syncthreads();                                     // wave totals visible; every word of the pass is in a register
        u64 pre[4] = {0, 0, 0, 0}, tot[4] = {0, 0, 0, 0};
        for (int w = 0; w < kCT / 64; ++w) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u64 t = wave_tot[w][q];
                tot[q] += t;
                pre[q] += w < wave ? t : 0ULL;
            }
        }
        u64 start[4] = {0, 0, 0, 0};
        unsigned run = 0;
#pragma unroll
        for (int d = 0; d < 16; ++d) {
            const int q = d >> 2, sh = (d & 3) * 16;
            const unsigned total = (unsigned)(tot[q] >> sh) & 0xFFFFu;
            const unsigned mine = run + ((unsigned)(pre[q] >> sh) & 0xFFFFu) + (((unsigned)(inc4[q] >> sh) & 0xFFFFu) - ((unsigned)(cnt[q] >> sh) & 0xFFFFu));
            start[q] |= (u64)mine << sh;
            run += total;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            if (e < E && base + e < n) {
                const int d = (int)(v[e] >> shift) & 15;
                const int q = d >> 2, sh = (d & 3) * 16;
                const u64 word = q == 0 ? start[0] : q == 1 ? start[1] : q == 2 ? start[2] : start[3];
                const int pos = (int)((word >> sh) & 0xFFFFu);
                const u64 inc = 1ULL << sh;
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) start[qq] += q == qq ? inc : 0ULL;
                s[sw(pos)] = v[e];
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(kCT) void seg_sort_batch(const Batch<BSort> b)
{
    extern __shared__ __attribute__((aligned(16))) u64 sort_lds[];
    const BSort& a = b.t[blockIdx.y];
    const int n = a.n, run = blockIdx.x;
    const int nruns = (n + kSortLdsMax - 1) / kSortLdsMax;
    if (n <= 0 || (run > 0 && run >= nruns)) return;
    // cells per axis: the cell of the cloud's upper bound (the expression of keys_kernel_body on the bound itself: (p - origin) / h is monotone in
    // p and the bound IS a point's coordinate, so this is the largest cell coordinate) + 1; walking the keys for their maxima cost a
    // global round trip and 3 k LDS atomics per call
    u64 dim[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        long v = (long)floor((a.bounds6[3 + d] - (a.bounds6[d] - a.shift)) / a.h);
        v = v < 0 ? 0 : (v > 2097151 ? 2097151 : v);
        dim[d] = (u64)v + 1;
    }
    const int ib = sort_idx_bits(n);
    const bool compact = nruns <= kSortRuns && (double)dim[0] * (double)dim[1] * (double)dim[2] <= (double)(1ULL << (63 - ib));
    if (run == 0 && threadIdx.x == 0) *a.flag = compact ? 1 : 0;
    if (compact) {
        const int r0 = run * kSortLdsMax, len = n - r0 < kSortLdsMax ? n - r0 : kSortLdsMax;
        for (int i = threadIdx.x; i < len; i += blockDim.x) {
            const u64 key = a.k_in[r0 + i];
            const u64 cx = key >> 42, cy = (key >> 21) & 2097151ULL, cz = key & 2097151ULL;
            sort_lds[sw(i)] = ((((cx * dim[1]) + cy) * dim[2] + cz) << ib) | (u64)(r0 + i);       // i_in[i] == i (keys_batch)
        }
        __syncthreads();
        {
            const u64 cells = dim[0] * dim[1] * dim[2];                       // exact: <= 2^(63 - ib)
            const int bits = cells > 1 ? 64 - __builtin_clzll(cells - 1) : 0;
            radix_lds(sort_lds, len, ib, bits);
        }
        if (nruns == 1) {
            for (int i = threadIdx.x; i < len; i += blockDim.x) {
                const u64 word = sort_lds[sw(i)];
                a.i_out[i] = (unsigned)(word & ((1ULL << ib) - 1));
                const u64 code = word >> ib, cxy = code / dim[2];      // the key back from its rank (no dependent gather of k_in)
                a.k_out[i] = pack_key((long)(cxy / dim[1]), (long)(cxy % dim[1]), (long)(code % dim[2]));
            }
        } else {
            for (int i = threadIdx.x; i < len; i += blockDim.x) a.k_scratch[r0 + i] = sort_lds[sw(i)];     // a sorted run; seg_merge_batch places it
        }
        return;
    }
    if (run != 0) return;
    // general form: the same network over global scratch copies, (key, index) compare; workgroup-scope fences order the exchanges
    for (int i = threadIdx.x; i < n; i += blockDim.x) { a.k_scratch[i] = a.k_in[i]; a.i_scratch[i] = a.i_in[i]; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    bitonic_network(n, [&](int i, int l) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const u64 kx = a.k_scratch[i], ky = a.k_scratch[l];
        const unsigned ix = a.i_scratch[i], iy = a.i_scratch[l];
        if (kx > ky || (kx == ky && ix > iy)) { a.k_scratch[i] = ky; a.k_scratch[l] = kx; a.i_scratch[i] = iy; a.i_scratch[l] = ix; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    });
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    for (int i = threadIdx.x; i < n; i += blockDim.x) { a.k_out[i] = a.k_scratch[i]; a.i_out[i] = a.i_scratch[i]; }
}

// clouds of 2..kSortRuns sorted runs: a word's place = its place in its run + the number of smaller words in every other run (the words
// are distinct: the index is part of them)
__global__ void seg_merge_batch(const Batch<BSort> b)
{
    const BSort& a = b.t[blockIdx.y];
    const int n = a.n;
    const int nruns = (n + kSortLdsMax - 1) / kSortLdsMax;
    if (nruns < 2 || (int)blockIdx.x >= a.gx || !*a.flag) return;
    const int ib = sort_idx_bits(n);
    for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < n; g += a.gx * blockDim.x) {
        const u64 w = a.k_scratch[g];
        const int r = g / kSortLdsMax;
        int pos = g - r * kSortLdsMax;
        for (int r2 = 0; r2 < nruns; ++r2) {
            if (r2 == r) continue;
            const u64* run = a.k_scratch + (size_t)r2 * kSortLdsMax;
            int lo = 0, hi = n - r2 * kSortLdsMax < kSortLdsMax ? n - r2 * kSortLdsMax : kSortLdsMax;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (run[mid] < w) lo = mid + 1; else hi = mid; }
            pos += lo;
        }
        const unsigned src = (unsigned)(w & ((1ULL << ib) - 1));
        a.i_out[pos] = src;
        a.k_out[pos] = a.k_in[src];
    }
}

template <class A, class K, class... X>
void launch_batch(K kern, const Batch<A>& b, int nb, int max_gx, int threads, hipStream_t st, X... extra)
{
    hipLaunchKernelGGL(kern, dim3(max_gx < 1 ? 1 : max_gx, nb), dim3(threads), 0, st, b, extra...);
}

// bounds -> keys -> one single-workgroup sort per cloud (ONE launch).  k0 / i0: concatenated [sum n] scratch, chain c at off[c]; the sorted
// keys / order go straight to k_out[c] / i_out[c] (per-cloud buffers); ks / is: scratch of the sort's general form.
int keys_and_sort(int nb, const double* const* pts, const int* n, const int* off, double h, double shift, double* const* origin, u64* k0, unsigned* i0,
                  u64* const* k_out, unsigned* const* i_out, u64* ks, unsigned* is, double* part, hipStream_t st)
{
    Batch<BPts> bp{};
    Batch<BKeys> bk{};
    Batch<BSort> bs{};
    int mg1 = 1, mg2 = 1, nmax = 1;
    for (int c = 0; c < nb; ++c) {
        const int g = n[c] > 0 ? grid_for(n[c], 1024) : 0;
        bp.t[c] = BPts{pts[c], part + (size_t)c * (1024 * 6 + 8), part + (size_t)c * (1024 * 6 + 8) + 1024 * 6, n[c], g};
        bk.t[c] = BKeys{pts[c], bp.t[c].out6, k0 + off[c], i0 + off[c], origin ? origin[c] : nullptr, n[c], n[c] > 0 ? grid_for(n[c]) : 0, h, shift};
        bs.t[c] = BSort{k0 + off[c], i0 + off[c], k_out[c], i_out[c], ks + off[c], is + off[c], reinterpret_cast<int*>(bp.t[c].out6 + 7), bp.t[c].out6, h, shift, n[c], bk.t[c].gx};
        mg1 = g > mg1 ? g : mg1;
        mg2 = bk.t[c].gx > mg2 ? bk.t[c].gx : mg2;
        nmax = n[c] > nmax ? n[c] : nmax;
    }
    launch_batch(bounds1_batch, bp, nb, mg1, kT, st);
    launch_batch(bounds2_batch, bp, nb, 1, 64, st);
    launch_batch(keys_batch, bk, nb, mg2, kT, st);
    static std::atomic<unsigned long long> attr_devs{0};                   // the dynamic-LDS attribute is per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return APE_ELAUNCH;
    if (!(attr_devs.load(std::memory_order_relaxed) >> dev & 1ULL)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(seg_sort_batch), hipFuncAttributeMaxDynamicSharedMemorySize, (kSortLdsMax + kSortLdsMax / 16) * 8) != hipSuccess) return APE_ELAUNCH;
        attr_devs.fetch_or(1ULL << dev, std::memory_order_relaxed);
    }
    const size_t lds_words = (size_t)(((nmax < kSortLdsMax ? nmax : kSortLdsMax) + 31) & ~31);
    const size_t lds = (lds_words + lds_words / 16) * 8;
    const int runs = (nmax + kSortLdsMax - 1) / kSortLdsMax;
    hipLaunchKernelGGL(seg_sort_batch, dim3(runs < 1 ? 1 : (runs > kSortRuns ? kSortRuns : runs), nb), dim3(kCT), lds, st, bs);
    if (runs > 1) launch_batch(seg_merge_batch, bs, nb, mg2, kT, st);
    return APE_OK;
}

}  // namespace

/* workspace for the batched entry points over `nb` clouds with `n_total` points in all (surface: n_total = nb * H * W pixels) */
extern "C" size_t ape_pc_batch_workspace_bytes(int nb, long n_total)
{
    if (nb < 1) nb = 1;
    if (n_total < 1) n_total = 1;
    const int nt = (int)(n_total > 0x7fffffffL ? 0x7fffffff : n_total);
    return 3 * align_up((size_t)nt * 8) + 5 * align_up((size_t)nt * 4) + align_up((size_t)nb * (1024 * 6 + 8) * 8) + align_up((size_t)nb * 512 * 29 * 8) + 8192;
}

#define APE_BATCH_CHECK(nb) if ((nb) < 1 || (nb) > kMaxBatch) return APE_EINVAL

/* ape_surface_points_f64 for nb views at once: label[c] / depth[c] [H][W], T16_host [nb][16], points[c] capacity H*W rows,
 * n_out [nb] on the device; pix_ws: nb * H * W ints */
extern "C" int ape_surface_points_batch_f64(int nb, const uint8_t* const* label, const uint16_t* const* depth, int H, int W, const double* intr4_host,
                                            const double* T16_host, double* const* points, int* n_out, int* pix_ws, void* stream)
{
    APE_BATCH_CHECK(nb);
    if (!label || !depth || !intr4_host || !T16_host || !points || !n_out || !pix_ws || H < 1 || W < 1) return APE_EINVAL;
    const int n = H * W;
    Batch<BSurf> b{};
    for (int c = 0; c < nb; ++c)
        b.t[c] = BSurf{label[c], depth[c], pix_ws + (size_t)c * n, n_out + c, points[c], load_mat(T16_host + c * 16), n, W, grid_for(n),
                       intr4_host[c * 4], intr4_host[c * 4 + 1], intr4_host[c * 4 + 2], intr4_host[c * 4 + 3]};
    hipStream_t st = (hipStream_t)stream;
    launch_batch(surface_compact_batch, b, nb, 1, kCT, st);
    launch_batch(surface_points_batch, b, nb, grid_for(n), kT, st);
    return ape::check_launch("ape_surface_points_batch_f64");
}

/* ape_voxel_down_sample_f64 for nb clouds: out[c] capacity n[c] rows, n_out [nb] on the device; n / out pointers are host arrays */
extern "C" int ape_voxel_down_sample_batch_f64(int nb, const double* const* pts, const int* n, double voxel, double* const* out, int* n_out, void* ws,
                                               size_t ws_bytes, void* stream)
{
    APE_BATCH_CHECK(nb);
    if (!pts || !n || !out || !n_out || !ws || !(voxel > 0)) return APE_EINVAL;
    int off[kMaxBatch + 1];
    off[0] = 0;
    for (int c = 0; c < nb; ++c) { if (n[c] < 0) return APE_EINVAL; off[c + 1] = off[c] + n[c]; }
    const int nt = off[nb];
    if (nt == 0) return APE_OK;
    if (ws_bytes < ape_pc_batch_workspace_bytes(nb, nt)) return APE_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    Carver cv{(char*)ws, (char*)ws + ws_bytes};
    u64* k0 = cv.take<u64>(nt); u64* k1 = cv.take<u64>(nt); u64* ks = cv.take<u64>(nt);
    unsigned* i0 = cv.take<unsigned>(nt); unsigned* i1 = cv.take<unsigned>(nt); unsigned* is = cv.take<unsigned>(nt);
    int* head = cv.take<int>(nt); int* scan = cv.take<int>(nt);
    double* part = cv.take<double>((size_t)nb * (1024 * 6 + 8));
    if (!part) return APE_EWORKSPACE;
    u64* ko[kMaxBatch]; unsigned* io[kMaxBatch];
    for (int c = 0; c < nb; ++c) { ko[c] = k1 + off[c]; io[c] = i1 + off[c]; }
    int rc = keys_and_sort(nb, pts, n, off, voxel, voxel * 0.5, nullptr, k0, i0, ko, io, ks, is, part, st);
    if (rc != APE_OK) return rc;
    Batch<BVox> bv{};
    int mg = 1;
    for (int c = 0; c < nb; ++c) {
        bv.t[c] = BVox{pts[c], k1 + off[c], i1 + off[c], head + off[c], scan + off[c], out[c], n_out + c, n[c], n[c] > 0 ? grid_for(n[c]) : 0};
        mg = bv.t[c].gx > mg ? bv.t[c].gx : mg;
    }
    launch_batch(heads_batch, bv, nb, mg, kT, st);
    launch_batch(scan_batch, bv, nb, 1, kCT, st);
    launch_batch(voxel_mean_batch, bv, nb, mg, kT, st);
    return ape::check_launch("ape_voxel_down_sample_batch_f64");
}

/* ape_grid_build_f64 for nb clouds; sorted / keys / order / origin: per-cloud device buffers (host arrays of pointers) */
extern "C" int ape_grid_build_batch_f64(int nb, const double* const* pts, const int* n, double cell, double* const* sorted,
                                        unsigned long long* const* keys, unsigned* const* order, double* const* origin3, void* ws, size_t ws_bytes,
                                        void* stream)
{
    APE_BATCH_CHECK(nb);
    if (!pts || !n || !sorted || !keys || !order || !origin3 || !ws || !(cell > 0)) return APE_EINVAL;
    int off[kMaxBatch + 1];
    off[0] = 0;
    for (int c = 0; c < nb; ++c) { if (n[c] < 0) return APE_EINVAL; off[c + 1] = off[c] + n[c]; }
    const int nt = off[nb];
    if (nt == 0) return APE_OK;
    if (ws_bytes < ape_pc_batch_workspace_bytes(nb, nt)) return APE_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    Carver cv{(char*)ws, (char*)ws + ws_bytes};
    u64* k0 = cv.take<u64>(nt); cv.take<u64>(nt); u64* ks = cv.take<u64>(nt);
    unsigned* i0 = cv.take<unsigned>(nt); cv.take<unsigned>(nt); unsigned* is = cv.take<unsigned>(nt);
    cv.take<int>(nt); cv.take<int>(nt);
    double* part = cv.take<double>((size_t)nb * (1024 * 6 + 8));
    if (!part) return APE_EWORKSPACE;
    int rc = keys_and_sort(nb, pts, n, off, cell, cell, origin3, k0, i0, (u64* const*)keys, order, ks, is, part, st);
    if (rc != APE_OK) return rc;
    Batch<BGather> bg{};
    int mg = 1;
    for (int c = 0; c < nb; ++c) {
        bg.t[c] = BGather{pts[c], order[c], sorted[c], n[c], n[c] > 0 ? grid_for(n[c]) : 0};
        mg = bg.t[c].gx > mg ? bg.t[c].gx : mg;
    }
    launch_batch(gather_batch, bg, nb, mg, kT, st);
    return ape::check_launch("ape_grid_build_batch_f64");
}

/* per-cloud grid description for the batched searches: the arguments of GRID_ARGS as arrays */
#define BGRID_ARGS const double* const* sorted, const unsigned long long* const* keys, const unsigned* const* order, const double* const* origin3, const int* gn, double cell
#define BGRID(c) Grid{sorted[c], (const u64*)keys[c], order[c], origin3[c], gn[c], cell}

/* op 0: radius count (count[c][nq]); op 1: hybrid normals (normals[c][nq][3], max_nn); op 2: k-NN mean distance of the grid's own points
 * (mean[c][gn], k) -- the queries q[c] (nq[c] rows) are ignored for op 2 */
extern "C" int ape_grid_query_batch_f64(int op, int nb, BGRID_ARGS, const double* const* q, const int* nq, double radius, int max_nn_or_k,
                                        int* const* count, double* const* normals, double* const* mean, void* stream)
{
    APE_BATCH_CHECK(nb);
    if (op < 0 || op > 2 || !sorted || !keys || !order || !origin3 || !gn || !(cell > 0)) return APE_EINVAL;
    if (op == 0 && (!q || !nq || !count || radius > cell)) return APE_EINVAL;
    if (op == 1 && (!q || !nq || !normals || radius > cell || max_nn_or_k < 1 || max_nn_or_k > kMaxNN)) return APE_EINVAL;
    if (op == 2 && (!mean || max_nn_or_k < 1 || max_nn_or_k > kMaxNN)) return APE_EINVAL;
    Batch<BGridQ> b{};
    int mg = 1;
    for (int c = 0; c < nb; ++c) {
        const int nqc = op == 2 ? gn[c] : nq[c];
        if (gn[c] < 0 || nqc < 0 || (op == 2 && gn[c] > 0 && max_nn_or_k > gn[c])) return APE_EINVAL;
        const int gx = (gn[c] > 0 && nqc > 0) ? ape::ceil_div((long)nqc * kG, (long)kT) : 0;
        b.t[c] = BGridQ{BGRID(c), op == 2 ? nullptr : q[c], nqc, gx, radius * radius, op == 0 ? count[c] : nullptr, op == 1 ? normals[c] : nullptr, max_nn_or_k,
                        op == 2 ? mean[c] : nullptr, max_nn_or_k};
        mg = gx > mg ? gx : mg;
    }
    hipStream_t st = (hipStream_t)stream;
    if (op == 0) launch_batch(radius_count_batch, b, nb, mg, kT, st);
    else if (op == 1) launch_batch(normals_batch, b, nb, mg, kT, st);
    else launch_batch(knn_mean_batch, b, nb, mg, kT, st);
    return ape::check_launch("ape_grid_query_batch_f64");
}

/* ordered row selection with the keep rule evaluated on the device.  mode 0: count[c][i] > thr_count (RemoveRadiusOutliers);
 * mode 1: mean[c][i] > 0 && mean[c][i] < thr_mean_host[c] (RemoveStatisticalOutliers).  out[c] capacity n[c] rows, sel_ws: sum n ints,
 * n_out [nb] on the device */
extern "C" int ape_select_points_batch_f64(int mode, int nb, const double* const* pts, const int* n, const int* const* count, int thr_count,
                                           const double* const* mean, const double* thr_mean_host, double* const* out, int* n_out, int* sel_ws,
                                           void* stream)
{
    APE_BATCH_CHECK(nb);
    if ((mode != 0 && mode != 1) || !pts || !n || !out || !n_out || !sel_ws) return APE_EINVAL;
    if ((mode == 0 && !count) || (mode == 1 && (!mean || !thr_mean_host))) return APE_EINVAL;
    Batch<BSel> b{};
    int mg = 1, off = 0;
    for (int c = 0; c < nb; ++c) {
        if (n[c] < 0) return APE_EINVAL;
        b.t[c] = BSel{pts[c], mode == 0 ? count[c] : nullptr, mode == 1 ? mean[c] : nullptr, sel_ws + off, n_out + c, out[c], n[c], n[c] > 0 ? grid_for(n[c]) : 0,
                      thr_count, mode == 1 ? thr_mean_host[c] : 0.0, mode, nullptr};
        off += n[c];
        mg = b.t[c].gx > mg ? b.t[c].gx : mg;
    }
    hipStream_t st = (hipStream_t)stream;
    launch_batch(select_compact_batch, b, nb, 1, kCT, st);
    launch_batch(select_rows_batch, b, nb, mg, kT, st);
    return ape::check_launch("ape_select_points_batch_f64");
}

/* ape_icp_sums_f64(kind 2) for nb clouds: out9 [nb][9] on the device; ws: nb * 512 * 9 doubles */
extern "C" int ape_moments_batch_f64(int nb, const double* const* pts, const int* n, double* out9, void* ws, size_t ws_bytes, void* stream)
{
    APE_BATCH_CHECK(nb);
    if (!pts || !n || !out9 || !ws || ws_bytes < (size_t)nb * 512 * 9 * 8) return APE_EINVAL;
    Batch<BMom> b{};
    int mg = 1;
    for (int c = 0; c < nb; ++c) {
        if (n[c] < 0) return APE_EINVAL;
        b.t[c] = BMom{pts[c], (double*)ws + (size_t)c * 512 * 9, out9 + c * 9, n[c], n[c] > 0 ? grid_for(n[c], 512) : 0};
        mg = b.t[c].gx > mg ? b.t[c].gx : mg;
    }
    hipStream_t st = (hipStream_t)stream;
    launch_batch(moments1_batch, b, nb, mg, kT, st);
    launch_batch(moments2_batch, b, nb, 1, 64, st);
    return ape::check_launch("ape_moments_batch_f64");
}

/* ape_mahalanobis_f64 for nb clouds: mc12_host [nb][12] */
extern "C" int ape_mahalanobis_batch_f64(int nb, const double* const* pts, const int* n, const double* mc12_host, double* const* out, void* stream)
{
    APE_BATCH_CHECK(nb);
    if (!pts || !n || !mc12_host || !out) return APE_EINVAL;
    // Vec12 by value is 96 B per chain: two launches of up to eight chains keep the kernel arguments under 4 KB
    hipStream_t st = (hipStream_t)stream;
    for (int c0 = 0; c0 < nb; c0 += 8) {
        Batch<BMaha> b{};
        int mg = 1, m = nb - c0 < 8 ? nb - c0 : 8;
        for (int c = 0; c < m; ++c) {
            if (n[c0 + c] < 0) return APE_EINVAL;
            b.t[c] = BMaha{pts[c0 + c], out[c0 + c], n[c0 + c], n[c0 + c] > 0 ? grid_for(n[c0 + c]) : 0, {}};
            for (int i = 0; i < 12; ++i) b.t[c].mc.v[i] = mc12_host[(c0 + c) * 12 + i];
            mg = b.t[c].gx > mg ? b.t[c].gx : mg;
        }
        launch_batch(mahalanobis_batch, b, m, mg, kT, st);
    }
    return ape::check_launch("ape_mahalanobis_batch_f64");
}

/* ape_transform_points_f64 for nb clouds in place: T16_host [nb][16]; normals[c] may be null */
extern "C" int ape_transform_points_batch_f64(int nb, double* const* pts, double* const* normals, const int* n, const double* T16_host, void* stream)
{
    APE_BATCH_CHECK(nb);
    if (!pts || !n || !T16_host) return APE_EINVAL;
    Batch<BXform> b{};
    int mg = 1;
    for (int c = 0; c < nb; ++c) {
        if (n[c] < 0) return APE_EINVAL;
        b.t[c] = BXform{pts[c], normals ? normals[c] : nullptr, load_mat(T16_host + c * 16), n[c], n[c] > 0 ? grid_for(n[c]) : 0};
        mg = b.t[c].gx > mg ? b.t[c].gx : mg;
    }
    launch_batch(transform_batch, b, nb, mg, kT, (hipStream_t)stream);
    return ape::check_launch("ape_transform_points_batch_f64");
}

/* out[c] = [a[c] (na[c] rows) | b[c] (nb_rows[c] rows)]; b may be null (plain copies) */
extern "C" int ape_concat_points_batch_f64(int nb, const double* const* a, const int* na, const double* const* b, const int* nb_rows, double* const* out,
                                           void* stream)
{
    APE_BATCH_CHECK(nb);
    if (!a || !na || !out) return APE_EINVAL;
    Batch<BCopy> bt{};
    int mg = 1;
    for (int c = 0; c < nb; ++c) {
        const int n2 = b && nb_rows ? nb_rows[c] : 0;
        if (na[c] < 0 || n2 < 0) return APE_EINVAL;
        const int tot = na[c] + n2;
        bt.t[c] = BCopy{a[c], b ? b[c] : nullptr, out[c], na[c], n2, tot > 0 ? grid_for(3L * tot) : 0};
        mg = bt.t[c].gx > mg ? bt.t[c].gx : mg;
    }
    launch_batch(concat_batch, bt, nb, mg, kT, (hipStream_t)stream);
    return ape::check_launch("ape_concat_points_batch_f64");
}

/* ape_icp_run_f64 for nb registrations of the same kind advancing together: per chain the target's grid, the moved source src[c] (ns[c]
 * rows, updated in place), target points / normals, scratch corr[c] / dist2[c] / sums[c] (29) and state[c] (40 doubles on the device, set
 * up as for ape_icp_run_f64); chains with ns[c] == 0 or gn[c] == 0 are skipped.  ws: nb * 512 * 29 doubles. */
extern "C" int ape_icp_run_batch_f64(int kind, int nb, BGRID_ARGS, double* const* src, const int* ns, const double* const* tgt,
                                     const double* const* tgt_normals, double max_dist, double rel_fitness, double rel_rmse, int max_iteration, int n_iter,
                                     int first_call, int* const* corr, double* const* dist2, double* const* sums, double* const* state, void* ws,
                                     size_t ws_bytes, void* stream)
{
    APE_BATCH_CHECK(nb);
    if (kind < 0 || kind > 1 || !sorted || !keys || !order || !origin3 || !gn || !src || !ns || !tgt || !corr || !dist2 || !sums || !state || !ws) return APE_EINVAL;
    if (kind == 1 && !tgt_normals) return APE_EINVAL;
    if (max_dist > cell || n_iter < 0 || max_iteration < 0 || ws_bytes < (size_t)nb * 512 * 29 * 8) return APE_EINVAL;
    Batch<BIcp> b{};
    int mg_nn = 1, mg_sum = 1;
    for (int c = 0; c < nb; ++c) {
        const bool on = ns[c] > 0 && gn[c] > 0;
        if (ns[c] < 0 || gn[c] < 0 || (on && kind == 1 && !tgt_normals[c])) return APE_EINVAL;
        b.t[c] = BIcp{BGRID(c), src[c], tgt[c], kind == 1 ? tgt_normals[c] : nullptr, corr[c], dist2[c], (double*)ws + (size_t)c * 512 * 29, sums[c], state[c],
                      ns[c], on ? ape::ceil_div((long)ns[c] * kG, (long)kT) : 0, on ? grid_for(ns[c], 512) : 0};
        mg_nn = b.t[c].gx_nn > mg_nn ? b.t[c].gx_nn : mg_nn;
        mg_sum = b.t[c].gx_sum > mg_sum ? b.t[c].gx_sum : mg_sum;
    }
    hipStream_t st = (hipStream_t)stream;
    auto block = [&](int apply) {
        launch_batch(icp_move_nn1_batch, b, nb, mg_nn, kT, st, max_dist * max_dist, apply);
        launch_batch(icp_sums_batch, b, nb, mg_sum, kT, st, kind);
        launch_batch(icp_reduce_step_batch, b, nb, 1, 64, st, kind, rel_fitness, rel_rmse, max_iteration);
    };
    if (first_call) block(0);
    for (int it = 0; it < n_iter; ++it) block(1);
    return ape::check_launch("ape_icp_run_batch_f64");
}


// ---------------------------------------------------------------------------------------------------------------------------------
// The one-cloud entry points: the batched ones above with one cloud (same kernels, same results as a batch slot)
extern "C" size_t ape_pc_workspace_bytes(int n)
{
    if (n < 1) n = 1;
    return ape_pc_batch_workspace_bytes(1, n) + align_up((size_t)n * 4);
}

/* label[H][W] u8, depth[H][W] u16 -> points[n][3] f64 in raster order (capacity H*W), *n_out on the device */
extern "C" int ape_surface_points_f64(const uint8_t* label, const uint16_t* depth, int H, int W, double fx, double fy, double ppx,
                                      double ppy, const double* T16_host, double* points, int* n_out, void* ws, size_t ws_bytes,
                                      void* stream)
{
    if (!label || !depth || !T16_host || !points || !n_out || !ws || H < 1 || W < 1) return APE_EINVAL;
    if (ws_bytes < (size_t)H * W * 4) return APE_EWORKSPACE;
    const double intr[4] = {fx, fy, ppx, ppy};
    return ape_surface_points_batch_f64(1, &label, &depth, H, W, intr, T16_host, &points, n_out, (int*)ws, stream);
}

/* out capacity n points; *n_out on the device */
extern "C" int ape_voxel_down_sample_f64(const double* pts, int n, double voxel, double* out, int* n_out, void* ws, size_t ws_bytes,
                                         void* stream)
{
    if (!pts || !out || !n_out || !ws || n < 1 || !(voxel > 0)) return APE_EINVAL;
    return ape_voxel_down_sample_batch_f64(1, &pts, &n, voxel, &out, n_out, ws, ws_bytes, stream);
}

/* Build the search grid of a cloud: sorted[n][3], keys[n], order[n], origin[3] (all caller-owned device buffers). */
extern "C" int ape_grid_build_f64(const double* pts, int n, double cell, double* sorted, unsigned long long* keys, unsigned* order,
                                  double* origin3, void* ws, size_t ws_bytes, void* stream)
{
    if (!pts || !sorted || !keys || !order || !origin3 || !ws || n < 1 || !(cell > 0)) return APE_EINVAL;
    return ape_grid_build_batch_f64(1, &pts, &n, cell, &sorted, &keys, &order, &origin3, ws, ws_bytes, stream);
}

/* out[i] = pts[sel[i]] for the rows with keep[i] != 0, in order (capacity n); *n_out on the device */
extern "C" int ape_select_points_f64(const double* pts, const uint8_t* keep, int n, double* out, int* sel_idx, int* n_out, void* ws,
                                     size_t ws_bytes, void* stream)
{
    (void)ws; (void)ws_bytes;
    if (!pts || !keep || !out || !sel_idx || !n_out || n < 1) return APE_EINVAL;
    Batch<BSel> b{};
    b.t[0] = BSel{pts, nullptr, nullptr, sel_idx, n_out, out, n, grid_for(n), 0, 0.0, 2, keep};
    hipStream_t st = (hipStream_t)stream;
    launch_batch(select_compact_batch, b, 1, 1, kCT, st);
    launch_batch(select_rows_batch, b, 1, b.t[0].gx, kT, st);
    return ape::check_launch("ape_select_points_f64");
}
