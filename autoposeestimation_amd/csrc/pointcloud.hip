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

__global__ __launch_bounds__(kT) void radius_count_group_kernel(Grid g, const double* __restrict__ q, int nq, double r2, int* __restrict__ count)
{
    const int lane = threadIdx.x % kG;
    const int i = (blockIdx.x * kT + threadIdx.x) / kG;
    int c = 0;
    if (i < nq) for_my_cell(g, q + (size_t)i * 3, lane, [&](int, double d2) { c += d2 < r2 ? 1 : 0; });
    for (int m = kG / 2; m >= 1; m >>= 1) c += __shfl_xor(c, m, kG);
    if (i < nq && lane == 0) count[i] = c;
}

__global__ __launch_bounds__(kT) void nn1_group_kernel(Grid g, const double* __restrict__ q, int nq, double r2, int* __restrict__ idx,
                                                       double* __restrict__ dist2, const double* __restrict__ skip)
{
    if (skip && skip[0] != 0.0) return;
    const int lane = threadIdx.x % kG;
    const int i = (blockIdx.x * kT + threadIdx.x) / kG;
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

__global__ __launch_bounds__(kT) void normals_kernel(Grid g, const double* __restrict__ q, int nq, double r2, int max_nn, double* __restrict__ normals)
{
    __shared__ double cand_d[kT / kG][kNrmCand];
    __shared__ int cand_j[kT / kG][kNrmCand];
    __shared__ int sel[kT / kG][kMaxNN];
    __shared__ int ncand[kT / kG];
    const int lane = threadIdx.x % kG, grp = threadIdx.x / kG;
    const int i = (blockIdx.x * kT + threadIdx.x) / kG;
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

__global__ __launch_bounds__(kT) void knn_mean_grid_kernel(Grid g, int k, double* __restrict__ mean)
{
    __shared__ double cand[kT / kG][kKnnCand];
    __shared__ int ncand[kT / kG];
    const int lane = threadIdx.x % kG, grp = threadIdx.x / kG;
    const int i = (blockIdx.x * kT + threadIdx.x) / kG;
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

__global__ void reduce_stage2(const double* __restrict__ part, int g, int nv, double* __restrict__ out, const double* __restrict__ skip = nullptr)
{
    if (skip && skip[0] != 0.0) return;
    const int v = threadIdx.x;
    if (v < nv) { double s = 0; for (int b = 0; b < g; ++b) s += part[b * nv + v]; out[v] = s; }
}

// out: [0] count, [1] sum d^2, [2..4] sum s, [5..7] sum t, [8..16] sum s_a t_b (row a, col b)
__global__ __launch_bounds__(kT) void p2p_sums_kernel(const double* __restrict__ src, const double* __restrict__ tgt, const int* __restrict__ corr,
                                                      const double* __restrict__ d2, int n, double* __restrict__ part,
                                                      const double* __restrict__ skip = nullptr)
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
    }, part);
}

// out: [0] count, [1] sum d^2, [2..22] upper triangle of J^T J (row major), [23..28] J^T r
__global__ __launch_bounds__(kT) void p2plane_sums_kernel(const double* __restrict__ src, const double* __restrict__ tgt,
                                                          const double* __restrict__ tn, const int* __restrict__ corr,
                                                          const double* __restrict__ d2, int n, double* __restrict__ part,
                                                          const double* __restrict__ skip = nullptr)
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
__global__ __launch_bounds__(64) void icp_reduce_step_kernel(const double* __restrict__ part, int g, int nv, double* __restrict__ sums, int kind,
                                                             double* __restrict__ st, int ns, double rel_fitness, double rel_rmse, int max_iteration)
{
    if (st[0] != 0.0) return;
    __shared__ double tot[32];
    const int v = threadIdx.x;
    if (v < nv) { double a = 0; for (int b = 0; b < g; ++b) a += part[b * nv + v]; tot[v] = a; sums[v] = a; }
    __syncthreads();
    if (v == 0) icp_step(kind, tot, st, ns, rel_fitness, rel_rmse, max_iteration);
}

// correspondence search of the ICP loop with the pending update applied on the way: the query is moved by state[21..36] (the update the
// previous step computed), written back, and searched -- kG lanes per query as in nn1_group_kernel (every lane moves its copy)
__global__ __launch_bounds__(kT) void icp_move_nn1_kernel(Grid g, double* __restrict__ src, int nq, double r2, int* __restrict__ idx,
                                                          double* __restrict__ dist2, const double* __restrict__ st, int apply)
{
    if (st[0] != 0.0) return;
    const int lane = threadIdx.x % kG;
    const int i = (blockIdx.x * kT + threadIdx.x) / kG;
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
