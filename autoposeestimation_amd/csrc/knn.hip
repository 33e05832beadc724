// k-NN for gfx950 -- replaces DenseFusion/lib/knn/src/{knn.h:12-66, cpu/knn_cpu.cpp, cuda/knn.cu}.
//
// The reference CUDA path materialises a ref_nb x query_nb distance matrix in HBM (knn.h:33, 4 GB at
// 1000 x 10^6) and then scans it column-strided (knn.cu:113-176).  Here distances never leave registers:
//
//   knn1_d3<S>  (dim = 3, k = 1: every call site in the reference)
//     - refs are staged once per workgroup through LDS as float4 (x,y,z,-) tiles: one ds_read_b128 per
//       ref per lane, a broadcast when S = 1, conflict-free consecutive 16-B slots when S = 64;
//     - each query is owned by S lanes (S = 1,4,16,64 picked on the host so that the launch has enough
//       wavefronts to fill 256 CUs even for 1000 queries); lane s scans refs s, s+S, ... in ascending
//       order with a strict '<', then the S partial minima are merged by wave shuffles with the
//       (distance, index) lexicographic rule -> exactly "lowest index wins ties" (knn_cpu.cpp:30).
//   knn1_d3_q<Q, G>  (the same, for query counts that fill the chip many times over -- the training loss's 10^6 queries, loss.py:38-47):
//     one lane owns Q queries, so one (broadcast) LDS read of a ref feeds Q pair evaluations (knn1_d3<1> reads 16 B from LDS per pair: the
//     LDS port was ~3/4 as busy as the vector units, 0.48 of the fp32 lane rate in round 1), and the refs go in groups of G = 8 whose
//     distances are reduced with v_min3_f32 before ONE compare-and-select per group: 8.9 vector operations per pair instead of 11, the
//     index inside the winning group recovered after the scan.  Same arithmetic per pair, a strict '<' between groups, the first equal
//     distance inside the group => the same indices bit for bit.  The file is built WITHOUT hipcc's SLP vectoriser (Makefile): it
//     pairs the x and z terms into v_pk_add_f32 / v_pk_mul_f32, which issue at half rate on gfx950 and cost moves and s_nops on top
//     (10.7 issue slots per pair).  10^6 queries x 1000 refs: 0.272 -> 0.18-0.19 ms (0.74-0.79 of the fp32 lane rate at 11 operations per
//     pair; 4 x 10^6: 0.95-0.99).
//   knn_general  (any dim, any k <= ref_nb): one query per lane, stable insertion into a 64-entry list, ceil(k / 64) passes.
//
// Bit-exactness: d = ((dx*dx) + (dy*dy)) + (dz*dz) with __fmul_rn/__fadd_rn (no FMA contraction), the
// same roundings as the reference's scalar x86 build (`dist = 0; dist += diff*diff` per dimension).
//
// Roofline: N_q*N_r pair evaluations at ~11 VALU lane-ops each; refs (12 B) and queries (12 B in, 8 B
// out) are read/written once -> fp32-VALU bound, not HBM bound.
#include "common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kTile = 2048;  // refs per LDS tile: 32 KB of float4

template <int S>
__global__ __launch_bounds__(kBlock) void knn1_d3(const float* __restrict__ ref, const float* __restrict__ query,
                                                  int64_t* __restrict__ idx, int ref_nb, int query_nb)
{
    __shared__ float4 tile[kTile];
    const int b = blockIdx.y;
    ref += (size_t)b * 3 * ref_nb;
    query += (size_t)b * 3 * query_nb;
    idx += (size_t)b * query_nb;

    constexpr int kQueriesPerBlock = kBlock / S;
    const int s = threadIdx.x % S;
    const int q = blockIdx.x * kQueriesPerBlock + threadIdx.x / S;
    const bool valid = q < query_nb;
    const int qc = valid ? q : query_nb - 1;
    const float qx = query[qc], qy = query[query_nb + qc], qz = query[2 * (size_t)query_nb + qc];

    float best = __builtin_inff();
    int besti = s;  // all-inf distances must still yield the lowest index (stable sort of equal keys)

    for (int t0 = 0; t0 < ref_nb; t0 += kTile) {
        const int n = min(kTile, ref_nb - t0);
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += kBlock)
            tile[i] = make_float4(ref[t0 + i], ref[ref_nb + t0 + i], ref[2 * (size_t)ref_nb + t0 + i], 0.f);
        __syncthreads();
#pragma unroll 4
        for (int r = s; r < n; r += S) {
            const float4 p = tile[r];
            const float dx = p.x - qx, dy = p.y - qy, dz = p.z - qz;
            const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
            if (d < best) { best = d; besti = t0 + r; }
        }
    }
#pragma unroll
    for (int off = S / 2; off > 0; off >>= 1) {
        const float od = __shfl_xor(best, off);
        const int oi = __shfl_xor(besti, off);
        if (od < best || (od == best && oi < besti)) { best = od; besti = oi; }
    }
    if (valid && s == 0) idx[q] = (int64_t)besti + 1;
}

// APE_ABLATIONS (make ablations -> libape_hip_abl.so, tools/mb_knn.py): the A/B forms -- four queries per lane, no group minima -- behind a
// run-time switch; the product build has the one form it launches and no switch in its tile loop
#ifdef APE_ABLATIONS
#define KNN_DBG_PARAM , int g_dbg
#define KNN_NO_GROUPS(n) ((g_dbg & 2) ? 0 : (n))
#else
#define KNN_DBG_PARAM
#define KNN_NO_GROUPS(n) (n)
#endif
template <int Q, int G>
__global__ __launch_bounds__(kBlock) void knn1_d3_q(const float* __restrict__ ref, const float* __restrict__ query,
                                                    int64_t* __restrict__ idx, int ref_nb, int query_nb KNN_DBG_PARAM)
{
    __shared__ float4 tile[kTile];
    const int b = blockIdx.y;
    ref += (size_t)b * 3 * ref_nb;
    query += (size_t)b * 3 * query_nb;
    idx += (size_t)b * query_nb;
    // query j of this lane: blockIdx.x * Q * kBlock + j * kBlock + threadIdx.x (consecutive lanes = consecutive queries: coalesced)
    const int q0 = blockIdx.x * (Q * kBlock) + threadIdx.x;
    float qx[Q], qy[Q], qz[Q], best[Q];
    int besti[Q];
#pragma unroll
    for (int j = 0; j < Q; ++j) {
        const int q = q0 + j * kBlock;
        const int qc = q < query_nb ? q : query_nb - 1;
        qx[j] = query[qc]; qy[j] = query[query_nb + qc]; qz[j] = query[2 * (size_t)query_nb + qc];
        best[j] = __builtin_inff();
        besti[j] = 0;
    }
    for (int t0 = 0; t0 < ref_nb; t0 += kTile) {
        const int n = min(kTile, ref_nb - t0);
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += kBlock)
            tile[i] = make_float4(ref[t0 + i], ref[ref_nb + t0 + i], ref[2 * (size_t)ref_nb + t0 + i], 0.f);
        __syncthreads();
        // Refs in groups of G: the G distances of a group are reduced with v_min3 / v_min (exact: a minimum does not round), and only the
        // group minimum goes through the compare-and-select against the running best -- 8 + 3/8 + 3/8 vector operations per pair instead of
        // 11.  A strict '<' on the group minimum keeps the FIRST group that holds the overall minimum; the index inside it is recovered
        // after the scan (below).  The refs behind the last full group of the last tile are groups of one.
        const int ng = KNN_NO_GROUPS(n / G * G);
#pragma unroll 1
        for (int r = 0; r < ng; r += G) {
            float d[Q][G];
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const float4 p = tile[r + i];       // every lane reads the same address: one broadcast
#pragma unroll
                for (int j = 0; j < Q; ++j) {
                    const float dx = p.x - qx[j], dy = p.y - qy[j], dz = p.z - qz[j];
                    d[j][i] = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
                }
            }
#pragma unroll
            for (int j = 0; j < Q; ++j) {
                float m = d[j][0];
#pragma unroll
                for (int i = 1; i + 1 < G; i += 2) m = __builtin_fminf(__builtin_fminf(m, d[j][i]), d[j][i + 1]);
                if (G % 2 == 0) m = __builtin_fminf(m, d[j][G - 1]);
                if (m < best[j]) { best[j] = m; besti[j] = t0 + r; }
            }
        }
        for (int r = ng; r < n; ++r) {
            const float4 p = tile[r];
#pragma unroll
            for (int j = 0; j < Q; ++j) {
                const float dx = p.x - qx[j], dy = p.y - qy[j], dz = p.z - qz[j];
                const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
                if (d < best[j]) { best[j] = d; besti[j] = t0 + r; }
            }
        }
    }
    // besti = first ref of the first group that holds the minimum: the neighbour is the first ref of refs besti .. besti + G - 1 at exactly
    // that distance (a group of one gives itself; all-inf / NaN distances keep besti, the reference's "lowest index").
#pragma unroll
    for (int j = 0; j < Q; ++j) {
        const int q = q0 + j * kBlock;
        if (q >= query_nb) continue;
        int found = besti[j];
        float dd[G];
#pragma unroll
        for (int i = 0; i < G; ++i) {          // (all 3 G loads of a query in flight together: clamped addresses, no branches)
            const int r = besti[j] + i < ref_nb ? besti[j] + i : ref_nb - 1;
            const float dx = ref[r] - qx[j], dy = ref[ref_nb + r] - qy[j], dz = ref[2 * (size_t)ref_nb + r] - qz[j];
            dd[i] = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        }
#pragma unroll
        for (int i = G - 1; i >= 0; --i)
            if (besti[j] + i < ref_nb && dd[i] == best[j]) found = besti[j] + i;
        idx[q] = (int64_t)found + 1;
    }
}

constexpr int kMaxK = 64;

__global__ __launch_bounds__(kBlock) void knn_general(const float* __restrict__ ref, const float* __restrict__ query,
                                                      int64_t* __restrict__ idx, int dim, int ref_nb, int query_nb, int k)
{
    const int b = blockIdx.y;
    ref += (size_t)b * dim * ref_nb;
    query += (size_t)b * dim * query_nb;
    idx += (size_t)b * k * query_nb;
    const int q = blockIdx.x * kBlock + threadIdx.x;
    if (q >= query_nb) return;

    // k <= 64 neighbours are kept in one sorted register list.  Larger k (the reference takes any k <= ref_nb, knn_cpu.cpp:18-44) runs
    // ceil(k / 64) passes over the refs: pass p keeps the 64 smallest (distance, index) pairs that come AFTER the last pair the
    // previous pass emitted -- the same ascending (distance, then ref index) order as the reference's stable bubble sort.
    float bd[kMaxK];
    int bi[kMaxK];
    float thr_d = -1.f;           // squared distances are >= 0: nothing is excluded in the first pass
    int thr_r = -1;
    for (int done = 0; done < k;) {
        const int want = k - done < kMaxK ? k - done : kMaxK;
        int filled = 0;
        for (int r = 0; r < ref_nb; ++r) {
            float d = 0.f;
            for (int h = 0; h < dim; ++h) {
                const float diff = ref[(size_t)h * ref_nb + r] - query[(size_t)h * query_nb + q];
                d = __fadd_rn(d, __fmul_rn(diff, diff));
            }
            if (!(d > thr_d || (d == thr_d && r > thr_r))) continue;       // emitted by an earlier pass
            // stable insertion: new entry goes after every entry with distance <= d
            if (filled < want) {
                int j = filled++;
                while (j > 0 && bd[j - 1] > d) { bd[j] = bd[j - 1]; bi[j] = bi[j - 1]; --j; }
                bd[j] = d; bi[j] = r;
            } else if (bd[want - 1] > d) {
                int j = want - 1;
                while (j > 0 && bd[j - 1] > d) { bd[j] = bd[j - 1]; bi[j] = bi[j - 1]; --j; }
                bd[j] = d; bi[j] = r;
            }
        }
        for (int i = 0; i < want; ++i) idx[(size_t)(done + i) * query_nb + q] = (int64_t)bi[i] + 1;
        thr_d = bd[want - 1];
        thr_r = bi[want - 1];
        done += want;
    }
}

#ifdef APE_ABLATIONS
int g_knn_q = 0;        // ape_knn_debug bits: 1 = never the several-queries-per-lane form, 2 = that form without the group minima, 4 / 8 = always four / two queries per lane (A/B)
#endif

template <int S>
void launch_knn1(const float* ref, const float* query, int64_t* idx, int batch, int ref_nb, int query_nb, hipStream_t st)
{
    dim3 grid(ape::ceil_div(query_nb, kBlock / S), batch);
    hipLaunchKernelGGL(knn1_d3<S>, grid, dim3(kBlock), 0, st, ref, query, idx, ref_nb, query_nb);
}

}  // namespace

#ifdef APE_ABLATIONS
extern "C" int ape_knn_debug(int bits) { g_knn_q = bits; return APE_OK; }
#endif

extern "C" int ape_knn_f32(const float* ref, const float* query, int64_t* idx,
                           int batch, int dim, int ref_nb, int query_nb, int k, void* stream)
{
    if (batch < 0 || dim < 1 || ref_nb < 1 || query_nb < 0 || k < 1 || k > ref_nb) return APE_EINVAL;
    if (batch == 0 || query_nb == 0) return APE_OK;
    if (!ref || !query || !idx) return APE_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dim == 3 && k == 1) {
        // lanes per query: aim for >= 8 waves per CU worth of work (256 CUs x 8 x 64 lanes)
        const long target = 256L * 8 * 64;
        const long total = (long)batch * query_nb;
        int s = 1;
        while (s < 64 && total * s < target && s * 4 <= ref_nb) s *= 4;
        // enough queries to give every lane several of them and still fill the chip: the knn1_d3_q forms
        // two queries per lane: faster than four at every size measured (10^6 queries: 0.177 vs 0.195 ms, 4 x 10^6: 0.568 vs 0.585 --
        // twice the waves hide the vector pipe's own latencies and the LDS reads' better than the halved LDS traffic pays)
#ifdef APE_ABLATIONS
        if (s == 1 && !(g_knn_q & 1) && total >= 4 * target && query_nb >= 4 * kBlock) {
            const bool four = (g_knn_q & 4) != 0;
            if (four && !(g_knn_q & 8)) {
                dim3 grid(ape::ceil_div(query_nb, 4 * kBlock), batch);
                hipLaunchKernelGGL((knn1_d3_q<4, 8>), grid, dim3(kBlock), 0, st, ref, query, idx, ref_nb, query_nb, g_knn_q);
            } else {
                dim3 grid(ape::ceil_div(query_nb, 2 * kBlock), batch);
                hipLaunchKernelGGL((knn1_d3_q<2, 8>), grid, dim3(kBlock), 0, st, ref, query, idx, ref_nb, query_nb, g_knn_q);
            }
            return ape::check_launch("ape_knn_f32");
        }
#else
        if (s == 1 && total >= 4 * target && query_nb >= 4 * kBlock) {
            dim3 grid(ape::ceil_div(query_nb, 2 * kBlock), batch);
            hipLaunchKernelGGL((knn1_d3_q<2, 8>), grid, dim3(kBlock), 0, st, ref, query, idx, ref_nb, query_nb);
            return ape::check_launch("ape_knn_f32");
        }
#endif
        switch (s) {
            case 1: launch_knn1<1>(ref, query, idx, batch, ref_nb, query_nb, st); break;
            case 4: launch_knn1<4>(ref, query, idx, batch, ref_nb, query_nb, st); break;
            case 16: launch_knn1<16>(ref, query, idx, batch, ref_nb, query_nb, st); break;
            default: launch_knn1<64>(ref, query, idx, batch, ref_nb, query_nb, st); break;
        }
    } else {
        dim3 grid(ape::ceil_div(query_nb, kBlock), batch);
        hipLaunchKernelGGL(knn_general, grid, dim3(kBlock), 0, st, ref, query, idx, dim, ref_nb, query_nb, k);
    }
    return ape::check_launch("ape_knn_f32");
}
