// Segmentation post-processing and crop/point selection, device resident (HBM-bound integer/byte work).
//
// Reference (per frame, on the host with numpy + cv2): pipeline/utils.py:430-469 and its twin
// label_generator/create_labels.py:127-147:
//     softmax -> argmax -> np.unique counts -> per class with >100 px: 8-connectivity components ->
//     component with the highest mean class probability (strict '>', first wins) -> mask {0,255};
// then pipeline/utils.py:524-561: get_bbox (myDatasetAugmented/dataset.py:342-380), candidate pixels =
// mask & depth != 0 inside the crop in raster order, N of them (random subset / wrap pad), back-projection
// in float32, crop normalisation.
//
// Here the [C,480,640] probability tensor never leaves the GPU (the reference copies ~16 MB per frame to the host,
// pipeline/utils.py:431-432).  One pass per stage over byte/int images:
//   seg_argmax     logits -> label u8 + softmax(softmax(logits))[argmax] f32          (reads C*4 B, writes 5 B / px)
//   ccl8_*         union-find labelling of ALL classes at once: pixels are linked iff they carry the same non-zero
//                  class; root = smallest raster index of the component (raster-order numbering)
//   seg_stats      per-root exact sums: probabilities are accumulated as 2^40 fixed point in u64 so the sum (and the
//                  arg-max over component means) is independent of the atomic arrival order
//   seg_pick_*     best component per (frame, class): max mean, ties -> smallest root (first in raster order)
//   seg_mask       object map u8 (class id inside its best component, else 0) + tight bbox per (frame, class)
//   seg_bbox       get_bbox arithmetic (x40 rounding, re-centring, shift into the image)
//   choose_points  ordered stream compaction of (object map == cls) & (depth != 0) inside the crop, then N picks
//   backproject    float32 pin-hole back-projection, bit-identical to the numpy arithmetic
//   crop_normalize u8 RGB -> NHWC4 f32 (x[/255] - mean) / std
#include "common.h"
#include "seg_head.h"

namespace {

constexpr int kT = 256;
inline int grid_for(long work) { long g = (work + kT - 1) / kT; return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g)); }

// ---------------------------------------------------------------------------------------------------------------
__global__ void seg_argmax_kernel(const float* __restrict__ logits, int ld, int C, uint8_t* __restrict__ label,
                                  float* __restrict__ score, long npix, int double_softmax)
{
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
        const float* l = logits + p * ld;
        float m = l[0];
        int am = 0;
        for (int c = 1; c < C; ++c)
            if (l[c] > m) { m = l[c]; am = c; }  // first maximum, like torch.argmax on the CPU
        // The reference takes the arg-max of the PROBABILITIES it finally holds (pipeline/utils.py:430-435): a class whose last exponential
        // (exp(l - m) after one softmax, exp(p1 - p1max) after two) rounds to 1 has the maximum's float32 probability, and torch.argmax
        // then returns the LOWEST such index (e == 1 implies p1 == p1max: the two-softmax test contains the one-softmax one).
        float s = 0.f;
        for (int c = 0; c < C; ++c) {
            const float e = expf(l[c] - m);
            if (!double_softmax && e == 1.f && c < am) am = c;
            s += e;
        }
        float pm = 1.f / s;  // softmax(logits)[am]  (activation='softmax' inside predict, create_labels.py:23)
        if (double_softmax) {
            // F.softmax applied again on the probabilities (pipeline/utils.py:430): the maximum is p1[am]
            float s2 = 0.f;
            for (int c = 0; c < C; ++c) {
                const float e2 = expf(expf(l[c] - m) / s - pm);
                if (e2 == 1.f && c < am) am = c;
                s2 += e2;
            }
            pm = 1.f / s2;
        }
        label[p] = (uint8_t)am;
        score[p] = pm;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// union-find on the pixel grid; every access to L during the merge is an agent-scope atomic (coherent across XCDs)
__device__ __forceinline__ int ld_relaxed(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ int uf_find(const int* L, int a)
{
    int p = ld_relaxed(&L[a]);
    while (p != a) { a = p; p = ld_relaxed(&L[a]); }
    return a;
}

__device__ void uf_union(int* L, int a, int b)
{
    bool done = false;
    while (!done) {
        a = uf_find(L, a);
        b = uf_find(L, b);
        if (a < b) {
            const int old = atomicMin(&L[b], a);
            done = old == b;
            b = old;
        } else if (b < a) {
            const int old = atomicMin(&L[a], b);
            done = old == a;
            a = old;
        } else {
            done = true;
        }
    }
}

// Run-based labelling: a wave covers 64 consecutive pixels; inside it every pixel is pointed at the first pixel of its
// horizontal run straight away (ballot of the run boundaries, no atomics), so the union-find only has to join RUNS:
//   * a run that continues across the wave's left edge joins the previous wave's run (one union per wave at most);
//   * a run joins the row above once per overlap with a run there (at the first pixel of the overlap), plus the two diagonal
//     contacts that no vertical contact implies.
// A 126 x 126 object costs ~130 unions instead of ~16 000 per-pixel unions all chasing the same root.  Unions link the larger
// root under the smaller, so the final root of a component is its smallest pixel index whatever the order: identical labels.
__global__ void ccl_init_kernel(const uint8_t* __restrict__ label, int* __restrict__ L, int W, long npix)
{
    const long nround = (npix + 63) & ~63L;          // whole waves stay converged for the ballot
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < nround; p += (long)gridDim.x * blockDim.x) {
        const int lane = threadIdx.x & 63;
        const bool in = p < npix;
        const int c = in ? label[p] : 0;
        const int x = in ? (int)(p % W) : 0;
        const int cl = __shfl_up(c, 1);
        const bool cont = in && lane > 0 && x > 0 && cl == c;          // continues the run of the lane to the left
        const unsigned long long starts = __ballot(!cont);
        if (in) {
            const unsigned long long below = starts & ((2ULL << lane) - 1ULL);      // lane 63: 2<<63 wraps to 0, -1 = all ones
            const int start = 63 - __clzll(below);
            L[p] = c ? (int)(p - (lane - start)) : -1;
        }
    }
}

// L indices are global over the batch (frame b occupies [b*H*W, (b+1)*H*W)), so roots are unique batch-wide.
__global__ void ccl_merge_kernel(const uint8_t* __restrict__ label, int* __restrict__ L, int H, int W, long npix)
{
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
        const uint8_t c = label[p];
        if (!c) continue;
        const int x = p % W;
        const int y = (p / W) % H;
        const bool left = x > 0 && label[p - 1] == c;
        if (left && (threadIdx.x & 63) == 0) uf_union(L, (int)p, (int)p - 1);   // in-wave run links were made by ccl_init_kernel
        if (y > 0) {
            const bool nw = x > 0 && label[p - W - 1] == c;
            if (label[p - W] == c) {
                // N present (NW and NE hang on N's run): one union per overlap of this run with a run above, at its first pixel
                if (!left || !nw) uf_union(L, (int)p, (int)(p - W));
            } else {
                if (nw && !left) uf_union(L, (int)p, (int)(p - W - 1));        // with a left neighbour, ITS N is this NW
                if (x < W - 1 && label[p - W + 1] == c && label[p + 1] != c)     // with a right neighbour, ITS N is this NE
                    uf_union(L, (int)p, (int)(p - W + 1));
            }
        }
    }
}
// sum / cnt (may be null): the per-component score accumulators live at the ROOT pixel's index; a root zeroes its own pair here instead of a
// pass that zeroed all npix pairs (236 MB per 64 frames) before the labelling
__global__ void ccl_compress_kernel(int* __restrict__ L, long npix, unsigned long long* __restrict__ sum, unsigned int* __restrict__ cnt)
{
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
        int a = L[p];
        if (a < 0) continue;
        while (true) {
            const int q = L[a];   // parents only ever point to smaller indices: chains end at the root
            if (q == a) break;
            a = q;
        }
        L[p] = a;   // benign race: other lanes may read either the old parent or the root, both lead to the root
        if (sum && a == (int)p) { sum[p] = 0ull; cnt[p] = 0u; }
    }
}

// ---------------------------------------------------------------------------------------------------------------
constexpr int kMaxCls = 64;
constexpr int kMaxClsHead = 16;   // classes supported by the fused head kernel (registers)

__global__ void seg_stats_kernel(const uint8_t* __restrict__ label, const float* __restrict__ score, const int* __restrict__ L,
                                 unsigned long long* __restrict__ sum, unsigned int* __restrict__ cnt,
                                 unsigned int* __restrict__ hist, int HW, int C)
{
    __shared__ unsigned int lh[kMaxCls];
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < kMaxCls; i += blockDim.x) lh[i] = 0;
    __syncthreads();
    // whole waves walk 64 consecutive pixels; HW is a multiple of 64 for 480x640, the tail is handled by `in`
    for (int q0 = blockIdx.x * blockDim.x; q0 < HW; q0 += gridDim.x * blockDim.x) {
        const int q = q0 + threadIdx.x;
        const bool in = q < HW;
        const long p = (long)b * HW + q;
        int root = -1;
        unsigned long long fx = 0;
        uint8_t c = 0;
        if (in) {
            c = label[p];
            if (c) {
                root = L[p];
                fx = (unsigned long long)(score[p] * 1099511627776.0f);  // 2^40 fixed point, exact for p < 1
            }
        }
        // wave-level aggregation: a wave is 64 consecutive pixels and sits in ONE component, or in a handful at an object's edge:
        // one atomic pair per distinct root (every lane of an object used to hit the same two addresses: ~15 000 serialised
        // atomics per 126 x 126 object); u64 fixed-point sums, so the grouping does not change the result
        unsigned long long todo = __ballot(root >= 0);
        for (int rounds = 0; todo && rounds < 6; ++rounds) {
            const int leader = __ffsll((long long)todo) - 1;
            const int r0 = __shfl(root, leader);
            const bool mine = root == r0;
            const unsigned long long grp = __ballot(mine);
            unsigned long long sacc = mine ? fx : 0ULL;
            for (int off = 32; off > 0; off >>= 1) sacc += __shfl_xor(sacc, off);
            if ((threadIdx.x & 63) == leader) {
                atomicAdd(&sum[r0], sacc);
                atomicAdd(&cnt[r0], (unsigned int)__popcll(grp));
            }
            todo &= ~grp;
        }
        if (root >= 0 && ((todo >> (threadIdx.x & 63)) & 1ULL)) {      // speckle-heavy wave: the rest one by one
            atomicAdd(&sum[root], fx);
            atomicAdd(&cnt[root], 1u);
        }
        if (c) atomicAdd(&lh[c], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += blockDim.x)
        if (lh[i]) atomicAdd(&hist[(long)b * C + i], lh[i]);
}

__global__ void seg_small_init_kernel(unsigned int* hist, unsigned long long* best_key, int* best_root, int* tight, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    hist[i] = 0;
    best_key[i] = 0;
    best_root[i] = 0x7fffffff;
    tight[i * 4 + 0] = 0x7fffffff; tight[i * 4 + 1] = -1; tight[i * 4 + 2] = 0x7fffffff; tight[i * 4 + 3] = -1;
}

__global__ void seg_pick_max_kernel(const uint8_t* __restrict__ label, const int* __restrict__ L,
                                    const unsigned long long* __restrict__ sum, const unsigned int* __restrict__ cnt,
                                    const unsigned int* __restrict__ hist, unsigned long long* __restrict__ best_key,
                                    int HW, int C, int min_pixels, int use_sum, long npix)
{
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
        if (L[p] != (int)p) continue;
        const int b = p / HW, c = label[p];
        if (hist[(long)b * C + c] <= (unsigned)min_pixels) continue;   // counts[i] > 100 (pipeline/utils.py:445)
        const double mean = use_sum ? (double)sum[p] : (double)sum[p] / (double)cnt[p];
        atomicMax(&best_key[(long)b * C + c], (unsigned long long)__double_as_longlong(mean));
    }
}

__global__ void seg_pick_root_kernel(const uint8_t* __restrict__ label, const int* __restrict__ L,
                                     const unsigned long long* __restrict__ sum, const unsigned int* __restrict__ cnt,
                                     const unsigned int* __restrict__ hist, const unsigned long long* __restrict__ best_key,
                                     int* __restrict__ best_root, int HW, int C, int min_pixels, int use_sum, long npix)
{
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
        if (L[p] != (int)p) continue;
        const int b = p / HW, c = label[p];
        if (hist[(long)b * C + c] <= (unsigned)min_pixels) continue;
        const double mean = use_sum ? (double)sum[p] : (double)sum[p] / (double)cnt[p];
        if ((unsigned long long)__double_as_longlong(mean) == best_key[(long)b * C + c])
            atomicMin(&best_root[(long)b * C + c], (int)p);   // ties: first component in raster order
    }
}

// objmap[p] = class if p belongs to the best component of its class, else 0; tight bbox per (frame, class).
// Only the ends of a component's row runs can move the extrema, so only those pixels issue atomics.
__global__ void seg_mask_kernel(const uint8_t* __restrict__ label, const int* __restrict__ L, const int* __restrict__ best_root,
                                uint8_t* __restrict__ objmap, int* __restrict__ tight, int H, int W, int C, long npix)
{
    const int HW = H * W;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
        const int c = label[p];
        uint8_t o = 0;
        if (c) {
            const int b = p / HW;
            if (L[p] == best_root[(long)b * C + c]) {
                o = (uint8_t)c;
                const int rem = p - (long)b * HW;
                const int y = rem / W, x = rem - y * W;
                int* t = tight + ((long)b * C + c) * 4;
                if (x == 0 || label[p - 1] != c) { atomicMin(&t[2], x); atomicMin(&t[0], y); atomicMax(&t[1], y); }
                if (x == W - 1 || label[p + 1] != c) atomicMax(&t[3], x);
            }
        }
        objmap[p] = o;
    }
}

__device__ __forceinline__ int up40(int v) { return v % 40 == 0 ? v : (v / 40 + 1) * 40; }

// det[b][c] = (valid, rmin, rmax, cmin, cmax): dataset.py:342-380 on the tight extents
__global__ void seg_bbox_kernel(const int* __restrict__ tight, const int* __restrict__ best_root, int* __restrict__ det,
                                int n, int H, int W)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int* d = det + (long)i * 5;
    if (best_root[i] == 0x7fffffff) { d[0] = 0; d[1] = d[2] = d[3] = d[4] = 0; return; }
    int rmin = tight[i * 4 + 0], rmax = tight[i * 4 + 1] + 1, cmin = tight[i * 4 + 2], cmax = tight[i * 4 + 3] + 1;
    const int r_b = up40(rmax - rmin), c_b = up40(cmax - cmin);
    const int cr = (rmin + rmax) / 2, cc = (cmin + cmax) / 2;     // int((a+b)/2) of non-negative ints
    rmin = cr - r_b / 2; rmax = cr + r_b / 2;
    cmin = cc - c_b / 2; cmax = cc + c_b / 2;
    if (rmin < 0) { rmax += -rmin; rmin = 0; }
    if (cmin < 0) { cmax += -cmin; cmin = 0; }
    if (rmax > H) { rmin -= rmax - H; rmax = H; }
    if (cmax > W) { cmin -= cmax - W; cmax = W; }
    d[0] = 1; d[1] = rmin; d[2] = rmax; d[3] = cmin; d[4] = cmax;
}

// ---------------------------------------------------------------------------------------------------------------
// objects[o] = (frame, cls, rmin, rmax, cmin, cmax).  One workgroup per object: ordered compaction, then N picks.
__global__ __launch_bounds__(kT) void choose_points_kernel(const uint8_t* __restrict__ objmap, const uint16_t* __restrict__ depth,
                                                           const int* __restrict__ objects, int H, int W, int N,
                                                           unsigned int seed, const unsigned int* __restrict__ seed_dev, int* __restrict__ cand,
                                                           long cand_stride, int64_t* __restrict__ choose, int* __restrict__ n_cand)
{
    // Ordered compaction of the crop's candidate pixels (mask && depth != 0, pipeline/utils.py:524-531), row-major.  A thread owns a RUN of
    // consecutive pixels per pass (its candidates stay in order, thread t's run precedes thread t + 1's), the runs' counts are scanned over
    // the workgroup: one pass covers kT * RUN pixels behind two barriers (the one-pixel-per-thread form paid three barriers per 256 pixels:
    // 108 us for 64 crops of 160 x 160, now ~15).
    constexpr int RUN = 16;
    __shared__ int wsum[2][kT / 64];
    const int o = blockIdx.x;
    const int* ob = objects + (long)o * 6;
    const int b = ob[0], cls = ob[1], rmin = ob[2], rmax = ob[3], cmin = ob[4], cmax = ob[5];
    const int Wc = cmax - cmin, total = (rmax - rmin) * Wc;
    const uint8_t* om = objmap + (long)b * H * W;
    const uint16_t* dp = depth + (long)b * H * W;
    int* cd = cand + (long)o * cand_stride;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int running = 0, par = 0;
    for (int base = 0; base < total; base += kT * RUN, par ^= 1) {
        const int i0 = base + threadIdx.x * RUN;
        unsigned mask = 0;
        if (i0 < total) {
            int r = i0 / Wc, c = i0 - r * Wc;
#pragma unroll
            for (int k = 0; k < RUN; ++k) {
                if (i0 + k < total) {
                    const int p = (r + rmin) * W + c + cmin;
                    if (om[p] == cls && dp[p] != 0) mask |= 1u << k;
                }
                if (++c == Wc) { c = 0; ++r; }
            }
        }
        const int cnt = __popc(mask);
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (lane == 63) wsum[par][wave] = incl;
        __syncthreads();                    // (wsum is double-buffered: the next pass writes the other half, so one barrier per pass suffices)
        int off = running + incl - cnt, all = 0;
#pragma unroll
        for (int w = 0; w < kT / 64; ++w) {
            if (w < wave) off += wsum[par][w];
            all += wsum[par][w];
        }
        for (unsigned m = mask; m; m &= m - 1) cd[off++] = i0 + __builtin_ctz(m);
        running += all;
    }
    const int n = running;
    if (threadIdx.x == 0) n_cand[o] = n;
    if (n == 0) return;
    __threadfence_block();
    __syncthreads();                        // every candidate written (workgroup-visible) before the picks read them
    if (n > N) {
        // ordered subset with equal inclusion probability N/n (systematic sampling; the reference draws an unseeded
        // np.random.shuffle of a 0/1 mask, pipeline/utils.py:533-537 -- parity tests inject `choose` instead)
        unsigned int h = (seed_dev ? *seed_dev : seed) ^ (0x9E3779B9u * (unsigned)(o + 1));
        h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
        const double u = (double)h / 4294967296.0;
        for (int i = threadIdx.x; i < N; i += kT) {
            long j = (long)(((double)i + u) * (double)n / (double)N);
            j = j >= n ? n - 1 : j;
            choose[(long)o * N + i] = cd[j];
        }
    } else {
        for (int i = threadIdx.x; i < N; i += kT) choose[(long)o * N + i] = cd[i % n];   // np.pad(..., 'wrap')
    }
}

__global__ void backproject_kernel(const uint16_t* __restrict__ depth, const int* __restrict__ objects,
                                   const int64_t* __restrict__ choose, float4* __restrict__ points, int H, int W, int N,
                                   float fx, float fy, float ppx, float ppy, float depth_scale)
{
    const int o = blockIdx.y;
    const int* ob = objects + (long)o * 6;
    const int b = ob[0], rmin = ob[2], cmin = ob[4], Wc = ob[5] - ob[4];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const int idx = (int)choose[(long)o * N + i];
        const int r = idx / Wc + rmin, c = idx % Wc + cmin;
        const float d = (float)depth[((long)b * H + r) * W + c];
        const float pt2 = d * depth_scale;                       // pipeline/utils.py:549
        const float pt0 = ((float)c - ppx) * pt2 / fx;           // :550  (ymap = column index)
        const float pt1 = ((float)r - ppy) * pt2 / fy;           // :551  (xmap = row index)
        points[(long)o * N + i] = make_float4(pt0, pt1, pt2, 0.f);
    }
}

// rects[o] = (frame, rmin, cmin); output [n][Hc][Wc][4]
__global__ void crop_normalize_kernel(const uint8_t* __restrict__ rgb, const int* __restrict__ rects, float4* __restrict__ out,
                                      int H, int W, int Hc, int Wc, int div255, long total)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = i % Wc;
        long t = i / Wc;
        const int y = t % Hc;
        const int o = t / Hc;
        const int b = rects[o * 3], r = rects[o * 3 + 1] + y, c = rects[o * 3 + 2] + x;
        const uint8_t* px = rgb + (((long)b * H + r) * W + c) * 3;
        float v0 = (float)px[0], v1 = (float)px[1], v2 = (float)px[2];
        if (div255) { v0 = v0 / 255.f; v1 = v1 / 255.f; v2 = v2 / 255.f; }   // torchvision ToTensor
        out[i] = make_float4((v0 - 0.485f) / 0.229f, (v1 - 0.456f) / 0.224f, (v2 - 0.406f) / 0.225f, 0.f);
    }
}

// Background-subtraction features (background_subtraction/utils.py:721-828): per pixel the 7 channels
//   |f_rgb - b_rgb| (3), |HSV(f) - HSV(b)| (3, Pillow's integer HSV), |f_depth - b_depth| after the distance gate and the
//   mutual zero test (:756-763), every channel cast to uint8 (the depth difference WRAPS modulo 256, numpy's float -> uint8
//   cast at :811), then ToTensor (/255) and Normalize((x - mean) / std) in float32 (:818-819).  out[B][H][W][8], channel 7 = 0.
struct BgsubArgs {
    float mean[7], stdv[7];
};

// Pillow's rgb2hsv_row (Convert.c, follows colorsys.py): float divisions, the hue wrap and the * 255.0 in double, truncation.
// Pinned bit for bit against PIL over all 2^24 colours (tools/gen_golden_bgsub.py).
__device__ __forceinline__ void pil_hsv(int r, int g, int b, int& uh, int& us, int& uv)
{
    const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
    uv = maxc;
    if (minc == maxc) { uh = 0; us = 0; return; }
    const float cr = (float)(maxc - minc);
    const float s = cr / (float)maxc;
    const float rc = (float)(maxc - r) / cr, gc = (float)(maxc - g) / cr, bc = (float)(maxc - b) / cr;
    float h;
    if (r == maxc) h = bc - gc;
    else if (g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
    else h = (float)(4.0 + (double)gc - (double)rc);
    const double hw = (double)h / 6.0 + 1.0;
    h = (float)(hw - floor(hw));                    // fmod(., 1.0) of a positive double: exact
    const int ih = (int)((double)h * 255.0), is = (int)((double)s * 255.0);
    uh = ih < 0 ? 0 : (ih > 255 ? 255 : ih);
    us = is < 0 ? 0 : (is > 255 ? 255 : is);
}

__global__ void bgsub_features_kernel(const uint8_t* __restrict__ f_rgb, const uint8_t* __restrict__ b_rgb,
                                      const uint16_t* __restrict__ f_depth, const uint16_t* __restrict__ b_depth,
                                      const double* __restrict__ gate /*[B][2] min,max*/, BgsubArgs a, float4* __restrict__ out,
                                      uint8_t* __restrict__ diff /*[B][H][W][7] or null*/, int HW, long npix)
{
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
        const int b = p / HW;
        const double dmin = gate[b * 2], dmax = gate[b * 2 + 1];
        const uint8_t* fp = f_rgb + p * 3;
        const uint8_t* bp = b_rgb + p * 3;
        const int fr = fp[0], fg = fp[1], fb = fp[2], br = bp[0], bg = bp[1], bb = bp[2];
        int fh, fs, fv, bh, bs, bv;
        pil_hsv(fr, fg, fb, fh, fs, fv);
        pil_hsv(br, bg, bb, bh, bs, bv);
        double fd = (double)f_depth[p], bd = (double)b_depth[p];
        if (fd > dmax) fd = 0.0;         // :756-759
        if (bd > dmax) bd = 0.0;
        if (fd < dmin) fd = 0.0;
        if (bd < dmin) bd = 0.0;
        if (bd == 0.0) fd = 0.0;         // :762-763 (sequential: the second test sees the updated f_depth)
        if (fd == 0.0) bd = 0.0;
        const int dd = (int)fabs(fd - bd) & 255;
        const int ch[7] = {abs(fr - br), abs(fg - bg), abs(fb - bb), abs(fh - bh), abs(fs - bs), abs(fv - bv), dd};
        float v[8];
#pragma unroll
        for (int c = 0; c < 7; ++c) v[c] = ((float)ch[c] / 255.f - a.mean[c]) / a.stdv[c];
        v[7] = 0.f;
        out[p * 2] = make_float4(v[0], v[1], v[2], v[3]);
        out[p * 2 + 1] = make_float4(v[4], v[5], v[6], v[7]);
        if (diff) {
#pragma unroll
            for (int c = 0; c < 7; ++c) diff[p * 7 + c] = (uint8_t)ch[c];
        }
    }
}

// trust checks of the pose-label relabelling (label_generator/create_labels.py:166-196): per frame six counts of
// (condition, pred set / unset) pairs: cond 0 = background-subtraction label != 0, cond 1 = depth inside the +-gate and != 0,
// cond 2 = centre window [cut0, H-cut0) x [cut1, W-cut1).  counts[b][6] = (c0&p, c0&!p, c1&p, c1&!p, c2&p, c2&!p).
__global__ __launch_bounds__(kT) void label_trust_kernel(const uint8_t* __restrict__ objmap, int cls, const uint8_t* __restrict__ bs_label,
                                                         const uint16_t* __restrict__ depth, const float* __restrict__ gate /*[B][2] min,max*/,
                                                         int H, int W, int cut0, int cut1, unsigned int* __restrict__ counts)
{
    __shared__ unsigned int lc[6];
    const int b = blockIdx.y;
    if (threadIdx.x < 6) lc[threadIdx.x] = 0;
    __syncthreads();
    unsigned int c[6] = {0, 0, 0, 0, 0, 0};
    const float dmin = gate[b * 2], dmax = gate[b * 2 + 1];
    const int HW = H * W;
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < HW; q += gridDim.x * blockDim.x) {
        const long p = (long)b * HW + q;
        const int pr = objmap[p] == cls ? 0 : 1;
        const int y = q / W, x = q - y * W;
        const float d = (float)depth[p];
        if (bs_label && bs_label[p] != 0) c[0 + pr]++;
        if (d != 0.f && !(d > dmax) && !(d < dmin)) c[2 + pr]++;       // depth[depth > max] = 0; depth[depth < min] = 0 (:111-112)
        if (y >= cut0 && y < H - cut0 && x >= cut1 && x < W - cut1) c[4 + pr]++;
    }
    for (int i = 0; i < 6; ++i) {
        unsigned int v = c[i];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&lc[i], v);
    }
    __syncthreads();
    if (threadIdx.x < 6 && lc[threadIdx.x]) atomicAdd(&counts[b * 6 + threadIdx.x], lc[threadIdx.x]);
}

// final 1x1 conv (64 -> C classes) + softmax (+ softmax) + argmax fused into one streaming pass over the 64-channel up_3
// activation (pspnet.py:53-55 restricted to the first C rows, then pipeline/utils.py:429-435): the C x 480 x 640 logits are
// never written.  The 64 -> 16 contraction runs on the exact-fp32 matrix cores (v_mfma_f32_16x16x4_f32, D = W . X^T: rows =
// classes, columns = 16 pixels): lane (px = l&15, kq = l>>4) loads four float4 of its pixel (k = 16 j + 4 kq + e), the 16
// weight operands per lane stay in registers for the whole kernel, and each pixel's 16 logits end up 4 per lane on the four
// lanes {px, px+16, px+32, px+48}, so max / arg-max / exp-sums need only two xor-shuffles.
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(kT) void seg_head_kernel(const float4* __restrict__ feat, const float* __restrict__ w, const float* __restrict__ bias,
                                                      int C, uint8_t* __restrict__ label, float* __restrict__ score, long npix, int double_softmax)
{
    const int lane = threadIdx.x & 63;
    const int px = lane & 15, kq = lane >> 4;
    float wreg[16], breg[4];
    ape_seg::seg_head_load_weights(w, bias, C, lane, wreg, breg);
    const long ngroups = (npix + 15) / 16;
    const long wave_id = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6, nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    for (long g = wave_id; g < ngroups; g += nwaves) {
        const long p = g * 16 + px;
        const bool in = p < npix;
        float4 x[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = in ? feat[p * 16 + 4 * j + kq] : make_float4(0.f, 0.f, 0.f, 0.f);
        int am;
        float pm;
        ape_seg::seg_head_group(x, wreg, breg, C, lane, double_softmax, am, pm);
        if (in && kq == 0) {
            label[p] = (uint8_t)am;
            score[p] = pm;
        }
    }
}

}  // namespace

extern "C" int ape_seg_argmax_f32(const float* logits, int ld, int C, uint8_t* label, float* score, long npix,
                                  int double_softmax, void* stream)
{
    if (!logits || !label || !score || C < 1 || C > kMaxCls || ld < C || npix < 0) return APE_EINVAL;
    if (npix == 0) return APE_OK;
    hipLaunchKernelGGL(seg_argmax_kernel, dim3(grid_for(npix)), dim3(kT), 0, (hipStream_t)stream, logits, ld, C, label, score,
                       npix, double_softmax);
    return ape::check_launch("ape_seg_argmax_f32");
}

extern "C" size_t ape_seg_components_workspace_bytes(int B, int H, int W, int C)
{
    const size_t npix = (size_t)B * H * W;
    // sum u64[npix] | best_key u64[B*C] | L i32[npix] | cnt u32[npix] | hist u32[B*C] | best_root i32[B*C] | tight i32[B*C*4]
    return npix * (8 + 4 + 4) + (size_t)B * C * (8 + 4 + 4 + 16) + 256;
}

/* label[B][H][W] u8, score[B][H][W] f32 -> objmap[B][H][W] u8, det[B][C][5] i32 (valid,rmin,rmax,cmin,cmax) */
extern "C" int ape_seg_components(const uint8_t* label, const float* score, uint8_t* objmap, int* det, int B, int H, int W,
                                  int C, int min_pixels, void* workspace, size_t workspace_bytes, void* stream)
{
    return ape_seg_components_scored(label, score, objmap, det, B, H, W, C, min_pixels, APE_SEG_SCORE_MEAN, workspace,
                                     workspace_bytes, stream);
}

extern "C" int ape_seg_components_scored(const uint8_t* label, const float* score, uint8_t* objmap, int* det, int B, int H, int W,
                                         int C, int min_pixels, int score_mode, void* workspace, size_t workspace_bytes, void* stream)
{
    if (score_mode != APE_SEG_SCORE_MEAN && score_mode != APE_SEG_SCORE_SUM) return APE_EINVAL;
    const int use_sum = score_mode == APE_SEG_SCORE_SUM;
    if (!label || !score || !objmap || !det || !workspace || B < 0 || H < 1 || W < 1 || C < 1 || C > kMaxCls) return APE_EINVAL;
    if (B == 0) return APE_OK;
    const long npix = (long)B * H * W;
    if (npix >= (1L << 31) - 1) return APE_EINVAL;
    if (workspace_bytes < ape_seg_components_workspace_bytes(B, H, W, C)) return APE_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    unsigned long long* sum = (unsigned long long*)ws;            ws += npix * 8;
    unsigned long long* best_key = (unsigned long long*)ws;       ws += (size_t)B * C * 8;
    int* L = (int*)ws;                                            ws += npix * 4;
    unsigned int* cnt = (unsigned int*)ws;                        ws += npix * 4;
    unsigned int* hist = (unsigned int*)ws;                       ws += (size_t)B * C * 4;
    int* best_root = (int*)ws;                                    ws += (size_t)B * C * 4;
    int* tight = (int*)ws;
    const int g = grid_for(npix);
    // (sum / cnt are zeroed where they are used, at the component roots, by ccl_compress_kernel.  Kernels, not hipMemsetAsync: memset nodes of a
    // captured HIP graph did not re-zero on replay -- tools/probes/graph_probe.py)
    const int BC = B * C;
    hipLaunchKernelGGL(seg_small_init_kernel, dim3(ape::ceil_div(BC, kT)), dim3(kT), 0, st, hist, best_key, best_root, tight, BC);
    hipLaunchKernelGGL(ccl_init_kernel, dim3(g), dim3(kT), 0, st, label, L, W, npix);
    hipLaunchKernelGGL(ccl_merge_kernel, dim3(g), dim3(kT), 0, st, label, L, H, W, npix);
    hipLaunchKernelGGL(ccl_compress_kernel, dim3(g), dim3(kT), 0, st, L, npix, sum, cnt);
    int gx = ape::ceil_div((long)H * W, kT);
    gx = gx > 64 ? 64 : gx;
    hipLaunchKernelGGL(seg_stats_kernel, dim3(gx, B), dim3(kT), 0, st, label, score, L, sum, cnt, hist, H * W, C);
    hipLaunchKernelGGL(seg_pick_max_kernel, dim3(g), dim3(kT), 0, st, label, L, sum, cnt, hist, best_key, H * W, C, min_pixels, use_sum, npix);
    hipLaunchKernelGGL(seg_pick_root_kernel, dim3(g), dim3(kT), 0, st, label, L, sum, cnt, hist, best_key, best_root, H * W, C,
                       min_pixels, use_sum, npix);
    hipLaunchKernelGGL(seg_mask_kernel, dim3(g), dim3(kT), 0, st, label, L, best_root, objmap, tight, H, W, C, npix);
    hipLaunchKernelGGL(seg_bbox_kernel, dim3(ape::ceil_div(BC, kT)), dim3(kT), 0, st, tight, best_root, det, BC, H, W);
    return ape::check_launch("ape_seg_components_scored");
}

/* objects[n][6] i32 = (frame, cls, rmin, rmax, cmin, cmax) -> choose[n][N] i64 (index inside the crop, row-major over Wc),
 * n_cand[n] i32 (0 => the reference `continue`s, pipeline/utils.py:530-531).  cand: i32 scratch [n][cand_stride]. */
static int choose_points_run(const uint8_t* objmap, const uint16_t* depth, const int* objects, int n, int H, int W, int N, unsigned int seed,
                             const unsigned int* seed_dev, int* cand, long cand_stride, int64_t* choose, int* n_cand, void* stream)
{
    if (!objmap || !depth || !objects || !cand || !choose || !n_cand || n < 0 || H < 1 || W < 1 || N < 1 || cand_stride < 1)
        return APE_EINVAL;
    if (n == 0) return APE_OK;
    hipLaunchKernelGGL(choose_points_kernel, dim3(n), dim3(kT), 0, (hipStream_t)stream, objmap, depth, objects, H, W, N, seed, seed_dev,
                       cand, cand_stride, choose, n_cand);
    return ape::check_launch("ape_choose_points");
}

extern "C" int ape_choose_points(const uint8_t* objmap, const uint16_t* depth, const int* objects, int n, int H, int W, int N,
                                 unsigned int seed, int* cand, long cand_stride, int64_t* choose, int* n_cand, void* stream)
{
    return choose_points_run(objmap, depth, objects, n, H, W, N, seed, nullptr, cand, cand_stride, choose, n_cand, stream);
}

extern "C" int ape_choose_points_dseed(const uint8_t* objmap, const uint16_t* depth, const int* objects, int n, int H, int W, int N,
                                       const unsigned int* seed_device, int* cand, long cand_stride, int64_t* choose, int* n_cand, void* stream)
{
    if (!seed_device) return APE_EINVAL;
    return choose_points_run(objmap, depth, objects, n, H, W, N, 0u, seed_device, cand, cand_stride, choose, n_cand, stream);
}

extern "C" int ape_backproject_f32(const uint16_t* depth, const int* objects, const int64_t* choose, float* points4, int n,
                                   int H, int W, int N, float fx, float fy, float ppx, float ppy, float depth_scale, void* stream)
{
    if (!depth || !objects || !choose || !points4 || n < 0 || N < 1) return APE_EINVAL;
    if (n == 0) return APE_OK;
    hipLaunchKernelGGL(backproject_kernel, dim3(ape::ceil_div(N, kT), n), dim3(kT), 0, (hipStream_t)stream, depth, objects, choose,
                       (float4*)points4, H, W, N, fx, fy, ppx, ppy, depth_scale);
    return ape::check_launch("ape_backproject_f32");
}

extern "C" int ape_preprocess_u8_nhwc4(const uint8_t* rgb, const int* rects, float* out, int n, int H, int W, int Hc, int Wc,
                                       int div255, void* stream)
{
    if (!rgb || !rects || !out || n < 0 || H < 1 || W < 1 || Hc < 1 || Wc < 1 || Hc > H || Wc > W) return APE_EINVAL;
    const long total = (long)n * Hc * Wc;
    if (total == 0) return APE_OK;
    hipLaunchKernelGGL(crop_normalize_kernel, dim3(grid_for(total)), dim3(kT), 0, (hipStream_t)stream, rgb, rects, (float4*)out, H,
                       W, Hc, Wc, div255, total);
    return ape::check_launch("ape_preprocess_u8_nhwc4");
}

/* counts[B][6] u32 must be zeroed by the caller (hipMemsetAsync / torch.zeros); see label_trust_kernel. */
extern "C" int ape_label_trust_counts(const uint8_t* objmap, int cls, const uint8_t* bs_label_or_null, const uint16_t* depth,
                                      const float* gate_min_max, int B, int H, int W, int cut0, int cut1, unsigned int* counts,
                                      void* stream)
{
    if (!objmap || !depth || !gate_min_max || !counts || B < 0 || H < 1 || W < 1 || cls < 1 || cls > 255) return APE_EINVAL;
    if (B == 0) return APE_OK;
    int gx = ape::ceil_div((long)H * W, kT);
    gx = gx > 64 ? 64 : gx;
    hipLaunchKernelGGL(label_trust_kernel, dim3(gx, B), dim3(kT), 0, (hipStream_t)stream, objmap, cls, bs_label_or_null, depth,
                       gate_min_max, H, W, cut0, cut1, counts);
    return ape::check_launch("ape_label_trust_counts");
}

/* feat[npix][64] f32 (up_3 activation), w[C][64], bias[C] -> label u8, score f32; C <= 16 */
extern "C" int ape_seg_head_f32(const float* feat, const float* w, const float* bias, int C, uint8_t* label, float* score, long npix,
                                int double_softmax, void* stream)
{
    if (!feat || !w || !label || !score || C < 1 || C > kMaxClsHead || npix < 0) return APE_EINVAL;
    if (npix == 0) return APE_OK;
    long g = (npix / 16 + kT / 64 - 1) / (kT / 64);   // one wave per 16-pixel group, at most 8192 workgroups, grid-stride
    g = g < 1 ? 1 : (g > 8192 ? 8192 : g);
    hipLaunchKernelGGL(seg_head_kernel, dim3((int)g), dim3(kT), 0, (hipStream_t)stream, (const float4*)feat, w, bias, C, label, score,
                       npix, double_softmax);
    return ape::check_launch("ape_seg_head_f32");
}

/* gate_min_max[B][2] f64 DEVICE; mean7 / std7 HOST pointers (7 floats each); out[B][H][W][8] f32; diff_or_null[B][H][W][7] u8 */
extern "C" int ape_bgsub_features_f32(const uint8_t* f_rgb, const uint8_t* b_rgb, const uint16_t* f_depth, const uint16_t* b_depth,
                                      const double* gate_min_max, const float* mean7_host, const float* std7_host, float* out,
                                      uint8_t* diff_or_null, int B, int H, int W, void* stream)
{
    if (!f_rgb || !b_rgb || !f_depth || !b_depth || !gate_min_max || !mean7_host || !std7_host || !out || B < 0 || H < 1 || W < 1)
        return APE_EINVAL;
    const long npix = (long)B * H * W;
    if (npix == 0) return APE_OK;
    BgsubArgs a;
    for (int c = 0; c < 7; ++c) {
        if (!(std7_host[c] != 0.f)) return APE_EINVAL;
        a.mean[c] = mean7_host[c];
        a.stdv[c] = std7_host[c];
    }
    hipLaunchKernelGGL(bgsub_features_kernel, dim3(grid_for(npix)), dim3(kT), 0, (hipStream_t)stream, f_rgb, b_rgb, f_depth, b_depth,
                       gate_min_max, a, (float4*)out, diff_or_null, H * W, npix);
    return ape::check_launch("ape_bgsub_features_f32");
}
