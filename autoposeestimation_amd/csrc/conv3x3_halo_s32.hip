// 3x3 / stride 1 / pad == dilation in {1,2,4} convolution on PRE-SPLIT ("S32", include/ape_hip.h) activations: the successor of
// conv3x3_halo.hip for the wide (Cout >= 128) layers of the segmentor.  Same tile (16 x 16 output pixels x 128 output channels per
// workgroup, 8 waves = 4 pixel-row groups x 2 channel halves, wave tile 64 pixels x 64 channels, v_mfma_f32_16x16x32_bf16 with the
// weights as the row operand), same products and the same K order (32-channel chunk outer, nine taps inner), hence the same
// accumulators bit for bit.  What changed is how the operands get to the matrix cores:
//   * no staging registers, no split / convert VALU, no ds_write: the halo and the weight tiles are LDS-DMA'd
//     (`buffer_load_dwordx4 ... lds`, 1 KB = 8 pixels (rows) x 128 B per wave-instruction); pixels outside the image and rows past
//     Cout lie outside the buffer descriptors' range and arrive as zeros;
//   * the halo lives in a RING OF 32 IMAGE ROWS (24 pixels x 128 B each): the (16 + 2d)-row image of chunk c+1 is streamed in while
//     chunk c is multiplied -- its first 16 - 2d rows (ring rows the current image does not use) during taps 1 and 2, d rows after the
//     taps ky = 0 (tap 3), d after ky = 1 (tap 6) -- and its LAST 2d rows, which land on rows the current chunk's final tap still
//     reads, only in tap 0 of chunk c+1 itself, i.e. behind the barrier that closes that final tap (they are first read for ky = 1,
//     from tap 2's second phase on, two closing barriers later).  A second whole image does not fit beside the weights for d = 4;
//   * PERSISTENT workgroups: one per CU walks the tiles  blockIdx, + gridDim, ...; the chunk stream simply continues into the next
//     tile's first chunk, so only a workgroup's FIRST tile pays the prologue (image + three weight tiles), and a tile's epilogue
//     (residual tile requested up front, stores fire-and-forget) is followed at once by the next tile's taps, whose operands are
//     already in LDS / registers;
//   * the weights of the next taps sit in a ring of four 16 KB tiles, issued three taps ahead;
//   * the fragments of tap t+1 are read from LDS (pinned inline asm, cf. conv_gemm_s32.hip) into a second register set while the
//     48 MFMAs of tap t run; one barrier per tap, counted vmcnt (never 0 in the steady state).
//   * RAGGED LAST TILE ROW: a tile that holds fewer than 13 image rows (the segmentor's 60-row maps: rows 48..59 of every frame, a quarter
//     of all tiles) gives each pixel-row group of waves THREE rows instead of four (`rpw`): the fourth row's 12 MFMAs per tap are skipped
//     instead of multiplying rows that do not exist;
// LDS image: a pixel (a weight row) is 128 B = chunks 0..3 hi | 4..7 lo, chunk c of pixel column hx in slot c ^ ((hx >> 1) & 7)
// (weights: row instead of hx); the permutation is applied to the per-lane DMA SOURCE address, inside the 128-B line.
#include <type_traits>
#include "common.h"

// timing-only ablation switches (results wrong when set): compiled IN only by `make ablations` (-DAPE_ABLATIONS -> ../libape_hip_abl.so, what
// tools/mb_*_abl.py load).  In the product build they cost: the run-time tests around the MFMA rows, waits and DMA statements of halo_s32
// were 1 ms of the 33 ms step (same-box A/B).
#ifdef APE_ABLATIONS
#define ABL(bit) (a.dbg & (bit))
#else
#define ABL(bit) 0
#endif

// S32 output: one 16-byte store per lane instead of two 8-byte ones (round 6, conv_gemm_s32.hip); APE_S32_WIDE_STORE=0 builds the two-store form
#ifndef APE_S32_WIDE_STORE
#define APE_S32_WIDE_STORE 1
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

struct HaloS32Args {
    const char* x;          // S32 activations [B][H][W][ldx]
    const char* w;          // S32K weights [Cout][9 * Cin / 32][hi 32 | lo 32], K order (tap, channel)
    const float* bias;
    const char* res;
    char* y;
    int B, H, W, Cin, Cout;
    int ldx, xoff, ldy, yoff, ldr, roff;
    int act;
    float alpha;
    int bias_bstride;
    int out_fmt, res_fmt;
    int tiles_x, tiles_y, n_tiles;
    int dbg;                // ape_conv3x3_halo_s32_debug: 1 = static priority 1 for waves 4-7, 2 = one workgroup per tile instead of the
                            // persistent walk, 16 = four rows per wave-row group in every tile (results unchanged by any of them);
                            // timing-only ablations: 4 = no epilogue stores, 8 = no residual loads, 32 = two of the three MFMAs per product, 64 = no fragment-read waits, 128 = no per-tap barrier, 256 = no per-tap DMA wait,
                            // 512 = no in-loop DMA, 1024 = no MFMAs
};

// NONE / RELU / PRELU as selects on loop-invariant scalars (a `switch` per element compiled to a cascade of scalar compares and branches
// per element: ~1.8 k scalar instructions in a 256-element epilogue); same values bit for bit (1 * v == v, also for -0 and NaN)
__device__ __forceinline__ float act_h(float v, int act, float alpha)
{
    if (act == APE_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
    const float neg = act == APE_ACT_RELU ? 0.f : (act == APE_ACT_PRELU ? alpha : 1.f) * v;
    return v > 0.f ? v : neg;
}

#if defined(__HIP_DEVICE_COMPILE__)
#define APE_DS_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#else
#define APE_DS_READ(dst, addr, off) (void)(addr)
#endif


// fragment reads of HALF a tap (asm operands cannot name variables captured by a generic lambda, hence a free function): two
// pixel rows (hi, lo) at their own ring-row addresses and two of the four 16-channel weight blocks (2 KB apart)
template <int JOFF>
__device__ __forceinline__ void ds_read_half(u32x4 (&A)[2][2], u32x4 (&Bf)[4][2], unsigned a0h, unsigned a0l, unsigned a1h, unsigned a1l,
                                             unsigned bh, unsigned bl)
{
    APE_DS_READ(A[0][0], a0h, 0); APE_DS_READ(A[0][1], a0l, 0);
    APE_DS_READ(A[1][0], a1h, 0); APE_DS_READ(A[1][1], a1l, 0);
    APE_DS_READ(Bf[JOFF][0], bh, JOFF * 2048);           APE_DS_READ(Bf[JOFF][1], bl, JOFF * 2048);
    APE_DS_READ(Bf[JOFF + 1][0], bh, JOFF * 2048 + 2048); APE_DS_READ(Bf[JOFF + 1][1], bl, JOFF * 2048 + 2048);
}

// PP form: one pixel row's two planes / the four weight blocks' two planes
__device__ __forceinline__ void ds_read_pair(u32x4 (&A)[2], unsigned ah, unsigned al)
{
    APE_DS_READ(A[0], ah, 0); APE_DS_READ(A[1], al, 0);
}
__device__ __forceinline__ void ds_read_b4(u32x4 (&Bf)[4][2], unsigned bh, unsigned bl)
{
    APE_DS_READ(Bf[0][0], bh, 0);    APE_DS_READ(Bf[0][1], bl, 0);
    APE_DS_READ(Bf[1][0], bh, 2048); APE_DS_READ(Bf[1][1], bl, 2048);
    APE_DS_READ(Bf[2][0], bh, 4096); APE_DS_READ(Bf[2][1], bl, 4096);
    APE_DS_READ(Bf[3][0], bh, 6144); APE_DS_READ(Bf[3][1], bl, 6144);
}

// keeps asm-read destinations allocated up to this point (a free function: asm operands cannot name variables captured by a generic
// lambda; device pass only: the host pass cannot check a "v" constraint)
__device__ __forceinline__ void keep_regs(const u32x4& a, const u32x4& b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" :: "v"(a), "v"(b));
#endif
}

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the immediate must be a literal)
__device__ __forceinline__ void wait_vmcnt(int n)
{
#if defined(__HIP_DEVICE_COMPILE__)
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#endif
}

constexpr int TS = 16;                  // output tile edge
constexpr int HWP = 24;                 // pixels per ring row (3 DMA pieces); the halo needs 16 + 2 d <= 24 of them
constexpr int ROW_B = HWP * 128;        // 3072 B
constexpr int RING_ROWS = 32;
constexpr int A_BYTES = RING_ROWS * ROW_B;          // 96 KB
constexpr int B_TILE = 128 * 128;                   // one tap's weights: 128 rows x (hi 64 B | lo 64 B)
constexpr int B_RING = 4;
constexpr int LDS_BYTES = A_BYTES + B_RING * B_TILE;   // 160 KB

// PP ("ping-pong", round 6): the two waves of a SIMD (wave w and w + 4) run a tap's two segments in OPPOSITE order instead of in lockstep.
// PMC of the lockstep form (profiles/r05_pmc_utilisation.txt): per tap and SIMD 1440 cycles with BOTH waves issuing matrix instructions and
// ~730 with both in their reads / DMA issue / waits -- matrix pipe 0.64 busy.  Here a tap is, for every wave, a matrix segment C (its 48
// MFMAs from ONE fragment set, nothing else in the stream) and a load segment L (the 16 fragment reads of its next tap, its DMA pieces,
// the waits), and there is still ONE barrier per tap and wave -- but waves 0-3 close their interval behind L (C(t) L(t+1) |) and waves 4-7
// behind C (L(t) C(t) |): inside an interval one wave of every SIMD multiplies while the other loads, and neither waits for the other
// in between.  Same products, same order per accumulator, same DMA duty per tap as the lockstep form: bit-identical outputs.
// LDS protocol (interval t = what lies between the barriers t-1 and t): tap X's fragments are read by waves 0-3 at the end of interval
// X-1 and by waves 4-7 at the start of interval X; the duties of tap t (they overwrite rows / the weight slot last read for tap t-1)
// are issued at the end of interval t (waves 0-3) or at the start of interval t+1 (waves 4-7: their L(t+1) opens that interval), retired
// by the issuing wave's vmcnt(0) in interval t+1 (waves 0-3: in front of their next duties; waves 4-7: behind their C, in front of
// their barrier) and so published by barrier t+1; the first reads of what they bring are those of tap t+3, at the end of interval t+2
// -- every duty of the schedule is needed no earlier.
// (Built first with TWO barriers per tap -- a slot for C beside a slot for L, waves 4-7 one slot behind: -9.5 % over the seven layer shapes
// against the lockstep form; this form -11.5 %.  With the fragment reads moved into C, one pair per MFMA triple on a second register set:
// slower than either, tools/attic/halo_s32_pp2_reads_in_matrix_slot.patch.  A static priority for C: no change.)
template <int D, bool PP>
__global__ __launch_bounds__(512, 2) void halo_s32_kernel(const HaloS32Args a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int I = TS + 2 * D;       // image rows of one chunk
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // ---- the tile walk: logical tile ids are dealt to the XCDs in contiguous runs (bijective remap of the dispatch id); a persistent
    // workgroup takes dispatch ids blockIdx, + gridDim, ... -- all on its own XCD's run, and (the host sizes the grid so) all with the
    // same channel tile n_tile, so neighbouring workgroups keep sharing one spatial tile's halo through their XCD's L2
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const int nwg = a.B * tiles_per_img * a.n_tiles;
    const int q = nwg / 8, r = nwg % 8;
    const int grid = (int)gridDim.x;
    const int my_tiles = (nwg - 1 - (int)blockIdx.x) / grid + 1;
    int cur_b, cur_y0, cur_x0, n0;
    // dbg bit 8 (experiment): ONE channel tile per XCD -- XCD x multiplies channel tile x % n_tiles for its share of the
    // spatial tiles, so its 2.4 MB of weights stay in its L2 while every halo is fetched by n_tiles XCDs (results unchanged)
    const bool xcd_ntile = (a.dbg & 8) && (8 % a.n_tiles) == 0 && ((a.B * tiles_per_img) % (8 / a.n_tiles)) == 0 && r == 0;
    auto decode = [&](int orig, int& tb, int& ty0, int& tx0, int& tn0) {
        const int xcd = orig % 8;
        int n_tile, mt;
        if (xcd_ntile) {
            const int groups = 8 / a.n_tiles, per = a.B * tiles_per_img / groups;
            n_tile = xcd % a.n_tiles;
            mt = (xcd / a.n_tiles) * per + orig / 8;
        } else {
            const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
            n_tile = logical % a.n_tiles;
            mt = logical / a.n_tiles;
        }
        tb = mt / tiles_per_img;
        const int trem = mt - tb * tiles_per_img;
        ty0 = (trem / a.tiles_x) * TS;
        tx0 = (trem % a.tiles_x) * TS;
        tn0 = n_tile * 128;
    };
    int cur_orig = blockIdx.x;
    decode(cur_orig, cur_b, cur_y0, cur_x0, n0);
    // rows per pixel-row group of waves in a tile at image row ty0 (4, or fewer when the tile holds fewer than 13 / 9 / 5 image rows), times
    // this wave's group index: the tile row this wave's first output row is (dbg bit 16: always 4)
    auto rows_per_wave = [&](int ty0) { const int rows = a.H - ty0 < TS ? a.H - ty0 : TS; return (a.dbg & 16) ? 4 : (rows + 3) >> 2; };
    int rpw = rows_per_wave(cur_y0), rpw_next = rpw;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int nchunks = a.Cin / 32;
    const int K = 9 * a.Cin;
    const int total_chunks = my_tiles * nchunks;
    const int total_taps = total_chunks * 9;

    // ---- DMA sources ---------------------------------------------------------------------------------------------------------
    const long x_bytes = (long)a.B * a.H * a.W * a.ldx * 4;
    const long w_bytes = (long)(a.Cout - n0) * K * 4;
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)(unsigned)x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.w + (long)n0 * K * 4), 0, (int)(w_bytes > 0x7FFFFFFFL ? 0x7FFFFFFFu : (unsigned)w_bytes), 0x00020000);
    // the tile whose image is being streamed in ("DMA tile"): the current tile until its last chunk's tap 0, then the next one.
    // a halo piece = 8 pixels of one image row: lanes 8 j .. 8 j + 7 fetch pixel j's 128-B line, chunks permuted by the column swizzle
    int dma_b = cur_b, dma_y0 = cur_y0;
    unsigned lane_x[3];
    auto set_dma_tile = [&](int tb, int ty0, int tx0) {
        dma_b = tb;
        dma_y0 = ty0;
        int ln = lane;          // (opaque: the lane-only parts of the offsets are loop-invariant, and hoisted out of the tile loop they get spilled --
        asm volatile("" : "+v"(ln));       // a scratch reload per tile whose vmcnt(0) drains the DMA pipeline)
#pragma unroll
        for (int xp = 0; xp < 3; ++xp) {
            const int hx = xp * 8 + (ln >> 3);
            const int gx = tx0 - D + hx;
            const bool ok = hx < TS + 2 * D && (unsigned)gx < (unsigned)a.W;
            lane_x[xp] = ok ? (unsigned)((gx * a.ldx + a.xoff) * 4 + (((ln & 7) ^ ((hx >> 1) & 7)) * 16)) : 0x80000000u;
        }
    };
    set_dma_tile(cur_b, cur_y0, cur_x0);
    unsigned vb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 8 + (lane >> 3);
        vb[i] = (unsigned)(row * K * 4 + (((lane & 7) ^ ((row >> 1) & 7)) * 16));
    }
    // piece (row, x-piece xp) of channel chunk `c` of the DMA tile's image, written to ring row (ring0 + row) & 31
    auto dma_a_piece = [&](int c, int ring0, int row, int xp) {      // xp: a literal at every call site
        const int gy = dma_y0 - D + row;
        const bool row_ok = (unsigned)gy < (unsigned)a.H;
        const unsigned row_off = (unsigned)(((dma_b * a.H + (row_ok ? gy : 0)) * a.W) * a.ldx * 4 + c * 128);
        unsigned voff = lane_x[xp] + row_off;
        if (!row_ok) voff = 0xFFFFFFFFu;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_void*)(smem + ((ring0 + row) & (RING_ROWS - 1)) * ROW_B + xp * 1024), 16, voff, 0, 0, 0);
    };
    // The 3 * nrows pieces of image rows [r0, r0 + nrows) are dealt to the waves by (row * 3 + xp) % 8, i.e. for x-piece xp a wave
    // takes the rows  r = 3 (wave - xp) mod 8, + 8, ...  (3 is its own inverse mod 8).  xp stays a compile-time index: lane_x[] must
    // not be indexed at run time (it would live in scratch, and a scratch reload's vmcnt(0) would drain the DMA pipeline).
    auto dma_a_rows_xp = [&](int c, int ring0, int r0, int nrows, auto xp_c) -> int {
        constexpr int xp = decltype(xp_c)::value;
        int n = 0;
        for (int row = (3 * (wave - xp + 8)) & 7; row < nrows; row += 8) {
            dma_a_piece(c, ring0, r0 + row, xp);
            ++n;
        }
        return n;
    };
    // the weights of tap `tap` of channel chunk `c` (k-group tap * nchunks + c) into ring slot `slot`
    auto dma_b_tap = [&](int c, int tap, int slot) {
        const int g = tap * nchunks + c;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_void*)(smem + A_BYTES + slot * B_TILE + (wave * 2 + i) * 1024), 16,
                                                     vb[i], g * 128, 0, 0);
    };

    // ---- fragment addresses --------------------------------------------------------------------------------------------------
    const int frow = lane & 15, fc = lane >> 4;
    unsigned a_lane[3][2];              // [kx][plane]: byte offset inside a ring row of this lane's chunk of pixel column frow + kx * D
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int hx = frow + kx * D;
        const int sw = (hx >> 1) & 7;
        a_lane[kx][0] = (unsigned)(hx * 128 + ((fc ^ sw) * 16));
        a_lane[kx][1] = (unsigned)(hx * 128 + (((4 + fc) ^ sw) * 16));
    }
    const int swb = (frow >> 1) & 7;
    const unsigned b_lane[2] = {(unsigned)(A_BYTES + (wn * 64 + frow) * 128 + ((fc ^ swb) * 16)),
                                (unsigned)(A_BYTES + (wn * 64 + frow) * 128 + (((4 + fc) ^ swb) * 16))};
    u32x4 Ah[2][2][2], Bf[2][4][2];     // Ah[half][pixel row in the half][plane] (one tap's rows 0,1 | 2,3); Bf[set][channel block][plane]
    // read: pixel rows 2 half, 2 half + 1 of (image at ring0, tap) and weight blocks 2 half, 2 half + 1 of ring slot (tgb & 3) into B set `bset`
    auto read_half = [&](auto half_c, auto bset_c, int ring0, int tap, int tgb) {
        constexpr int half = decltype(half_c)::value, bset = decltype(bset_c)::value;
        const int ky = tap / 3, kx = tap - ky * 3;
        const unsigned ah = a_lane[kx][0], al = a_lane[kx][1];
        const int rb = ring0 + rpw * wm + ky * D + 2 * half;
        const unsigned r0 = (unsigned)(((rb + 0) & (RING_ROWS - 1)) * ROW_B), r1 = (unsigned)(((rb + 1) & (RING_ROWS - 1)) * ROW_B);
        const unsigned so = (unsigned)((tgb & (B_RING - 1)) * B_TILE);
        ds_read_half<2 * half>(Ah[half], Bf[bset], ah + r0, al + r0, ah + r1, al + r1, b_lane[0] + so, b_lane[1] + so);
    };
    f32x4 acc[4][4];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
    };
    zero_acc();
    // 12 MFMAs of pixel row i of one tap on register set `set`
    auto mfma_row = [&](auto set_c, auto ic) {
        constexpr int set = decltype(set_c)::value, i = decltype(ic)::value;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (ABL(1024)) continue;
            const bf16x8 bh = __builtin_bit_cast(bf16x8, Bf[set][j][0]), bl = __builtin_bit_cast(bf16x8, Bf[set][j][1]);
            const bf16x8 ah = __builtin_bit_cast(bf16x8, Ah[i >> 1][i & 1][0]), al = __builtin_bit_cast(bf16x8, Ah[i >> 1][i & 1][1]);
            // weights as the row operand (D[channel 4 fc + e][pixel frow]); product order of conv3x3_halo.hip
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, al, acc[i][j], 0, 0, 0);
            if (!ABL(32)) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, ah, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah, acc[i][j], 0, 0, 0);
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    auto phase_end = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        if (!ABL(64)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    // the asm reads' destinations stay allocated until the wait behind them (conv_gemm_s32.hip keep_a / keep_b: where hipcc finds such
    // outputs dead -- the look-ahead reads of a workgroup's last tap -- it re-uses their registers while the LDS data is on its way)
    auto keep_half = [&](auto half_c, auto bset_c) {
        constexpr int half = decltype(half_c)::value, bset = decltype(bset_c)::value;
        keep_regs(Ah[half][0][0], Ah[half][0][1]);
        keep_regs(Ah[half][1][0], Ah[half][1][1]);
#pragma unroll
        for (int j = 0; j < 4; ++j) keep_regs(Bf[bset][j][0], Bf[bset][j][1]);
    };

    u32x4 Af[4][2], Bs[4][2];           // PP: the ONE fragment set (pixel rows 0..3 x plane, channel blocks 0..3 x plane)
    auto read_all = [&](int ring, int tap, int rp, int tgb) {
        const int ky = tap / 3, kx = tap - ky * 3;
        const unsigned ah = a_lane[kx][0], al = a_lane[kx][1];
        const int rb = ring + rp * wm + ky * D;
        const unsigned so = (unsigned)((tgb & (B_RING - 1)) * B_TILE);
        const unsigned bh = b_lane[0] + so, bl = b_lane[1] + so;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned ro = (unsigned)(((rb + i) & (RING_ROWS - 1)) * ROW_B);
            ds_read_pair(Af[i], ah + ro, al + ro);
        }
        ds_read_b4(Bs, bh, bl);
    };
    auto keep_all = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) { keep_regs(Af[i][0], Af[i][1]); keep_regs(Bs[i][0], Bs[i][1]); }
    };
    // one wave of every SIMD pair at priority 1 for the whole kernel pays in the GEMM kernel (conv_gemm_s32.hip, -1.5 .. -3 %) but not
    // here (+-0 .. +1 % on all six layer shapes, tools/mb_halo_s32.py): off unless the debug bit asks for it
    if ((a.dbg & 1) && __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(1);
    // ---- prologue (a workgroup's first tile only): the whole first image and the first three weight tiles --------------------------
    dma_a_rows_xp(0, 0, 0, I, I0{});
    dma_a_rows_xp(0, 0, 0, I, I1{});
    dma_a_rows_xp(0, 0, 0, I, I2{});
    dma_b_tap(0, 0, 0);
    if (total_taps > 1) dma_b_tap(0, 1, 1);
    if (total_taps > 2) dma_b_tap(0, 2, 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const bool group_b = PP && __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256;       // waves 4-7: the second wave of every SIMD
    if constexpr (PP) {
        read_all(0, 0, rpw, 0);
        phase_end();
        keep_all();
        __builtin_amdgcn_sched_barrier(0);
    } else {
    read_half(I0{}, I0{}, 0, 0, 0);
    u32x4 dummy[2][2];
    {   // second half of tap 0's weights (its pixel rows 2, 3 are read during the tap's first phase)
        const unsigned a0 = a_lane[0][0];
        ds_read_half<2>(dummy, Bf[0], a0, a0, a0, a0, b_lane[0], b_lane[1]);
    }
    phase_end();
    keep_half(I0{}, I0{});
    keep_regs(dummy[0][0], dummy[0][1]);
    keep_regs(dummy[1][0], dummy[1][1]);
    }

    // ---- one tap = two phases of 24 MFMAs (pixel rows 0,1 | 2,3 against the four weight blocks of B set P).  While a phase runs, the
    // fragments of the NEXT phase are read: phase 0 fetches this tap's rows 2,3 and weight blocks 0,1 of tap t+1 (into the other B
    // set), phase 1 fetches rows 0,1 and weight blocks 2,3 of tap t+1 (across a chunk or tile seam: of the next image's tap 0).  The
    // DMA pieces (image rows [ar0, ar0 + arn) of chunk dc, ring position dring) and the weights of tap t+3 go out between the MFMA
    // rows.  The closing barrier makes the weights of tap t+2 (issued one tap ago) and every older piece visible: only what this tap
    // issued may still be in flight.
    //   tg = flattened tap index over all of this workgroup's chunks; (c, tap) = chunk in the tile, tap in the chunk
    auto tap_body = [&](auto set_c, int tg, int c, int tap, bool last_of_tile, int ring_cur, int ring_next, int dc, int dring, int ar0, int arn) __attribute__((always_inline)) {
        constexpr int P = decltype(set_c)::value;
        const bool wrap = tap == 8;
        const int nring = wrap ? ring_next : ring_cur, ntap = wrap ? 0 : tap + 1;
        const bool more_a = arn > 0 && !ABL(512);
        const bool more_b = tg + 3 < total_taps && !ABL(512);
        int issued = 0;
        // phase 0
        {
            const int ky = tap / 3, kx = tap - ky * 3;
            const unsigned ah = a_lane[kx][0], al = a_lane[kx][1];
            const int rb = ring_cur + rpw * wm + ky * D + 2;
            const unsigned r0 = (unsigned)(((rb + 0) & (RING_ROWS - 1)) * ROW_B), r1 = (unsigned)(((rb + 1) & (RING_ROWS - 1)) * ROW_B);
            const unsigned so = (unsigned)(((tg + 1) & (B_RING - 1)) * B_TILE);
            ds_read_half<0>(Ah[1], Bf[P ^ 1], ah + r0, al + r0, ah + r1, al + r1, b_lane[0] + so, b_lane[1] + so);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma_row(set_c, I0{});
        __builtin_amdgcn_sched_barrier(0);
        if (more_a) issued += dma_a_rows_xp(dc, dring, ar0, arn, I0{});
        __builtin_amdgcn_sched_barrier(0);
        mfma_row(set_c, I1{});
        __builtin_amdgcn_sched_barrier(0);
        if (more_a) issued += dma_a_rows_xp(dc, dring, ar0, arn, I1{});
        phase_end();
        keep_half(I1{}, std::integral_constant<int, P ^ 1>{});
        // phase 1
        {
            const int ky = ntap / 3, kx = ntap - ky * 3;
            const unsigned ah = a_lane[kx][0], al = a_lane[kx][1];
            const int rb = nring + ((wrap && last_of_tile) ? rpw_next : rpw) * wm + ky * D;      // (across a tile seam: the next tile's row of this wave)
            const unsigned r0 = (unsigned)(((rb + 0) & (RING_ROWS - 1)) * ROW_B), r1 = (unsigned)(((rb + 1) & (RING_ROWS - 1)) * ROW_B);
            const unsigned so = (unsigned)(((tg + 1) & (B_RING - 1)) * B_TILE);
            ds_read_half<2>(Ah[0], Bf[P ^ 1], ah + r0, al + r0, ah + r1, al + r1, b_lane[0] + so, b_lane[1] + so);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (rpw > 2) mfma_row(set_c, I2{});
        __builtin_amdgcn_sched_barrier(0);
        if (more_a) issued += dma_a_rows_xp(dc, dring, ar0, arn, I2{});
        __builtin_amdgcn_sched_barrier(0);
        if (rpw > 3) mfma_row(set_c, I3{});
        __builtin_amdgcn_sched_barrier(0);
        if (more_b) {
            const int t3 = tap + 3;          // (chunk, tap) of flattened tap tg + 3: the same weights for every tile of this workgroup
            const int c3 = t3 < 9 ? c : (last_of_tile ? 0 : c + 1);
            dma_b_tap(c3, t3 < 9 ? t3 : t3 - 9, (tg + 3) & (B_RING - 1));
            issued += 2;
        }
        phase_end();
        keep_half(I0{}, std::integral_constant<int, P ^ 1>{});
        if (!ABL(256)) wait_vmcnt(issued);
        if (!ABL(128)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };


    // ---- PP form of one tap (see the note at the kernel's head): C(t) | barrier | L(t+1) + the duties of tap t | barrier ------------
    auto mfma_all = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i >= rpw || ABL(1024) || (ABL(32) && i >= 2)) continue;             // (a ragged tile: rows past rpw do not exist; ABL 32: half the MFMAs)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bf16x8 bh = __builtin_bit_cast(bf16x8, Bs[j][0]), bl = __builtin_bit_cast(bf16x8, Bs[j][1]);
                const bf16x8 ah = __builtin_bit_cast(bf16x8, Af[i][0]), al = __builtin_bit_cast(bf16x8, Af[i][1]);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, al, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, ah, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah, acc[i][j], 0, 0, 0);
            }
        }
    };
    // the halo duty of one tap in the PP form: image rows [AR0, AR0 + NR) of chunk dc (NR <= 8 in every duty of the schedule, so a wave has at
    // most ONE row per x-piece: row roww[xp] = 3 (wave - xp) mod 8 if that is < NR -- the same dealing as dma_a_rows_xp).  Lean on purpose: a
    // wave issues one instruction per four cycles at best, and the L slot is the longer one -- the row part of the source address goes
    // through the instruction's SCALAR offset (no vector add), everything per tap is formed once for the three pieces.
    const unsigned rowstride = (unsigned)(a.W * a.ldx * 4);
    int roww[3];
    unsigned rsw[3];
#pragma unroll
    for (int xp = 0; xp < 3; ++xp) {
        roww[xp] = __builtin_amdgcn_readfirstlane((3 * (wave - xp + 8)) & 7);
        rsw[xp] = (unsigned)roww[xp] * rowstride;
    }
    auto duty_pp = [&](auto ar0_c, auto nr_c, bool en, int dc, int dring) __attribute__((always_inline)) {
        constexpr int AR0 = decltype(ar0_c)::value, NR = decltype(nr_c)::value;
        static_assert(NR <= 8, "one row per wave and x-piece");
        if constexpr (NR > 0) {
            if (en && !ABL(512)) {
                const int gy0 = dma_y0 - D + AR0;
                const unsigned s0 = (unsigned)(((dma_b * a.H + gy0) * a.W) * a.ldx * 4 + dc * 128);
                const int ring_b = dring + AR0;
                auto piece = [&](auto xp_c) __attribute__((always_inline)) {
                    constexpr int xp = decltype(xp_c)::value;
                    if (roww[xp] < NR) {
                        const bool ok = (unsigned)(gy0 + roww[xp]) < (unsigned)a.H;
                        const unsigned voff = ok ? lane_x[xp] : 0x80000000u;         // (a row outside the image: every lane out of range -> zeros)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_void*)(smem + ((ring_b + roww[xp]) & (RING_ROWS - 1)) * ROW_B + xp * 1024), 16,
                                                                 voff, s0 + rsw[xp], 0, 0);
                    }
                };
                piece(I0{});
                piece(I1{});
                piece(I2{});
            }
        }
    };
    auto tap_body_pp = [&](auto ar0_c, auto nr_c, bool en, int tg, int c, int tap, bool last_of_tile, int ring_cur, int ring_next, int dc, int dring) __attribute__((always_inline)) {
        const bool wrap = tap == 8;
        const int nring = wrap ? ring_next : ring_cur, ntap = wrap ? 0 : tap + 1;
        const bool more_b = tg + 3 < total_taps && !ABL(512);
        // C(t)
        mfma_all();
        __builtin_amdgcn_sched_barrier(0);
        // waves 4-7 close their interval here, behind C (waves 0-3: behind L, below).  They issued their pieces at the START of this interval and
        // retire them now: this barrier publishes them
        if (group_b) {
            if (!ABL(256)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!ABL(128)) __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_sched_barrier(0);
        // L(t+1): the fragments of the next tap (across a chunk or tile seam: of the next image's tap 0) ...
        if (!ABL(2048)) read_all(nring, ntap, (wrap && last_of_tile) ? rpw_next : rpw, tg + 1);
        __builtin_amdgcn_sched_barrier(0);
        // ... the pieces this wave issued in its previous L slot (a whole tap ago) have landed: the barrier below publishes them ...
        if (!group_b && !ABL(256)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // ... and this tap's DMA duties
        duty_pp(ar0_c, nr_c, en, dc, dring);
        if (more_b) {
            const int t3 = tap + 3;          // (chunk, tap) of flattened tap tg + 3: the same weights for every tile of this workgroup
            const int c3 = t3 < 9 ? c : (last_of_tile ? 0 : c + 1);
            dma_b_tap(c3, t3 < 9 ? t3 : t3 - 9, (tg + 3) & (B_RING - 1));
        }
        phase_end();
        keep_all();
        if (!group_b && !ABL(128)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- epilogue of the current tile straight from the registers: lane (frow, fc) holds channels 16 j + 4 fc .. + 3 of pixel
    // (4 wm + i, frow).  The whole residual tile is requested before the first use (16 loads in flight: one memory round trip, not one
    // per pixel row -- the stores of row i may alias the loads of row i + 1 for all the compiler knows); the stores are not waited
    // for here: the next tile's taps run while they drain (a tap's closing vmcnt wait covers them, they are older than its pieces).
    // One instance of the store loop per output format; the activation of the common layers is a select, not a switch.
    auto epilogue = [&]() __attribute__((always_inline)) {
        // opaque copies of the lane coordinates: the epilogue's 64-bit address parts must be formed HERE -- hoisted in front of the tile loop
        // (they are loop-invariant) they are spilled around the 250-register tap bodies and reloaded from scratch here
        int frow_e = frow, fc_e = fc;
        asm volatile("" : "+v"(frow_e), "+v"(fc_e));
        const int nq = n0 + wn * 64 + fc_e * 4;
        const float* bp = a.bias ? a.bias + (a.bias_bstride ? (size_t)cur_b * a.bias_bstride : 0) : nullptr;
        const int gx = cur_x0 + frow_e;
        // 16 residual bytes per (pixel row i, channel block j), fetched as two 8-byte halves for BOTH formats (same instruction stream):
        // S32: hi 4 x bf16 | lo 4 x bf16 (64 B apart); fp32: floats 0,1 | floats 2,3 (8 B apart)
        uint2 rlo[4][4], rhi[4][4];
        if (a.res && !(ABL(8))) {
            const bool rs32 = a.res_fmt == APE_FMT_S32;
            const long second = rs32 ? 64 : 8;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int gy = cur_y0 + rpw * wm + i;
                const bool pix_ok = i < rpw && gy < a.H && gx < a.W;
                const size_t m = ((size_t)cur_b * a.H + (pix_ok ? gy : 0)) * a.W + (pix_ok ? gx : 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = nq + j * 16;
                    const int cr = a.roff + (n < a.Cout ? n : 0);       // (a clamped, never-used address for channels past Cout)
                    const char* rp = a.res + (rs32 ? m * a.ldr * 4 + (size_t)((cr >> 5) * 128 + (cr & 31) * 2) : (m * a.ldr + cr) * 4);
                    rlo[i][j] = *reinterpret_cast<const uint2*>(rp);
                    rhi[i][j] = *reinterpret_cast<const uint2*>(rp + second);
                }
            }
        }
        float4 b4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = nq + j * 16;
            b4[j] = (bp && n < a.Cout) ? *reinterpret_cast<const float4*>(bp + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const ape::ActFast af = ape::act_fast_make(a.act, a.alpha);        // (common.h: no scalar compare-and-branch cascade per element)
        const bool sigmoid = a.act == APE_ACT_SIGMOID;
#if APE_S32_WIDE_STORE
        const bool wide = a.Cout % 8 == 0 && a.yoff % 8 == 0;              // (a pair of lanes = two adjacent channel quads: both inside Cout or both outside)
#endif
        auto store_tile = [&](auto s32_c) __attribute__((always_inline)) {
            constexpr bool OUT_S32 = decltype(s32_c)::value != 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int gy = cur_y0 + rpw * wm + i;
                if (i >= rpw || gy >= a.H || gx >= a.W) continue;
                const size_t m = ((size_t)cur_b * a.H + gy) * a.W + gx;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = nq + j * 16;
                    if (n >= a.Cout || (ABL(4))) continue;
                    float vv[4] = {acc[i][j][0] + b4[j].x, acc[i][j][1] + b4[j].y, acc[i][j][2] + b4[j].z, acc[i][j][3] + b4[j].w};
                    if (a.res && !(ABL(8))) {
                        if (a.res_fmt == APE_FMT_S32) {
                            const bf16x4 h = __builtin_bit_cast(bf16x4, rlo[i][j]), l = __builtin_bit_cast(bf16x4, rhi[i][j]);
#pragma unroll
                            for (int e = 0; e < 4; ++e) vv[e] += (float)h[e] + (float)l[e];
                        } else {
                            vv[0] += __uint_as_float(rlo[i][j].x);
                            vv[1] += __uint_as_float(rlo[i][j].y);
                            vv[2] += __uint_as_float(rhi[i][j].x);
                            vv[3] += __uint_as_float(rhi[i][j].y);
                        }
                    }
                    if (sigmoid) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) vv[e] = 1.f / (1.f + __expf(-vv[e]));
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) vv[e] = ape::act_fast(vv[e], af);
                    }
                    if (OUT_S32) {
                        const int cy = a.yoff + n;
                        char* yp = a.y + m * a.ldy * 4 + (cy >> 5) * 128 + (cy & 31) * 2;
                        bf16x4 h, l;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { h[e] = (__bf16)vv[e]; l[e] = (__bf16)(vv[e] - (float)h[e]); }
#if APE_S32_WIDE_STORE
                        if (wide) {
                            // ONE 16-byte store per lane (cf. conv_gemm_s32.hip): the lanes fc and fc ^ 1 of a pixel hold adjacent channel quads;
                            // after the swap of the odd 16-lane rows of `hi` with the even rows of `lo` an even-fc lane owns the hi halves of both
                            // quads (16 contiguous bytes at its own hi address), the odd-fc lane both lo halves (at its own hi address - 8 + 64)
                            uint2 hu = __builtin_bit_cast(uint2, h), lu = __builtin_bit_cast(uint2, l);
                            const auto r0 = __builtin_amdgcn_permlane16_swap(hu.x, lu.x, false, false);
                            const auto r1 = __builtin_amdgcn_permlane16_swap(hu.y, lu.y, false, false);
                            *reinterpret_cast<uint4*>((fc_e & 1) ? yp + 56 : yp) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
                        } else
#endif
                        {
                            *reinterpret_cast<bf16x4*>(yp) = h;
                            *reinterpret_cast<bf16x4*>(yp + 64) = l;
                        }
                    } else {
                        *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.y) + m * a.ldy + a.yoff + n) = make_float4(vv[0], vv[1], vv[2], vv[3]);
                    }
                }
            }
        };
        if (a.out_fmt == APE_FMT_S32) store_tile(I1{}); else store_tile(I0{});
    };

    // ---- one channel chunk = nine taps on alternating register sets (P0 = the set its tap 0 multiplies); gc = flattened chunk index.
    int c = 0;                  // chunk inside the current tile
    bool tile_done = false;     // the chunk just multiplied was its tile's last: the epilogue follows at the end of the iteration
    auto chunk = [&](auto p0_c, int gc, int ring_cur, int ring_next) __attribute__((always_inline)) {
        constexpr int P0 = decltype(p0_c)::value;
        using PA = std::integral_constant<int, P0>;
        using PB = std::integral_constant<int, P0 ^ 1>;
        const bool last_of_tile = c + 1 == nchunks;
        const bool has_next = gc + 1 < total_chunks;       // a further chunk (of this tile or of the workgroup's next tile) follows
        const int tg = gc * 9;
        // tap 0: the last 2 D rows of THIS chunk's image (skipped for the workgroup's very first chunk: the prologue loaded all of it)
        // (PP: the duty's row range is a compile-time pair, its run-time part is the enable)
        auto tap_any = [&](auto set_c, auto ar0_c, auto nr_c, bool en, int tg_, int tap_, int dc_, int dring_) __attribute__((always_inline)) {
            if constexpr (PP) tap_body_pp(ar0_c, nr_c, en, tg_, c, tap_, last_of_tile, ring_cur, ring_next, dc_, dring_);
            else tap_body(set_c, tg_, c, tap_, last_of_tile, ring_cur, ring_next, dc_, dring_, decltype(ar0_c)::value, en ? decltype(nr_c)::value : 0);
        };
        constexpr int FREE = RING_ROWS - I;                // rows of the next image that land on ring rows no image uses now
        constexpr int H1 = (FREE + 1) / 2;
        using Z = std::integral_constant<int, 0>;
        tap_any(PA{}, std::integral_constant<int, I - 2 * D>{}, std::integral_constant<int, 2 * D>{}, gc > 0, tg + 0, 0, c, ring_cur);
        // from here on the DMA side works on the next chunk's image: chunk c + 1 of this tile, or chunk 0 of the workgroup's next tile
        int dc = c + 1;
        if (last_of_tile) {
            dc = 0;
            if (has_next) {
                int tb, ty0, tx0, tn0;
                decode(cur_orig + grid, tb, ty0, tx0, tn0);
                set_dma_tile(tb, ty0, tx0);
                rpw_next = rows_per_wave(ty0);
            }
        }
        tap_any(PB{}, Z{}, std::integral_constant<int, H1>{}, has_next, tg + 1, 1, dc, ring_next);
        tap_any(PA{}, std::integral_constant<int, H1>{}, std::integral_constant<int, FREE - H1>{}, has_next, tg + 2, 2, dc, ring_next);
        tap_any(PB{}, std::integral_constant<int, FREE>{}, std::integral_constant<int, D>{}, has_next, tg + 3, 3, dc, ring_next);
        tap_any(PA{}, Z{}, Z{}, false, tg + 4, 4, dc, ring_next);
        tap_any(PB{}, Z{}, Z{}, false, tg + 5, 5, dc, ring_next);
        tap_any(PA{}, std::integral_constant<int, FREE + D>{}, std::integral_constant<int, D>{}, has_next, tg + 6, 6, dc, ring_next);
        tap_any(PB{}, Z{}, Z{}, false, tg + 7, 7, dc, ring_next);
        tap_any(PA{}, Z{}, Z{}, false, tg + 8, 8, dc, ring_next);
        tile_done = last_of_tile;
        c = last_of_tile ? 0 : c + 1;
    };
    int ring = 0;       // ring row of the current chunk's image row 0
#pragma unroll 1
    for (int gc = 0; gc < total_chunks; gc += 2) {
        // two chunks per iteration: 18 taps, so the register sets alternate with a compile-time parity.  A tile ends only at the end of
        // an iteration: the host walks several tiles per workgroup only for an even chunk count (Cin % 64 == 0); with an odd count
        // every workgroup has ONE tile, which ends with the iteration's first chunk, the second being skipped.
        const int ring1 = (ring + I) & (RING_ROWS - 1), ring2 = (ring1 + I) & (RING_ROWS - 1);
        chunk(I0{}, gc, ring, ring1);
        if (gc + 1 < total_chunks) chunk(I1{}, gc + 1, ring1, ring2);
        ring = ring2;
        if (tile_done) {
            epilogue();
            zero_acc();
            cur_orig += grid;
            if (gc + 2 < total_chunks) { int tn0; decode(cur_orig, cur_b, cur_y0, cur_x0, tn0); rpw = rows_per_wave(cur_y0); }
        }
    }
#endif
}
#undef APE_DS_READ

template <int D, bool PP>
int launch_halo_s32(const HaloS32Args& a, hipStream_t st)
{
    auto kern = halo_s32_kernel<D, PP>;
    static ape::DeviceOnce once;       // (per kernel instantiation)
    int ncu = 256;
    if (int rc = ape::device_once(once, reinterpret_cast<const void*>(kern), LDS_BYTES, &ncu)) return rc;
    // one workgroup fills a CU (160 KB of LDS, 8 waves x 256 registers): launch one per CU and let each walk its share of the tiles.
    // The walk keeps a workgroup on ONE channel tile (its weight descriptor is built once) when the tile stride grid / 8 is a multiple
    // of n_tiles; otherwise, and for grids smaller than the chip, every tile gets its own workgroup as before.
    const int nwg = a.B * a.tiles_x * a.tiles_y * a.n_tiles;
    int grid = nwg;
    const int unit = 8 * a.n_tiles;
    if (!(a.dbg & 2) && nwg > ncu && ncu >= unit && (a.Cin / 32) % 2 == 0) grid = (ncu / unit) * unit;    // (kernel: a tile ends on an even chunk)
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS_BYTES, st, a);
    return ape::check_launch("ape_conv3x3_halo_s32");
}

bool halo_s32_supported(const ape_conv_params& p)
{
    if (p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != p.dil || (p.dil != 1 && p.dil != 2 && p.dil != 4) || p.ups != 0) return false;
    if (p.B < 0 || p.H < 1 || p.W < 1 || p.Ho != p.H || p.Wo != p.W) return false;
    if (p.Cin < 32 || p.Cin % 32 || p.ldx % 32 || p.xoff % 32 || p.xoff + p.Cin > p.ldx) return false;
    if (p.Cout < 128 || p.Cout % 4 || p.yoff + p.Cout > p.ldy || p.ldy % 4 || p.yoff % 4) return false;
    if (p.act < APE_ACT_NONE || p.act > APE_ACT_SIGMOID) return false;
    if ((long)p.B * p.H * p.W * p.ldx * 4 >= (1L << 31) || 128L * 9 * p.Cin * 4 >= (1L << 31)) return false;
    return true;
}

}  // namespace

static int g_halo_s32_dbg = 0;
extern "C" int ape_conv3x3_halo_s32_debug(int bits) { g_halo_s32_dbg = bits; return APE_OK; }

extern "C" int ape_conv3x3_halo_s32_supported(const ape_conv_params* params) { return params && halo_s32_supported(*params) ? 1 : 0; }

extern "C" int ape_conv3x3_halo_s32(const void* x_s32, const void* w_s32k, const float* bias, const void* residual, int res_fmt, void* y,
                                    int out_fmt, const ape_conv_params* params, void* stream)
{
    if (!x_s32 || !w_s32k || !y || !params) return APE_EINVAL;
    const ape_conv_params& p = *params;
    if (!halo_s32_supported(p)) return APE_EINVAL;
    if ((out_fmt != APE_FMT_F32 && out_fmt != APE_FMT_S32) || (residual && res_fmt != APE_FMT_F32 && res_fmt != APE_FMT_S32)) return APE_EINVAL;
    if (out_fmt == APE_FMT_S32 && p.ldy % 32) return APE_EINVAL;
    if (residual && (p.roff + p.Cout > p.ldr || p.ldr % 4 || p.roff % 4 || (res_fmt == APE_FMT_S32 && p.ldr % 32))) return APE_EINVAL;
    if (p.B == 0) return APE_OK;
    HaloS32Args a;
    a.dbg = g_halo_s32_dbg;
    a.x = (const char*)x_s32; a.w = (const char*)w_s32k; a.bias = bias; a.res = (const char*)residual; a.y = (char*)y;
    a.B = p.B; a.H = p.H; a.W = p.W; a.Cin = p.Cin; a.Cout = p.Cout;
    a.ldx = p.ldx; a.xoff = p.xoff; a.ldy = p.ldy; a.yoff = p.yoff; a.ldr = p.ldr; a.roff = p.roff;
    a.act = p.act; a.alpha = p.alpha; a.bias_bstride = p.bias_bstride; a.out_fmt = out_fmt; a.res_fmt = res_fmt;
    a.tiles_x = ape::ceil_div(p.W, TS); a.tiles_y = ape::ceil_div(p.H, TS); a.n_tiles = ape::ceil_div(p.Cout, 128);
    hipStream_t st = (hipStream_t)stream;
    if (g_halo_s32_dbg & 4096) {      // the one-barrier-per-tap form of rounds 2-5 (A/B: tools/mb_halo_pp.py; bit-identical outputs)
        if (p.dil == 1) return launch_halo_s32<1, false>(a, st);
        if (p.dil == 2) return launch_halo_s32<2, false>(a, st);
        return launch_halo_s32<4, false>(a, st);
    }
    if (p.dil == 1) return launch_halo_s32<1, true>(a, st);
    if (p.dil == 2) return launch_halo_s32<2, true>(a, st);
    return launch_halo_s32<4, true>(a, st);
}
