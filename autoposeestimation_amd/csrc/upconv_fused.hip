// PSPUpsample with 64 output channels (DenseFusion/lib/pspnet.py:27-37: nn.Upsample(x2, bilinear, align_corners=True) -> Conv2d 3x3 pad 1
// -> PReLU; the segmentor's up_3, pspnet.py:51) as ONE kernel, optionally followed by the segmentation head (final 1x1 conv rows + softmax
// (+ softmax) + arg-max, pspnet.py:53-55, pipeline/utils.py:429-435) in the same launch.
//
// The direct form (conv3x3_halo.hip with the fused up-sampling) multiplies at HIGH resolution: 2 * 9 * Cin * 64 flop per output pixel.
// Conv and bilinear resize are both linear, so the channel mixing can run at LOW resolution -- z[p][tap][co] = sum_ci W[co][ci][tap] x[p][ci],
// 4x fewer flops -- and the output is  out(Y, X) = act(bias + sum_{ky,kx} lerp2d(z_{ky,kx})(Y + ky - 1, X + kx - 1))  (engine.UpConv:
// ape_conv_gemm_s32 + ape_upconv3x3_gather_f32 for up_1 / up_2).  For up_3 the tap tensor z would be 9 * 64 channels at 240 x 320:
// 11 GB per 64 frames.  Here it never leaves the CU:
//   * a workgroup (12 waves) owns a 16 x 24 output tile = the 10 x 16 low-resolution pixels under it (with the floor pattern of the
//     align_corners source index, iy0(q) = (q - 1) >> 1, the 18 up-sampled rows Y0 - 1 .. Y0 + 16 touch exactly ten low-resolution rows
//     and the 26 columns at most fourteen); their S32 rows are LDS-DMA'd once (XT, swizzled like conv3x3_halo_s32.hip's image);
//   * wave (kx, slab) multiplies the 160 pixels with the weights of the three taps (ky = 0..2, kx) of its 16 output channels:
//     v_mfma_f32_16x16x32_bf16, weights as the row operand, split-bf16 products in ape_conv_gemm_s32's order -- the accumulators ARE
//     z, bit for bit.  A lane holds ONE low-resolution column (lane & 15) and four channels in all ten rows, so the ROW interpolation and
//     the sum over ky  S_kx(Y, ix) = sum_ky lerp_y(z_{ky,kx}(., ix), Y + ky - 1)  is register arithmetic on the accumulators with
//     wave-uniform weights (static register indices: slot (t >> 1), (t >> 1) + 1 for t = y + ky);
//   * S goes to LDS (12 waves x 8 rows x 1 KB, the tile's top and bottom half in turn), and each wave then finishes one group of 16
//     output pixels x 64 channels per half: column interpolation + sum over kx from LDS, bias, activation, and either the store or the
//     head's sixteen exact-fp32 matrix instructions (seg_head.h, in seg_head_group's order) + soft-max + arg-max.
// Arithmetic = ape_conv_gemm_s32 followed by ape_upconv3x3_gather_ex(fma) operation for operation (tests/test_gpu_upfuse.py compares
// bit for bit); FMA picks the chained fused-multiply-add form of both interpolations.
// Persistent workgroups (one per CU) walk the tiles in an XCD-contiguous order; the next tile's pixels are requested as soon as the
// current tile's matrix phase is over and land during its gather phases.
#include <type_traits>
#include "common.h"
#include "s32.h"
#include "seg_head.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

// diagnostic build only (make stamps -> libape_hip_stamps.so, tools/stamp_upfuse.py): s_memtime sums per (workgroup, wave) and segment of the tile loop
#ifdef APE_UPFUSE_STAMPS
#define STAMP(i)                                                                                              \
    do {                                                                                                      \
        unsigned long long t_;                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                            \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        st_sum[i] += t_ - st_last;                                                                            \
        st_last = t_;                                                                                         \
    } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

// timing-only ablation switches (results wrong when set): compiled into the diagnostic build only
#ifdef APE_UPFUSE_ABLATIONS
#define ABL(bit) (a.dbg & (bit))
#else
#define ABL(bit) 0
#endif

struct UpFuseArgs {
    const char* x;          // S32 activations [B][h][w][G * 32]
    const char* w;          // S32K weights [9 * 64][G][hi 32 | lo 32], row = tap * 64 + co
    const float* bias;      // [64] or null
    char* y;                // !HEAD: out[B][2h][2w][64] in out_fmt
    const float* head_w;    // HEAD: [C][64]
    const float* head_b;    // HEAD: [C] or null
    uint8_t* label;         // HEAD: [B][2h][2w]
    float* score;
    int B, h, w_;
    int act;
    float alpha;
    float sh, sw;
    int out_fmt, head_c, head_dsm;
    int tiles_x, tiles_y;
    unsigned long long* stamps;     // diagnostic build: [workgroup][wave][16] cycle sums (null otherwise)
    int dbg;                // ape_upconv3x3_fused_debug: timing-only ablations (WRONG results): 1 no matrix instructions, 2 no row interpolation / S stores,
                            // 4 no gather phases, 8 no head (matrix instructions + soft-max / arg-max), 16 no pixel DMA, 32 no weight loads, 64 no S reads in the gather
};

constexpr int TY = 16, TX = 24;         // output tile
constexpr int COLS = 16;                  // low-resolution pixels under it (slot r <-> row clamp(Y0 / 2 - 1 + r), slot s <-> column X0 / 2 - 1 + s)
constexpr int RA = 6, RB = 4;           // ... held as two row groups: slots 0..5 (the tile's top half needs exactly these) and 6..9
constexpr int NW = 12;                  // waves: (kx = wave % 3, slab = wave / 3)
constexpr int SB_BYTES = NW * 8 * 1024; // S_kx rows of half a tile: [kx][slab][y & 7][channel quad 4][column slot 16] float4 (a wave stores a quad's 16 slots as 256
                                        // contiguous bytes; the 16 lanes of a gather read hit 8 consecutive slots: no bank conflict either way)
constexpr int HWS = 68;                 // floats per class row of the head weights in LDS (272 B: sixteen classes hit sixteen different bank slots)
constexpr int HW_BYTES = 16 * HWS * 4, HB_BYTES = 64, CB_BYTES = 256;
constexpr int xa_bytes(int G) { return G * RA * COLS * 128; }
constexpr int xb_bytes(int G) { return G * RB * COLS * 128; }
constexpr int lds_bytes(int G) { return xa_bytes(G) + xb_bytes(G) + SB_BYTES + HW_BYTES + HB_BYTES + CB_BYTES; }

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// Both interpolations are built from the hand-written two-channel instructions below (v_pk_fma_f32 / v_pk_mul_f32 with an op_sel weight
// broadcast); APE_NO_ASM_MATH = 1 writes the same arithmetic in C (hipcc's SLP vectoriser then emits packed instructions of its own, with two
// v_mov per broadcast in places: 1.5 % slower at bench size, same bits).
//
// THE BROADCAST OPERAND MUST BE src0.  Rounds 1-4 carried an intermittent fault here (and, with the same signature, in round 1's fused
// up-sampling of conv3x3_halo.hip): with the weight pair as the SECOND source -- `v_pk_fma_f32 d, data, w, d op_sel:[0,1,0]` -- the LOW half of
// the result is occasionally computed from a wrong weight in lanes 48-63.  Round 5 isolated it on the GPU (tools/stress_upfuse.py,
// `make variant`): only the column interpolation (per-lane weights in a VGPR pair; the row interpolation's SGPR pairs are fine), only its
// top-half instance, which a wave enters right after issuing its two LDS-DMA pieces, only waves 8-11 (the third wave of every SIMD: last at
// the barrier, so it runs alone and issues back to back), only j = 3 / lanes 48-63 / the low half of a pair; 8 wait states in front of AND
// behind every statement hide it, pads on one side do not; no matrix instruction is within 16 wait states of any statement
// (tools/isa_audit.mfma_asm_hazards) and every LDS / vector-memory wait is in place.  What decides it is the operand slot alone: the same
// instruction with the swizzled pair as src0 (`v_pk_fma_f32 d, w, data, d op_sel:[1,0,0]`, the form hipcc itself emits for every broadcast)
// gave 0 wrong pixels in 600 stress launches where the src1 form fails in EVERY launch (2 x 240 x 320, non-fma head build: 10^3..10^4 pixels),
// and without any op_sel (weights broadcast by v_mov) both operand orders are clean.  So: an op_sel / op_sel_hi broadcast of a VGPR pair in
// the src1 slot of a packed-f32 instruction is not safe on gfx950 under back-to-back issue -- no LLVM hazard entry, no ISA-manual rule; kept
// out of the tree by tests/test_isa_waits.py (tools/isa_audit.pk_src1_swizzles).  APE_ASM_SRC1_BCAST = 1 rebuilds the faulty form (the
// audit's red case and tools/stress_upfuse.py's positive control; never the product).
#ifndef APE_NO_ASM_MATH
#define APE_NO_ASM_MATH 0
#endif
#ifndef APE_ASM_SRC1_BCAST
#define APE_ASM_SRC1_BCAST 0
#endif
#ifndef APE_NO_FAST_ROWS
#define APE_NO_FAST_ROWS 0
#endif
#ifndef APE_NO_FAST_COLS
#define APE_NO_FAST_COLS 0
#endif

// One LDS-DMA piece (64 lanes x 16 B -> 1 KB at LDS byte address lds_addr) as inline assembly: issued through the builtin, hipcc knows of a
// pending LDS write and puts `s_waitcnt vmcnt(0)` in front of the next LDS read it schedules -- here the first S read of the gather phase
// that the transfer is meant to run under (it cannot see the hand-placed waits + barriers that order the transfer against its real
// readers).  M0 (compiler-reserved) is saved and restored inside the statement.
__device__ __forceinline__ void dma_piece(unsigned voff, const i32x4& rs, unsigned lds_addr)
{
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned keep;
    // (s_nop 4 first: hipcc may reload a spilled SGPR operand -- v_readlane, a VALU write of an SGPR -- right in front of the statement, and a
    // vector-memory instruction that reads such an SGPR is owed five wait states, which hipcc pays for its own instructions only.  Found in the
    // streamed experiment of this kernel (branch exp/upfuse-streamed), where a weight read took a stale offset that way, deterministically.)
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds_addr) : "memory");
#endif
}

__device__ __forceinline__ float lane_value(float v, int l)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

// Two channels per instruction with ONE interpolation weight for both: v_pk_fma_f32 / v_pk_mul_f32 with op_sel picking the low (lo) or
// high (hi) dword of the 64-bit weight operand w = {l0, l1} for BOTH halves.  hipcc builds such a broadcast operand with two v_mov per
// use; the kernel is vector-issue-bound, so the moves cost as much as the arithmetic.  Two independent IEEE operations per instruction:
// bit for bit the scalar fmaf / multiply.  WC = "v" (a VGPR pair, the gather's per-lane weights) or "s" (an SGPR pair, the row weights).
// The statements are VOLATILE on purpose: hipcc's hazard recognizer does not look inside an asm statement, and a register-only asm is free
// to move above the `s_barrier` statement in front of it -- right behind the matrix instruction that produces its operand, inside the wait
// states software owes between an MFMA's write and a vector read.  Volatile statements keep their order against the barrier statements,
// which puts dozens of instructions between the last matrix instruction and the first read (checked per build by
// tools/isa_audit.mfma_asm_hazards).
#if defined(__HIP_DEVICE_COMPILE__)
#if APE_ASM_SRC1_BCAST
// (the historic, FAULTY operand order: data src0, weight pair src1 -- see the note at APE_NO_ASM_MATH)
#define APE_PK_OPS(SUF, WC, WT, VOL)                                                                                                       \
    __device__ __forceinline__ void pk_fma0_lo_##SUF(f32x2& d, const f32x2 v, const WT w)                                                  \
    { asm volatile("v_pk_fma_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(v), WC(w)); }                                           \
    __device__ __forceinline__ void pk_fma_lo_##SUF(f32x2& d, const f32x2 v, const WT w)                                                   \
    { asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(d) : "v"(v), WC(w)); }                                          \
    __device__ __forceinline__ void pk_fma_hi_##SUF(f32x2& d, const f32x2 v, const WT w)                                                   \
    { asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(d) : "v"(v), WC(w)); }                           \
    __device__ __forceinline__ f32x2 pk_mul_lo_##SUF(const f32x2 v, const WT w)                                                            \
    { f32x2 d; asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(v), WC(w)); return d; }                             \
    __device__ __forceinline__ f32x2 pk_mul_hi_##SUF(const f32x2 v, const WT w)                                                            \
    { f32x2 d; asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(d) : "v"(v), WC(w)); return d; }
#else
// the weight pair is src0, the operand slot hipcc itself uses for a broadcast (products commute: bit for bit the same results).  VOL: the row
// interpolation's statements (SGPR weights, operands = the MFMA accumulators) are volatile, see above; the column interpolation's are NOT --
// their data operands come from LDS loads behind the barrier statement's memory clobber, so they cannot move in front of it, and a volatile
// statement is a scheduling barrier: hipcc then cannot issue the next taps' ds_reads while the current ones multiply (two loads in flight
// instead of six).
#define APE_PK_OPS(SUF, WC, WT, VOL)                                                                                                       \
    __device__ __forceinline__ void pk_fma0_lo_##SUF(f32x2& d, const f32x2 v, const WT w)                                                  \
    { asm VOL("v_pk_fma_f32 %0, %2, %1, 0 op_sel_hi:[0,1,0]" : "=v"(d) : "v"(v), WC(w)); }                                                \
    __device__ __forceinline__ void pk_fma_lo_##SUF(f32x2& d, const f32x2 v, const WT w)                                                   \
    { asm VOL("v_pk_fma_f32 %0, %2, %1, %0 op_sel_hi:[0,1,1]" : "+v"(d) : "v"(v), WC(w)); }                                               \
    __device__ __forceinline__ void pk_fma_hi_##SUF(f32x2& d, const f32x2 v, const WT w)                                                   \
    { asm VOL("v_pk_fma_f32 %0, %2, %1, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(d) : "v"(v), WC(w)); }                                \
    __device__ __forceinline__ f32x2 pk_mul_lo_##SUF(const f32x2 v, const WT w)                                                            \
    { f32x2 d; asm VOL("v_pk_mul_f32 %0, %2, %1 op_sel_hi:[0,1]" : "=v"(d) : "v"(v), WC(w)); return d; }                                  \
    __device__ __forceinline__ f32x2 pk_mul_hi_##SUF(const f32x2 v, const WT w)                                                            \
    { f32x2 d; asm VOL("v_pk_mul_f32 %0, %2, %1 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(d) : "v"(v), WC(w)); return d; }
#endif
#ifndef APE_COLS_VOLATILE
#define APE_COLS_VOLATILE
#endif
APE_PK_OPS(v, "v", f32x2, APE_COLS_VOLATILE)
APE_PK_OPS(s, "s", unsigned long long, volatile)
#undef APE_PK_OPS
// acc (+)= l0 * v0 + l1 * v1 for two channels, in the unfused kernel's arithmetic: FMA: fma(l1, v1, fma(l0, v0, acc)); else acc + (l0 * v0 + l1 * v1).
// FIRST: acc is the zero the unfused kernel starts from
#define APE_LERP_TERM(SUF, WT)                                                                                                             \
    template <bool FMA, bool FIRST>                                                                                                        \
    __device__ __forceinline__ void lerp_term_##SUF(f32x2& acc, const f32x2 v0, const f32x2 v1, const WT w)                                \
    {                                                                                                                                      \
        if (APE_NO_ASM_MATH) {                                                                                                             \
            const f32x2 wf = __builtin_bit_cast(f32x2, w);                                                                                 \
            for (int e = 0; e < 2; ++e) {                                                                                                  \
                const float a0 = FIRST ? 0.f : acc[e];                                                                                     \
                if (FMA) acc[e] = __builtin_fmaf(wf[1], v1[e], __builtin_fmaf(wf[0], v0[e], a0));                                           \
                else acc[e] = a0 + (wf[0] * v0[e] + wf[1] * v1[e]);                                                                         \
            }                                                                                                                              \
        } else if (FMA) {                                                                                                                  \
            if (FIRST) pk_fma0_lo_##SUF(acc, v0, w); else pk_fma_lo_##SUF(acc, v0, w);                                                     \
            pk_fma_hi_##SUF(acc, v1, w);                                                                                                   \
        } else {                                                                                                                           \
            const f32x2 t = pk_mul_lo_##SUF(v0, w) + pk_mul_hi_##SUF(v1, w);                                                               \
            acc = FIRST ? f32x2{0.f, 0.f} + t : acc + t;                                                                                   \
        }                                                                                                                                  \
    }
APE_LERP_TERM(v, f32x2)
APE_LERP_TERM(s, unsigned long long)
#undef APE_LERP_TERM
#endif

__device__ __forceinline__ f32x2 lo2(const f32x4& v) { return __builtin_shufflevector(v, v, 0, 1); }
__device__ __forceinline__ f32x2 hi2(const f32x4& v) { return __builtin_shufflevector(v, v, 2, 3); }

// One workgroup per CU walks its tiles through a two-stage pipeline; per tile
//     row interpolation, top half (accumulator rows 0..5) -> S | barrier | [next tile's rows 0..5 requested]
//     gather top half (S -> pixels -> head) ; matrix instructions of rows 6..9 | barrier
//     row interpolation, bottom half (rows 4..9) -> S | barrier | [next tile's rows 6..9 requested]
//     gather bottom half ; matrix instructions of the NEXT tile's rows 0..5 | barrier
// so a gather half (vector + LDS work, long dependency chains) and a matrix half sit in the same barrier interval and the waves drift
// from one into the other, and every pixel request has a whole interval to land.  The weights of a wave (three taps x two channel groups
// x hi | lo = 48 registers) stay in registers for the whole kernel.  The kernel is bound by vector-instruction issue (about four cycles
// per wave-instruction and SIMD), so both interpolations are written two channels per instruction (above) and the common case -- a tile
// whose taps all lie inside the image -- runs without per-term validity tests and with per-lane LDS offsets computed once per kernel.
template <int G, bool HEAD, bool FMA>
__global__ __launch_bounds__(NW * 64) void upconv_fused_kernel(const UpFuseArgs a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS: the small constant tables first (every access then is one lane base + an immediate below 64 KB), the pixel rows, S last
    float* const HW = reinterpret_cast<float*>(smem);
    float* const HB = HW + 16 * HWS;
    float* const CB = HB + 16;
    constexpr int XA_OFF = HW_BYTES + HB_BYTES + CB_BYTES, XB_OFF = XA_OFF + xa_bytes(G), SB_OFF = XB_OFF + xb_bytes(G);
    char* const XA = smem + XA_OFF;
    char* const XB = smem + XB_OFF;
    char* const SB = smem + SB_OFF;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kx = wave % 3, slab = wave / 3;
    const int H2 = 2 * a.h, W2 = 2 * a.w_;

    // ---- constants of the gather phases: head weights (class-major, zero rows past C), head bias, conv bias
    if (HEAD) {
        for (int i = tid; i < 16 * 64; i += NW * 64) HW[(i >> 6) * HWS + (i & 63)] = (i >> 6) < a.head_c ? a.head_w[i] : 0.f;
        if (tid < 16) HB[tid] = (tid < a.head_c && a.head_b) ? a.head_b[tid] : 0.f;
    }
    if (tid < 64) CB[tid] = a.bias ? a.bias[tid] : 0.f;
    const bool has_bias = a.bias != nullptr;
    const bool prelu_max = a.act == APE_ACT_PRELU && a.alpha >= 0.f && a.alpha <= 1.f;     // then x > 0 ? x : alpha x == max(x, alpha x), bit for bit
    const unsigned long long alpha2 = (unsigned long long)__builtin_bit_cast(unsigned, a.alpha);

    // ---- tile walk: logical tile ids are dealt to the XCDs in contiguous runs (bijective remap of the dispatch id, as halo_s32_kernel)
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const int nt = a.B * tiles_per_img;
    const int q8 = nt / 8, r8 = nt % 8;
    const int grid = (int)gridDim.x;
    auto decode = [&](int orig, int& tb, int& ty0, int& tx0) {
        const int xcd = orig % 8;
        const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + orig / 8;
        tb = logical / tiles_per_img;
        const int trem = logical - tb * tiles_per_img;
        const int ty = trem / a.tiles_x;
        ty0 = ty * TY;
        tx0 = (trem - ty * a.tiles_x) * TX;
    };
    // NR rows (slots r0 .. r0 + NR - 1) x 16 pixels of a tile, 32-channel group by group: piece p = 8 pixels of one row (1 KB per
    // wave-instruction, wave w takes pieces w and w + 12); lanes 8 j .. 8 j + 7 fetch pixel j's 128-B line with the 16-B chunks permuted by
    // the column swizzle (chunk c of column slot sp lives in slot c ^ ((sp >> 1) & 7)).  Rows AND columns are clamped into the image: a
    // clamped column slot holds a real pixel that no interpolation ever refers to (the slots of a tile's taps are inside by the floor pattern)
    const long frame = (long)a.h * a.w_ * G * 128;
    const unsigned lds0 = (unsigned)reinterpret_cast<size_t>((lds_void*)smem);
    auto dma_rows = [&](int dst_off, int r0, auto nr_c, int tb, int ty0, int tx0) __attribute__((always_inline)) {
        constexpr int NR = decltype(nr_c)::value, NP = G * NR * 2;
        static_assert(NP >= NW && NP <= 2 * NW, "two pieces per wave at most");
        if (ABL(16)) return;
        const int iyb = ty0 / 2 - 1, ixb = tx0 / 2 - 1;
        const unsigned long long base = reinterpret_cast<unsigned long long>(a.x) + (unsigned long long)tb * (unsigned long long)frame;
        const i32x4 rs = {(int)(unsigned)base, (int)(unsigned)((base >> 32) & 0xFFFFu), (int)frame, 0x00020000};
        int ln_d = lane;
        asm volatile("" : "+v"(ln_d));
        auto piece = [&](int p) __attribute__((always_inline)) {
            const int g = p / (NR * 2), rem = p - g * (NR * 2), rr = rem >> 1, half = rem & 1;
            int row = iyb + r0 + rr;
            row = row < 0 ? 0 : (row > a.h - 1 ? a.h - 1 : row);
            const int sp = half * 8 + (ln_d >> 3);
            int col = ixb + sp;
            col = col < 0 ? 0 : (col > a.w_ - 1 ? a.w_ - 1 : col);
            const unsigned voff = (unsigned)(((row * a.w_ + col) * G + g) * 128 + (((ln_d & 7) ^ ((sp >> 1) & 7)) * 16));
            dma_piece(voff, rs, lds0 + (unsigned)(dst_off + p * 1024));
        };
        piece(wave);
        if (NP == 2 * NW || wave + NW < NP) piece(wave + NW);
    };

    int orig = blockIdx.x;
    int b = 0, Y0 = 0, X0 = 0;
    decode(orig, b, Y0, X0);               // (the host launches no more workgroups than tiles)
    dma_rows(XA_OFF, 0, std::integral_constant<int, RA>{}, b, Y0, X0);
    dma_rows(XB_OFF, RA, std::integral_constant<int, RB>{}, b, Y0, X0);
    // ---- the weights of this wave's (kx, slab): three taps x 16 rows x G channel groups, hi | lo
    bf16x8 wh[G][3], wl[G][3];
    {
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, 9 * 64 * G * 128, 0x00020000);
        const unsigned wv = (unsigned)((lane & 15) * (G * 128) + (lane >> 4) * 16);
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int soff = (((ky * 3 + kx) * 64 + slab * 16) * G + g) * 128;
                if (ABL(32)) {
                    const unsigned u = wv + soff;
                    wh[g][ky] = __builtin_bit_cast(bf16x8, make_uint4(u, u, u, u));
                    wl[g][ky] = wh[g][ky];
                    continue;
                }
                wh[g][ky] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, wv, soff, 0));
                wl[g][ky] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, wv + 64, soff, 0));
            }
    }
    // z of this wave's three taps x 16 channels for NR rows x 16 pixels.  The lane-derived addresses are formed from an opaque copy of the
    // lane id inside the phase: hoisted in front of the tile loop (they are loop-invariant) they would stay allocated through every phase
    auto mfma_rows = [&](const char* src, auto nr_c, f32x4 (&acc)[3][decltype(nr_c)::value]) __attribute__((always_inline)) {
        constexpr int NR = decltype(nr_c)::value;
        int ln_m = lane;
        asm volatile("" : "+v"(ln_m));
        const int s_m = ln_m & 15, fc_m = ln_m >> 4, sw7 = (s_m >> 1) & 7;
        const unsigned xoff_h = (unsigned)(s_m * 128 + ((fc_m ^ sw7) * 16)), xoff_l = (unsigned)(s_m * 128 + (((4 + fc_m) ^ sw7) * 16));
#pragma unroll
        for (int g = 0; g < G; ++g) {
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const char* xp = src + (g * NR + r) * (COLS * 128);
                const bf16x8 xh = *reinterpret_cast<const bf16x8*>(xp + xoff_h), xl = *reinterpret_cast<const bf16x8*>(xp + xoff_l);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    if (ABL(1)) {
                        asm volatile("" :: "v"(xh), "v"(xl));
                        if (g == 0) acc[ky][r] = f32x4{0.f, 0.f, 0.f, 0.f};
                        continue;
                    }
                    // weights as the row operand (D[channel 4 fc + e][pixel s]); product order of conv_gemm_s32.hip
                    const f32x4 c0 = g == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[ky][r];
                    acc[ky][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[g][ky], xl, c0, 0, 0, 0);
                    acc[ky][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[g][ky], xh, acc[ky][r], 0, 0, 0);
                    acc[ky][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[g][ky], xh, acc[ky][r], 0, 0, 0);
                }
            }
        }
    };

    f32x4 accA[3][RA], accB[3][RB];
#ifdef APE_UPFUSE_STAMPS
    unsigned long long st_sum[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last;
#endif
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    mfma_rows(XA, std::integral_constant<int, RA>{}, accA);

#ifdef APE_UPFUSE_STAMPS
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
#endif
#pragma unroll 1
    for (; orig < nt; orig += grid) {
        const int ixb = X0 / 2 - 1;
        const bool has_next = orig + grid < nt;
        int nb = b, ny0 = Y0, nx0 = X0;
        if (has_next) decode(orig + grid, nb, ny0, nx0);

        // ---- row interpolation weights of the 18 up-sampled rows q = Y0 - 1 + t (lane t computes, everybody reads them as scalars)
        float l0v, l1v;
        bool okv;
        {
            const int qv = Y0 - 1 + lane;
            okv = lane < TY + 2 && qv >= 0 && qv < H2;
            const float fy = a.sh * (float)(okv ? qv : 0);
            const int iy0 = (int)fy;
            l1v = fy - (float)iy0;
            l0v = 1.f - l1v;
        }
        const unsigned okm = (unsigned)__ballot(okv);
        const bool top_tile = Y0 == 0;
        const bool rows_inside = !APE_NO_FAST_ROWS && okm == 0x3FFFFu && !top_tile;     // every one of the 18 rows is in the image and follows the floor pattern
        // S_kx(Y0 + y, column slot s) = sum_ky lerp_y(z_{ky,kx}(., s), q = Y0 + y + ky - 1): iy0(q) - (Y0 / 2 - 1) = t >> 1 for t = y + ky
        // (q = 0, the top tile's t = 1, is the one exception of the pattern: iy0 = 0 is slot 1 like slot 0 (clamped), iy1 = 1 is slot 2)
        auto row_of = [&](auto ky_c, auto r_c) -> const f32x4& {
            constexpr int ky = decltype(ky_c)::value, r = decltype(r_c)::value;
            if constexpr (r < RA) return accA[ky][r];
            else return accB[ky][r - RA];
        };
        auto ylerp_rows = [&](auto half_c) __attribute__((always_inline)) {
            constexpr int y_lo = decltype(half_c)::value * 8;
            int ln_y = lane;
            asm volatile("" : "+v"(ln_y));
            const int s_y = ln_y & 15, fc_y = ln_y >> 4;
            const unsigned s_wr = (unsigned)((kx * 4 + slab) * 8192 + fc_y * 256 + s_y * 16);     // this lane's float4 in row 0 of its wave's S block (kx-major:
                                                                                                   // a gather lane reaches the four slabs of a kx by immediates)
            if (rows_inside) {
                // the common case: ten weight pairs as scalars, then 96 two-channel instructions without a branch
                unsigned long long w2[10];
#pragma unroll
                for (int i = 0; i < 10; ++i)
                    w2[i] = (unsigned long long)__builtin_bit_cast(unsigned, lane_value(l0v, y_lo + i)) |
                            ((unsigned long long)__builtin_bit_cast(unsigned, lane_value(l1v, y_lo + i)) << 32);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_nop 1" ::: "memory");      // (a lane read writes an SGPR: two wait states before a vector instruction hipcc cannot see reads it)
                __builtin_amdgcn_sched_barrier(0);
                auto one_row = [&](auto yy_c) __attribute__((always_inline)) {
                    constexpr int yy = decltype(yy_c)::value, y = y_lo + yy;
                    f32x2 Sa, Sb;
                    auto term = [&](auto ky_c) __attribute__((always_inline)) {
                        constexpr int ky = decltype(ky_c)::value, t = y + ky;
                        const f32x4& v0 = row_of(ky_c, std::integral_constant<int, (t >> 1)>{});
                        const f32x4& v1 = row_of(ky_c, std::integral_constant<int, (t >> 1) + 1>{});
                        lerp_term_s<FMA, ky == 0>(Sa, lo2(v0), lo2(v1), w2[t - y_lo]);
                        lerp_term_s<FMA, ky == 0>(Sb, hi2(v0), hi2(v1), w2[t - y_lo]);
                    };
                    term(std::integral_constant<int, 0>{});
                    term(std::integral_constant<int, 1>{});
                    term(std::integral_constant<int, 2>{});
                    *reinterpret_cast<f32x4*>(SB + s_wr + yy * 1024) = f32x4{Sa[0], Sa[1], Sb[0], Sb[1]};
                };
                one_row(std::integral_constant<int, 0>{}); one_row(std::integral_constant<int, 1>{});
                one_row(std::integral_constant<int, 2>{}); one_row(std::integral_constant<int, 3>{});
                one_row(std::integral_constant<int, 4>{}); one_row(std::integral_constant<int, 5>{});
                one_row(std::integral_constant<int, 6>{}); one_row(std::integral_constant<int, 7>{});
                return;
            }
            // tiles on the first / last image rows: a term per valid up-sampled row only (the reference skips the taps outside the image)
            auto one_row = [&](auto yy_c) __attribute__((always_inline)) {
                constexpr int yy = decltype(yy_c)::value, y = y_lo + yy;
                f32x4 S = {0.f, 0.f, 0.f, 0.f};
                auto term = [&](auto ky_c) __attribute__((always_inline)) {
                    constexpr int ky = decltype(ky_c)::value, t = y + ky;
                    if ((okm >> t) & 1u) {
                        const float l0 = lane_value(l0v, t), l1 = lane_value(l1v, t);
                        const f32x4 v0 = row_of(ky_c, std::integral_constant<int, (t >> 1)>{});
                        f32x4 v1 = row_of(ky_c, std::integral_constant<int, (t >> 1) + 1>{});
                        if constexpr (t == 1) {
                            if (top_tile) v1 = row_of(ky_c, std::integral_constant<int, 2>{});
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (FMA) S[e] = __builtin_fmaf(l1, v1[e], __builtin_fmaf(l0, v0[e], S[e]));
                            else S[e] = S[e] + (l0 * v0[e] + l1 * v1[e]);
                        }
                    }
                };
                term(std::integral_constant<int, 0>{});
                term(std::integral_constant<int, 1>{});
                term(std::integral_constant<int, 2>{});
                *reinterpret_cast<f32x4*>(SB + s_wr + yy * 1024) = S;
            };
            one_row(std::integral_constant<int, 0>{}); one_row(std::integral_constant<int, 1>{});
            one_row(std::integral_constant<int, 2>{}); one_row(std::integral_constant<int, 3>{});
            one_row(std::integral_constant<int, 4>{}); one_row(std::integral_constant<int, 5>{});
            one_row(std::integral_constant<int, 6>{}); one_row(std::integral_constant<int, 7>{});
        };
        // ---- one group of 16 output pixels x 64 channels per wave and half tile: column interpolation + sum over kx, bias, activation,
        // then the store or the head.  A tap column outside the image contributes with both weights zero (the reference skips it: the
        // same sum)
        auto gather_half = [&](int hh) __attribute__((always_inline)) {
            // lane = pixel g_s of group `wave`, channel quad g_fc.  For a tile whose tap columns q all satisfy 1 <= q <= 2w - 2 (floor pattern:
            // ix0 = (q - 1) >> 1, ix1 = ix0 + 1) the column slot s0 = ix0 - (X0 / 2 - 1) = (Xl + k) >> 1 does not depend on the tile.  (Formed here from an
            // opaque lane id: hoisted out of the tile loop these few values would be spilled around the matrix phases.)
            int ln_g = lane;
            asm volatile("" : "+v"(ln_g));
            const int g_s = ln_g & 15, g_fc = ln_g >> 4;
            const int g_lin = wave * 16 + g_s;
            const int g_yl = g_lin / TX, g_xl = g_lin - g_yl * TX;
            const unsigned so_base = (unsigned)(g_yl * 1024 + g_fc * 256 + (g_xl >> 1) * 16);      // k = 0; k = 1: + 16 (Xl & 1); k = 2: + 16; ix1: + 16 more
            const unsigned so_odd = (unsigned)((g_xl & 1) * 16);
            const int Y = Y0 + hh * 8 + g_yl;
            int X = X0 + g_xl;
            const bool pix_ok = Y < H2 && X < W2;
            X = X < W2 ? X : W2 - 1;
            f32x2 lx[3];            // {weight of ix0, weight of ix1}
            unsigned o0[3], o1[3];
            if (!APE_NO_FAST_COLS && X0 >= 2 && X0 + TX <= W2 - 2) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float fx = a.sw * (float)(X + k - 1);
                    const float l1 = fx - (float)(ixb + ((g_xl + k) >> 1));
                    lx[k] = f32x2{1.f - l1, l1};
                }
                o0[0] = so_base; o0[1] = so_base + so_odd; o0[2] = so_base + 16;
                o1[0] = o0[0] + 16; o1[1] = o0[1] + 16; o1[2] = o0[2] + 16;
            } else {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int qx = X + k - 1;
                    const bool okx = (unsigned)qx < (unsigned)W2;
                    const float fx = a.sw * (float)(okx ? qx : 0);
                    const int ix0 = (int)fx, ix1 = ix0 + (ix0 < a.w_ - 1 ? 1 : 0);
                    const float l1 = fx - (float)ix0;
                    lx[k] = f32x2{okx ? 1.f - l1 : 0.f, okx ? l1 : 0.f};
                    const int s0 = (ix0 - ixb) & 15, s1 = (ix1 - ixb) & 15;
                    o0[k] = (unsigned)(g_yl * 1024 + g_fc * 256 + s0 * 16);
                    o1[k] = (unsigned)(g_yl * 1024 + g_fc * 256 + s1 * 16);
                }
            }
            STAMP(12);
            const ape::ActFast af = ape::act_fast_make(a.act, a.alpha);
            float4 xv[4];           // HEAD: the pixel's 64 activated channels in seg_head_group's layout (x[j] = channels 16 j + 4 fc .. + 3)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x2 oa, ob;       // channels 16 j + 4 fc + {0, 1}, {2, 3}
                auto tap = [&](auto k_c) __attribute__((always_inline)) {
                    constexpr int k = decltype(k_c)::value;
                    const char* sp = SB + (k * 4 + j) * 8192;
                    f32x4 a0, a1;
                    if (ABL(64)) {
                        a0 = f32x4{lx[k][0], lx[k][1], lx[k][0], lx[k][1]};
                        a1 = a0;
                    } else {
                        a0 = *reinterpret_cast<const f32x4*>(sp + o0[k]);
                        a1 = *reinterpret_cast<const f32x4*>(sp + o1[k]);
                    }
                    lerp_term_v<FMA, k == 0>(oa, lo2(a0), lo2(a1), lx[k]);
                    lerp_term_v<FMA, k == 0>(ob, hi2(a0), hi2(a1), lx[k]);
                };
                tap(std::integral_constant<int, 0>{});
                tap(std::integral_constant<int, 1>{});
                tap(std::integral_constant<int, 2>{});
                if (has_bias) {
                    const float4 cb = *reinterpret_cast<const float4*>(CB + 16 * j + 4 * g_fc);
                    oa += f32x2{cb.x, cb.y};
                    ob += f32x2{cb.z, cb.w};
                }
                float o[4];
                if (prelu_max) {
                    const f32x2 ta = pk_mul_lo_s(oa, alpha2), tb = pk_mul_lo_s(ob, alpha2);
                    // (v_max_f32 by hand: hipcc canonicalises both operands of fmaxf with a v_max each first)
                    asm volatile("v_max_f32 %0, %4, %5\n\tv_max_f32 %1, %6, %7\n\tv_max_f32 %2, %8, %9\n\tv_max_f32 %3, %10, %11"
                                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3])
                                 : "v"(oa[0]), "v"(ta[0]), "v"(oa[1]), "v"(ta[1]), "v"(ob[0]), "v"(tb[0]), "v"(ob[1]), "v"(tb[1]));
                } else {
                    o[0] = ape::act_fast(oa[0], af); o[1] = ape::act_fast(oa[1], af);
                    o[2] = ape::act_fast(ob[0], af); o[3] = ape::act_fast(ob[1], af);
                }
                if (HEAD) {
                    xv[j] = make_float4(o[0], o[1], o[2], o[3]);
                } else if (pix_ok) {
                    const size_t pix = ((size_t)b * H2 + Y) * W2 + X;
                    const float4 ov = make_float4(o[0], o[1], o[2], o[3]);
                    if (a.out_fmt == APE_FMT_S32) ape::s32_store4(a.y, (long)pix, 16, 4 * j + g_fc, ov);
                    else *reinterpret_cast<float4*>(a.y + (pix * 64 + 16 * j + 4 * g_fc) * 4) = ov;
                }
            }
            STAMP(13);
            if (HEAD) {
                // the head proper: seg_head_group on the pixel's 64 channels, exactly as seg_head_kernel and the halo kernel's epilogue run it
                float wreg[16], hbias[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 wv = *reinterpret_cast<const float4*>(HW + g_s * HWS + 16 * j + 4 * g_fc);
                    wreg[4 * j] = wv.x; wreg[4 * j + 1] = wv.y; wreg[4 * j + 2] = wv.z; wreg[4 * j + 3] = wv.w;
                }
                {
                    const float4 hb = *reinterpret_cast<const float4*>(HB + g_fc * 4);
                    hbias[0] = hb.x; hbias[1] = hb.y; hbias[2] = hb.z; hbias[3] = hb.w;
                }
                int am;
                float pm;
                if (ABL(8)) { am = (int)xv[0].x; pm = xv[1].y + wreg[3] + hbias[1]; }
                else ape_seg::seg_head_group(xv, wreg, hbias, a.head_c, ln_g, a.head_dsm, am, pm);
                if (g_fc == 0 && pix_ok) {
                    const size_t pix = ((size_t)b * H2 + Y) * W2 + X;
                    a.label[pix] = (uint8_t)am;
                    a.score[pix] = pm;
                }
#ifdef APE_UPFUSE_DUMP
                // diagnostic build only (tools/dump_upfuse.py): the activations the head consumed, [pixel][64] floats behind the stamps pointer
                if (a.stamps && pix_ok) {
                    float* dd = reinterpret_cast<float*>(a.stamps) + (((size_t)b * H2 + Y) * W2 + X) * 64;
#pragma unroll
                    for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(dd + 16 * j + 4 * g_fc) = xv[j];
                }
#endif
            }
        };

        // (S is free: the previous tile's bottom gather ended behind the barrier that closed the last interval)
        STAMP(0);
        if (!ABL(2)) ylerp_rows(std::integral_constant<int, 0>{});
        STAMP(1);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        STAMP(2);
        // every wave has multiplied rows 0..5 of this tile: their LDS image takes the next tile's
        if (has_next) dma_rows(XA_OFF, 0, std::integral_constant<int, RA>{}, nb, ny0, nx0);
        STAMP(3);
        if (!ABL(4)) gather_half(0);
        STAMP(4);
        mfma_rows(XB, std::integral_constant<int, RB>{}, accB);
        STAMP(5);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        STAMP(6);
        if (!ABL(2)) ylerp_rows(std::integral_constant<int, 1>{});
        STAMP(7);
        // the next tile's rows 0..5 (requested an interval ago) must have landed before anyone multiplies them
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        STAMP(8);
        if (has_next) dma_rows(XB_OFF, RA, std::integral_constant<int, RB>{}, nb, ny0, nx0);
        if (!ABL(4)) gather_half(1);
        STAMP(9);
        // (the last tile multiplies its own rows 0..5 once more, unused: an unconditional redefinition keeps the old rows dead from here on)
        mfma_rows(XA, std::integral_constant<int, RA>{}, accA);
        STAMP(10);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        STAMP(11);
        b = nb; Y0 = ny0; X0 = nx0;
    }
#ifdef APE_UPFUSE_STAMPS
    if (a.stamps && lane == 0)
        for (int i = 0; i < 16; ++i) a.stamps[((size_t)blockIdx.x * NW + wave) * 16 + i] = st_sum[i];
#endif
#endif
}

// the register form's floor pattern: (int)(scale * q) == (q - 1) >> 1 for q >= 1 (0 for q = 0), scale = (n - 1) / (2 n - 1) in fp32
bool floor_pattern_ok(int n)
{
    const float sc = 2 * n > 1 ? (float)(n - 1) / (float)(2 * n - 1) : 0.f;
    for (int q = 0; q < 2 * n; ++q) {
        const int want = q >= 1 ? (q - 1) >> 1 : 0;
        if ((int)(sc * (float)q) != want) return false;
    }
    return true;
}

bool fused_supported(int h, int w, int Cin, int Cout)
{
    if (Cin != 64 || Cout != 64 || h < 1 || w < 1) return false;
    if ((long)h * w * Cin * 4 >= (1L << 31)) return false;
    return floor_pattern_ok(h) && floor_pattern_ok(w);
}

template <int G, bool HEAD, bool FMA>
int launch_fused(const UpFuseArgs& a, hipStream_t st)
{
    auto kern = upconv_fused_kernel<G, HEAD, FMA>;
    static ape::DeviceOnce once;       // (per kernel instantiation)
    int ncu = 256;
    if (int rc = ape::device_once(once, reinterpret_cast<const void*>(kern), lds_bytes(G), &ncu)) return rc;
    const long nt = (long)a.B * a.tiles_x * a.tiles_y;
    int grid = (int)(nt < ncu ? nt : ncu);
    if (grid > 8) grid -= grid % 8;             // whole XCD rounds: dispatch id % 8 labels a workgroup's XCD for every tile it walks
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds_bytes(G), st, a);
    return ape::check_launch("ape_upconv3x3_fused");
}

int g_fused_dbg = 0;
unsigned long long* g_fused_stamps = nullptr;

int run_fused(const void* x, const void* w9, const float* bias, void* out, int out_fmt, int B, int h, int w, int Cin, int act, float alpha,
              int fma, const float* head_w, const float* head_b, int C, uint8_t* label, float* score, int dsm, bool head, void* stream)
{
    if (!x || !w9 || B < 0 || (fma != 0 && fma != 1)) return APE_EINVAL;
    if (!fused_supported(h, w, Cin, 64)) return APE_EINVAL;
    if (act < APE_ACT_NONE || act > APE_ACT_PRELU) return APE_EINVAL;
    if (head) {
        if (!head_w || !label || !score || C < 1 || C > 16) return APE_EINVAL;
    } else {
        if (!out || (out_fmt != APE_FMT_F32 && out_fmt != APE_FMT_S32)) return APE_EINVAL;
    }
    if ((long)B * 4 * h * w >= (1L << 31)) return APE_EINVAL;
    if (B == 0) return APE_OK;
    UpFuseArgs a;
    a.x = (const char*)x; a.w = (const char*)w9; a.bias = bias; a.y = (char*)out;
    a.head_w = head_w; a.head_b = head_b; a.label = label; a.score = score;
    a.B = B; a.h = h; a.w_ = w; a.act = act; a.alpha = alpha;
    a.sh = 2 * h > 1 ? (float)(h - 1) / (float)(2 * h - 1) : 0.f;
    a.sw = 2 * w > 1 ? (float)(w - 1) / (float)(2 * w - 1) : 0.f;
    a.out_fmt = out_fmt; a.head_c = C; a.head_dsm = dsm;
    a.tiles_x = ape::ceil_div(2 * w, TX); a.tiles_y = ape::ceil_div(2 * h, TY);
    a.dbg = g_fused_dbg;
    a.stamps = g_fused_stamps;
    hipStream_t st = (hipStream_t)stream;
    if (head) return fma ? launch_fused<2, true, true>(a, st) : launch_fused<2, true, false>(a, st);
    return fma ? launch_fused<2, false, true>(a, st) : launch_fused<2, false, false>(a, st);
}

}  // namespace

extern "C" int ape_upconv3x3_fused_debug(int bits) { g_fused_dbg = bits; return APE_OK; }
extern "C" int ape_upconv3x3_fused_stamps(void* device_buffer) { g_fused_stamps = (unsigned long long*)device_buffer; return APE_OK; }

extern "C" int ape_upconv3x3_fused_supported(int h, int w, int Cin, int Cout) { return fused_supported(h, w, Cin, Cout) ? 1 : 0; }

extern "C" int ape_upconv3x3_fused_s32(const void* x_s32, const void* w9_s32k, const float* bias, void* out, int out_fmt, int B, int h, int w,
                                       int Cin, int act, float alpha, int fma, void* stream)
{
    return run_fused(x_s32, w9_s32k, bias, out, out_fmt, B, h, w, Cin, act, alpha, fma, nullptr, nullptr, 0, nullptr, nullptr, 0, false, stream);
}

extern "C" int ape_upconv3x3_fused_seghead_s32(const void* x_s32, const void* w9_s32k, const float* bias, int B, int h, int w, int Cin, int act,
                                               float alpha, int fma, const float* head_w, const float* head_b, int C, uint8_t* label, float* score,
                                               int double_softmax, void* stream)
{
    return run_fused(x_s32, w9_s32k, bias, nullptr, APE_FMT_F32, B, h, w, Cin, act, alpha, fma, head_w, head_b, C, label, score, double_softmax, true,
                     stream);
}
