/* ape_hip.h -- C ABI of libape_hip.so, the MI355X (gfx950) hot path behind the Python call signatures of
 * KochPJ/AutoPoseEstimation's seg -> DenseFusion -> ICP slice.
 *
 * Conventions ("Ownership / errors / threading"):
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless its name ends in `_host`;
 *   - the caller owns every buffer, nothing is allocated behind its back (workspaces are passed in);
 *   - every entry point enqueues on the hipStream_t passed as `void* stream` (NULL = default stream)
 *     and returns without synchronising;  it is re-entrant per stream;
 *   - return value: APE_OK (0) or a negative APE_E* code; nothing throws, nothing prints.
 *
 * Each entry names the reference interface it replaces (file:line relative to the reference repo).
 * The Python side (autoposeestimation_amd/_lib.py) binds exactly these symbols with ctypes;
 * INTEGRATION.md shows the binding a reference maintainer would add.
 */
#ifndef APE_HIP_H
#define APE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define APE_OK 0
#define APE_EINVAL (-1)   /* bad shape / unsupported parameter combination */
#define APE_ELAUNCH (-2)  /* hipGetLastError() != hipSuccess after the launch */
#define APE_EWORKSPACE (-3) /* workspace too small */

/* ABI version: bumped whenever a signature below changes. */
int ape_abi_version(void);
/* Name of the last HIP error seen by this thread's failing call (static storage). */
const char* ape_last_error(void);

/* ---- k-NN -------------------------------------------------------------------------------------------
 * Replaces `int knn(at::Tensor& ref, at::Tensor& query, at::Tensor& idx)`
 *   DenseFusion/lib/knn/src/knn.h:12-66 (dispatcher), src/cpu/knn_cpu.cpp:4-55 (semantics),
 *   src/cuda/knn.cu:217-263 (the CUDA path this supersedes), bound at src/vision.cpp:3-5.
 * ref[batch][dim][ref_nb] f32, query[batch][dim][query_nb] f32 (contiguous, as knn.h:23-24 assumes),
 * idx[batch][k][query_nb] i64, filled with 1-BASED ref indices of the k smallest squared-L2 distances
 * in ascending order, ties in ascending ref index (knn_cpu.cpp:30 swaps only on '>').
 * Distances are accumulated in float32 dimension by dimension, one rounding per multiply and add
 * (no FMA), so indices are bit-identical to the reference CPU build.
 * Returns APE_OK (the reference returns 1; the Python wrapper keeps that convention). */
int ape_knn_f32(const float* ref, const float* query, int64_t* idx,
                int batch, int dim, int ref_nb, int query_nb, int k, void* stream);

/* ---- dense contractions: conv2d / 1x1 / Linear, exact fp32 on the matrix cores ---------------------------
 * One entry point replaces every torch.nn.Conv2d / Conv1d(k=1) / Linear forward on the path:
 *   DenseFusion/lib/extractors.py:14-16,82-89,101-105 (ResNet convs), pspnet.py:12-17,30-33,53-55 (PSP, up-convs,
 *   final 1x1), network.py:42-49,76-92,139-146,175-182 (PointNet / head 1x1 chains, refiner Linear stacks),
 * with the following pointwise op fused: bias, residual add (extractors.py:40), ReLU / PReLU / sigmoid.
 *
 * Layouts: x[B][H][W][ldx] f32 (NHWC, the layer reads channels xoff .. xoff+Cin-1), w[Cout][KH][KW][Cin] f32,
 * y[B][Ho][Wo][ldy] (writes channels yoff .. yoff+Cout-1), residual like y with (ldr, roff).
 * bias: NULL, or bias[Cout] (bias_bstride = 0), or per-image bias[B][bias_bstride] (first Cout entries used).
 * Constraints: Cin, ldx, xoff multiples of 4 (16-byte loads); Ho/Wo must equal the conv arithmetic result. */
enum { APE_ACT_NONE = 0, APE_ACT_RELU = 1, APE_ACT_PRELU = 2, APE_ACT_SIGMOID = 3 };
typedef struct ape_conv_params {
    int32_t B, H, W, Cin, ldx, xoff;
    int32_t Ho, Wo, Cout, ldy, yoff;
    int32_t KH, KW, stride, pad, dil;
    int32_t act;            /* APE_ACT_* */
    float alpha;            /* PReLU slope (pspnet.py:33, single parameter) */
    int32_t bias_bstride;
    int32_t ldr, roff;
    int32_t ups;            /* 1: x is the LOW-resolution tensor [B][H/2][W/2][ldx] and the conv reads its bilinear x2
                               (align_corners=True) up-sampling on the fly (nn.Upsample + Conv2d of pspnet.py:30-32 fused);
                               only ape_conv3x3_halo_bf16 accepts it (H, W even), everything else requires 0 */
} ape_conv_params;
int ape_conv2d_nhwc_f32(const float* x, const float* w, const float* bias, const float* residual, float* y,
                        const ape_conv_params* params_host, void* stream);

/* ---- the same contraction on the bf16 matrix cores (fp32 accumulate, fp32 activations in HBM) ------------------
 * nsplit = 3: split-bf16 operands (x = hi + lo, three MFMA products per term, ~2^-16 relative operand error);
 * nsplit = 1: plain bf16 operands.  Same params / fused epilogue / layouts as ape_conv2d_nhwc_f32, except that the
 * weights are the packed planes produced by ape_pack_weights_bf16 from w[Cout][KH*KW*Cin] f32:
 * hi plane [Cout][Kp] bf16 then lo plane [Cout][Kp], Kp = K rounded up to 8 (ape_packed_weights_bf16_elems elements). */
long ape_packed_weights_bf16_elems(int cout, int K);
int ape_pack_weights_bf16(const float* w, void* out, int cout, int K, void* stream);
int ape_conv2d_nhwc_bf16(const float* x, const void* w_packed, const float* bias, const float* residual, float* y,
                         const ape_conv_params* params_host, int nsplit, void* stream);
/* Cin % 32 == 0 successor of ape_conv2d_nhwc_bf16 (same arguments, packed weights, numerics class and fused epilogue):
 * 16x16x32 MFMA, swizzled double-buffered LDS stages, register staging that writes tile t+1 after the barrier and re-issues
 * tile t+2 at once.  Takes every 1x1 conv / Conv1d(k=1) / Linear of the path (pspnet.py:12-17,53, network.py:42-49,76-92,
 * 139-146,175-182) and the k x k convs the LDS-halo kernel leaves.  variant: 0 = block shape chosen from (M, Cout, K),
 * 1 = 256x256, 2 = 128x128, 3 = 256x64, 4 = 256x192 (tuning aid).  ape_conv_gemm_supported(params) says whether it applies. */
int ape_conv_gemm_supported(const ape_conv_params* params_host);
int ape_conv_gemm_bf16(const float* x, const void* w_packed, const float* bias, const float* residual, float* y,
                       const ape_conv_params* params_host, int nsplit, int variant, void* stream);
/* ---- pre-split ("S32") activations --------------------------------------------------------------------------------------
 * The split-bf16 operands of the bf16 kernels above, made ONCE by the producing kernel's epilogue instead of by every consumer
 * tile: a pixel's C channels (C % 32 == 0) are C/32 groups of 128 bytes [hi: 32 x bf16 | lo: 32 x bf16], hi = bf16(v),
 * lo = bf16(v - hi) -- 4 bytes per channel, so (ld, offset) arithmetic in channels is unchanged.  Weights for S32 consumers use
 * the same grouping along K ("S32K": [Cout][K/32][hi 32 | lo 32], ape_pack_weights_s32k from w[Cout][K] f32).
 * ape_conv_gemm_s32: the 1x1 / stride-1 layers (pspnet.py:12-17,22-24 and the low-resolution channel mixing of :27-37) with
 * x in S32, operands streamed HBM -> LDS by LDS-DMA (no staging registers, no split VALU), y and the residual in either format.
 * Needs Cin % 32 == 0, ldx % 32 == 0, xoff % 32 == 0, Cout >= 128 and % 4; ape_conv_gemm_s32_supported(params) says so. */
enum { APE_FMT_F32 = 0, APE_FMT_S32 = 1 };
int ape_pack_weights_s32k(const float* w, void* out, int cout, int K, void* stream);
/* x[rows][C] f32 -> y[rows][C] S32 (to_s32 = 1) or back (to_s32 = 0: hi + lo); C % 32 == 0.  Tests and format boundaries only. */
int ape_convert_s32(const void* x, void* y, long rows, int C, int to_s32, void* stream);
int ape_conv_gemm_s32_supported(const ape_conv_params* params_host);
int ape_conv3x3_halo_s32_debug(int bits);   /* development aid, results unchanged by any bit: 1 = a static wave priority for waves 4-7 (lockstep kernel); 2 = one workgroup per tile; 8 = one channel tile per XCD (weights L2-resident, halos fetched n_tiles times: DESIGN.md 6f); 16 = four rows per wave-row group in every tile; 4096 = launch the lockstep (one-barrier, in-phase) kernel of rounds 2-5 instead of the ping-pong one (tools/mb_halo_pp.py, tests/test_gpu_s32.py) */
int ape_conv_gemm_s32_debug(int bits);   /* development aid: 16 / 32 / 64 / 8192 leave the results unchanged (k-tile rotation, no static priority, one tile per workgroup, 8192 = launch the ping-pong kernels: bit-identical, slower, tools/mb_gemm_pp.py); the ablation build (make ablations) also knows timing-only bits that break the results */
int ape_conv_gemm_s32(const void* x_s32, const void* w_s32k, const float* bias, const void* residual, int res_fmt, void* y,
                      int out_fmt, const ape_conv_params* params_host, void* stream);
/* The same contraction with PER-IMAGE weights: image i of the batch (params->H * params->W rows) multiplies with the S32K matrix at
 * w_s32k + i * w_image_stride_bytes (a multiple of 16), and no 256-row tile holds rows of two images.  No residual operand.
 * Used by the PSP module (DenseFusion/lib/pspnet.py:12-24), whose prior sum is folded into the bottleneck's contraction:
 * ape_psp_fold_operands builds both operands' extra 64 K-columns -- channels [Cin, Cin + 64) of every pixel of x_s32 (ld >= Cin + 64
 * channels per pixel) get the pixel's bilinear (align_corners = False) coefficients of the 1 + 4 + 9 + 36 prior cells, and
 * w_out[B][Cout][Cin/32 + 2 groups] gets the shared weights wf_s32k[Cout][Cin/32 groups] followed by frame b's prior values
 * z_s[b][cell][co] (z1..z6: the merged stage x bottleneck-column convs of the pooled maps, fp32 [B][s][s][Cout]) -- so that
 *     relu(W_f . f + sum_s upsample(Z_s) + bias)  ==  ape_conv_gemm_s32_per_image over K = Cin + 64
 * and the 4 * B * h * w * Cout-byte prior-sum tensor (ape_psp_prior_sum_f32) is neither written nor read back. */
int ape_conv_gemm_s32_per_image(const void* x_s32, const void* w_s32k, long w_image_stride_bytes, const float* bias, void* y, int out_fmt,
                                const ape_conv_params* params_host, void* stream);
int ape_psp_fold_operands(const void* wf_s32k, const float* z1, const float* z2, const float* z3, const float* z6, void* w_out, void* x_s32,
                          int B, int h, int w, int ld, int Cin, int Cout, void* stream);
/* Format-aware forms of three fp32 entry points, for the tensors that cross between fp32 and S32 kernels: ape_conv_gemm_bf16 with the
 * OUTPUT in either format (the stride-2 convs that feed the first S32 3x3 layer), ape_adaptive_avgpool_multi_nhwc_f32 with the INPUT
 * in either format (the PSP pools of the S32 layer-4 map), ape_upconv3x3_gather_f32 with the OUTPUT in either format (up_1's result
 * feeds up_2's S32 channel mixing).  Declared further down next to their fp32 forms' documentation. */
int ape_conv_gemm_bf16_fmt(const float* x, const void* w_packed, const float* bias, const float* residual, void* y, int out_fmt,
                           const ape_conv_params* params_host, int nsplit, int variant, void* stream);
/* The PSP module's stage convolutions (pspnet.py:15-18, 22: `stage(feats) for stage in self.stages` -- four bias-free 1x1 convolutions of the
 * 1x1 / 2x2 / 3x3 / 6x6 pooled maps) are independent problems of 1 .. 18 tiles each: n <= 4 such 1x1 / stride-1 convolutions (fp32 in, fp32 out,
 * no residual) in ONE launch.  Problem i multiplies x[i] by w_packed[i] (ape_pack_weights_bf16 layout) under params_host[i]; bias may be NULL
 * or hold NULL entries.  Every output element is bit for bit what ape_conv_gemm_bf16 gives for that problem alone. */
int ape_conv_gemm_bf16_multi(int n, const float* const* x, const void* const* w_packed, const float* const* bias, float* const* y,
                             const ape_conv_params* params_host, int nsplit, void* stream);
/* Split-K form for the training tape's batch-1 layers (train.py:205-238 runs ONE 160x160 crop per step: its 20 x 20 feature maps are 4
 * row tiles of the 128 x 128 block, K up to 4608): the k-tiles are dealt to several workgroups per output tile, raw sums go to the
 * workspace, a second pass adds them in a fixed order with bias / residual / activation.  Same products, another summation order than
 * ape_conv_gemm_bf16; the inference path never takes it.  ..._workspace_bytes returns 0 when the shape is not worth splitting. */
size_t ape_conv_gemm_splitk_workspace_bytes(const ape_conv_params* params);
int ape_conv_gemm_bf16_splitk(const float* x, const void* w_packed, const float* bias, const float* residual, float* y,
                              const ape_conv_params* params, int nsplit, void* workspace, size_t workspace_bytes, void* stream);
int ape_adaptive_avgpool_multi_nhwc_fmt(const void* x, int in_fmt, float* const* ys_host, const int* sizes_host, int nsizes, int B, int H, int W,
                                        int C, void* workspace, size_t workspace_bytes, void* stream);
/* ... of the first C channels of a map that holds ldx channels per pixel (the PSP module's feature map with the 64 spare channels of
 * ape_psp_fold_operands behind its 512: pspnet.py:15 pools the 512); outputs [B,s,s,C] fp32 */
int ape_adaptive_avgpool_multi_nhwc_ld(const void* x, int in_fmt, float* const* ys_host, const int* sizes_host, int nsizes, int B, int H, int W,
                                       int C, int ldx, void* workspace, size_t workspace_bytes, void* stream);
int ape_upconv3x3_gather_fmt(const float* z, const float* bias, void* out, int out_fmt, int B, int h, int w, int C, int act, float alpha,
                             void* stream);
/* ... and with the interpolation arithmetic chosen: fma = 0 separately rounded products like ape_bilinear_nhwc_f32 (what the two entry
 * points above use), fma = 1 chained fused multiply-adds acc = fma(l1, v1, fma(l0, v0, acc)) -- the unfused twin of
 * ape_upconv3x3_fused_* called with fma = 1 */
int ape_upconv3x3_gather_ex(const float* z, const float* bias, void* out, int out_fmt, int B, int h, int w, int C, int act, float alpha,
                            int fma, void* stream);
/* Channel counts that are multiples of 64 take a second kernel with the same arithmetic (bit-identical outputs): strips of `rows` output
 * rows per workgroup, the z rows of every tap held in registers while the strip moves down (each z element is loaded ~1.2x instead of
 * ~5x).  rows >= 1 sets the strip height (default 30), 0 routes every call to the one-row kernel, < 0 only queries; returns the
 * previous value.  A process-wide tuning / test switch, not per stream. */
int ape_upconv3x3_gather_strip_rows(int rows);
/* PSPUpsample with 64 output channels (DenseFusion/lib/pspnet.py:27-37: nn.Upsample x2 align_corners=True -> Conv2d 3x3 pad 1 -> PReLU;
 * up_2 and up_3 of pspnet.py:50-51) as ONE kernel on an S32 input x[B][h][w][Cin] (Cin = 64): the low-resolution channel mixing
 * z = W9 . x (W9 S32K [9*64][Cin], row = tap*64 + co, as ape_upconv3x3_gather_f32 expects) runs on the matrix cores for the 10 x 16
 * low-resolution pixels under a 16 x 24 output tile, the row interpolation + tap-row sum happen on the accumulators (a lane holds one
 * low-resolution column in all rows), the column interpolation + tap-column sum through LDS; the 9*64-channel tensor z never exists
 * in memory.  Values are bit for bit those of ape_conv_gemm_s32 (z) followed by ape_upconv3x3_gather_ex(fma) on the same operands.
 * ..._fused_s32 writes out[B][2h][2w][64] in out_fmt; ..._fused_seghead_s32 feeds the pixels straight into the segmentation head
 * (ape_seg_head_f32: final 1x1 conv rows 0..C-1 + softmax(+softmax) + arg-max, pspnet.py:53-55, pipeline/utils.py:429-435) and writes
 * label[B][2h][2w] u8 / score f32 only.  ..._supported: 1 when the geometry is served (Cin == 64, Cout == 64, and the floor pattern of the
 * align_corners source index that the register form relies on holds for h and w; always true for the sizes of the reference). */
int ape_upconv3x3_fused_supported(int h, int w, int Cin, int Cout);
int ape_upconv3x3_fused_stamps(void* device_buffer);   /* diagnostic build (make stamps) only: [workgroups][12 waves][16] u64 cycle sums; NULL = off */
int ape_upconv3x3_fused_debug(int bits);    /* timing-only ablations for tools/mb_upfuse.py (0 = off; anything else gives WRONG results) */
int ape_upconv3x3_fused_s32(const void* x_s32, const void* w9_s32k, const float* bias, void* out, int out_fmt, int B, int h, int w, int Cin,
                            int act, float alpha, int fma, void* stream);
int ape_upconv3x3_fused_seghead_s32(const void* x_s32, const void* w9_s32k, const float* bias, int B, int h, int w, int Cin, int act, float alpha,
                                    int fma, const float* head_w, const float* head_b, int C, uint8_t* label, float* score, int double_softmax,
                                    void* stream);
/* 3x3 / stride 1 / pad == dilation in {1,2,4} convolutions with Cout >= 128 on S32 activations (extractors.py:29-43 blocks of layers
 * 2-4): the LDS-halo kernel with the halo rows and the weight tiles streamed by LDS-DMA into rings and every fragment read
 * prefetched one tap ahead.  Same accumulators as ape_conv3x3_halo_bf16(nsplit = 3) on the fp32 form of x; weights S32K in the
 * (tap, channel) K order of the packed layout; y / residual in either format. */
int ape_conv3x3_halo_s32_supported(const ape_conv_params* params_host);
int ape_conv3x3_halo_s32(const void* x_s32, const void* w_s32k, const float* bias, const void* residual, int res_fmt, void* y,
                         int out_fmt, const ape_conv_params* params_host, void* stream);
/* The same convolution (extractors.py:29-43; layer 4's 512 -> 512 blocks) with HALF the matrix passes: operands in the "F16M6" line format
 * (autoposeestimation_amd/mx6.py: per 32 channels fp16 values | block-scaled e2m3 codes of them | block-scaled e2m3 codes of the fp16
 * residual), x . w ~ x1 w1 + Q(x1) Q(w2) + Q(x2) Q(w1): one v_mfma_f32_16x16x32_f16 per tap and block + one
 * v_mfma_scale_f32_16x16x128_f8f6f4 per tap PAIR and block.  x from ape_s32_to_f16m6, w from mx6.pack_conv_weights; y / residual /
 * bias / activation as ape_conv3x3_halo_s32.  NOT bit-identical to the bf16x3 kernels: DESIGN.md 6e prices the difference. */
int ape_conv3x3_halo_mx_supported(const ape_conv_params* params_host);
int ape_conv3x3_halo_mx(const void* x_f16m6, const void* w_f16m6, const float* bias, const void* residual, int res_fmt, void* y,
                        int out_fmt, const ape_conv_params* params_host, void* stream);
int ape_conv3x3_halo_mx_debug(int bits);
int ape_s32_to_f16m6(const void* x_s32, void* y_f16m6, long pixels, int C, void* stream);
/* The ResNet stem in one kernel: Conv2d(3 -> 64, 7x7, stride 2, pad 3) + ReLU + MaxPool2d(3, 2, 1)  (extractors.py:82-85, 111-117).
 * x[B][H][W][4] f32 (RGB + a zero channel), w[64][7][7][4] f32 (the UNPACKED ape_conv2d_nhwc_f32 layout: the kernel splits its
 * own weight fragments), bias[64] or NULL -> y[B][Hp][Wp][64] with Ho = (H - 1) / 2 + 1, Hp = (Ho - 1) / 2 + 1 (same for W).
 * Operands as in ape_conv2d_nhwc_bf16 (nsplit 3 = split-bf16, 1 = plain bf16); the half-resolution 64-channel activation
 * between the convolution and the pool is never written. */
int ape_stem_conv_pool_bf16(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int nsplit,
                            void* stream);
/* ... and straight from the uint8 frames (pipeline/utils.py:421-427,556-560: ToTensor / Normalize of the frame or of the crop): crop o = the
 * Hc x Wc window of frame rects[o][0] at (row rects[o][1], column rects[o][2]) of rgb[n_frames][Hf][Wf][3] (rects NULL: the n = n_frames whole frames); the
 * normalisation (ape_preprocess_u8_nhwc4's arithmetic, div255 as there) runs on the way into LDS, so y is bit for bit
 * ape_stem_conv_pool_bf16(ape_preprocess_u8_nhwc4(...)) and the fp32 image is never written. */
int ape_stem_conv_pool_u8(const uint8_t* rgb, int n_frames, const int* rects, const float* w, const float* bias, float* y, int n, int Hf, int Wf,
                          int Hc, int Wc, int div255, int nsplit, void* stream);
/* 3x3 / stride 1 / pad == dilation in {1,2,4} / Cin % 32 == 0 specialisation of ape_conv2d_nhwc_bf16: the input halo of
 * a 16x16-pixel tile is staged once per 32-channel chunk in LDS and shared by the nine taps (3x less operand traffic).
 * Same arguments, packed weights, numerics and epilogue; ape_conv3x3_halo_supported(params) says whether it applies. */
int ape_conv3x3_halo_supported(const ape_conv_params* params_host);
int ape_conv3x3_halo_bf16(const float* x, const void* w_packed, const float* bias, const float* residual, float* y,
                          const ape_conv_params* params_host, int nsplit, void* stream);

/* ---- HBM-bound glue of the PSPNet / PointNet graphs (NHWC f32, C multiple of 4) -----------------------------
 * nn.MaxPool2d(3, 2, 1)                      DenseFusion/lib/extractors.py:85,117.   y[B][Ho][Wo][C], Ho=(H-1)/2+1 */
int ape_maxpool3x3s2_nhwc_f32(const float* x, float* y, int B, int H, int W, int C, void* stream);
/* nn.AdaptiveAvgPool2d((S,S))                DenseFusion/lib/pspnet.py:15.           y[B][S][S][C] */
int ape_adaptive_avgpool_nhwc_f32(const float* x, float* y, int B, int H, int W, int C, int S, void* stream);
/* The PSP module's pools (pspnet.py:15: sizes 2, 3, 6 of one map) in ONE pass over the map: the bin edges of all sizes cut each axis
 * into at most 12 "atoms" (APE_EINVAL otherwise: use the per-size entry point); every atom is summed once into `workspace`
 * (ape_adaptive_avgpool_multi_workspace_bytes(B, C)), then each bin adds up its atoms.  ys_host / sizes_host: host arrays of nsizes
 * <= 4 device output pointers y_i[B][S_i][S_i][C] and sizes S_i <= 8. */
size_t ape_adaptive_avgpool_multi_workspace_bytes(int B, int C);
int ape_adaptive_avgpool_multi_nhwc_f32(const float* x, float* const* ys_host, const int* sizes_host, int nsizes, int B, int H, int W,
                                        int C, void* workspace, size_t workspace_bytes, void* stream);
/* F.upsample(size=(Ho,Wo), 'bilinear') / nn.Upsample(x2, align_corners=True)   pspnet.py:22,31.
 * Reads x[B][H][W][ldx] channels 0..C-1, writes y[B][Ho][Wo][ldy] channels yoff..yoff+C-1 (a slice of the PSP
 * concat buffer, pspnet.py:22-23); accumulate != 0 adds into y instead of overwriting. */
int ape_bilinear_nhwc_f32(const float* x, float* y, int B, int H, int W, int C, int ldx, int Ho, int Wo, int ldy,
                          int yoff, int align_corners, int accumulate, void* stream);
/* sum over the four PSP priors of F.upsample(prior_s, size=(h,w), 'bilinear') (pspnet.py:22, align_corners=False), z_s[B][s][s][C]
 * for s = 1,2,3,6 -> out[B][h][w][C]; one pass instead of four accumulating ape_bilinear_nhwc_f32 passes, same summation order */
int ape_psp_prior_sum_f32(const float* z1, const float* z2, const float* z3, const float* z6, float* out, int B, int h,
                          int w, int C, void* stream);
/* PSPUpsample (nn.Upsample x2 align_corners=True -> Conv2d 3x3 pad 1 -> PReLU, pspnet.py:27-37) restructured: the 3x3 conv's
 * channel mixing runs at LOW resolution as a 1x1 conv producing z[B][h][w][9*C] (channel = tap*C + c, tap = ky*3+kx, made by
 * ape_conv2d_* from the repacked weights W'[(tap,co)][ci]); this kernel resizes, shifts by the tap, sums, adds bias and
 * applies the activation: out[B][2h][2w][C].  Exact up to fp32 rounding (conv and bilinear resize are both linear). */
int ape_upconv3x3_gather_f32(const float* z, const float* bias, float* out, int B, int h, int w, int C, int act, float alpha,
                             void* stream);
/* torch.gather(emb, 2, choose)               DenseFusion/lib/network.py:100-102.     y[b][i][:] = x[b][index[b][i]][:] */
int ape_gather_rows_f32(const float* x, const int64_t* index, float* y, int B, int rows_in, int n, int C, void* stream);
/* 3x3 patches of nn.Upsample(x2, align_corners=True)(x) at chosen pixels only: x[B][h][w][C] (C % 4 == 0), index[B][n] i64 = pixel
 * y * 2w + x of the up-sampled image -> out[B*n][9*C], column = (ky*3+kx)*C + c, zero where the tap falls outside the 2h x 2w
 * image.  With the up_3 weights read as a [Cout][9*C] matrix (their packed layout already is one) a single 1x1 contraction
 * then evaluates pspnet.py:30-33 at the N points network.py:100-102 keeps, instead of at all Hc*Wc pixels of the crop. */
int ape_ups_patch_gather_f32(const float* x, const int64_t* index, float* out, int B, int h, int w, int C, int n, void* stream);
/* nn.LogSoftmax() over the channel run       DenseFusion/lib/pspnet.py:55.           rows x C, C contiguous */
int ape_log_softmax_rows_f32(const float* x, float* y, long rows, int C, void* stream);
/* nn.AvgPool1d(num_points)                   DenseFusion/lib/network.py:51,65,149,166.  x[B][n][C] -> y[B][C] */
int ape_mean_rows_f32(const float* x, float* y, int B, int n, int C, void* stream);
/* x[rows][3] -> y[rows][4] (w = 0): lets the first 1x1 conv (network.py:42,139) use 16-byte loads */
int ape_pad3to4_f32(const float* x, float* y, long rows, void* stream);
/* conv4_r / conv4_t / conv4_c + sigmoid + index_select(obj)   DenseFusion/lib/network.py:115-126 (and the refiner's
 * conv3_r / conv3_t, :197-204, with wc = bc = NULL): only the selected object's 8 output rows are evaluated.
 * h[B*n][ldh] holds the K-wide inputs of the three heads at channel offsets off_r/off_t/off_c; obj[B] i64;
 * out[B][n][8] = (qw,qx,qy,qz, tx,ty,tz, sigmoid(c)). */
int ape_head_select_f32(const float* h, int ldh, int off_r, int off_t, int off_c, const float* wr, const float* br,
                        const float* wt, const float* bt, const float* wc, const float* bc, const int64_t* obj,
                        float* out, int B, int n, int K, void* stream);

/* ---- pose extraction / composition, device resident ----------------------------------------------------------
 * my_estimator_prediction + get_new_points   DenseFusion/tools/utils.py:7-18,43-86.
 * heads[B][n][8] (from ape_head_select_f32), points4[B][n][4] -> pose[B][7] f64 (qw,qx,qy,qz,tx,ty,tz),
 * which[B] (arg-max confidence, may be NULL), new_points4[B][n][4] (may be NULL). */
int ape_pose_select_f32(const float* heads, const float* points4, double* pose, int* which, float* new_points4,
                        int B, int n, void* stream);
/* my_refined_prediction                      DenseFusion/tools/utils.py:20-40 (+ transformations.py:1254-1278,1320-1363).
 * pose[B][7] f64 is updated in place with the refiner residual ref_r[B][ldr>=4], ref_t[B][ldt>=3] (f32). */
int ape_pose_compose_f64(double* pose, const float* ref_r, int ldr, const float* ref_t, int ldt, int B, void* stream);
/* cloud re-centring of the iterative loop    DenseFusion/tools/eval_ycb.py:205-210. */
int ape_pose_recentre_f32(const float* points4, const double* pose, float* new_points4, int B, int n, void* stream);

/* ---- ADD / ADD-S and the DenseFusion loss forward -------------------------------------------------------------
 * loss_calculation               DenseFusion/lib/loss.py:12-73, lib/loss_refiner.py:12-64, tools/eval_linemod.py:118-130.
 * pred[n][m] = R(pred_r[n]/|pred_r[n]|) . model[m] + pred_t[n] (+ points[n] when points != NULL, loss.py:38);
 * dis[n] = mean_m |pred[n][m] - target[m']|, m' = m, or for symmetric objects the nearest target with the k-NN
 * kernel's float32 arithmetic and lowest-index tie rule (loss.py:42-47); stdv[n] = unbiased std of those norms (may be
 * NULL); pred_out[N][M][3] is written only when non-NULL.  pred_r[N][4], pred_t[N][3], points[N][3], model/target[M][3]. */
int ape_adds_dis_f32(const float* pred_r, const float* pred_t, const float* points, const float* model,
                     const float* target, int N, int M, int symmetric, float* pred_out, float* dis, float* stdv,
                     void* stream);
/* The evaluation form (DenseFusion/tools/eval_linemod.py:118-130, experiments/eval.py:75-95) for a BATCH of objects in one launch: pose b
 * (pred_r[b] unnormalised wxyz, pred_t[b]) places ITS model cloud model[b][M][3] and is scored against ITS target[b][M][3]; symmetric:
 * every predicted point against its nearest target (the k-NN kernel's arithmetic and tie rule, several lanes per point like ape_knn_f32).
 * dis[b] = ADD / ADD-S in the clouds' unit, bit for bit ape_adds_dis_f32's value for that object alone. */
int ape_adds_dis_batched_f32(const float* pred_r, const float* pred_t, const float* model, const float* target, int B, int M, int symmetric,
                             float* workspace /* B * M floats */, float* dis, void* stream);
/* loss = mean((dis + 2 std) c - w log c), which = argmax c (first), out9 = (loss, dis[which], pred_r[which][0..3],
 * pred_t[which] + points[which])   loss.py:50-59 */
int ape_adds_select_f32(const float* dis, const float* stdv, const float* pred_c, const float* pred_r,
                        const float* pred_t, const float* points, int N, float w, float* out9, int* which,
                        void* stream);
/* out[i] = (pts[i] - t) . ori_base(q/|q|), qt7 = (q[4], t[3]) on the device   loss.py:61-69, loss_refiner.py:51-60 */
int ape_recentre_qt_f32(const float* pts, const float* qt7, float* out, int n, void* stream);

/* ---- segmentation post-processing, crop / point selection (byte and index work, bit-exact) ----------------------
 * softmax(+softmax) / argmax                 pipeline/utils.py:429-435 (predict's softmax activation, create_labels.py:23,
 * then F.softmax again, then torch.argmax).  logits[npix][ld] f32 (first C channels) -> label[npix] u8,
 * score[npix] f32 = probability of the arg-max class after one (double_softmax=0) or two softmaxes. */
int ape_seg_argmax_f32(const float* logits, int ld, int C, uint8_t* label, float* score, long npix,
                       int double_softmax, void* stream);
/* Fused segmentation head: the 64 -> C final 1x1 conv (pspnet.py:53-55, first C rows) + softmax(+softmax) + argmax in one
 * pass over feat[npix][64] f32, without the logits tensor.  C <= 16.  Against ape_conv2d_* followed by ape_seg_argmax_f32: the head
 * sums the 64 channels on the matrix cores (another fp32 order) and takes its exponentials / its reciprocal from v_exp_f32 / v_rcp_f32
 * (1 ulp) where ape_seg_argmax_f32 uses expf and IEEE divisions, so scores agree to ~1e-6 and labels agree except where two classes'
 * final float32 probabilities are within 1 ulp of each other (the tie band torch.argmax resolves to the lowest index; for a top
 * probability below 0.5 one ulp of p decides exp(p - pmax) == 1).  Every fused form of this head in the library (this entry,
 * ape_conv3x3_halo_seghead_bf16, ape_upconv3x3_fused_seghead_s32) runs the SAME code (csrc/seg_head.h) and they agree bit for bit;
 * tests/test_gpu_segpost.py and bench.py's parity block count label differences outside the band only. */
int ape_seg_head_f32(const float* feat, const float* w, const float* bias, int C, uint8_t* label, float* score, long npix,
                     int double_softmax, void* stream);
/* np.unique counts + cv2.connectedComponents(8) + best mean-probability component + mask + get_bbox
 *   pipeline/utils.py:437-469, DenseFusion/datasets/myDatasetAugmented/dataset.py:338-380.
 * label/score[B][H][W] -> objmap[B][H][W] u8 (class id inside the winning component of that class, else 0; the
 * reference's per-class mask is (objmap == cls) * 255) and det[B][C][5] i32 = (valid, rmin, rmax, cmin, cmax).
 * Classes with <= min_pixels pixels are skipped (reference: 100). */
size_t ape_seg_components_workspace_bytes(int B, int H, int W, int C);
int ape_seg_components(const uint8_t* label, const float* score, uint8_t* objmap, int* det, int B, int H, int W,
                       int C, int min_pixels, void* workspace, size_t workspace_bytes, void* stream);
/* Same with the component score selectable: APE_SEG_SCORE_MEAN = mean probability (pipeline/utils.py:456-462),
 * APE_SEG_SCORE_SUM = summed probability, the rule of do_cca (background_subtraction/utils.py:199-222: label 0/1,
 * biggest = arg max_u sum(max-prob[labels == u]), first component wins ties). */
#define APE_SEG_SCORE_MEAN 0
#define APE_SEG_SCORE_SUM 1
int ape_seg_components_scored(const uint8_t* label, const float* score, uint8_t* objmap, int* det, int B, int H, int W,
                              int C, int min_pixels, int score_mode, void* workspace, size_t workspace_bytes, void* stream);
/* Background-subtraction network input   background_subtraction/utils.py:721-828 (get_mask_prediction's per-frame block):
 * f_rgb/b_rgb[B][H][W][3] u8 and f_depth/b_depth[B][H][W] u16 of the object frame and of the empty-scene frame taken from
 * the same view point -> the 7 channels |dRGB|, |dHSV| (Pillow's 8-bit HSV), |d depth| (gate [min,max] per frame, :747-763,
 * gate_min_max[B][2] f64 on the DEVICE), each cast to uint8 as numpy does (:811, the depth channel wraps mod 256), then
 * ToTensor + Normalize(mean7, std7) (:818-819; HOST pointers).  out[B][H][W][8] f32 NHWC (channel 7 = 0);
 * diff_or_null[B][H][W][7] receives the uint8 channels when not NULL. */
int ape_bgsub_features_f32(const uint8_t* f_rgb, const uint8_t* b_rgb, const uint16_t* f_depth, const uint16_t* b_depth,
                           const double* gate_min_max, const float* mean7_host, const float* std7_host, float* out,
                           uint8_t* diff_or_null, int B, int H, int W, void* stream);
/* trust checks of the relabelling loop   label_generator/create_labels.py:166-196.  objmap from ape_seg_components
 * (min_pixels = 0), cls = target class; counts[B][6] u32 (zeroed by the caller) = (bs&pred, bs&!pred, depth&pred,
 * depth&!pred, centre&pred, centre&!pred) with the depth gate [min,max] per frame (:106-112) and the 30/50 px centre window. */
int ape_label_trust_counts(const uint8_t* objmap, int cls, const uint8_t* bs_label_or_null, const uint16_t* depth,
                           const float* gate_min_max, int B, int H, int W, int cut0, int cut1, unsigned int* counts,
                           void* stream);
/* choose = mask[rmin:rmax, cmin:cmax].flatten().nonzero(); > N -> ordered subset, <= N -> wrap pad
 *   pipeline/utils.py:524-539.  objects[n][6] i32 = (frame, cls, rmin, rmax, cmin, cmax);
 * choose[n][N] i64 indices inside the crop; n_cand[n] (0 => object dropped, :530-531); cand: scratch [n][cand_stride]. */
int ape_choose_points(const uint8_t* objmap, const uint16_t* depth, const int* objects, int n, int H, int W, int N,
                      unsigned int seed, int* cand, long cand_stride, int64_t* choose, int* n_cand, void* stream);
/* the same with the sampling seed read from DEVICE memory when the kernel runs: a launch captured in a HIP graph then follows a seed that
 * the host updates between replays (pipeline/utils.py FramePipeline(pose_graphs=True)) */
int ape_choose_points_dseed(const uint8_t* objmap, const uint16_t* depth, const int* objects, int n, int H, int W, int N,
                            const unsigned int* seed_device, int* cand, long cand_stride, int64_t* choose, int* n_cand, void* stream);
/* float32 pin-hole back-projection          pipeline/utils.py:542-553.  points4[n][N][4] = (x, y, z, 0) */
int ape_backproject_f32(const uint16_t* depth, const int* objects, const int64_t* choose, float* points4, int n,
                        int H, int W, int N, float fx, float fy, float ppx, float ppy, float depth_scale, void* stream);
/* ToTensor(/255)+Normalize (pipeline/utils.py:421-427, div255=1) or raw-scale crop normalisation (:559-560, div255=0):
 * rgb[B][H][W][3] u8, rects[n][3] i32 = (frame, row0, col0) -> out[n][Hc][Wc][4] f32 (channel 3 = 0). */
int ape_preprocess_u8_nhwc4(const uint8_t* rgb, const int* rects, float* out, int n, int H, int W, int Hc, int Wc,
                            int div255, void* stream);

/* ---- pose-label path: point clouds in float64 (pc_reconstruction/open3d_utils.py, open3d 0.9 semantics) -------
 * Workspace for any of the calls below on clouds of up to n points (H*W for ape_surface_points_f64). */
size_t ape_pc_workspace_bytes(int n);
/* get_surface's pixel loop        pc_reconstruction/open3d_utils.py:171-192: pixels with label != 0 and depth != 0, in raster
 * order, back-projected (mm, no depth scale) and moved to the robot frame by T (row-major 4x4, HOST pointer).
 * points[H*W][3] capacity, *n_out (device int). */
int ape_surface_points_f64(const uint8_t* label, const uint16_t* depth, int H, int W, double fx, double fy, double ppx,
                           double ppy, const double* T16_host, double* points, int* n_out, void* ws, size_t ws_bytes,
                           void* stream);
/* PointCloud.transform            in place; normals (may be NULL) get the rotation part */
int ape_transform_points_f64(double* pts, double* normals_or_null, int n, const double* T16_host, void* stream);
/* PointCloud.voxel_down_sample    open3d_utils.py:21,198,157: per-voxel mean, output ordered by voxel key */
int ape_voxel_down_sample_f64(const double* pts, int n, double voxel, double* out, int* n_out, void* ws, size_t ws_bytes,
                              void* stream);
/* Uniform search grid over a cloud (cell >= every radius later asked of it): caller-owned sorted[n][3], keys[n],
 * order[n], origin3[3].  Replaces open3d's KDTreeFlann for the bounded-radius searches of the path. */
int ape_grid_build_f64(const double* pts, int n, double cell, double* sorted, unsigned long long* keys, unsigned* order,
                       double* origin3, void* ws, size_t ws_bytes, void* stream);
/* remove_radius_outlier's neighbour count (d < radius, self included)   open3d_utils.py:203 */
int ape_grid_radius_count_f64(const double* sorted, const unsigned long long* keys, const unsigned* order, const double* origin3,
                              int n, double cell, const double* q, int nq, double radius, int* count, void* stream);
/* registration_icp's correspondence search: nearest target point within max_dist, idx = -1 if none   :98-117 */
int ape_grid_nn1_f64(const double* sorted, const unsigned long long* keys, const unsigned* order, const double* origin3,
                     int n, double cell, const double* q, int nq, double max_dist, int* idx, double* dist2, void* stream);
/* estimate_normals(KDTreeSearchParamHybrid(radius, max_nn))   open3d_utils.py:25-27; normals oriented towards +z */
int ape_grid_normals_f64(const double* sorted, const unsigned long long* keys, const unsigned* order, const double* origin3,
                         int n, double cell, const double* q, int nq, double radius, int max_nn, double* normals, void* stream);
/* remove_statistical_outlier's per-point mean distance to its k nearest neighbours (self included)   :208-211 */
int ape_knn_mean_dist_f64(const double* pts, int n, int k, double* mean, void* stream);
/* The same means for the points of a grid-indexed cloud (the grid's own points are the queries), searched shell by shell through the
 * uniform grid instead of over all pairs: bitwise equal to ape_knn_mean_dist_f64 for any cell size; fastest when a cell holds a few
 * points (cell ~ the radius that contains k neighbours).  k <= 64, k <= n. */
int ape_grid_knn_mean_dist_f64(const double* sorted, const unsigned long long* keys, const unsigned* order, const double* origin3,
                               int n, double cell, int k, double* mean, void* stream);
/* One-pass ICP reductions (bitwise reproducible).  kind 0: point-to-point (Umeyama) out[17] = count, sum d^2, sum s[3],
 * sum t[3], sum s_a t_b[9];  kind 1: point-to-plane out[29] = count, sum d^2, upper triangle of J^T J[21], J^T r[6];
 * kind 2: moments of src, out[9] = sum p[3], sum p_a p_b upper triangle[6] (get_center / compute_mahalanobis_distance). */
int ape_icp_sums_f64(int kind, const double* src, const double* tgt, const double* tgt_normals, const int* corr,
                     const double* dist2, int n, double* out, void* ws, size_t ws_bytes, void* stream);
/* registration_icp's LOOP on the device (open3d_utils.py:96-117; open3d 0.9 RegistrationICP): `n_iter` iterations of
 * [move src by the pending update + correspondence search] -> partial sums -> [reduce + step: fitness / rmse / convergence test, Umeyama
 * 3x3 SVD (kind 0) or 6x6 solve (kind 1), T <- update . T], three launches each, enqueued at once; every launch is a no-op once
 * state[0] != 0.  first_call = 1 prepends the evaluation before open3d's loop and the step that computes the first update.  `src` is the
 * source already moved by the initial guess, updated in place.  state[40] doubles on the device: [0] done, [1] updates applied,
 * [2] fitness, [3] inlier rmse, [4] correspondences, [5..20] T row major, [21..36] last update, [37] stop reason (1 converged,
 * 2 too few correspondences, 3 iteration limit), [38] internal.  Before the first call of a registration (first_call = 1) the caller
 * zeroes it and writes the initial T; one copy of it back per call tells whether to enqueue more.  Grid arguments as above
 * (cell >= max_dist); ws: 512 x 29 doubles.  sums[29], corr[ns], dist2[ns]: caller-owned scratch. */
int ape_icp_run_f64(int kind, const double* sorted, const unsigned long long* keys, const unsigned* order, const double* origin3, int n,
                    double cell, double* src, int ns, const double* tgt, const double* tgt_normals, double max_dist, double rel_fitness,
                    double rel_rmse, int max_iteration, int n_iter, int first_call, int* corr, double* dist2, double* sums, double* state,
                    void* ws, size_t ws_bytes, void* stream);
/* compute_mahalanobis_distance    mean_cinv12_host = (mean[3], inverse covariance[9]) */
int ape_mahalanobis_f64(const double* pts, int n, const double* mean_cinv12_host, double* out, void* stream);
/* ordered row selection (outlier filters): out = pts[keep != 0], sel_idx = kept indices, *n_out on the device */
int ape_select_points_f64(const double* pts, const uint8_t* keep, int n, double* out, int* sel_idx, int* n_out, void* ws,
                          size_t ws_bytes, void* stream);

/* ---- BATCHED forms of the point-cloud kernels above (pose-label path, SURVEY.md 8e: the (object, direction) chains of
 * pc_reconstruction/create_pointcloud.py:276-312 are independent of each other).  ONE launch advances up to 16 clouds: blockIdx.y selects
 * the cloud's argument record, blockIdx.x walks that cloud's own grid -- the same device code with the same per-cloud grid sizes as the
 * one-cloud entry points, hence bit-identical results per cloud.  Per-cloud arguments are HOST arrays (of device pointers / sizes) of
 * length nb <= 16; clouds with n = 0 are skipped.  hipCUB's radix sort becomes a single-workgroup LDS sort per cloud (the cell keys re-coded
 * as a 32-bit lexicographic rank, packed with the index: stable, the same permutation), its select / scan calls a single-workgroup ordered
 * compaction / scan per cloud. */
size_t ape_pc_batch_workspace_bytes(int nb, long n_total);
/* label[c] / depth[c] [H][W]; intr4_host [nb][4] = fx, fy, ppx, ppy; T16_host [nb][16]; points[c] capacity H*W rows;
 * n_out [nb] on the device; pix_ws: nb * H * W ints of scratch */
int ape_surface_points_batch_f64(int nb, const uint8_t* const* label, const uint16_t* const* depth, int H, int W, const double* intr4_host,
                                 const double* T16_host, double* const* points, int* n_out, int* pix_ws, void* stream);
int ape_voxel_down_sample_batch_f64(int nb, const double* const* pts, const int* n, double voxel, double* const* out, int* n_out, void* ws,
                                    size_t ws_bytes, void* stream);
int ape_grid_build_batch_f64(int nb, const double* const* pts, const int* n, double cell, double* const* sorted, unsigned long long* const* keys,
                             unsigned* const* order, double* const* origin3, void* ws, size_t ws_bytes, void* stream);
/* op 0: ape_grid_radius_count_f64 (count[c][nq[c]]); op 1: ape_grid_normals_f64 (normals[c][nq[c]][3], max_nn); op 2:
 * ape_grid_knn_mean_dist_f64 (mean[c][gn[c]], k; q / nq unused) */
int ape_grid_query_batch_f64(int op, int nb, const double* const* sorted, const unsigned long long* const* keys, const unsigned* const* order,
                             const double* const* origin3, const int* gn, double cell, const double* const* q, const int* nq, double radius,
                             int max_nn_or_k, int* const* count, double* const* normals, double* const* mean, void* stream);
/* ape_select_points_f64 with the keep rule evaluated on the device: mode 0 count[c][i] > thr_count (RemoveRadiusOutliers), mode 1
 * mean[c][i] > 0 && mean[c][i] < thr_mean_host[c] (RemoveStatisticalOutliers); sel_ws: sum n ints; n_out [nb] on the device */
int ape_select_points_batch_f64(int mode, int nb, const double* const* pts, const int* n, const int* const* count, int thr_count,
                                const double* const* mean, const double* thr_mean_host, double* const* out, int* n_out, int* sel_ws, void* stream);
/* ape_icp_sums_f64(kind 2): out9 [nb][9] on the device; ws: nb * 512 * 9 doubles */
int ape_moments_batch_f64(int nb, const double* const* pts, const int* n, double* out9, void* ws, size_t ws_bytes, void* stream);
int ape_mahalanobis_batch_f64(int nb, const double* const* pts, const int* n, const double* mc12_host, double* const* out, void* stream);
int ape_transform_points_batch_f64(int nb, double* const* pts, double* const* normals, const int* n, const double* T16_host, void* stream);
/* out[c] = [a[c] | b[c]] (b may be NULL: copies) */
int ape_concat_points_batch_f64(int nb, const double* const* a, const int* na, const double* const* b, const int* nb_rows, double* const* out,
                                void* stream);
/* ape_icp_run_f64 for nb registrations of one kind advancing together (state[c]: 40 doubles on the device each); ws: nb * 512 * 29 doubles */
int ape_icp_run_batch_f64(int kind, int nb, const double* const* sorted, const unsigned long long* const* keys, const unsigned* const* order,
                          const double* const* origin3, const int* gn, double cell, double* const* src, const int* ns, const double* const* tgt,
                          const double* const* tgt_normals, double max_dist, double rel_fitness, double rel_rmse, int max_iteration, int n_iter,
                          int first_call, int* const* corr, double* const* dist2, double* const* sums, double* const* state, void* ws,
                          size_t ws_bytes, void* stream);

/* ape_conv3x3_halo_bf16 for a 64-channel layer (the segmentor's up_3, pspnet.py:51) with ape_seg_head_f32 fused into its
 * epilogue: the [B][H][W][64] activation is never written, label[B][H][W] u8 / score[B][H][W] f32 are (bit-identical to the
 * unfused pair).  params->Cout must be 64, no residual; params->ups as in ape_conv3x3_halo_bf16. */
int ape_conv3x3_halo_seghead_bf16(const float* x, const void* w_packed, const float* bias, const ape_conv_params* params, int nsplit,
                                  const float* head_w, const float* head_b, int C, uint8_t* label, float* score, int double_softmax,
                                  void* stream);


/* ---- training step: the backward kernels behind DenseFusion/tools/train.py:205-238 -------------------
 * `loss.backward()` / `dis.backward()` there run torch autograd over cuDNN; each entry below is one backward rule of the ops
 * the estimator / refiner / losses are built from (DenseFusion/lib/{network,pspnet,extractors,loss,loss_refiner}.py). */
/* Convolution weight gradient (nn.Conv2d / Conv1d / Linear backward w.r.t. weight): x, dy NHWC as in ape_conv2d_nhwc_f32 with
 * the same ape_conv_params; dw[Cout][KH][KW][Cin] in the packed forward layout (Cin % 4 == 0).  The input gradient is the
 * forward kernel applied to dy with the flipped, transposed weights. */
size_t ape_conv2d_wgrad_workspace_bytes(const ape_conv_params* params);
int ape_conv2d_wgrad_nhwc_f32(const float* x, const float* dy, float* dw, const ape_conv_params* params, void* workspace,
                              size_t workspace_bytes, void* stream);
/* the same sums written as the reference's parameter tensor: dw[Cout][cin_param][KH][KW] (cin_param <= Cin: the zero channels that pad x
 * to a multiple of 4 drop out) -- what `weight.grad` holds after train.py:221 */
int ape_conv2d_wgrad_param_f32(const float* x, const float* dy, float* dw, const ape_conv_params* params, int cin_param, void* workspace,
                               size_t workspace_bytes, void* stream);
/* dx = dy * act'(ref): ref = the OUTPUT for APE_ACT_RELU / APE_ACT_SIGMOID, the INPUT for APE_ACT_PRELU (nn.ReLU, nn.PReLU
 * pspnet.py:33, torch.sigmoid network.py:117); in place allowed */
int ape_act_bwd_f32(const float* dy, const float* ref, float* dx, long n, int act, float alpha, void* stream);
/* nn.PReLU with one slope (pspnet.py:33) as a separate op of the training forward, and its slope gradient
 * dalpha[0] = sum dy * x * [x <= 0]; scratch1024: 1024 floats */
int ape_prelu_f32(const float* x, float* y, long n, float alpha, void* stream);
int ape_prelu_dalpha_f32(const float* dy, const float* x, float* dalpha, long n, float* scratch1024, void* stream);
/* bias gradient: out[C] = column sums of x[rows][ld] at channel offset off; scratch: 64 * C floats */
int ape_colsum_f32(const float* x, float* out, long rows, int C, int ld, int off, float* scratch, void* stream);
/* nn.MaxPool2d(3, 2, 1) backward (extractors.py:91), ATen's first-maximum tie rule */
int ape_maxpool3x3s2_bwd_nhwc_f32(const float* x, const float* dy, float* dx, int B, int H, int W, int C, void* stream);
/* nn.AdaptiveAvgPool2d((S, S)) backward (pspnet.py:16) */
int ape_adaptive_avgpool_bwd_nhwc_f32(const float* dy, float* dx, int B, int H, int W, int C, int S, void* stream);
/* F.upsample(..., mode='bilinear') backward, both conventions of ape_bilinear_nhwc_f32 (pspnet.py:22, :37) */
int ape_bilinear_bwd_nhwc_f32(const float* dy, float* dx, int B, int H, int W, int C, int Ho, int Wo, int align_corners,
                              void* stream);
/* nn.LogSoftmax backward per row (pspnet.py:55) */
int ape_log_softmax_bwd_rows_f32(const float* dy, const float* y, float* dx, long rows, int C, void* stream);
/* torch.gather backward (network.py:100-102): dx[B][rows_in][C] = scatter-add of dy[B][n][C] at index[B][n] */
int ape_scatter_add_rows_f32(const float* dy, const int64_t* index, float* dx, int B, int rows_in, int n, int C, void* stream);
/* AvgPool1d(num_points) backward (network.py:64): dx[B][n][C] = dy[B][C] / n */
int ape_mean_rows_bwd_f32(const float* dy, float* dx, int B, int n, int C, void* stream);
/* Gradient of Loss (full = 1; loss.py:50-53) or of Loss_refine's dis (full = 0; loss_refiner.py:47) w.r.t. pred_r[N][4],
 * pred_t[N][3] (and pred_c[N]); dis / stdv from ape_adds_dis_f32, gscale = DEVICE scalar upstream gradient */
int ape_adds_grad_f32(const float* pred_r, const float* pred_t, const float* points, const float* model, const float* target,
                      const float* pred_c, const float* dis, const float* stdv, const float* gscale, int N, int M, int symmetric,
                      int full, float w, float* d_r, float* d_t, float* d_c, void* stream);
/* optim.Adam(lr) update of one flat parameter buffer (train.py:109,113; torch defaults betas (0.9, 0.999), eps 1e-8) */
int ape_adam_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1,
                      float beta2, float eps, int step, float weight_decay, void* stream);

/* The same update for n parameter buffers at once (64 per launch): optimizer.step() of train.py:231-232 over all parameters of the
 * estimator / refiner.  bc1 = 1 - beta1^step, bc2_sqrt = sqrt(1 - beta2^step) of each buffer's own step count. */
typedef struct ape_adam_job {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    long n;
    float bc1, bc2_sqrt;
} ape_adam_job;
int ape_adam_step_multi_f32(int n, const ape_adam_job* jobs_host, float lr, float beta1, float beta2, float eps, float weight_decay,
                            void* stream);
/* After an optimizer step the conv kernels' weight operands are stale: one launch re-packs many parameters ([Cout][Cin][taps] f32, the
 * reference's nn.Conv2d / Conv1d / Linear layout) into f32 [N][taps][C4] (C4 = C rounded up to 4, zero channels) and, when dst_bf16 is
 * set, the split-bf16 planes of ape_pack_weights_bf16.  transpose = 0: N = Cout, C = Cin (forward operand); 1: N = Cin, C = Cout, taps
 * reversed (the flipped, transposed weights whose forward conv is the input gradient).  jobs: DEVICE array. */
typedef struct ape_pack_job {
    const float* src;
    float* dst_f32;
    void* dst_bf16;
    int cout, cin, taps, transpose;
    int src_ld;            /* elements between two output-channel rows of src (cin * taps when src is the whole parameter; larger for a
                              column block of a wider one, network.py:104-121 feeds conv1_r/t/c two column blocks of one weight) */
    int reserved;
} ape_pack_job;
int ape_pack_train_weights(int n, const ape_pack_job* jobs_device, long max_elems, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* APE_HIP_H */
