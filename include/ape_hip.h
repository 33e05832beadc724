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

#ifdef __cplusplus
}
#endif
#endif /* APE_HIP_H */
