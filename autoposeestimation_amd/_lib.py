"""ctypes binding of libape_hip.so (the C ABI declared in include/ape_hip.h).

There is NO CPU fallback: if the shared library is missing or a tensor is not on the GPU the call
raises.  torch is used only for device memory and streams (tensor.data_ptr(), current stream)."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("APE_HIP_LIB", os.path.join(_HERE, "libape_hip.so"))   # env override: kernel A/B experiments

_c = ctypes
_P, _I, _F, _L, _D = _c.c_void_p, _c.c_int, _c.c_float, _c.c_int64, _c.c_double

# symbol -> argtypes (restype is int unless listed in _RESTYPES); mirrors include/ape_hip.h one to one
SIGNATURES = {
    "ape_abi_version": [],
    "ape_last_error": [],
    "ape_knn_f32": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "ape_conv2d_nhwc_f32": [_P, _P, _P, _P, _P, _P, _P],
    "ape_packed_weights_bf16_elems": [_I, _I],
    "ape_pack_weights_bf16": [_P, _P, _I, _I, _P],
    "ape_conv2d_nhwc_bf16": [_P, _P, _P, _P, _P, _P, _I, _P],
    "ape_conv_gemm_supported": [_P],
    "ape_conv_gemm_bf16": [_P, _P, _P, _P, _P, _P, _I, _I, _P],
    "ape_pack_weights_s32k": [_P, _P, _I, _I, _P],
    "ape_convert_s32": [_P, _P, _c.c_long, _I, _I, _P],
    "ape_conv_gemm_s32_supported": [_P],
    "ape_conv_gemm_s32_debug": [_I],
    "ape_conv3x3_halo_s32_debug": [_I],
    "ape_conv_gemm_s32": [_P, _P, _P, _P, _I, _P, _I, _P, _P],
    "ape_conv_gemm_s32_per_image": [_P, _P, _c.c_long, _P, _P, _I, _P, _P],
    "ape_psp_fold_operands": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "ape_conv_gemm_bf16_fmt": [_P, _P, _P, _P, _P, _I, _P, _I, _I, _P],
    "ape_conv_gemm_splitk_workspace_bytes": [_P],
    "ape_conv_gemm_bf16_splitk": [_P, _P, _P, _P, _P, _P, _I, _P, _c.c_size_t, _P],
    "ape_conv_gemm_bf16_multi": [_I, _P, _P, _P, _P, _P, _I, _P],
    "ape_adaptive_avgpool_multi_nhwc_fmt": [_P, _I, _P, _P, _I, _I, _I, _I, _I, _P, _c.c_size_t, _P],
    "ape_adaptive_avgpool_multi_nhwc_ld": [_P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _P, _c.c_size_t, _P],
    "ape_upconv3x3_gather_fmt": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P],
    "ape_upconv3x3_gather_ex": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _I, _P],
    "ape_upconv3x3_gather_strip_rows": [_I],
    "ape_upconv3x3_fused_supported": [_I, _I, _I, _I],
    "ape_upconv3x3_fused_debug": [_I],
    "ape_upconv3x3_fused_stamps": [_P],
    "ape_upconv3x3_fused_s32": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _I, _P],
    "ape_upconv3x3_fused_seghead_s32": [_P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P, _P, _I, _P, _P, _I, _P],
    "ape_conv3x3_halo_s32_supported": [_P],
    "ape_conv3x3_halo_s32": [_P, _P, _P, _P, _I, _P, _I, _P, _P],
    "ape_conv3x3_halo_mx_supported": [_P],
    "ape_conv3x3_halo_mx": [_P, _P, _P, _P, _I, _P, _I, _P, _P],
    "ape_conv3x3_halo_mx_debug": [_I],
    "ape_s32_to_f16m6": [_P, _P, _c.c_long, _I, _P],
    "ape_stem_conv_pool_bf16": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "ape_stem_conv_pool_u8": [_P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "ape_conv3x3_halo_supported": [_P],
    "ape_conv3x3_halo_bf16": [_P, _P, _P, _P, _P, _P, _I, _P],
    "ape_maxpool3x3s2_nhwc_f32": [_P, _P, _I, _I, _I, _I, _P],
    "ape_adaptive_avgpool_nhwc_f32": [_P, _P, _I, _I, _I, _I, _I, _P],
    "ape_adaptive_avgpool_multi_workspace_bytes": [_I, _I],
    "ape_adaptive_avgpool_multi_nhwc_f32": [_P, _P, _P, _I, _I, _I, _I, _I, _P, _c.c_size_t, _P],
    "ape_bilinear_nhwc_f32": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "ape_psp_prior_sum_f32": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "ape_upconv3x3_gather_f32": [_P, _P, _P, _I, _I, _I, _I, _I, _F, _P],
    "ape_gather_rows_f32": [_P, _P, _P, _I, _I, _I, _I, _P],
    "ape_ups_patch_gather_f32": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "ape_log_softmax_rows_f32": [_P, _P, _c.c_long, _I, _P],
    "ape_mean_rows_f32": [_P, _P, _I, _I, _I, _P],
    "ape_pad3to4_f32": [_P, _P, _c.c_long, _P],
    "ape_head_select_f32": [_P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "ape_pose_select_f32": [_P, _P, _P, _P, _P, _I, _I, _P],
    "ape_pose_compose_f64": [_P, _P, _I, _P, _I, _I, _P],
    "ape_pose_recentre_f32": [_P, _P, _P, _I, _I, _P],
    "ape_adds_dis_f32": [_P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P],
    "ape_adds_dis_batched_f32": [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P],
    "ape_adds_select_f32": [_P, _P, _P, _P, _P, _P, _I, _F, _P, _P, _P],
    "ape_recentre_qt_f32": [_P, _P, _P, _I, _P],
    "ape_seg_argmax_f32": [_P, _I, _I, _P, _P, _c.c_long, _I, _P],
    "ape_seg_head_f32": [_P, _P, _P, _I, _P, _P, _c.c_long, _I, _P],
    "ape_seg_components_workspace_bytes": [_I, _I, _I, _I],
    "ape_seg_components": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _c.c_size_t, _P],
    "ape_conv3x3_halo_seghead_bf16": [_P, _P, _P, _P, _I, _P, _P, _I, _P, _P, _I, _P],
    "ape_seg_components_scored": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _c.c_size_t, _P],
    "ape_bgsub_features_f32": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "ape_conv2d_wgrad_workspace_bytes": [_P],
    "ape_conv2d_wgrad_nhwc_f32": [_P, _P, _P, _P, _P, _c.c_size_t, _P],
    "ape_act_bwd_f32": [_P, _P, _P, _c.c_long, _I, _F, _P],
    "ape_prelu_f32": [_P, _P, _c.c_long, _F, _P],
    "ape_prelu_dalpha_f32": [_P, _P, _P, _c.c_long, _P, _P],
    "ape_colsum_f32": [_P, _P, _c.c_long, _I, _I, _I, _P, _P],
    "ape_maxpool3x3s2_bwd_nhwc_f32": [_P, _P, _P, _I, _I, _I, _I, _P],
    "ape_adaptive_avgpool_bwd_nhwc_f32": [_P, _P, _I, _I, _I, _I, _I, _P],
    "ape_bilinear_bwd_nhwc_f32": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "ape_log_softmax_bwd_rows_f32": [_P, _P, _P, _c.c_long, _I, _P],
    "ape_scatter_add_rows_f32": [_P, _P, _P, _I, _I, _I, _I, _P],
    "ape_mean_rows_bwd_f32": [_P, _P, _I, _I, _I, _P],
    "ape_adds_grad_f32": [_P] * 9 + [_I, _I, _I, _I, _F, _P, _P, _P, _P],
    "ape_adam_step_f32": [_P, _P, _P, _P, _c.c_long, _F, _F, _F, _F, _I, _F, _P],
    "ape_adam_step_multi_f32": [_I, _P, _F, _F, _F, _F, _F, _P],
    "ape_pack_train_weights": [_I, _P, _c.c_long, _P],
    "ape_conv2d_wgrad_param_f32": [_P, _P, _P, _P, _I, _P, _c.c_size_t, _P],
    "ape_label_trust_counts": [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P],
    "ape_choose_points": [_P, _P, _P, _I, _I, _I, _I, _c.c_uint, _P, _c.c_long, _P, _P, _P],
    "ape_choose_points_dseed": [_P, _P, _P, _I, _I, _I, _I, _P, _P, _c.c_long, _P, _P, _P],
    "ape_backproject_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _F, _F, _F, _P],
    "ape_preprocess_u8_nhwc4": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "ape_pc_workspace_bytes": [_I],
    "ape_surface_points_f64": [_P, _P, _I, _I, _D, _D, _D, _D, _P, _P, _P, _P, _c.c_size_t, _P],
    "ape_transform_points_f64": [_P, _P, _I, _P, _P],
    "ape_voxel_down_sample_f64": [_P, _I, _D, _P, _P, _P, _c.c_size_t, _P],
    "ape_grid_build_f64": [_P, _I, _D, _P, _P, _P, _P, _P, _c.c_size_t, _P],
    "ape_grid_radius_count_f64": [_P, _P, _P, _P, _I, _D, _P, _I, _D, _P, _P],
    "ape_grid_nn1_f64": [_P, _P, _P, _P, _I, _D, _P, _I, _D, _P, _P, _P],
    "ape_grid_normals_f64": [_P, _P, _P, _P, _I, _D, _P, _I, _D, _I, _P, _P],
    "ape_knn_mean_dist_f64": [_P, _I, _I, _P, _P],
    "ape_grid_knn_mean_dist_f64": [_P, _P, _P, _P, _I, _D, _I, _P, _P],
    "ape_icp_sums_f64": [_I, _P, _P, _P, _P, _P, _I, _P, _P, _c.c_size_t, _P],
    "ape_icp_run_f64": [_I, _P, _P, _P, _P, _I, _D, _P, _I, _P, _P, _D, _D, _D, _I, _I, _I, _P, _P, _P, _P, _P, _c.c_size_t, _P],
    "ape_mahalanobis_f64": [_P, _I, _P, _P, _P],
    "ape_select_points_f64": [_P, _P, _I, _P, _P, _P, _P, _c.c_size_t, _P],
    # batched forms (blockIdx.y = cloud; per-cloud arguments as host arrays of pointers / sizes)
    "ape_pc_batch_workspace_bytes": [_I, _c.c_long],
    "ape_surface_points_batch_f64": [_I, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P],
    "ape_voxel_down_sample_batch_f64": [_I, _P, _P, _D, _P, _P, _P, _c.c_size_t, _P],
    "ape_grid_build_batch_f64": [_I, _P, _P, _D, _P, _P, _P, _P, _P, _c.c_size_t, _P],
    "ape_grid_query_batch_f64": [_I, _I, _P, _P, _P, _P, _P, _D, _P, _P, _D, _I, _P, _P, _P, _P],
    "ape_select_points_batch_f64": [_I, _I, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P],
    "ape_moments_batch_f64": [_I, _P, _P, _P, _P, _c.c_size_t, _P],
    "ape_mahalanobis_batch_f64": [_I, _P, _P, _P, _P, _P],
    "ape_transform_points_batch_f64": [_I, _P, _P, _P, _P, _P],
    "ape_concat_points_batch_f64": [_I, _P, _P, _P, _P, _P, _P],
    "ape_icp_run_batch_f64": [_I, _I, _P, _P, _P, _P, _P, _D, _P, _P, _P, _P, _D, _D, _D, _I, _I, _I, _P, _P, _P, _P, _P, _c.c_size_t, _P],
}


class AdamJob(_c.Structure):
    """Mirror of `ape_adam_job` (include/ape_hip.h)."""
    _fields_ = [("param", _c.c_void_p), ("grad", _c.c_void_p), ("exp_avg", _c.c_void_p), ("exp_avg_sq", _c.c_void_p), ("n", _c.c_long),
                ("bc1", _c.c_float), ("bc2_sqrt", _c.c_float)]


class PackJob(_c.Structure):
    """Mirror of `ape_pack_job` (include/ape_hip.h)."""
    _fields_ = [("src", _c.c_void_p), ("dst_f32", _c.c_void_p), ("dst_bf16", _c.c_void_p), ("cout", _c.c_int32), ("cin", _c.c_int32),
                ("taps", _c.c_int32), ("transpose", _c.c_int32), ("src_ld", _c.c_int32), ("reserved", _c.c_int32)]


class ConvParams(_c.Structure):
    """Mirror of `ape_conv_params` (include/ape_hip.h)."""
    _fields_ = [(n, _c.c_int32) for n in ("B", "H", "W", "Cin", "ldx", "xoff", "Ho", "Wo", "Cout", "ldy", "yoff",
                                          "KH", "KW", "stride", "pad", "dil", "act")] + \
               [("alpha", _c.c_float), ("bias_bstride", _c.c_int32), ("ldr", _c.c_int32), ("roff", _c.c_int32), ("ups", _c.c_int32)]


ACT_NONE, ACT_RELU, ACT_PRELU, ACT_SIGMOID = 0, 1, 2, 3
_RESTYPES = {"ape_last_error": _c.c_char_p, "ape_adaptive_avgpool_multi_workspace_bytes": _c.c_size_t, "ape_seg_components_workspace_bytes": _c.c_size_t,
             "ape_packed_weights_bf16_elems": _c.c_long, "ape_pc_workspace_bytes": _c.c_size_t, "ape_pc_batch_workspace_bytes": _c.c_size_t,
             "ape_conv2d_wgrad_workspace_bytes": _c.c_size_t, "ape_conv_gemm_splitk_workspace_bytes": _c.c_size_t}

_lib = None


class ApeError(RuntimeError):
    pass


def lib():
    """Load libape_hip.so once; raise loudly when it is absent (never fall back to a CPU path)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C autoposeestimation_amd/csrc` (hipcc, gfx950). There is no CPU fallback." % LIB_PATH)
        h = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(h, name)  # AttributeError here = header and library disagree
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, _c.c_int)
        _lib = h
    return _lib


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_get_device = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device


def stream_ptr():
    """the HIP stream torch is currently enqueuing on (per thread, per device).  torch.cuda.current_stream() builds a Stream object
    through several Python layers (~12 us); the raw accessor is one C call -- at ~20 launches per registration / ~140 per frame batch
    this is a visible share of the host time"""
    if _raw_stream is not None:
        return _c.c_void_p(_raw_stream(_get_device()))
    return _c.c_void_p(torch.cuda.current_stream().cuda_stream)


def dptr(t, dtype=None):
    """Device pointer of a contiguous CUDA(HIP) tensor; refuses host tensors."""
    if t is None:
        return _c.c_void_p(0)
    if not t.is_cuda:
        raise ApeError("tensor is on %s; the HIP path needs device memory (no CPU fallback)" % t.device)
    if not t.is_contiguous():
        raise ApeError("tensor must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise ApeError("expected %s, got %s" % (dtype, t.dtype))
    return _c.c_void_p(t.data_ptr())


def check(rc, what):
    if rc != 0:
        msg = lib().ape_last_error()
        raise ApeError("%s failed: code %d (%s)" % (what, rc, msg.decode() if msg else ""))
