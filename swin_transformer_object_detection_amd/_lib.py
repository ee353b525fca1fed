"""ctypes binding of include/swin_hip.h.  No fallback: a missing library is an error."""
import ctypes
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
# The library exists in two builds of the same sources and ABI (csrc/common.h): libswin_hip.so, whose 16-bit type is bfloat16, and
# libswin_hip_f16.so (-DSWIN_HALF), whose 16-bit type is IEEE half -- the reference's mixed precision is fp16 (apex O1,
# mmdet/apis/train.py:82-89).  A process works with ONE of them: chosen by SWIN_HALF_DTYPE=fp16 in the environment or by
# set_half_dtype() before the library is first used (build_detector(compute_dtype=torch.float16) does that).
_HALF = torch.float16 if os.environ.get("SWIN_HALF_DTYPE", "bf16").lower() in ("fp16", "f16", "float16", "half") else torch.bfloat16


def _default_path():
    return os.path.join(HERE, "lib", "libswin_hip_f16.so" if _HALF == torch.float16 else "libswin_hip.so")


# SWIN_HIP_LIB selects an alternative build of the SAME ABI (A/B experiments, the -DSWIN_DEV build); default: the in-tree library
LIB_PATH = os.environ.get("SWIN_HIP_LIB") or _default_path()


def half_dtype():
    """torch dtype of the loaded library's 16-bit type (what SWIN_BF16 means in this process)"""
    return _HALF


def set_half_dtype(dtype):
    """Choose the 16-bit type of this process (torch.bfloat16 or torch.float16).  Only before the library is first used."""
    global _HALF, LIB_PATH
    if dtype == _HALF:
        return
    if dtype not in (torch.bfloat16, torch.float16):
        raise SwinHipError(f"the 16-bit compute type is torch.bfloat16 or torch.float16, not {dtype}")
    if _lib is not None:
        raise SwinHipError(f"the library is already loaded with {_HALF} as its 16-bit type; choose {dtype} before first use "
                           "(SWIN_HALF_DTYPE=fp16, or build the model first)")
    _HALF = dtype
    if not os.environ.get("SWIN_HIP_LIB"):
        LIB_PATH = _default_path()

SWIN_F32, SWIN_BF16 = 0, 1
_ERR = {1: "SWIN_ERR_BAD_ARG", 2: "SWIN_ERR_UNSUPPORTED", 3: "SWIN_ERR_LAUNCH"}

_p, _i, _i64, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float

# name -> argtypes, mirrors include/swin_hip.h one to one
SIGNATURES = {
    "swin_hip_abi_version": [],
    "swin_hip_half_type": [],
    "swin_layernorm_fwd": [_p, _p, _p, _p, _p, _p, _i64, _i, _f, _i, _p],
    "swin_layernorm_bwd": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _p, _p, _i64, _i, _i, _p, _p],
    "swin_layernorm_bwd_workspace_bytes": [_i64, _i, _i],
    "swin_add_layernorm_fwd": [_p, _p, _p, _i64, _p, _p, _p, _p, _p, _p, _i64, _i, _f, _i, _p],
    "swin_window_attn_fwd": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _i, _p],
    "swin_window_attn_bwd_workspace_bytes": [_i, _i, _i, _i, _i],
    "swin_window_attn_bwd": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _i, _p],
    "swin_rel_bias_expand": [_p, _p, _i, _p],
    "swin_rel_bias_expand_multi": [_p, _p, _p, _i, _p],
    "swin_rel_bias_reduce": [_p, _p, _i, _p],
    "swin_tail_reduce": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _p],
    "swin_bias_gelu_fwd": [_p, _p, _p, _i64, _i, _i, _p],
    "swin_bias_gelu_bwd": [_p, _p, _p, _p, _p, _i64, _i, _i, _p],
    "swin_patch_merge_ln_fwd": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _i, _p],
    "swin_patch_merge_ln_bwd": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p],
    "swin_patch_im2row": [_p, _p, _i, _i, _i, _i, _p],
    "fpn_upsample_add_fwd": [_p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "fpn_upsample_add_bwd": [_p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "fpn_upsample_add_out_fwd": [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "fpn_upsample_add_out_bwd": [_p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "roi_align_fwd": [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _i, _i, _i, _i, _p],
    "roi_align_bwd": [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _i, _i, _i, _p],
    "conv3x3_nhwc_bf16": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "wgrad_linear_bf16": [_p, _p, _p, _p, _i64, _i, _i, _p],
    "wgrad_conv3x3_nhwc_bf16": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "swin_wgrad_record": [_p, _p, _p, _p, _i64, _i, _i],
    "swin_wgrad_pending": [_p],
    "swin_wgrad_flush": [_p],
    "swin_wgrad96_group": [_p, _p, _p, _p, _p, _p, _p, _i, _p],
    "swin_nms_workspace_bytes": [_i64],
    "nms_sorted": [_p, _i64, _f, _i, _i, _p, _p, _p, _i, _p, _p],
    "nms_sorted_batch": [_p, _i, _i64, _f, _i, _i, _p, _p, _p, _i, _p, _p],
    "nms_sorted_batch_grouped": [_p, _p, _p, _i, _i64, _i, _i, _f, _i, _i, _p, _p, _p, _i, _p, _p],
    "nms_grouped_workspace_bytes": [_i, _i64, _i, _i],
    "roi_align_multilevel_fwd": [_p, _p, _p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "roi_align_multilevel_bwd": [_p, _p, _p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "roi_align_gather_workspace_bytes": [_p, _p, _i, _i, _i],
    "roi_align_multilevel_bwd_gather": [_p, _p, _p, _p, _i, _i, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _i64, _p],
    "det_assign_workspace_bytes": [_i64, _i],
    "det_max_iou_assign": [_p, _i64, _p, _i, _p, _f, _f, _f, _i, _i, _p, _p, _p, _p, _p, _p],
    "det_random_sample_workspace_bytes": [],
    "det_random_sample": [_p, _i64, _i, _i, ctypes.c_uint64, _p, _p, _p, _p, _p],
    "swin_set_u64": [_p, ctypes.c_uint64, _p],
    "det_bbox_targets": [_p, _p, _p, _p, _p, _i, _p, _i64, _p, _p, _i, _p, _p, _p, _p, _p],
    "det_roi_targets_pack": [_p, _p, _p, _p, _p, _i, _p, _i64, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _i, _i, _f, _f,
                             _p, _p, _p, _p, _p],
    "det_delta2bbox": [_p, _p, _i64, _p, _p, _f, _f, _f, _p, _p],
    "swin_block_fwd": [_p, _p, _p, _p],
    "swin_block_bwd": [_p, _p, _p, _p],
    "conv_dgrad_layout_multi": [_p, _p, _p, _p, _i, _p],
    "swin_set_aux_stream": [_p],
    "swin_fork_stream": [_p, _p],
    "swin_stream_create_low_priority": [_p],
    "conv3x3_nhwc_bf16_gated": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "conv3x3_halo_nhwc_bf16": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "narrow_dgrad_gated_bf16": [_p, _p, _p, _p, _i64, _i, _i, _p],
    "conv3x3_splitk_workspace_bytes": [_i, _i, _i, _i, _i],
    "conv3x3_nhwc_bf16_ws": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _i64, _p],
    "swin_gemm_workspace_bytes": [],
    "swin_gemm_bf16": [_p, _p, _p, _p, _i64, _i, _i, _i, _p, _p],
    "swin_adamw_step": [_p, _p, _i, _p, _p, _i, _f, _f, _f, _f, _f, _p],
    "swin_adamw_chunk_elems": [],
    "swin_adamw_set_state": [_p, _p, _p, _i, _f, _f, _p],
    "swin_adamw_step_dev": [_p, _p, _i, _p, _f, _f, _f, _p],
    "swin_loss_scale_begin": [_p, _p],
    "swin_grad_check_finite": [_p, _i64, _p, _p],
    "swin_loss_scale_update": [_p, _f, _f, _i, _f, _f, _p],
    "swin_linear_hip_bf16": [_p, _p, _p, _p, _i64, _i, _i, _i, _p],
    "swin_linear_gelu_hip_bf16": [_p, _p, _p, _p, _p, _i64, _i, _i, _p],
    "swin_linear_dgelu_hip_bf16": [_p, _p, _p, _p, _p, _i64, _i, _i, _p],
    "linear_t_layout_multi": [_p, _p, _p, _p, _i, _p],
    "swin_gemm_plans_export": [_p, _i],
    "swin_gemm_plans_import": [_p, _i],
    "det_rpn_loss_fwd": [_p, _p, _i, _i64, _i, _p, _p, _p, _f, _p, _i, _p],
    "det_rpn_loss_bwd": [_p, _p, _i, _i64, _i, _p, _p, _p, _f, _p, _p, _p, _p, _i, _p],
    "det_bbox_loss_fwd": [_p, _p, _i, _i, _p, _p, _p, _i, _i, _f, _f, _p, _p, _p, _p, _p, _i, _p],
    "det_bbox_loss_bwd": [_p, _p, _i, _i, _p, _p, _p, _i, _i, _f, _f, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p],
    "det_regress_by_class": [_p, _p, _p, _p, _i64, _i, _i, _p, _p, _f, _f, _p, _i, _p],
    "det_bn_workspace_bytes": [_i],
    "det_bn_stats": [_p, _i64, _i, _p, _p, _i, _p],
    "det_bn_finalize": [_p, _i, _f, _f, _p, _p, _p, _p],
    "det_bn_apply": [_p, _p, _i64, _i, _p, _p, _p, _i, _i, _p],
    "det_bn_bwd_reduce": [_p, _p, _i64, _i, _p, _p, _p, _i, _p, _p, _i, _p],
    "det_bn_bwd_apply": [_p, _p, _p, _i64, _i, _p, _p, _p, _i, _p, _p, _i, _p],
    "det_mask_loss_fwd": [_p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _i, _p],
    "det_mask_loss_bwd": [_p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _i, _p],
    "det_rpn_flatten_fwd": [_p, _p, _i, _i, _i, _i, _p, _p, _i, _p],
    "det_rpn_flatten_bwd": [_p, _p, _i, _i, _i, _i, _p, _p, _i, _p],
    "swin_mlp_fwd_bf16": [_p, _p, _p, _p, _p, _p, _i64, _i, _p],
    "swin_mlp_bwd_bf16": [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i, _p],
    "swin_mlp_add_ln_fwd_bf16": [_p, _p, _p, _p, _p, _p, _p, _i64, _p, _p, _p, _p, _p, _p, _i64, _i, _f, _p],
    "swin_mlp_ln_bwd_partial_rows": [_i64, _i],
    "swin_mlp_ln_bwd_bf16": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _p, _p, _p, _i64, _i, _p],
    "swin_mlp_ln2_bwd_bf16": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i, _p],
    "swin_ts_linear_bf16": [_p, _p, _p, _p, _i64, _i, _i, _i, _p],
    "swin_ts_proj_add_ln_bf16": [_p, _p, _p, _p, _p, _i64, _p, _p, _p, _p, _p, _p, _i64, _i, _f, _p],
    "nms_prepare_workspace_bytes": [_i, _i64],
    "nms_prepare_sorted_batch": [_p, _p, _p, _i, _i64, _p, _p, _p, _p],
    "nms_gather_dets": [_p, _p, _p, _p, _i, _i64, _i, _p, _p, _p],
    "det_map_roi_levels": [_p, _p, _i64, _i, _f, _p, _p],
    "det_rpn_topk_decode_workspace_bytes": [_i64, _i64],
    "det_rpn_topk_decode": [_p, _p, _p, _p, _i, _i64, _i, _p, _p, _f, _f, _p, _p, _p, _p, _i, _p],
    "det_paste_masks": [_p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _i, _i, _p, _p],
}
_RESTYPE = {"swin_nms_workspace_bytes": _i64, "swin_gemm_workspace_bytes": _i64, "swin_layernorm_bwd_workspace_bytes": _i64, "swin_window_attn_bwd_workspace_bytes": _i64,
            "det_assign_workspace_bytes": _i64, "det_random_sample_workspace_bytes": _i64, "det_bn_workspace_bytes": _i64,
            "det_rpn_topk_decode_workspace_bytes": _i64, "nms_prepare_workspace_bytes": _i64, "nms_grouped_workspace_bytes": _i64,
            "conv3x3_splitk_workspace_bytes": _i64, "roi_align_gather_workspace_bytes": _i64, "swin_mlp_ln_bwd_partial_rows": _i64}

_lib = None


class SwinHipError(RuntimeError):
    pass


def lib():
    """The loaded library.  Raises if it has not been built -- there is no CPU fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SwinHipError(
                f"{LIB_PATH} is missing: build it with `python -m swin_transformer_object_detection_amd.build` "
                "(or __graft_entry__.build()).  The HIP path has no fallback.")
        l = ctypes.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(l, name)          # AttributeError here == header/library mismatch
            fn.argtypes = args
            fn.restype = _RESTYPE.get(name, _i)
        if l.swin_hip_half_type() != (1 if _HALF == torch.float16 else 0):
            raise SwinHipError(f"{LIB_PATH} was built for the other 16-bit type than {_HALF}")
        _lib = l
    return _lib


def call(name, *args):
    """Call an int-status entry point; raise on a non-zero status."""
    st = getattr(lib(), name)(*args)
    if st != 0:
        raise SwinHipError(f"{name} failed: {_ERR.get(st, st)}")
