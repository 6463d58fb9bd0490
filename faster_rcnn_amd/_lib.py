"""ctypes binding of libfrcnn_hip.so (include/frcnn_hip.h).

The library is the product: there is NO CPU fallback.  Importing this module never touches
the GPU; the first compute call on a machine without the built library or without a HIP
device raises ``FrcnnError``.
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_int, c_size_t, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
# (FRCNN_LIB_PATH: another BUILD of this same library -- A/B runs of two kernel versions on one box; there is still no CPU fallback)
LIB_PATH = os.environ.get("FRCNN_LIB_PATH") or os.path.join(HERE, "libfrcnn_hip.so")


class FrcnnError(RuntimeError):
    pass


ABI_VERSION = 108       # include/frcnn_hip.h FRCNN_ABI_VERSION (tests/test_abi.py holds the two together)
P = c_void_p
I = c_int
# name -> (restype, argtypes).  Must list every symbol include/frcnn_hip.h declares
# (tests/test_abi.py parses the header and checks this table and the .so against it).
SIGNATURES = {
    "frcnn_last_error": (c_char_p, []),
    "frcnn_version": (I, []),
    "frcnn_device_count": (I, []),
    "frcnn_preprocess_u8": (I, [P, c_size_t, P, P, P]),
    "frcnn_resize_cubic_taps": (I, [I, I, P]),
    "frcnn_resize_cubic_u8": (I, [P, I, I, P, P, I, I, I, P, P]),
    "frcnn_anchors_image": (I, [I, I, P, I, I, P, P]),
    "frcnn_anchors_conv": (I, [I, I, P, I, P, P]),
    "frcnn_cross_ious_f32": (I, [P, I, P, I, P, P]),
    "frcnn_cross_ious_i16": (I, [P, I, P, I, P, P]),
    "frcnn_rpn_assign_workspace_bytes": (c_size_t, [I, I, I, I]),
    "frcnn_rpn_assign": (I, [I, I, P, I, I, P, I, I, I, P, P, P, P, P, c_size_t, P]),
    "frcnn_rpn_sample_lists": (I, [P, P, I, P, P, P, P]),
    "frcnn_rpn_pack_targets": (I, [P, P, P, I, I, P, I, P, I, P, I, P, I, P, P, P]),
    "frcnn_host_mt_sample_range": (I, [P, P, I, I, I, P]),
    "frcnn_decode_proposals": (I, [P, I, I, P, I, P, P, P]),
    "frcnn_transform_inplace": (I, [P, P, I, P]),
    "frcnn_preprocess_u8_canvas": (I, [P, I, I, I, I, I, I, P, P, P]),
    "frcnn_zero_outside": (I, [P, I, I, I, I, P, P]),
    "frcnn_decode_proposals_canvas": (I, [P, I, I, P, I, P, P, P, P]),
    "frcnn_topk_workspace_bytes": (c_size_t, [I]),
    "frcnn_topk_order": (I, [P, P, I, I, P, P, P, c_size_t, P]),
    "frcnn_gather_candidates": (I, [P, P, P, P, I, P, P, P]),
    "frcnn_nms_workspace_bytes": (c_size_t, [I]),
    "frcnn_nms_i16": (I, [P, P, I, c_double, I, P, P, P, c_size_t, P]),
    "frcnn_nms_f64": (I, [P, P, I, c_double, I, P, P, P, c_size_t, P]),
    "frcnn_gather_rois": (I, [P, P, P, I, I, P, P]),
    "frcnn_roi_targets": (I, [P, I, P, P, P, I, I, P, P, P, P]),
    "frcnn_roi_crop_resize_fwd": (I, [P, I, I, I, P, I, I, P, P]),
    "frcnn_roi_crop_resize_fwd_ex": (I, [P, I, I, I, P, I, I, P, I, I, P, P]),
    "frcnn_roi_crop_resize_bwd": (I, [P, I, I, I, P, I, I, P, P]),
    "frcnn_conv_packed_k": (I, [I, I, I]),
    "frcnn_pack_conv_weights": (I, [P, I, I, I, I, P, P]),
    "frcnn_conv2d_fwd": (I, [P, P, P, P, P, P, P, P]),
    "frcnn_conv2d_config": (I, [P]),
    "frcnn_conv2d_fwd_masked": (I, [P, P, P, P, P, P, P, P, P]),
    "frcnn_conv2d_workspace_bytes": (c_size_t, [P]),
    "frcnn_conv2d_fwd_ws": (I, [P, P, P, P, P, P, P, P, P, c_size_t, P]),
    "frcnn_conv2d_dual_workspace_bytes": (c_size_t, [P]),
    "frcnn_conv2d_fwd_dual": (I, [P, P, P, P, P, P, I, I, P, I, P, c_size_t, P]),
    "frcnn_conv2d_dual_config": (I, [P, I]),
    "frcnn_conv2d_x6_config": (I, [P, I]),
    "frcnn_pack_conv_weights_x6": (I, [P, I, I, P, P]),
    "frcnn_refresh_x6_planes": (I, [P, I, P]),
    "frcnn_refresh_h3_planes": (I, [P, I, P]),
    "frcnn_conv2d_x6_workspace_bytes": (c_size_t, [P]),
    "frcnn_conv2d_fwd_x6": (I, [P, P, P, P, P, P, P, P, P, c_size_t, P]),
    "frcnn_conv2d_fwd_dual_x6": (I, [P, P, P, P, P, P, I, I, P, I, P]),
    "frcnn_conv2d_engine": (I, [P, I, I]),
    "frcnn_conv_h3_planes_bytes": (c_size_t, [I, I]),
    "frcnn_pack_conv_weights_h3": (I, [P, I, I, P, P]),
    "frcnn_amax_record_floats": (I, []),
    "frcnn_amax_clear": (I, [P, I, P]),
    "frcnn_amax_f32": (I, [P, c_size_t, P, P]),
    "frcnn_amax_merge": (I, [P, P, ctypes.c_float, P, P]),
    "frcnn_amax_status": (I, [P, I, P, P]),
    "frcnn_roi_crop_resize_fwd_planes": (I, [P, I, I, I, P, I, I, P, I, I, P, P]),
    "frcnn_roi_crop_resize_fwd_batch": (I, [P, I, I, I, P, I, I, I, P, I, I, P, P, P]),
    "frcnn_conv2d_h3_config": (I, [P, I]),
    "frcnn_conv2d_h3_workspace_bytes": (c_size_t, [P]),
    "frcnn_conv2d_fwd_h3": (I, [P, P, P, P, P, P, P, P, P, P, P, c_size_t, P]),
    "frcnn_conv2d_fwd_dual_h3": (I, [P, P, P, P, P, P, P, I, I, P, P, I, P, P]),
    "frcnn_stem_h3_packed_bytes": (c_size_t, []),
    "frcnn_pack_stem_weights_h3": (I, [P, P, P]),
    "frcnn_stem_h3_fwd": (I, [P, P, I, I, I, P, P, P, P, P, P]),
    "frcnn_conv2d_fwd_h3_planes": (I, [P, P, P, P, P, P, P, P, P, P, P, P, ctypes.c_float, ctypes.c_float, P]),
    "frcnn_conv2d_fwd_h3_planes_res": (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, ctypes.c_float, ctypes.c_float, P]),
    "frcnn_conv2d_fwd_ws_amax": (I, [P, P, P, P, P, P, P, P, P, P, c_size_t, P]),
    "frcnn_pack_conv_weights_dgrad": (I, [P, P, I, I, I, I, P, P]),
    "frcnn_conv2d_wgrad_workspace_bytes": (c_size_t, [P]),
    "frcnn_conv2d_wgrad": (I, [P, P, P, P, P, P, P, c_size_t, P]),
    "frcnn_conv2d_wgrad_batch_workspace_bytes": (c_size_t, [P, I]),
    "frcnn_conv2d_wgrad_batch": (I, [P, I, P, c_size_t, P]),
    "frcnn_refresh_packed": (I, [P, I, P]),
    "frcnn_colsum_batch": (I, [P, I, P]),
    "frcnn_pool2d_fwd": (I, [P, I, I, I, I, I, I, I, P, P]),
    "frcnn_pool2d_fwd_planes": (I, [P, I, I, I, I, I, I, I, P, P]),
    "frcnn_avgpool_pos_major": (I, [P, I, I, I, P, P]),
    "frcnn_softmax_rows": (I, [P, I, I, I, P, I, P]),
    "frcnn_dense_heads_split": (I, [P, I, I, I, I, P, P, P]),
    "frcnn_loss_rpn_cls": (I, [P, P, I, I, P, P, P]),
    "frcnn_loss_rpn_reg": (I, [P, P, I, I, P, P, P]),
    "frcnn_loss_workspace_bytes": (ctypes.c_size_t, []),
    "frcnn_loss_rpn_cls_ws": (I, [P, P, I, I, P, P, P, P]),
    "frcnn_loss_rpn_reg_ws": (I, [P, P, I, I, P, P, P, P]),
    "frcnn_loss_det_cls": (I, [P, P, I, I, P, P, I, P]),
    "frcnn_loss_det_reg": (I, [P, P, I, I, P, P, I, P]),
    "frcnn_relu_bwd_inplace": (I, [P, P, c_size_t, P]),
    "frcnn_avgpool_bwd_masked": (I, [P, P, I, I, I, P, P]),
    "frcnn_maxpool_bwd": (I, [P, P, P, I, I, I, I, I, P, P]),
    "frcnn_sgd_momentum": (I, [P, P, P, c_size_t, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, P]),
    "frcnn_adam": (I, [P, P, P, P, c_size_t, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, I, ctypes.c_float, ctypes.c_float, P]),
    "frcnn_sumsq_workspace_bytes": (c_size_t, []),
    "frcnn_sumsq": (I, [P, c_size_t, P, P, c_size_t, P]),
    "frcnn_fold_bias": (I, [P, P, P, P, I, P]),
    "frcnn_stem_bf16_packed_elems": (I, []),
    "frcnn_pack_stem_weights_bf16": (I, [P, P, P]),
    "frcnn_stem_bf16_fwd": (I, [P, I, I, I, P, P, P, P, P]),
    "frcnn_conv_packed_k_bf16": (I, [I, I, I]),
    "frcnn_pack_conv_weights_bf16": (I, [P, I, I, I, I, P, P]),
    "frcnn_conv2d_fwd_bf16": (I, [P, P, P, P, P, P, P, I, P]),
    "frcnn_conv2d_workspace_bytes_bf16": (c_size_t, [P]),
    "frcnn_conv2d_fwd_bf16_ws": (I, [P, P, P, P, P, P, P, I, P, c_size_t, P]),
    "frcnn_conv2d_fwd_bf16_masked": (I, [P, P, P, P, P, P, P, P, I, P, c_size_t, P]),
    "frcnn_refresh_packed_bf16": (I, [P, I, P]),
    "frcnn_conv2d_wgrad_bf16": (I, [P, P, P, P, P, P, P, c_size_t, P]),
    "frcnn_cast_bf16_to_f32": (I, [P, c_size_t, P, P]),
    "frcnn_relu_bwd_inplace_bf16": (I, [P, P, c_size_t, P]),
    "frcnn_avgpool_bwd_masked_bf16": (I, [P, P, I, I, I, P, P]),
    "frcnn_roi_crop_resize_bwd_bf16": (I, [P, I, I, I, P, I, I, P, P]),
    "frcnn_cast_f32_to_bf16": (I, [P, c_size_t, P, P]),
    "frcnn_avgpool_bf16_to_f32": (I, [P, I, I, I, P, P]),
    "frcnn_avgpool_bf16_to_f32_ex": (I, [P, I, I, I, I, P, P]),
    "frcnn_roi_crop_resize_fwd_bf16": (I, [P, I, I, I, P, I, I, P, P]),
    "frcnn_roi_crop_resize_fwd_bf16_ex": (I, [P, I, I, I, P, I, I, P, I, I, P, P]),
    "frcnn_roi_crop_resize_fwd_bf16_batch": (I, [P, I, I, I, I, P, I, I, P, I, I, P, P]),
    "frcnn_detections": (I, [P, P, I, P, P, I, I, c_double, c_double, c_double, c_double, P, P, P, P, P, P]),
    "frcnn_detections_dyn": (I, [P, P, I, I, P, P, I, I, c_double, c_double, P, P, P, P, P, P, P]),
}



class ConvDesc(ctypes.Structure):
    """frcnn_conv_desc (include/frcnn_hip.h)."""
    _fields_ = [(k, ctypes.c_int32) for k in (
        "n", "h", "w", "cin", "cout", "kh", "kw", "stride", "pad_top", "pad_left", "ho", "wo",
        "act", "ldy", "ldres", "tile", "layout")]


class PackJob(ctypes.Structure):
    """frcnn_pack_job (include/frcnn_hip.h)."""
    _fields_ = [(k, c_void_p) for k in ("w_hwio", "packed", "packed_dgrad", "bias", "scale", "shift_const", "shift")] + \
               [(k, ctypes.c_int32) for k in ("kh", "kw", "cin", "cout")]


class WgradJob(ctypes.Structure):
    """frcnn_wgrad_job (include/frcnn_hip.h)."""
    _fields_ = [("d", ConvDesc)] + [(k, c_void_p) for k in ("x", "g", "scale", "dw")] + [(k, ctypes.c_int32) for k in ("in_bf16", "reserved")]


class X6Job(ctypes.Structure):
    """frcnn_x6_job (include/frcnn_hip.h)."""
    _fields_ = [("w_packed", c_void_p), ("planes_bf16", c_void_p), ("rows", ctypes.c_int32), ("kpad", ctypes.c_int32)]


H3_UNDER, H3_SATURATED, H3_NONFINITE = 1, 2, 4      # FRCNN_H3_* status bits (include/frcnn_hip.h)


class H3Planes(ctypes.Structure):
    """frcnn_h3_planes (include/frcnn_hip.h)."""
    _fields_ = [("planes", c_void_p), ("exponent", c_void_p), ("status", c_void_p)]


class ColsumJob(ctypes.Structure):
    """frcnn_colsum_job (include/frcnn_hip.h)."""
    _fields_ = [(k, c_void_p) for k in ("g", "scale", "out")] + [(k, ctypes.c_int32) for k in ("m", "cout", "g_is_bf16", "reserved")]


_lib = None


def load():
    """Load the shared library (once) and bind every declared symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FrcnnError(
            f"{LIB_PATH} is missing: build it with `python -m faster_rcnn_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    # torch bundles its own HIP runtime (same SONAME as /opt/rocm's).  The process must use
    # ONE runtime -- streams and device pointers are shared with torch -- so make sure
    # torch's copy is loaded first; ours then binds to it by SONAME.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library drift
        fn.restype = res
        fn.argtypes = args
    if lib.frcnn_version() != ABI_VERSION:
        raise FrcnnError(f"{LIB_PATH} speaks ABI revision {lib.frcnn_version()}, this binding {ABI_VERSION} (include/frcnn_hip.h "
                         "FRCNN_ABI_VERSION): rebuild with `python -m faster_rcnn_amd.build`")
    _lib = lib
    return lib


def check(code, what=""):
    if code != 0:
        msg = load().frcnn_last_error()
        raise FrcnnError(f"{what or 'frcnn call'} failed ({code}): {msg.decode() if msg else ''}")


def call(name, *args):
    """Call an int-returning entry point and raise FrcnnError on a non-zero status."""
    code = getattr(_lib or load(), name)(*args)
    if code != 0:
        check(code, name)
