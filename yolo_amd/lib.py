"""ctypes binding of libyolo_amd.so (the C ABI declared in include/yolo_amd.h).

There is deliberately no fallback: if the shared library is absent or a symbol is missing the
import of this module raises, so a GPU test can never pass on a silent CPU path.
"""
import ctypes
import os
from ctypes import c_double, c_float, c_int, c_int32, c_int64, c_longlong, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libyolo_amd.so")

ACT_NONE, ACT_SILU, ACT_LEAKY, ACT_MISH, ACT_RELU = 0, 1, 2, 3, 4
DECODE_V5, DECODE_V3, DECODE_POSE, DECODE_FACE = 0, 1, 2, 3

# every symbol include/yolo_amd.h declares
EXPORTS = [
    "yamd_last_error", "yamd_version", "yamd_init", "yamd_conv_desc_init", "yamd_conv_desc_tune", "yamd_conv_pack_weights", "yamd_conv2d_bias_act", "yamd_conv2d_bias_act_split", "yamd_conv2d_bias_act_cat2",
    "yamd_focus_pack", "yamd_nchw_to_nhwc16", "yamd_upsample2x", "yamd_maxpool2", "yamd_maxpool5_cascade", "yamd_maxpool3_cascade",
    "yamd_copy_channels", "yamd_depth_to_space2", "yamd_decode_ltrb_points",
    "yamd_decode_anchor", "yamd_decode_anchor_levels", "yamd_decode_dfl", "yamd_decode_dfl_unit_conf", "yamd_decode_dfl_points", "yamd_nms_workspace_bytes", "yamd_nms_batched", "yamd_decode_nms_anchor", "yamd_nms_static_capacity_bytes", "yamd_nms_batched_static", "yamd_compact_rows", "yamd_letterbox_workspace_bytes", "yamd_letterbox_u8", "yamd_detect_match", "yamd_stem_packed_elems", "yamd_stem_pack_weights", "yamd_stem_conv", "yamd_stem_conv2_packed_elems", "yamd_stem_conv2_pack_weights", "yamd_stem_conv2",
    "yamd_bottleneck_packed_elems", "yamd_bottleneck_pack_weights", "yamd_bottleneck",
    "yamd_c3_packed_elems", "yamd_c3_pack_weights", "yamd_c3_fused",
    "yamd_microbench_mfma_bf16", "yamd_microbench_copy", "yamd_microbench_clock_probe",
    "yamd_plan_create", "yamd_plan_run", "yamd_plan_size", "yamd_plan_destroy",
]

# yamd_plan_op.kind
OP_CONV, OP_CONV_SPLIT, OP_STEM, OP_STEM_CONV2, OP_FOCUS, OP_NHWC16, OP_UPSAMPLE2X, OP_MAXPOOL5_CASCADE, \
    OP_MAXPOOL3_CASCADE, OP_MAXPOOL2, OP_DEPTH_TO_SPACE2, OP_COPY_CHANNELS, OP_CONV_CAT2, OP_BOTTLENECK, OP_C3 = range(1, 16)


class ConvDesc(ctypes.Structure):
    """Mirror of ``yamd_conv_desc``."""
    _fields_ = [("ksize", c_int32), ("stride", c_int32), ("cin", c_int32), ("cout", c_int32), ("act", c_int32),
                ("out_f32", c_int32), ("nrep", c_int32), ("ck", c_int32), ("mrep", c_int32), ("cout_pad", c_int32),
                ("packed_w_elems", c_int64), ("nwaves", c_int32), ("residual_scale", ctypes.c_float), ("stages", c_int32)]


class PlanOp(ctypes.Structure):
    """Mirror of ``yamd_plan_op``."""
    _fields_ = [("kind", c_int32), ("i", c_int32 * 9), ("p", c_void_p * 7), ("desc", ConvDesc)]


class YamdError(RuntimeError):
    pass


def _load() -> ctypes.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m yolo_amd.build` (hipcc, gfx950). "
            "yolo_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise ImportError(f"{LIB_PATH} does not export {name}")
    P = ctypes.POINTER
    lib.yamd_last_error.restype = ctypes.c_char_p
    lib.yamd_last_error.argtypes = []
    lib.yamd_version.restype = c_int
    lib.yamd_init.argtypes = []
    lib.yamd_conv_desc_init.argtypes = [P(ConvDesc), c_int, c_int, c_int, c_int, c_int, c_int]
    lib.yamd_conv_desc_tune.argtypes = [P(ConvDesc), c_int, c_int, c_int, c_int, c_int]
    lib.yamd_conv_pack_weights.argtypes = [P(ConvDesc), c_void_p, c_int, c_void_p, c_void_p, c_void_p]
    lib.yamd_conv2d_bias_act.argtypes = [P(ConvDesc), c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                         c_int, c_int, c_int, c_int, c_void_p]
    lib.yamd_conv2d_bias_act_split.argtypes = [P(ConvDesc), c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                               c_int, c_int, c_int, c_int, c_void_p]
    lib.yamd_conv2d_bias_act_cat2.argtypes = [P(ConvDesc), c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                              c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]
    lib.yamd_focus_pack.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]
    lib.yamd_nchw_to_nhwc16.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]
    lib.yamd_upsample2x.argtypes = [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]
    lib.yamd_maxpool2.argtypes = [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]
    lib.yamd_depth_to_space2.argtypes = [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]
    lib.yamd_maxpool5_cascade.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                          c_int, c_void_p]
    lib.yamd_maxpool3_cascade.argtypes = lib.yamd_maxpool5_cascade.argtypes
    lib.yamd_copy_channels.argtypes = [c_void_p, c_int, c_void_p, c_int, c_longlong, c_int, c_void_p]
    lib.yamd_decode_anchor.argtypes = [c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int,
                                       c_float, c_int, c_int, c_void_p, c_longlong, c_longlong, c_void_p]
    lib.yamd_decode_anchor_levels.argtypes = [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                              c_int, c_float, c_int, c_void_p, c_void_p]
    lib.yamd_decode_dfl.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                    c_void_p, c_void_p]
    lib.yamd_decode_dfl_unit_conf.argtypes = lib.yamd_decode_dfl.argtypes
    lib.yamd_decode_ltrb_points.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                            c_void_p, c_void_p]
    lib.yamd_decode_dfl_points.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                           c_void_p, c_void_p]
    lib.yamd_detect_match.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_float, c_float, c_float,
                                      c_void_p, c_void_p]
    lib.yamd_stem_packed_elems.restype = c_int64
    lib.yamd_stem_packed_elems.argtypes = [c_int, c_int]
    lib.yamd_stem_pack_weights.argtypes = [c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p]
    lib.yamd_stem_conv.argtypes = [c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p]
    lib.yamd_stem_conv2_packed_elems.restype = c_int64
    lib.yamd_stem_conv2_packed_elems.argtypes = [c_int, c_int]
    lib.yamd_stem_conv2_pack_weights.argtypes = [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]
    lib.yamd_stem_conv2.argtypes = [c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                    c_int, c_int, c_void_p, c_int, c_void_p]
    lib.yamd_bottleneck_packed_elems.restype = c_int64
    lib.yamd_bottleneck_packed_elems.argtypes = [c_int, c_int, c_int]
    lib.yamd_bottleneck_pack_weights.argtypes = [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.yamd_bottleneck.argtypes = [c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                    c_int, c_int, c_int, c_void_p]
    lib.yamd_c3_packed_elems.restype = c_int64
    lib.yamd_c3_packed_elems.argtypes = [c_int, c_int, c_int]
    lib.yamd_c3_pack_weights.argtypes = [c_int, c_int, c_int] + [c_void_p] * 12
    lib.yamd_c3_fused.argtypes = lib.yamd_bottleneck.argtypes
    lib.yamd_microbench_mfma_bf16.argtypes = [c_void_p, c_void_p, c_int, c_int, c_void_p]
    lib.yamd_microbench_copy.argtypes = [c_void_p, c_void_p, c_size_t, c_int, c_void_p]
    lib.yamd_microbench_clock_probe.argtypes = [c_void_p, c_int, c_int, c_void_p]
    lib.yamd_letterbox_workspace_bytes.restype = c_size_t
    lib.yamd_letterbox_workspace_bytes.argtypes = [c_int, c_int, c_int, c_int]
    lib.yamd_letterbox_u8.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_size_t, c_void_p]
    lib.yamd_nms_workspace_bytes.restype = c_size_t
    lib.yamd_nms_workspace_bytes.argtypes = [c_int, c_int, c_int, c_int]
    lib.yamd_nms_batched.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_double, c_void_p,
                                     c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]
    lib.yamd_decode_nms_anchor.argtypes = [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                           c_float, c_int, c_void_p, c_float, c_double, c_void_p, c_void_p, c_int, c_void_p,
                                           c_void_p, c_size_t, c_void_p]
    lib.yamd_compact_rows.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]
    lib.yamd_nms_static_capacity_bytes.restype = c_size_t
    lib.yamd_nms_static_capacity_bytes.argtypes = []
    lib.yamd_nms_batched_static.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_double,
                                            c_void_p, c_void_p, c_int, c_void_p, c_void_p]
    lib.yamd_plan_create.argtypes = [P(PlanOp), c_int, c_int, c_int, c_int, P(c_void_p)]
    lib.yamd_plan_run.argtypes = [c_void_p, c_void_p, c_void_p, P(c_int)]
    lib.yamd_plan_size.argtypes = [c_void_p]
    lib.yamd_plan_destroy.argtypes = [c_void_p]
    lib.yamd_plan_destroy.restype = None
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name not in ("yamd_plan_destroy","yamd_last_error", "yamd_nms_workspace_bytes", "yamd_letterbox_workspace_bytes", "yamd_stem_packed_elems", "yamd_stem_conv2_packed_elems", "yamd_bottleneck_packed_elems", "yamd_c3_packed_elems",
                        "yamd_nms_static_capacity_bytes"):
            fn.restype = c_int
    return lib


LIB = _load()


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise YamdError(f"{what or 'yolo_amd call'} failed ({rc}): {LIB.yamd_last_error().decode()}")


def current_stream() -> int:
    """Raw hipStream_t of torch's current stream (0 = default stream)."""
    return int(torch.cuda.current_stream().cuda_stream)


def require_gpu(t: torch.Tensor, dtype=None, name: str = "tensor") -> None:
    if not t.is_cuda:
        raise YamdError(f"{name} must live on the GPU: yolo_amd has no CPU path")
    if dtype is not None and t.dtype != dtype:
        raise YamdError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise YamdError(f"{name} must be contiguous")
