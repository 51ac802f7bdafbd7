"""TEST INFRASTRUCTURE ONLY -- ctypes front-end of oracle/yolo_oracle.c.

Restates luoyang/utils/torch_utils.py:101-241 (the three non_max_suppression variants) on the
CPU: one call handles a whole (B, A, C) float32 batch and returns the reference's
``List[Optional[ndarray]]`` (``None`` for an image with no candidate).
"""
import ctypes
import os
import subprocess
from typing import List, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libyolo_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile oracle/yolo_oracle.c with gcc (a few hundred ms)."""
    src = os.path.join(_HERE, "yolo_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.yo_nms.restype = ctypes.c_int64
        _lib.yo_nms.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_double,
                                ctypes.c_void_p]
        _lib.yo_filter_nms.restype = ctypes.c_int64
        _lib.yo_filter_nms.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_float,
                                       ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_double,
                                       ctypes.c_void_p, ctypes.c_void_p]
    return _lib


def nms(boxes: np.ndarray, scores: np.ndarray, iou_threshold: float) -> np.ndarray:
    """torchvision.ops.nms CPU semantics; returns kept indices (int64, descending score)."""
    boxes = np.ascontiguousarray(boxes, dtype=np.float32).reshape(-1, 4)
    scores = np.ascontiguousarray(scores, dtype=np.float32).reshape(-1)
    n = boxes.shape[0]
    keep = np.empty(max(n, 1), dtype=np.int64)
    k = lib().yo_nms(boxes.ctypes.data, scores.ctypes.data, n, float(iou_threshold), keep.ctypes.data)
    return keep[:k].copy()


def non_max_suppression(prediction: np.ndarray, img_size: float, in_fmt: str = "cxcywh",
                        conf_threshold: float = 0.5, nms_threshold: float = 0.3,
                        point_num: int = 0, point_conf: bool = False,
                        return_index: bool = False):
    """All three reference variants: point_num=0 -> plain; point_conf=False -> 2P extra columns
    (torch_utils.py:144), True -> 3P extra columns (torch_utils.py:194)."""
    pred = np.ascontiguousarray(prediction, dtype=np.float32)
    assert pred.ndim == 3
    B, A, C = pred.shape
    n_extra = point_num * (3 if point_conf else 2)
    W = 6 + n_extra + 1
    out: List[Optional[np.ndarray]] = []
    idx: List[Optional[np.ndarray]] = []
    rows = np.empty((max(A, 1), W), dtype=np.float32)
    src = np.empty(max(A, 1), dtype=np.int64)
    for b in range(B):
        k = lib().yo_filter_nms(pred[b].ctypes.data, A, C, float(img_size), 1 if in_fmt == "cxcywh" else 0,
                                n_extra, float(np.float32(conf_threshold)), float(nms_threshold),
                                rows.ctypes.data, src.ctypes.data)
        out.append(rows[:k].copy() if k else None)
        idx.append(src[:k].copy() if k else None)
    return (out, idx) if return_index else out
