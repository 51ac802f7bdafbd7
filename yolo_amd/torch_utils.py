"""Drop-in counterparts of luoyang/utils/torch_utils.py for the inference path.

``non_max_suppression*`` keep the reference signatures and return type
(``List[Optional[Tensor]]``, ``None`` for an image without candidates) but run the whole batch in
the HIP kernels behind ``yamd_nms_batched`` -- no per-image Python loop, one device->host copy of
the counts.  ``fuse_conv_and_bn`` is the host-side BatchNorm folding used when a plan is compiled.
"""
import ctypes
import threading
from typing import List, Optional, Tuple, Union

import numpy as np
import torch
import torch.nn as nn

from . import lib as _lib

LIB = _lib.LIB

_ws_cache = {}
_ws_lock = threading.Lock()


def _workspace(nbytes: int, device) -> torch.Tensor:
    """NMS scratch, one per (device, stream): calls on different streams (or from different host threads, each on
    its own stream) never share a workspace; calls on one stream are ordered, so they can."""
    key = (str(device), int(torch.cuda.current_stream(device).cuda_stream))
    with _ws_lock:
        ws = _ws_cache.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
            _ws_cache[key] = ws
    return ws


class Arena:
    """Named device buffers that persist between calls, so that a step re-uses the same addresses every time
    (no allocator traffic per batch; a step recorded under stream capture writes to memory the model still owns).
    A buffer is re-allocated only when its shape, dtype or device changes."""

    def __init__(self):
        self._bufs = {}

    def get(self, name: str, shape, dtype, device) -> torch.Tensor:
        t = self._bufs.get(name)
        shape = tuple(int(v) for v in shape)
        if t is None or tuple(t.shape) != shape or t.dtype != dtype or t.device != torch.device(device):
            t = torch.empty(shape, dtype=dtype, device=device)
            self._bufs[name] = t
        return t

    def clear(self) -> None:
        self._bufs.clear()


def _out(arena: Optional[Arena], name: str, shape, dtype, device) -> torch.Tensor:
    return torch.empty(shape, dtype=dtype, device=device) if arena is None else arena.get(name, shape, dtype, device)


def nms_batched(prediction: torch.Tensor, img_size: float, in_fmt: str, conf_threshold: float,
                nms_threshold: float, n_extra: int = 0, max_out: Optional[int] = None,
                want_index: bool = False, arena: Optional[Arena] = None
                ) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
    """Device-side result: rows (B, max_out, 7+n_extra) float32, counts (B,) int32, index (B, max_out) int32.
    With an ``arena`` the outputs and the scratch live in its persistent buffers (overwritten by the next call)."""
    _lib.require_gpu(prediction, torch.float32, "prediction")
    if prediction.dim() != 3:
        raise _lib.YamdError("prediction must be (batch, anchors, columns)")
    if in_fmt not in ("cxcywh", "xyxy"):
        raise ValueError(f"unsupported in_fmt {in_fmt}")
    B, A, C = prediction.shape
    max_out = A if max_out is None else int(max_out)
    dev = prediction.device
    rows = _out(arena, "nms_rows", (B, max_out, 7 + n_extra), torch.float32, dev)
    counts = _out(arena, "nms_counts", (B,), torch.int32, dev)
    index = _out(arena, "nms_index", (B, max_out), torch.int32, dev) if want_index else None
    nbytes = LIB.yamd_nms_workspace_bytes(B, A, C, n_extra)
    if nbytes == 0:
        raise _lib.YamdError(f"bad NMS shape B={B} A={A} C={C} extra={n_extra}")
    ws = _workspace(nbytes, dev) if arena is None else arena.get("nms_workspace", (nbytes,), torch.uint8, dev)
    # the comparison 'score >= conf_threshold' is done in float32, as torch does for a python scalar
    conf32 = float(torch.tensor(conf_threshold, dtype=torch.float32))
    _lib.check(LIB.yamd_nms_batched(prediction.data_ptr(), B, A, C, 1 if in_fmt == "cxcywh" else 0, n_extra,
                                    float(img_size), conf32, float(nms_threshold), rows.data_ptr(),
                                    None if index is None else index.data_ptr(), max_out, counts.data_ptr(),
                                    ws.data_ptr(), ws.numel(), _lib.current_stream()), "nms_batched")
    return rows, counts, index


def rows_to_host(rows: torch.Tensor, counts: torch.Tensor, arena: Optional[Arena] = None
                 ) -> List[Optional["np.ndarray"]]:
    """Kept rows of the whole batch -> the reference's per-image list of ndarrays (None when empty) with
    ONE packed device->host copy (plus the (B+1)-entry offset table) instead of one copy per image."""
    B, max_out, W = rows.shape
    compact = _out(arena, "compact_rows", (B * max_out, W), torch.float32, rows.device)
    offsets = _out(arena, "compact_offsets", (B + 1,), torch.int64, rows.device)
    _lib.check(LIB.yamd_compact_rows(rows.data_ptr(), counts.data_ptr(), B, max_out, W, compact.data_ptr(),
                                     offsets.data_ptr(), _lib.current_stream()), "compact_rows")
    off = offsets.cpu().tolist()
    host = compact[:off[-1]].cpu().numpy()
    return [None if off[b + 1] == off[b] else host[off[b]:off[b + 1]] for b in range(B)]


def _split(rows: torch.Tensor, counts: torch.Tensor) -> List[Optional[torch.Tensor]]:
    host_counts = counts.cpu().tolist()        # the single synchronisation of the call
    max_out = rows.shape[1]
    return [None if k == 0 else rows[b, :min(k, max_out)] for b, k in enumerate(host_counts)]


def non_max_suppression(prediction: torch.Tensor, img_size: float, in_fmt: str = "cxcywh",
                        conf_threshold: float = 0.5, nms_threshold: float = 0.3
                        ) -> List[Union[None, torch.Tensor]]:
    """luoyang/utils/torch_utils.py:101-141.  Rows: [x1,y1,x2,y2,obj_conf,class_conf,class_id]."""
    rows, counts, _ = nms_batched(prediction, img_size, in_fmt, conf_threshold, nms_threshold, 0)
    return _split(rows, counts)


def non_max_suppression_with_point(prediction: torch.Tensor, img_size: float, point_num: int,
                                   in_fmt: str = "cxcywh", conf_threshold: float = 0.5,
                                   nms_threshold: float = 0.3) -> List[Union[None, torch.Tensor]]:
    """luoyang/utils/torch_utils.py:144-191.  Rows: [box, obj, class_conf, 2*P points, class_id]."""
    rows, counts, _ = nms_batched(prediction, img_size, in_fmt, conf_threshold, nms_threshold, 2 * point_num)
    return _split(rows, counts)


def non_max_suppression_with_point_conf(prediction: torch.Tensor, img_size: float, point_num: int,
                                        in_fmt: str = "cxcywh", conf_threshold: float = 0.5,
                                        nms_threshold: float = 0.3) -> List[Union[None, torch.Tensor]]:
    """luoyang/utils/torch_utils.py:194-241.  Rows: [box, obj, class_conf, 2*P points, P point conf, class_id]."""
    rows, counts, _ = nms_batched(prediction, img_size, in_fmt, conf_threshold, nms_threshold, 3 * point_num)
    return _split(rows, counts)


def decode_anchor_levels(raws: List[torch.Tensor], anchors, strides, img_size: int, kind: int, point_num: int = 0,
                         clamp_points: bool = False, arena: Optional[Arena] = None) -> torch.Tensor:
    """raws[i]: (B,h,w,na*C) float32 NHWC head logits, ordered like ``strides``.  Returns the merged
    (B, A, C) tensor of the reference's ``_forward`` (cxcywh, input pixels)."""
    B = raws[0].shape[0]
    na = len(anchors) // len(raws)
    C = raws[0].shape[-1] // na
    A = sum(na * r.shape[1] * r.shape[2] for r in raws)
    out = _out(arena, "decoded", (B, A, C), torch.float32, raws[0].device)
    n = len(raws)
    for r in raws:
        _lib.require_gpu(r, torch.float32, "raw head")
    ptrs = (ctypes.c_void_p * n)(*[r.data_ptr() for r in raws])
    hs = (ctypes.c_int * n)(*[r.shape[1] for r in raws])
    ws = (ctypes.c_int * n)(*[r.shape[2] for r in raws])
    ss = (ctypes.c_int * n)(*[int(s) for s in strides])
    wh = (ctypes.c_float * (2 * na * n))(*[float(v) for a in anchors[:n * na] for v in a])
    _lib.check(LIB.yamd_decode_anchor_levels(kind, ptrs, hs, ws, ss, wh, n, B, na, C, point_num, float(img_size),
                                             int(clamp_points), out.data_ptr(), _lib.current_stream()), "decode_anchor_levels")
    return out


def decode_nms_anchor_levels(raws: List[torch.Tensor], anchors, strides, img_size: int, kind: int, point_num: int,
                             clamp_points: bool, conf_threshold: float, nms_threshold: float, want_index: bool = False,
                             arena: Optional[Arena] = None) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
    """``nms_batched(decode_anchor_levels(raws, ...), img_size, "cxcywh", ...)`` as one library call in which the decode
    kernel filters its own rows (``yamd_decode_nms_anchor``): the same rows, counts and indices bit for bit, without
    writing and re-reading the full decoded tensor.  The predict path's form; ``_forward`` keeps the plain decode."""
    B = raws[0].shape[0]
    n = len(raws)
    na = len(anchors) // n
    C = raws[0].shape[-1] // na
    A = sum(na * r.shape[1] * r.shape[2] for r in raws)
    dev = raws[0].device
    for r in raws:
        _lib.require_gpu(r, torch.float32, "raw head")
    n_extra = {_lib.DECODE_POSE: 3 * point_num, _lib.DECODE_FACE: 2 * point_num}.get(kind, 0)
    pred = _out(arena, "decoded", (B, A, C), torch.float32, dev)
    rows = _out(arena, "nms_rows", (B, A, 7 + n_extra), torch.float32, dev)
    counts = _out(arena, "nms_counts", (B,), torch.int32, dev)
    index = _out(arena, "nms_index", (B, A), torch.int32, dev) if want_index else None
    nbytes = LIB.yamd_nms_workspace_bytes(B, A, C, n_extra)
    if nbytes == 0:
        raise _lib.YamdError(f"bad NMS shape B={B} A={A} C={C} extra={n_extra}")
    ws = _workspace(nbytes, dev) if arena is None else arena.get("nms_workspace", (nbytes,), torch.uint8, dev)
    ptrs = (ctypes.c_void_p * n)(*[r.data_ptr() for r in raws])
    hs = (ctypes.c_int * n)(*[r.shape[1] for r in raws])
    wsz = (ctypes.c_int * n)(*[r.shape[2] for r in raws])
    ss = (ctypes.c_int * n)(*[int(s_) for s_ in strides])
    wh = (ctypes.c_float * (2 * na * n))(*[float(v) for a in anchors[:n * na] for v in a])
    conf32 = float(torch.tensor(conf_threshold, dtype=torch.float32))
    _lib.check(LIB.yamd_decode_nms_anchor(kind, ptrs, hs, wsz, ss, wh, n, B, na, C, point_num, float(img_size), int(clamp_points),
                                          pred.data_ptr(), conf32, float(nms_threshold), rows.data_ptr(),
                                          None if index is None else index.data_ptr(), A, counts.data_ptr(), ws.data_ptr(),
                                          ws.numel(), _lib.current_stream()), "decode_nms_anchor")
    return rows, counts, index


def decode_anchor_head(raw: torch.Tensor, anchors, stride: int, kind: int, point_num: int = 0) -> torch.Tensor:
    """One head module's output: (B, na, h, w, C) in grid units (e.g. yolov5/YoloV5Head.py:23-60)."""
    _lib.require_gpu(raw, torch.float32, "raw head")
    B, h, w, ch = raw.shape
    na = len(anchors)
    C = ch // na
    out = torch.empty((B, na, h, w, C), dtype=torch.float32, device=raw.device)
    wh = (ctypes.c_float * (2 * na))(*[float(v) for a in anchors for v in a])
    _lib.check(LIB.yamd_decode_anchor(kind, raw.data_ptr(), B, h, w, na, C, wh, int(stride), point_num, 0.0, 0, 0,
                                      out.data_ptr(), 0, 0, _lib.current_stream()), "decode_anchor")
    return out


def decode_dfl(raws: List[torch.Tensor], strides, reg_max: int, nc: int, out_fmt: str, unit_conf: bool = False,
               arena: Optional[Arena] = None) -> torch.Tensor:
    """raws[i]: (B,h,w,4*reg_max+nc) float32 NHWC.  Returns (B, A, 5+nc): [box, max score, scores]; with
    ``unit_conf`` (YOLOv6 m/l, reg_max = 17 bins) column 4 is the constant 1.0."""
    B = raws[0].shape[0]
    n = len(raws)
    for r in raws:
        _lib.require_gpu(r, torch.float32, "raw head")
    A = sum(r.shape[1] * r.shape[2] for r in raws)
    out = _out(arena, "decoded", (B, A, 5 + nc), torch.float32, raws[0].device)
    ptrs = (ctypes.c_void_p * n)(*[r.data_ptr() for r in raws])
    hs = (ctypes.c_int * n)(*[r.shape[1] for r in raws])
    ws = (ctypes.c_int * n)(*[r.shape[2] for r in raws])
    ss = (ctypes.c_int * n)(*[int(s) for s in strides])
    fn = LIB.yamd_decode_dfl_unit_conf if unit_conf else LIB.yamd_decode_dfl
    _lib.check(fn(ptrs, hs, ws, ss, n, B, reg_max, nc, 1 if out_fmt == "cxcywh" else 0, out.data_ptr(),
                  _lib.current_stream()), "decode_dfl")
    return out


def decode_ltrb_points(raws: List[torch.Tensor], strides, point_num: int, nc: int, out_fmt: str, bins: int = 1,
                       arena: Optional[Arena] = None) -> torch.Tensor:
    """raws[i]: (B,h,w,4*bins+2P+nc) float32 NHWC [ltrb (bins > 1: 4 x bins DFL logits) | landmarks | cls].  Returns
    (B, A, 5+2P+nc): [box, 1.0, landmarks, scores] (luoyang/yolov6_face/YoloV6Face.py:42-58 and :66-74)."""
    B = raws[0].shape[0]
    n = len(raws)
    for r in raws:
        _lib.require_gpu(r, torch.float32, "raw head")
    A = sum(r.shape[1] * r.shape[2] for r in raws)
    out = _out(arena, "decoded", (B, A, 5 + 2 * point_num + nc), torch.float32, raws[0].device)
    ptrs = (ctypes.c_void_p * n)(*[r.data_ptr() for r in raws])
    hs = (ctypes.c_int * n)(*[r.shape[1] for r in raws])
    ws = (ctypes.c_int * n)(*[r.shape[2] for r in raws])
    ss = (ctypes.c_int * n)(*[int(s) for s in strides])
    cx = 1 if out_fmt == "cxcywh" else 0
    if bins > 1:
        _lib.check(LIB.yamd_decode_dfl_points(ptrs, hs, ws, ss, n, B, bins, point_num, nc, cx, out.data_ptr(),
                                              _lib.current_stream()), "decode_dfl_points")
    else:
        _lib.check(LIB.yamd_decode_ltrb_points(ptrs, hs, ws, ss, n, B, point_num, nc, cx, out.data_ptr(),
                                               _lib.current_stream()), "decode_ltrb_points")
    return out


def fold_conv_bn(conv_w: torch.Tensor, conv_b: Optional[torch.Tensor], bn: nn.BatchNorm2d
                 ) -> Tuple[torch.Tensor, torch.Tensor]:
    """W' = diag(g/sqrt(var+eps)) W ; b' = beta - g*mean/sqrt(var+eps) + diag(..) b  (eval-mode BN)."""
    scale = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
    w = conv_w.detach().double() * scale.view(-1, 1, 1, 1)
    b = bn.bias.detach().double() - bn.running_mean.detach().double() * scale
    if conv_b is not None:
        b = b + conv_b.detach().double() * scale
    return w.float(), b.float()


def fuse_conv_and_bn(conv: nn.Conv2d, bn: nn.BatchNorm2d) -> nn.Conv2d:
    """luoyang/utils/torch_utils.py:344-370: a Conv2d with bias equivalent to bn(conv(x)) in eval mode."""
    assert isinstance(conv, nn.Conv2d) and isinstance(bn, nn.BatchNorm2d)
    fused = nn.Conv2d(conv.in_channels, conv.out_channels, kernel_size=conv.kernel_size, stride=conv.stride,
                      padding=conv.padding, groups=conv.groups, bias=True).requires_grad_(False).to(conv.weight.device)
    w, b = fold_conv_bn(conv.weight, conv.bias, bn)
    fused.weight.copy_(w)
    fused.bias.copy_(b)
    return fused
