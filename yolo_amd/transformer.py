"""Inference pre/post-processing with the reference's names (luoyang/transformer/transformer.py).

``letterbox_batch`` is the device counterpart of ``ResizeImage(is_train=False)`` + ``ImageNorm`` applied to a list
of images (:134-178, :204-232): uint8 HWC RGB images of any size -> one (B,3,H,W) float32 CUDA tensor in [0,1],
resized exactly like ``PIL.Image.resize(..., Image.BICUBIC)`` and centred on a (128,128,128) canvas, plus the
per-image bookkeeping dictionaries the reference stores in ``data_dict``.  ``RestoreSize`` maps predicted boxes back
to the original image (:260-295) with the reference's integer semantics.
"""
import ctypes
from typing import Any, Dict, List, Sequence, Tuple, Union

import numpy as np
import torch

from . import lib as _lib

LIB = _lib.LIB


def letterbox_batch(images: Sequence[Union[np.ndarray, torch.Tensor]], target_height: int, target_width: int,
                    device="cuda") -> Tuple[torch.Tensor, List[Dict[str, Any]]]:
    dev = torch.device(device)
    out = torch.empty((len(images), 3, target_height, target_width), dtype=torch.float32, device=dev)
    metas: List[Dict[str, Any]] = []
    st = _lib.current_stream()
    need = 0
    srcs = []
    for im in images:
        t = torch.as_tensor(im)
        if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
            raise _lib.YamdError("images must be uint8 arrays of shape (H, W, 3)")
        t = t.to(dev).contiguous()
        srcs.append(t)
        nb = LIB.yamd_letterbox_workspace_bytes(t.shape[0], t.shape[1], target_height, target_width)
        if nb == 0:
            raise _lib.YamdError(f"cannot letterbox a {t.shape[0]}x{t.shape[1]} image to {target_height}x{target_width}")
        need = max(need, nb)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)          # reused image after image (stream ordered)
    geom = (ctypes.c_int * 4)()
    ratio = ctypes.c_double()
    for i, t in enumerate(srcs):
        h, w = int(t.shape[0]), int(t.shape[1])
        _lib.check(LIB.yamd_letterbox_u8(t.data_ptr(), h, w, 3 * w, target_height, target_width, out[i].data_ptr(), geom,
                                         ctypes.byref(ratio), ws.data_ptr(), ws.numel(), st), "letterbox_u8")
        metas.append(dict(ori_height=h, ori_width=w, resize_ratio=float(ratio.value), resize_w=int(geom[0]),
                          resize_h=int(geom[1]), dw=int(geom[2]), dh=int(geom[3])))
    return out, metas


class RestoreSize(object):
    """``RestoreSize().transformer(data_dict)`` as in the reference: reads ``predict`` (ndarray (K, W) or None) and the
    ResizeImage bookkeeping, adds ``predict_boxs`` (int32 (K,4)) and ``predict_label`` (int32 (K,))."""

    def transformer(self, data_dict: Dict[str, Any]):
        predict = data_dict["predict"]
        if predict is not None:
            resize_ratio = data_dict["resize_ratio"]
            dw, dh = data_dict["dw"], data_dict["dh"]
            bbox = predict[:, :4].astype(np.int32)              # truncation happens before the un-letterboxing
            bbox[:, 0] = (bbox[:, 0] - dw) / resize_ratio
            bbox[:, 1] = (bbox[:, 1] - dh) / resize_ratio
            bbox[:, 2] = (bbox[:, 2] - dw) / resize_ratio
            bbox[:, 3] = (bbox[:, 3] - dh) / resize_ratio
            bbox[:, 0] = np.minimum(np.maximum(0, bbox[:, 0]), data_dict["ori_width"])
            bbox[:, 2] = np.minimum(np.maximum(0, bbox[:, 2]), data_dict["ori_width"])
            bbox[:, 1] = np.minimum(np.maximum(0, bbox[:, 1]), data_dict["ori_height"])
            bbox[:, 3] = np.minimum(np.maximum(0, bbox[:, 3]), data_dict["ori_height"])
            data_dict["predict_boxs"] = bbox
            data_dict["predict_label"] = predict[:, -1].astype(np.int32)
        else:
            data_dict["predict_boxs"] = []
            data_dict["predict_label"] = []

    def batch_transformer(self, data_dict_list: List[Dict[str, Any]]):
        for d in data_dict_list:
            self.transformer(d)
