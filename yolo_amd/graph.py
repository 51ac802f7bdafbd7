"""Shared pieces of the model descriptions: parameter holders with the reference's attribute names,
the recorder that turns them into fused conv ops, and the TaskLayer-shaped base class.

A model module (``yolov5``, ``yolov8`` ...) builds a tree of bare ``nn.Module`` holders whose
``state_dict`` keys equal the reference's, then walks it once per image size to record an
``engine.Program``.  BatchNorm folding, RepVGG re-parameterisation and bf16 packing all happen at
that point ("compile"), never per batch.
"""
from typing import Any, Dict, List, Optional

import torch
import torch.nn as nn

from . import lib as _lib
from . import torch_utils
from .engine import Program, View


def conv_bn(cin: int, cout: int, k: int = 1, s: int = 1, eps: float = 1e-3, momentum: float = 0.03,
            conv_name: str = "conv", bn_name: str = "bn") -> nn.Module:
    """Holder of a bias-free conv followed by BatchNorm under the given attribute names."""
    m = nn.Module()
    setattr(m, conv_name, nn.Conv2d(cin, cout, (k, k), (s, s), k // 2, bias=False))
    setattr(m, bn_name, nn.BatchNorm2d(cout, eps=eps, momentum=momentum))
    return m


def folded(m: nn.Module, conv_name: str = "conv", bn_name: str = "bn"):
    """(weight, bias, stride) of a conv(+bn) holder with the BatchNorm folded in fp64; a holder whose bn was
    removed by ``fuse()`` carries its bias on the conv."""
    conv = getattr(m, conv_name)
    bn = getattr(m, bn_name, None)
    if isinstance(bn, nn.BatchNorm2d):
        w, b = torch_utils.fold_conv_bn(conv.weight, conv.bias, bn)
    else:
        w = conv.weight.detach()
        b = None if conv.bias is None else conv.bias.detach()
    return w, b, conv.stride[0]


class Recorder:
    """Records conv(+bn)+activation holders into a Program."""

    def __init__(self, prog: Program, act: int):
        self.p = prog
        self.act = act

    def cba(self, name: str, m: nn.Module, x: View, out: Optional[View] = None, residual: Optional[View] = None,
            act: Optional[int] = None, conv_name: str = "conv", bn_name: str = "bn") -> View:
        w, b, s = folded(m, conv_name, bn_name)
        return self.p.conv(name, x, w, b, s, self.act if act is None else act, out=out, residual=residual)

    def head(self, name: str, conv: nn.Conv2d, x: View, out: Optional[View] = None) -> View:
        """Bias-only 1x1 conv producing float32 logits."""
        return self.p.conv(name, x, conv.weight.detach(), None if conv.bias is None else conv.bias.detach(), 1,
                           _lib.ACT_NONE, out=out, out_f32=True)


class TaskModel(nn.Module):
    """Plan cache + the parts of the reference's TaskLayer protocol that do not depend on the model."""

    def __init__(self):
        super().__init__()
        self._program: Optional[Program] = None
        self._program_key = None
        self._arena = torch_utils.Arena()      # decode / NMS buffers of predict_device, re-used from call to call

    def invalidate(self) -> None:
        """Drop the compiled plan (call after changing weights in place)."""
        self._program = None

    def load_state_dict(self, *a, **k):
        self.invalidate()
        return super().load_state_dict(*a, **k)

    def _plan_key(self):
        """Everything the compiled plan depends on: the identity and version counter of every parameter and buffer
        (an in-place update, ``load_state_dict`` on the model or on any submodule, ``.data`` assignment and ``fuse()``
        all change one of the two) and the Param fields ``_compile`` and the decode read."""
        p = self.param
        weights = tuple((t.data_ptr(), t._version) for t in list(self.parameters()) + list(self.buffers()))
        cfg = (p.img_size, getattr(p, "class_num", None), repr(getattr(p, "anchors", None)),
               tuple(getattr(p, "strides", ())), getattr(p, "m_type", None))
        return cfg, weights

    def program(self) -> Program:
        key = self._plan_key()
        if self._program is None or self._program_key != key:
            if self.param.img_size % max(self.param.strides):
                raise _lib.YamdError(f"img_size must be a multiple of {max(self.param.strides)}")
            self._program = self._compile(self.param.img_size)
            self._program_key = key
        return self._program

    def _compile(self, size: int) -> Program:      # pragma: no cover - abstract
        raise NotImplementedError

    def fuse_conv_bn_holders(self, conv_name: str = "conv", bn_name: str = "bn") -> None:
        """Fold every (conv, bn) holder pair into a conv with bias and delete the bn, which is what the
        reference's per-model ``fuse()`` leaves in the ``state_dict``."""
        for m in self.modules():
            if isinstance(getattr(m, bn_name, None), nn.BatchNorm2d) and isinstance(getattr(m, conv_name, None), nn.Conv2d):
                setattr(m, conv_name, torch_utils.fuse_conv_and_bn(getattr(m, conv_name), getattr(m, bn_name)))
                delattr(m, bn_name)
        self.invalidate()

    # the three stages of predict(): raw head logits -> decoded (B,A,C) tensor -> kept rows, all on the device
    def raw_heads(self, images: torch.Tensor) -> List[torch.Tensor]:      # pragma: no cover - abstract
        raise NotImplementedError

    def decode_predict(self, raws, arena=None) -> torch.Tensor:           # pragma: no cover - abstract
        raise NotImplementedError

    def nms_predict(self, pred: torch.Tensor, arena=None):                # pragma: no cover - abstract
        raise NotImplementedError

    def predict_device(self, images: torch.Tensor):
        """Whole path on the device: returns (rows (B,A,W), counts (B,)) without synchronising.

        Activations, the decoded tensor, the NMS scratch and both results live in buffers that persist between
        calls (``Program`` arena + ``self._arena``): a call with the same batch size allocates nothing and writes
        to the same addresses as the one before, so the results are valid until the next call on this model.
        (The call also records into a HIP graph as kernel nodes only, but launching one such graph repeatedly is
        not supported: see ``yamd_init`` in include/yolo_amd.h.)"""
        return self.nms_predict(self.decode_predict(self.raw_heads(images), self._arena), self._arena)

    def fuse(self):
        """Models without a re-parameterisation step in the reference raise there too (model/Layer.py TaskLayer.fuse);
        BatchNorm is folded when the plan is compiled, so callers have nothing to do."""
        raise NotImplementedError(f"{type(self).__name__}.fuse is not implemented in the reference")

    def to_onnx(self, onnx_path: str):
        raise NotImplementedError("ONNX export is outside the accelerated path; use the reference for export")


class AnchorTaskModel(TaskModel):
    """Anchor-based families (v3, v4, v5, v7 pose/face): three raw heads -> per-level decode -> merge -> NMS.
    Subclasses provide ``_compile`` (outputs "P3","P4","P5") and set the class attributes below.  The methods
    follow e.g. luoyang/yolov3/YoloV3.py: ``_feed_forward`` :66-83, ``predict`` :118-148, ``_forward`` :191-213 and
    luoyang/yolov7_pose/YoloV7Pose.py:57-83, :132-168 (keypoint columns scaled too, clamped in ``predict`` only)."""

    DECODE_KIND = _lib.DECODE_V5
    POINT_NUM = 0            # keypoints / landmarks per box
    POINT_CONF = False       # True: one confidence column per keypoint (pose)
    CLAMP_POINTS_IN_PREDICT = None   # None: as POINT_CONF; the face models set True (their predict() clamps landmarks too)

    def raw_heads(self, images: torch.Tensor) -> List[torch.Tensor]:
        """Raw head logits [P3, P4, P5], float32 NHWC (B,h,w,3*C)."""
        out = self.program().run(images)
        return [out["P3"], out["P4"], out["P5"]]

    def _level_anchors(self, level: int):
        na = self.param.anchor_num
        return self.param.anchors[level * na:(level + 1) * na]

    def _merge(self, raws, clamp_points: bool, arena=None) -> torch.Tensor:
        return torch_utils.decode_anchor_levels(raws, self.param.anchors, self.param.strides, self.param.img_size,
                                                self.DECODE_KIND, self.POINT_NUM, clamp_points, arena=arena)

    def _forward(self, images: torch.Tensor) -> List[torch.Tensor]:
        """[(B, A, C)] merged over levels, cxcywh in input pixels."""
        return [self._merge(self.raw_heads(images), False)]

    def _feed_forward(self, tensor_dict: Dict[str, Any]) -> Dict[str, Any]:
        raws = self.raw_heads(tensor_dict["batch_images"])
        for i, name in enumerate(("head3_predicts", "head4_predicts", "head5_predicts")):
            tensor_dict[name] = torch_utils.decode_anchor_head(raws[i], self._level_anchors(i), self.param.strides[i],
                                                               self.DECODE_KIND, self.POINT_NUM)
        tensor_dict["_raw_heads"] = raws
        return tensor_dict

    def forward(self, tensor_dict: Dict[str, Any]) -> Dict[str, Any]:
        """Inference keys only (``head3/4/5_predicts``); target assignment belongs to training."""
        tensor_dict = self._feed_forward(tensor_dict)
        tensor_dict.pop("_raw_heads")
        return tensor_dict

    def _n_extra(self) -> int:
        return (3 if self.POINT_CONF else 2) * self.POINT_NUM

    def decode_predict(self, raws, arena=None) -> torch.Tensor:
        clamp = self.POINT_CONF if self.CLAMP_POINTS_IN_PREDICT is None else self.CLAMP_POINTS_IN_PREDICT
        return self._merge(raws, self.POINT_NUM > 0 and clamp, arena)

    def nms_predict(self, pred: torch.Tensor, arena=None):
        rows, counts, _ = torch_utils.nms_batched(pred, self.param.img_size, "cxcywh", self.param.conf_threshold,
                                                  self.param.nms_threshold, self._n_extra(), arena=arena)
        return rows, counts

    def decode_nms_predict(self, raws, arena=None):
        """``nms_predict(decode_predict(raws))`` with the confidence filter inside the decode kernel: same rows and counts."""
        clamp = self.POINT_CONF if self.CLAMP_POINTS_IN_PREDICT is None else self.CLAMP_POINTS_IN_PREDICT
        rows, counts, _ = torch_utils.decode_nms_anchor_levels(raws, self.param.anchors, self.param.strides, self.param.img_size,
                                                               self.DECODE_KIND, self.POINT_NUM, self.POINT_NUM > 0 and clamp,
                                                               self.param.conf_threshold, self.param.nms_threshold, arena=arena)
        return rows, counts

    def predict_device(self, images: torch.Tensor):
        return self.decode_nms_predict(self.raw_heads(images), self._arena)

    def predict(self, tensor_dict: Dict[str, Any]) -> Dict[str, Any]:
        """Adds ``predicts``: per image ``None`` or ndarray (K, W) in the reference's row layout."""
        tensor_dict = self._feed_forward(tensor_dict)
        rows, counts = self.decode_nms_predict(tensor_dict.pop("_raw_heads"))
        tensor_dict["predicts"] = torch_utils.rows_to_host(rows, counts)     # one packed D2H for the batch
        return tensor_dict
