"""TEST INFRASTRUCTURE ONLY -- imports the reference (/root/reference) in the BUILD container.

torchvision is not installed here, and the reference imports ``torchvision.ops`` (nms, box_convert,
box_area) at luoyang/utils/torch_utils.py:5-6.  This module registers a stand-in that restates the
published torchvision semantics in plain torch (a deliberately naive, loop-based NMS that shares no
code with oracle/yolo_oracle.c, so the two cross-check each other).  Never imported on the GPU box.
"""
import sys
import types

import torch

REFERENCE_ROOT = "/root/reference"


def _nms(boxes: torch.Tensor, scores: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    n = boxes.shape[0]
    if n == 0:
        return torch.zeros((0,), dtype=torch.int64)
    boxes = boxes.float()
    order = torch.sort(scores.float(), stable=True, descending=True)[1].tolist()
    x1, y1, x2, y2 = [boxes[:, i] for i in range(4)]
    area = (x2 - x1) * (y2 - y1)
    thr = torch.tensor(iou_threshold, dtype=torch.float64)
    dead = [False] * n
    keep = []
    for a, i in enumerate(order):
        if dead[i]:
            continue
        keep.append(i)
        rest = [j for j in order[a + 1:] if not dead[j]]
        if not rest:
            continue
        r = torch.tensor(rest, dtype=torch.int64)
        xx1 = torch.maximum(x1[i], x1[r])
        yy1 = torch.maximum(y1[i], y1[r])
        xx2 = torch.minimum(x2[i], x2[r])
        yy2 = torch.minimum(y2[i], y2[r])
        w = torch.clamp(xx2 - xx1, min=0)
        h = torch.clamp(yy2 - yy1, min=0)
        inter = w * h
        ovr = inter / ((area[i] + area[r]) - inter)          # float32, one rounding per op
        sup = ovr.double() > thr                              # compared against the double threshold
        for j, s in zip(rest, sup.tolist()):
            if s:
                dead[j] = True
    return torch.tensor(keep, dtype=torch.int64)


def _box_convert(boxes: torch.Tensor, in_fmt: str, out_fmt: str) -> torch.Tensor:
    if in_fmt == out_fmt:
        return boxes.clone()
    if in_fmt == "cxcywh" and out_fmt == "xyxy":
        cx, cy, w, h = boxes.unbind(-1)
        return torch.stack((cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h), dim=-1)
    if in_fmt == "xyxy" and out_fmt == "cxcywh":
        x1, y1, x2, y2 = boxes.unbind(-1)
        return torch.stack(((x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1), dim=-1)
    raise ValueError((in_fmt, out_fmt))


def _box_area(boxes: torch.Tensor) -> torch.Tensor:
    return (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])


def install() -> None:
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        ops = types.ModuleType("torchvision.ops")
        ops.nms, ops.box_convert, ops.box_area = _nms, _box_convert, _box_area
        tv.ops = ops
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.ops"] = ops
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
