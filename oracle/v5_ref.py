"""TEST INFRASTRUCTURE ONLY -- PyTorch-CPU fp32 restatement of the YOLOv5 conv graph.

A functional interpreter over a ``state_dict`` that carries the reference's key names, so the
same weights drive the reference, this oracle and the HIP path.

Follows luoyang/yolov5/YoloV5Body.py: Conv :26-51, Focus :54-77, Bottleneck :80-93, C3 :96-111,
SPPF (parallel k=5,9,13) :114-132, CSPDarknet :135-168, YoloV5Body.forward :223-252.
BatchNorm is eval-mode with eps=1e-3 (:42); a fused checkpoint (conv.bias present, no bn.*)
is accepted too (luoyang/yolov5/YoloV5.py:171-176).
"""
from typing import Dict, List

import torch
import torch.nn.functional as F

BN_EPS = 1e-3


def _conv_block(sd: Dict[str, torch.Tensor], p: str, x: torch.Tensor, s: int = 1) -> torch.Tensor:
    w = sd[p + ".conv.weight"]
    k = w.shape[-1]
    if p + ".bn.weight" in sd:
        y = F.conv2d(x, w, None, stride=s, padding=k // 2)
        y = F.batch_norm(y, sd[p + ".bn.running_mean"], sd[p + ".bn.running_var"], sd[p + ".bn.weight"],
                         sd[p + ".bn.bias"], False, 0.0, BN_EPS)
    else:
        y = F.conv2d(x, w, sd[p + ".conv.bias"], stride=s, padding=k // 2)
    return y * torch.sigmoid(y)


def _depth(sd, p: str) -> int:
    n = 0
    while f"{p}.m.{n}.cv1.conv.weight" in sd:
        n += 1
    return n


def _c3(sd, p: str, x: torch.Tensor, shortcut: bool) -> torch.Tensor:
    a = _conv_block(sd, p + ".cv1", x)
    for i in range(_depth(sd, p)):
        y = _conv_block(sd, f"{p}.m.{i}.cv2", _conv_block(sd, f"{p}.m.{i}.cv1", a))
        a = a + y if shortcut else y
    b = _conv_block(sd, p + ".cv2", x)
    return _conv_block(sd, p + ".cv3", torch.cat((a, b), 1))


def _spp(sd, p: str, x: torch.Tensor, ks=(5, 9, 13)) -> torch.Tensor:
    x = _conv_block(sd, p + ".cv1", x)
    pools = [F.max_pool2d(x, k, 1, k // 2) for k in ks]
    return _conv_block(sd, p + ".cv2", torch.cat([x] + pools, 1))


def body_forward(sd: Dict[str, torch.Tensor], images: torch.Tensor, taps: Dict[str, torch.Tensor] = None
                 ) -> List[torch.Tensor]:
    """images (B,3,S,S) fp32 -> raw head tensors [P3 (stride 8), P4, P5], each (B, 3*(5+nc), H, W)."""
    bb = "body_net.backbone."
    x = torch.cat((images[..., ::2, ::2], images[..., 1::2, ::2], images[..., ::2, 1::2], images[..., 1::2, 1::2]), 1)
    x = _conv_block(sd, bb + "stem.conv", x)
    if taps is not None:
        taps["stem"] = x
    x = _c3(sd, bb + "dark2.1", _conv_block(sd, bb + "dark2.0", x, 2), True)
    if taps is not None:
        taps["dark2"] = x
    feat1 = _c3(sd, bb + "dark3.1", _conv_block(sd, bb + "dark3.0", x, 2), True)
    feat2 = _c3(sd, bb + "dark4.1", _conv_block(sd, bb + "dark4.0", feat1, 2), True)
    x = _conv_block(sd, bb + "dark5.0", feat2, 2)
    x = _spp(sd, bb + "dark5.1", x)
    feat3 = _c3(sd, bb + "dark5.2", x, False)
    if taps is not None:
        taps.update(feat1=feat1, feat2=feat2, feat3=feat3)
    n = "body_net."
    p5 = _conv_block(sd, n + "conv_for_feat3", feat3)
    p4 = torch.cat((F.interpolate(p5, scale_factor=2, mode="nearest"), feat2), 1)
    p4 = _c3(sd, n + "conv3_for_up_sample1", p4, False)
    p4 = _conv_block(sd, n + "conv_for_feat2", p4)
    p3 = torch.cat((F.interpolate(p4, scale_factor=2, mode="nearest"), feat1), 1)
    p3 = _c3(sd, n + "conv3_for_up_sample2", p3, False)
    p4 = torch.cat((_conv_block(sd, n + "down_sample1", p3, 2), p4), 1)
    p4 = _c3(sd, n + "conv3_for_down_sample1", p4, False)
    p5 = torch.cat((_conv_block(sd, n + "down_sample2", p4, 2), p5), 1)
    p5 = _c3(sd, n + "conv3_for_down_sample2", p5, False)
    if taps is not None:
        taps.update(p3=p3, p4=p4, p5=p5)
    outs = []
    for name, t in (("yolo_head_P3", p3), ("yolo_head_P4", p4), ("yolo_head_P5", p5)):
        outs.append(F.conv2d(t, sd[n + name + ".weight"], sd[n + name + ".bias"]))
    return outs


def face_body_forward(sd: Dict[str, torch.Tensor], images: torch.Tensor) -> List[torch.Tensor]:
    """The v5 face / pose graph (luoyang/yolov5_face/Yolov5FaceBody.py: StemBlock :54-69, SPPF k=3,5,7 :106-125,
    CSPDarknet :128-160, body forward :217-248; yolov5_pose/YoloV5PoseBody.py is the same graph)."""
    bb = "body_net.backbone."
    s1 = _conv_block(sd, bb + "stem.stem_1", images, 2)
    s2 = _conv_block(sd, bb + "stem.stem_2b", _conv_block(sd, bb + "stem.stem_2a", s1), 2)
    x = _conv_block(sd, bb + "stem.stem_3", torch.cat((s2, F.max_pool2d(s1, 2, 2, ceil_mode=True)), 1))
    x = _c3(sd, bb + "dark2.0", x, True)
    feat1 = _c3(sd, bb + "dark3.1", _conv_block(sd, bb + "dark3.0", x, 2), True)
    feat2 = _c3(sd, bb + "dark4.1", _conv_block(sd, bb + "dark4.0", feat1, 2), True)
    x = _spp(sd, bb + "dark5.1", _conv_block(sd, bb + "dark5.0", feat2, 2), (3, 5, 7))
    feat3 = _c3(sd, bb + "dark5.2", x, False)
    n = "body_net."
    p5 = _conv_block(sd, n + "conv_for_feat3", feat3)
    p4 = _c3(sd, n + "conv3_for_up_sample1", torch.cat((F.interpolate(p5, scale_factor=2, mode="nearest"), feat2), 1), False)
    p4 = _conv_block(sd, n + "conv_for_feat2", p4)
    p3 = _c3(sd, n + "conv3_for_up_sample2", torch.cat((F.interpolate(p4, scale_factor=2, mode="nearest"), feat1), 1), False)
    p4 = _c3(sd, n + "conv3_for_down_sample1", torch.cat((_conv_block(sd, n + "down_sample1", p3, 2), p4), 1), False)
    p5 = _c3(sd, n + "conv3_for_down_sample2", torch.cat((_conv_block(sd, n + "down_sample2", p4, 2), p5), 1), False)
    return [F.conv2d(t, sd[n + name + ".weight"], sd[n + name + ".bias"])
            for name, t in (("yolo_head_P3", p3), ("yolo_head_P4", p4), ("yolo_head_P5", p5))]
