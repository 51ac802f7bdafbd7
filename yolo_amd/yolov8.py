"""YOLOv8 detect on the HIP path, behind the reference's TaskLayer interface.

Mirrors luoyang/yolov8/YoloV8.py (``_forward`` :39-57, ``_feed_forward`` :59-73, ``predict`` :108-134) and the
module tree of luoyang/yolov8/YoloV8Body.py (Conv :26-63, Bottleneck :66-80, C2f :83-99, SPPF :102-114,
Backbone :117-151, YoloV8Body :226-308) plus the three constant Parameters of luoyang/yolov8/YoloV8Head.py:20-26,
so ``state_dict`` keys are identical and reference checkpoints load unchanged.
"""
from typing import Any, Dict, List

import torch
import torch.nn as nn

from . import lib as _lib
from . import torch_utils
from .engine import Program, View
from .graph import Recorder, TaskModel, conv_bn
from .param import Yolo8Param

_DEPTH = {"n": 0.33, "s": 0.33, "m": 0.67, "l": 1.00, "x": 1.00}
_WIDTH = {"n": 0.25, "s": 0.50, "m": 0.75, "l": 1.00, "x": 1.25}
_DEEP = {"n": 1.00, "s": 1.00, "m": 0.75, "l": 0.50, "x": 0.50}
REG_MAX = 16


def c2f_block(c1: int, c2: int, n: int) -> nn.Module:
    """C2f (YoloV8Body.py:83-99): cv1 1x1 to 2c, n bottlenecks of two 3x3 convs on c channels, cv2 1x1."""
    m = nn.Module()
    m.c = int(c2 * 0.5)
    m.cv1 = conv_bn(c1, 2 * m.c)
    m.cv2 = conv_bn((2 + n) * m.c, c2)
    bots = []
    for _ in range(n):
        b = nn.Module()
        b.cv1, b.cv2 = conv_bn(m.c, m.c, 3), conv_bn(m.c, m.c, 3)
        bots.append(b)
    m.m = nn.ModuleList(bots)
    return m


def sppf_block(c1: int, c2: int) -> nn.Module:
    m = nn.Module()
    m.cv1, m.cv2 = conv_bn(c1, c1 // 2), conv_bn(c1 // 2 * 4, c2)
    return m


def _tree(nc: int, phi: str) -> nn.Module:
    bc, bd, dm = int(_WIDTH[phi] * 64), max(round(_DEPTH[phi] * 3), 1), _DEEP[phi]
    c5 = int(bc * 16 * dm)
    body = nn.Module()
    bb = nn.Module()
    bb.stem = conv_bn(3, bc, 3, 2)
    bb.dark2 = nn.Sequential(conv_bn(bc, bc * 2, 3, 2), c2f_block(bc * 2, bc * 2, bd))
    bb.dark3 = nn.Sequential(conv_bn(bc * 2, bc * 4, 3, 2), c2f_block(bc * 4, bc * 4, bd * 2))
    bb.dark4 = nn.Sequential(conv_bn(bc * 4, bc * 8, 3, 2), c2f_block(bc * 8, bc * 8, bd * 2))
    bb.dark5 = nn.Sequential(conv_bn(bc * 8, c5, 3, 2), c2f_block(c5, c5, bd), sppf_block(c5, c5))
    body.backbone = bb
    body.conv3_for_up_sample1 = c2f_block(c5 + bc * 8, bc * 8, bd)
    body.conv3_for_up_sample2 = c2f_block(bc * 8 + bc * 4, bc * 4, bd)
    body.down_sample1 = conv_bn(bc * 4, bc * 4, 3, 2)
    body.conv3_for_down_sample1 = c2f_block(bc * 8 + bc * 4, bc * 8, bd)
    body.down_sample2 = conv_bn(bc * 8, bc * 8, 3, 2)
    body.conv3_for_down_sample2 = c2f_block(c5 + bc * 8, c5, bd)
    ch = [bc * 4, bc * 8, c5]
    c2, c3 = max(16, ch[0] // 4, REG_MAX * 4), max(ch[0], nc)
    body.cv2 = nn.ModuleList(nn.Sequential(conv_bn(x, c2, 3), conv_bn(c2, c2, 3), nn.Conv2d(c2, 4 * REG_MAX, 1))
                             for x in ch)
    body.cv3 = nn.ModuleList(nn.Sequential(conv_bn(x, c3, 3), conv_bn(c3, c3, 3), nn.Conv2d(c3, nc, 1)) for x in ch)
    body.dfl = nn.Module()
    body.dfl.conv = nn.Conv2d(REG_MAX, 1, 1, bias=False).requires_grad_(False)       # unused by the forward pass
    body.dfl.conv.weight.data[:] = torch.arange(REG_MAX, dtype=torch.float).view(1, REG_MAX, 1, 1)
    return body


def make_anchors(feature_size: List[int], strides: List[int], offset: float = 0.5):
    """Cell centres (A,2) as (x,y) in grid units and the per-anchor stride (A,1) (YoloV8Body.py:204-223)."""
    pts, st = [], []
    for f, s in zip(feature_size, strides):
        ax = torch.arange(f) + offset
        sy, sx = torch.meshgrid(ax, ax, indexing="ij")
        pts.append(torch.stack((sx, sy), -1).view(-1, 2))
        st.append(torch.full((f * f, 1), s))
    return torch.cat(pts), torch.cat(st)


class YoloV8(TaskModel):
    """``YoloV8(param)`` as in the reference.  Images must be CUDA float32 NCHW in [0,1]."""

    def __init__(self, param: Yolo8Param):
        super().__init__()
        self.param = param
        self.body_net = _tree(param.class_num, param.m_type)
        feature_size = [param.img_size // s for s in param.strides]
        head = nn.Module()
        head.proj = nn.Parameter(torch.arange(REG_MAX, dtype=torch.float), requires_grad=False)
        pts, st = make_anchors(feature_size, param.strides)
        head.anchor_points = nn.Parameter(pts, requires_grad=False)
        head.stride_tensor = nn.Parameter(st, requires_grad=False)
        self.head = head
        self.eval()

    def _c2f(self, R: Recorder, name: str, m: nn.Module, x: View, shortcut: bool) -> View:
        ys = [R.cba(f"{name}.cv1", m.cv1, x)]
        last = ys[0].slice(m.c, m.c)
        for i, b in enumerate(m.m):
            t = R.cba(f"{name}.m.{i}.cv1", b.cv1, last)
            last = R.cba(f"{name}.m.{i}.cv2", b.cv2, t, residual=last if shortcut else None)
            ys.append(last)
        return R.cba(f"{name}.cv2", m.cv2, R.p.cat(ys))

    def _compile(self, S: int) -> Program:
        body, bb = self.body_net, self.body_net.backbone
        P = Program(S, S)
        R = Recorder(P, _lib.ACT_SILU)
        x = R.cba("stem", bb.stem, P.image16(3))
        x = self._c2f(R, "dark2.1", bb.dark2[1], R.cba("dark2.0", bb.dark2[0], x), True)
        feat1 = self._c2f(R, "dark3.1", bb.dark3[1], R.cba("dark3.0", bb.dark3[0], x), True)
        feat2 = self._c2f(R, "dark4.1", bb.dark4[1], R.cba("dark4.0", bb.dark4[0], feat1), True)
        x = self._c2f(R, "dark5.1", bb.dark5[1], R.cba("dark5.0", bb.dark5[0], feat2), True)
        sp = bb.dark5[2]
        hc = sp.cv1.conv.out_channels
        cat = P.buffer(x.h, x.w, 4 * hc, name="cat_sppf")
        R.cba("sppf.cv1", sp.cv1, x, out=cat.slice(0, hc))
        P.maxpool5_cascade(cat.slice(0, hc), cat.slice(hc, hc), cat.slice(2 * hc, hc), cat.slice(3 * hc, hc))
        feat3 = R.cba("sppf.cv2", sp.cv2, cat)

        p4 = self._c2f(R, "conv3_for_up_sample1", body.conv3_for_up_sample1, P.cat([P.upsample2x(feat3), feat2]), False)
        p3 = self._c2f(R, "conv3_for_up_sample2", body.conv3_for_up_sample2, P.cat([P.upsample2x(p4), feat1]), False)
        d = R.cba("down_sample1", body.down_sample1, p3)
        p4 = self._c2f(R, "conv3_for_down_sample1", body.conv3_for_down_sample1, P.cat([d, p4]), False)
        d = R.cba("down_sample2", body.down_sample2, p4)
        p5 = self._c2f(R, "conv3_for_down_sample2", body.conv3_for_down_sample2, P.cat([d, feat3]), False)
        nc = self.param.class_num
        for i, t in enumerate((p3, p4, p5)):
            raw = P.buffer(t.h, t.w, 4 * REG_MAX + nc, torch.float32, name=f"raw{i}")    # [reg 64 | cls nc]
            for br, off, c in ((body.cv2[i], 0, 4 * REG_MAX), (body.cv3[i], 4 * REG_MAX, nc)):
                tag = "cv2" if off == 0 else "cv3"
                a = R.cba(f"{tag}.{i}.0", br[0], t)
                a = R.cba(f"{tag}.{i}.1", br[1], a)
                R.head(f"{tag}.{i}.2", br[2], a, out=raw.slice(off, c))
            P.outputs[f"L{i}"] = raw
        return P

    # ---- reference interface -------------------------------------------------------------------
    def raw_heads(self, images: torch.Tensor) -> List[torch.Tensor]:
        """Raw head tensors per level, float32 NHWC (B,h,w,64+nc), channel order [reg | cls]."""
        out = self.program().run(images)
        return [out["L0"], out["L1"], out["L2"]]

    def _decode(self, raws, fmt: str, strides=None, arena=None) -> torch.Tensor:
        return torch_utils.decode_dfl(raws, self.param.strides if strides is None else strides, REG_MAX,
                                      self.param.class_num, fmt, arena=arena)

    def _forward(self, images: torch.Tensor) -> List[torch.Tensor]:
        """[(B, A, 5+nc)]: cxcywh in input pixels, max class score, class scores (YoloV8.py:39-57)."""
        return [self._decode(self.raw_heads(images), "cxcywh")]

    def _feed_forward(self, tensor_dict: Dict[str, Any]) -> Dict[str, Any]:
        raws = self.raw_heads(tensor_dict["batch_images"])
        B = raws[0].shape[0]
        unit = self._decode(raws, "xyxy", strides=[1] * len(raws))         # boxes in grid units
        tensor_dict["predict_scores"] = unit[..., 5:]
        tensor_dict["predict_scaled_bounding_boxes"] = unit[..., :4]
        dev = raws[0].device
        tensor_dict["stride_tensor"] = self.head.stride_tensor.to(dev)
        tensor_dict["anchor_points"] = self.head.anchor_points.to(dev)
        tensor_dict["predict_distribute"] = torch.cat([r[..., :4 * REG_MAX].reshape(B, -1, 4 * REG_MAX) for r in raws], 1)
        tensor_dict["_raw_heads"] = raws
        return tensor_dict

    def forward(self, tensor_dict: Dict[str, Any]) -> Dict[str, Any]:
        """Inference keys of YoloV8.py:59-73 (target assignment of :75-106 belongs to training)."""
        tensor_dict = self._feed_forward(tensor_dict)
        tensor_dict.pop("_raw_heads")
        return tensor_dict

    def decode_predict(self, raws, arena=None) -> torch.Tensor:
        return self._decode(raws, "xyxy", arena=arena)

    def nms_predict(self, pred: torch.Tensor, arena=None):
        rows, counts, _ = torch_utils.nms_batched(pred, self.param.img_size, "xyxy", self.param.conf_threshold,
                                                  self.param.nms_threshold, arena=arena)
        return rows, counts

    def predict(self, tensor_dict: Dict[str, Any]) -> Dict[str, Any]:
        """Adds ``predicts``: per image ``None`` or ndarray (K,7) [x1,y1,x2,y2,conf,cls_conf,cls_id]."""
        tensor_dict = self._feed_forward(tensor_dict)
        rows, counts = self.nms_predict(self.decode_predict(tensor_dict.pop("_raw_heads")))
        tensor_dict["predicts"] = torch_utils.rows_to_host(rows, counts)
        return tensor_dict
