"""YOLOv7 pose (17 keypoints) on the HIP path, behind the reference's TaskLayer interface.

Mirrors luoyang/yolov7_pose/YoloV7Pose.py (``_forward`` :57-83, ``predict`` :132-168, ``fuse`` :219-227) and the module
tree of luoyang/yolov7_pose/YoloV7PoseBody.py (Conv :24-48, MultiConcatBlock :51-73, MP/TransitionBlock :76-100,
Backbone :103-148, SppCspC :175-192, RepConv :195-380, YoloV7PoseBody :402-509), so ``state_dict`` keys are identical
before and after ``fuse()``.  RepConv (3x3+BN, 1x1+BN and optional identity BN branches) is re-parameterised into
one 3x3 conv with bias when the plan is compiled.  Both widths of the reference are built: l, and x (40-channel base
width, five-way concats, plain convs in place of RepConv, :402-478).
"""
from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import lib as _lib
from . import torch_utils
from .engine import Program, View
from .graph import AnchorTaskModel, Recorder, conv_bn, folded
from .param import Yolo7PoseParam
from .yolov3 import _seq

EPS, MOM = 1e-3, 0.03


def _mcb(c1: int, c2: int, c3: int, n: int, e: int, ids: List[int]) -> nn.Module:
    m = nn.Module()
    c_ = int(c2 * e)
    m.ids = list(ids)
    m.cv1, m.cv2 = conv_bn(c1, c_), conv_bn(c1, c_)
    m.cv3 = nn.ModuleList([conv_bn(c_ if i == 0 else c2, c2, 3) for i in range(n)])
    m.cv4 = conv_bn(c_ * 2 + c2 * (len(ids) - 2), c3)
    return m


def _transition(c1: int, c2: int) -> nn.Module:
    m = nn.Module()
    m.cv1, m.cv2, m.cv3 = conv_bn(c1, c2), conv_bn(c1, c2), conv_bn(c2, c2, 3, 2)
    return m


def _sppcspc(c1: int, c2: int) -> nn.Module:
    m = nn.Module()
    c_ = int(2 * c2 * 0.5)
    m.cv1, m.cv2, m.cv3, m.cv4 = conv_bn(c1, c_), conv_bn(c1, c_), conv_bn(c_, c_, 3), conv_bn(c_, c_)
    m.cv5, m.cv6, m.cv7 = conv_bn(4 * c_, c_), conv_bn(c_, c_, 3), conv_bn(2 * c_, c2)
    return m


def rep_conv(c1: int, c2: int) -> nn.Module:
    """Training-form RepConv (YoloV7PoseBody.py:225-245): rbr_dense = 3x3 conv + BN, rbr_1x1 = 1x1 conv + BN,
    rbr_identity = BN when c1 == c2 (stride is 1 everywhere in this model)."""
    m = nn.Module()
    m.is_rep_conv = True
    m.rbr_identity = nn.BatchNorm2d(c1, eps=EPS, momentum=MOM) if c1 == c2 else None
    m.rbr_dense = _seq(nn.Conv2d(c1, c2, 3, 1, 1, bias=False), nn.BatchNorm2d(c2, eps=EPS, momentum=MOM))
    m.rbr_1x1 = _seq(nn.Conv2d(c1, c2, 1, 1, 0, bias=False), nn.BatchNorm2d(c2, eps=EPS, momentum=MOM))
    return m


def rep_conv_equivalent(m: nn.Module):
    """(weight (c2,c1,3,3), bias) of the single 3x3 conv equal to the three RepConv branches
    (get_equivalent_kernel_bias, YoloV7PoseBody.py:256-300), or the already re-parameterised conv."""
    if getattr(m, "rbr_reparam", None) is not None:
        return m.rbr_reparam.weight.detach(), m.rbr_reparam.bias.detach()
    w3, b3 = torch_utils.fold_conv_bn(getattr(m.rbr_dense, "0").weight, None, getattr(m.rbr_dense, "1"))
    w1, b1 = torch_utils.fold_conv_bn(getattr(m.rbr_1x1, "0").weight, None, getattr(m.rbr_1x1, "1"))
    w, b = w3 + F.pad(w1, [1, 1, 1, 1]), b3 + b1
    if isinstance(m.rbr_identity, nn.BatchNorm2d):
        c = w.shape[1]
        eye = torch.zeros(c, c, 3, 3)
        eye[torch.arange(c), torch.arange(c), 1, 1] = 1.0
        wi, bi = torch_utils.fold_conv_bn(eye, None, m.rbr_identity)
        w, b = w + wi, b + bi
    return w, b


def _tree(param, anchor_num: int, cols: int) -> nn.Module:
    phi = param.m_type
    if phi not in ("l", "x"):
        raise NotImplementedError(f"YOLOv7 m_type={phi!r}: the reference defines the l and x widths")
    tc = {"l": 32, "x": 40}[phi]
    bc = 32
    pc = {"l": 32, "x": 64}[phi]
    e = {"l": 2, "x": 1}[phi]
    n = {"l": 4, "x": 6}[phi]
    bids = {"l": [-1, -3, -5, -6], "x": [-1, -3, -5, -7, -8]}[phi]
    nids = {"l": [-1, -2, -3, -4, -5, -6], "x": [-1, -3, -5, -7, -8]}[phi]
    body = nn.Module()
    bb = nn.Module()
    bb.stem = nn.Sequential(conv_bn(3, tc, 3, 1), conv_bn(tc, tc * 2, 3, 2), conv_bn(tc * 2, tc * 2, 3, 1))
    bb.dark2 = nn.Sequential(conv_bn(tc * 2, tc * 4, 3, 2), _mcb(tc * 4, bc * 2, tc * 8, n, 1, bids))
    bb.dark3 = nn.Sequential(_transition(tc * 8, tc * 4), _mcb(tc * 8, bc * 4, tc * 16, n, 1, bids))
    bb.dark4 = nn.Sequential(_transition(tc * 16, tc * 8), _mcb(tc * 16, bc * 8, tc * 32, n, 1, bids))
    bb.dark5 = nn.Sequential(_transition(tc * 32, tc * 16), _mcb(tc * 32, bc * 8, tc * 32, n, 1, bids))
    body.backbone = bb
    body.spp_csp_c = _sppcspc(tc * 32, tc * 16)
    body.conv_for_P5 = conv_bn(tc * 16, tc * 8)
    body.conv_for_feat2 = conv_bn(tc * 32, tc * 8)
    body.conv3_for_up_sample1 = _mcb(tc * 16, pc * 4, tc * 8, n, e, nids)
    body.conv_for_P4 = conv_bn(tc * 8, tc * 4)
    body.conv_for_feat1 = conv_bn(tc * 16, tc * 4)
    body.conv3_for_up_sample2 = _mcb(tc * 8, pc * 2, tc * 4, n, e, nids)
    body.down_sample1 = _transition(tc * 4, tc * 4)
    body.conv3_for_down_sample1 = _mcb(tc * 16, pc * 4, tc * 8, n, e, nids)
    body.down_sample2 = _transition(tc * 8, tc * 8)
    body.conv3_for_down_sample2 = _mcb(tc * 32, pc * 8, tc * 16, n, e, nids)
    mk = rep_conv if phi == "l" else (lambda a, b: conv_bn(a, b, 3, 1))
    body.rep_conv_1, body.rep_conv_2, body.rep_conv_3 = mk(tc * 4, tc * 8), mk(tc * 8, tc * 16), mk(tc * 16, tc * 32)
    out = anchor_num * cols        # per-anchor columns: 5 + classes (+ 2 or 3 per keypoint)
    body.yolo_head_P3 = nn.Conv2d(tc * 8, out, 1)
    body.yolo_head_P4 = nn.Conv2d(tc * 16, out, 1)
    body.yolo_head_P5 = nn.Conv2d(tc * 32, out, 1)
    return body


class YoloV7Pose(AnchorTaskModel):
    """``YoloV7Pose(param)`` as in the reference.  Images must be CUDA float32 NCHW in [0,1].
    ``predicts`` rows: [x1,y1,x2,y2,obj,cls_conf, 17 x (x,y), 17 keypoint confidences, cls_id]."""

    DECODE_KIND = _lib.DECODE_POSE
    POINT_CONF = True

    def __init__(self, param: Yolo7PoseParam, anchor_num: int = 3):
        super().__init__()
        self.param = param
        self.POINT_NUM = param.point_num
        self.body_output_size = 5 + 3 * param.point_num + param.class_num
        self.body_net = _tree(param, anchor_num, self.body_output_size)
        self.eval()

    # ---- graph -----------------------------------------------------------------------------------
    @staticmethod
    def _mcb(R: Recorder, name: str, m: nn.Module, x: View) -> View:
        n = 2 + len(m.cv3)
        ids = [i % n for i in m.ids]
        if ids[-2:] != [1, 0]:
            xs = [R.cba(f"{name}.cv1", m.cv1, x), R.cba(f"{name}.cv2", m.cv2, x)]
            for i, cv in enumerate(m.cv3):
                xs.append(R.cba(f"{name}.cv3.{i}", cv, xs[-1]))
            return R.cba(f"{name}.cv4", m.cv4, R.p.cat([xs[i] for i in ids]))
        # cv1 and cv2 are 1x1 convs of the same input whose outputs end the concat as [.., cv2, cv1]: they run as one
        # conv that reads the input once and writes both slices
        (w1, b1, _), (w2, b2, _) = folded(m.cv1), folded(m.cv2)
        c_, c2 = w1.shape[0], m.cv3[0].conv.out_channels
        cat = R.p.buffer(x.h, x.w, 2 * c_ + c2 * (len(ids) - 2))
        place = {j: cat.slice(k * c2, c2) for k, j in enumerate(ids[:-2])}
        tail = cat.slice(c2 * (len(ids) - 2), 2 * c_)
        R.p.conv(f"{name}.cv2+1", x, torch.cat((w2, w1), 0), torch.cat((b2, b1), 0), 1, R.act, out=tail)
        t = tail.slice(0, c_)
        for i, cv in enumerate(m.cv3):
            t = R.cba(f"{name}.cv3.{i}", cv, t, out=place.get(2 + i))
        return R.cba(f"{name}.cv4", m.cv4, cat)

    @staticmethod
    def _transition(R: Recorder, name: str, m: nn.Module, x: View) -> View:
        x1 = R.cba(f"{name}.cv1", m.cv1, R.p.maxpool2(x))
        x2 = R.cba(f"{name}.cv3", m.cv3, R.cba(f"{name}.cv2", m.cv2, x))
        return R.p.cat([x2, x1])

    @staticmethod
    def _rep(R: Recorder, name: str, m: nn.Module, x: View) -> View:
        if not getattr(m, "is_rep_conv", False):
            return R.cba(name, m, x)
        w, b = rep_conv_equivalent(m)
        return R.p.conv(name, x, w, b, 1, R.act)

    def _compile(self, S: int) -> Program:
        body, bb = self.body_net, self.body_net.backbone
        P = Program(S, S)
        R = Recorder(P, _lib.ACT_SILU)
        x = P.image16(3)
        for i in range(3):
            w, b, st = folded(bb.stem[i])
            # the x width starts with 40 channels; activations are stored in multiples of 16, so that layer computes 8
            # more output channels with zero weights and zero bias (SiLU(0) = 0) and the next one ignores them
            cin_pad, cout_pad = x.c - w.shape[1] if i else 0, -w.shape[0] % 16
            if cin_pad or cout_pad:
                w = F.pad(w, [0, 0, 0, 0, 0, cin_pad, 0, cout_pad])
                b = F.pad(b, [0, cout_pad])
            x = P.conv(f"stem.{i}", x, w, b, st, R.act)
        x = self._mcb(R, "dark2.1", bb.dark2[1], R.cba("dark2.0", bb.dark2[0], x))
        feat1 = self._mcb(R, "dark3.1", bb.dark3[1], self._transition(R, "dark3.0", bb.dark3[0], x))
        feat2 = self._mcb(R, "dark4.1", bb.dark4[1], self._transition(R, "dark4.0", bb.dark4[0], feat1))
        feat3 = self._mcb(R, "dark5.1", bb.dark5[1], self._transition(R, "dark5.0", bb.dark5[0], feat2))

        sp = body.spp_csp_c
        c_ = sp.cv4.conv.out_channels
        cat = P.buffer(feat3.h, feat3.w, 4 * c_, name="cat_spp")
        # cv1 and cv2 both read feat3: one conv, cv1's half to a buffer of its own, cv2's half into cv7's concat input
        (w1, b1, _), (w2, b2, _) = folded(sp.cv1), folded(sp.cv2)
        cat7 = P.buffer(feat3.h, feat3.w, 2 * c_, name="cat_spp_out")
        t = P.buffer(feat3.h, feat3.w, c_)
        P.conv("spp.cv1+2", feat3, torch.cat((w1, w2), 0), torch.cat((b1, b2), 0), 1, R.act, out=t, out2=cat7.slice(c_, c_))
        t = R.cba("spp.cv3", sp.cv3, t)
        R.cba("spp.cv4", sp.cv4, t, out=cat.slice(0, c_))
        P.maxpool5_cascade(cat.slice(0, c_), cat.slice(c_, c_), cat.slice(2 * c_, c_), cat.slice(3 * c_, c_))
        R.cba("spp.cv6", sp.cv6, R.cba("spp.cv5", sp.cv5, cat), out=cat7.slice(0, c_))
        p5 = R.cba("spp.cv7", sp.cv7, cat7)

        up = P.upsample2x(R.cba("conv_for_P5", body.conv_for_P5, p5))
        p4 = self._mcb(R, "conv3_for_up_sample1", body.conv3_for_up_sample1,
                       P.cat([R.cba("conv_for_feat2", body.conv_for_feat2, feat2), up]))
        up = P.upsample2x(R.cba("conv_for_P4", body.conv_for_P4, p4))
        p3 = self._mcb(R, "conv3_for_up_sample2", body.conv3_for_up_sample2,
                       P.cat([R.cba("conv_for_feat1", body.conv_for_feat1, feat1), up]))
        p4 = self._mcb(R, "conv3_for_down_sample1", body.conv3_for_down_sample1,
                       P.cat([self._transition(R, "down_sample1", body.down_sample1, p3), p4]))
        p5 = self._mcb(R, "conv3_for_down_sample2", body.conv3_for_down_sample2,
                       P.cat([self._transition(R, "down_sample2", body.down_sample2, p4), p5]))
        for name, rep, head, t in (("P3", body.rep_conv_1, body.yolo_head_P3, p3),
                                   ("P4", body.rep_conv_2, body.yolo_head_P4, p4),
                                   ("P5", body.rep_conv_3, body.yolo_head_P5, p5)):
            P.outputs[name] = R.head(f"yolo_head_{name}", head, self._rep(R, f"rep_conv_{name}", rep, t))
        return P

    # ---- re-parameterisation -----------------------------------------------------------------------
    def fuse(self):
        """RepConv -> one ``rbr_reparam`` conv with bias, Conv -> conv with bias and no bn (YoloV7Pose.py:219-227)."""
        for m in self.modules():
            if getattr(m, "is_rep_conv", False):
                if getattr(m, "rbr_reparam", None) is None:
                    w, b = rep_conv_equivalent(m)
                    conv = nn.Conv2d(w.shape[1], w.shape[0], 3, 1, 1, bias=True).requires_grad_(False)
                    conv.weight.copy_(w)
                    conv.bias.copy_(b)
                    m.rbr_reparam = conv
                    m.rbr_identity = m.rbr_1x1 = m.rbr_dense = None
            elif isinstance(getattr(m, "bn", None), nn.BatchNorm2d) and isinstance(getattr(m, "conv", None), nn.Conv2d):
                m.conv = torch_utils.fuse_conv_and_bn(m.conv, m.bn)
                delattr(m, "bn")
        self.invalidate()
        return self
