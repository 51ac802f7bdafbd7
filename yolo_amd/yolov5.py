"""YOLOv5 detect on the HIP path, behind the reference's TaskLayer interface.

Mirrors luoyang/yolov5/YoloV5.py (``forward`` :88-95, ``_forward`` :48-71, ``predict`` :97-127,
``fuse`` :171-176) and the module tree of luoyang/yolov5/YoloV5Body.py so that ``state_dict`` keys are
identical (``body_net.backbone.stem.conv.conv.weight`` ...) and reference checkpoints load unchanged.
The modules below only hold parameters: the conv graph is compiled into a ``engine.Program`` of
fused HIP kernels (BN folded, concat in place, bf16 NHWC), decode and NMS run in their own kernels.
"""
from typing import Optional

import torch
import torch.nn as nn

from . import lib as _lib
from . import torch_utils
from .engine import Program, View
from .graph import AnchorTaskModel
from .param import Yolo5Param

_DEPTH = {"s": 0.33, "m": 0.67, "l": 1.00, "x": 1.33}
_WIDTH = {"s": 0.50, "m": 0.75, "l": 1.00, "x": 1.25}
BN_EPS, BN_MOM = 1e-3, 0.03


def conv_block(cin: int, cout: int, k: int = 1, s: int = 1) -> nn.Module:
    """Parameter holder of Conv (YoloV5Body.py:26-51): ``.conv`` (no bias) + ``.bn``; SiLU follows."""
    m = nn.Module()
    m.conv = nn.Conv2d(cin, cout, (k, k), (s, s), k // 2, bias=False)
    m.bn = nn.BatchNorm2d(cout, eps=BN_EPS, momentum=BN_MOM)
    return m


def c3_block(cin: int, cout: int, n: int) -> nn.Module:
    """C3 (YoloV5Body.py:96-111): cv1/cv2 1x1 to cout/2, n bottlenecks (1x1, 3x3), cv3 1x1."""
    m = nn.Module()
    h = int(cout * 0.5)
    m.cv1, m.cv2, m.cv3 = conv_block(cin, h), conv_block(cin, h), conv_block(2 * h, cout)
    bots = []
    for _ in range(n):
        b = nn.Module()
        b.cv1, b.cv2 = conv_block(h, h, 1), conv_block(h, h, 3)
        bots.append(b)
    m.m = nn.Sequential(*bots)
    return m


def _tree(nc: int, na: int, wid: float, dep: float) -> nn.Module:
    bc, bd = int(wid * 64), max(round(dep * 3), 1)
    body = nn.Module()
    bb = nn.Module()
    bb.stem = nn.Module()
    bb.stem.conv = conv_block(12, bc, 3)
    bb.dark2 = nn.Sequential(conv_block(bc, bc * 2, 3, 2), c3_block(bc * 2, bc * 2, bd))
    bb.dark3 = nn.Sequential(conv_block(bc * 2, bc * 4, 3, 2), c3_block(bc * 4, bc * 4, bd * 3))
    bb.dark4 = nn.Sequential(conv_block(bc * 4, bc * 8, 3, 2), c3_block(bc * 8, bc * 8, bd * 3))
    spp = nn.Module()
    spp.cv1, spp.cv2 = conv_block(bc * 16, bc * 8), conv_block(bc * 8 * 4, bc * 16)
    bb.dark5 = nn.Sequential(conv_block(bc * 8, bc * 16, 3, 2), spp, c3_block(bc * 16, bc * 16, bd))
    body.backbone = bb
    body.conv_for_feat3 = conv_block(bc * 16, bc * 8)
    body.conv3_for_up_sample1 = c3_block(bc * 16, bc * 8, bd)
    body.conv_for_feat2 = conv_block(bc * 8, bc * 4)
    body.conv3_for_up_sample2 = c3_block(bc * 8, bc * 4, bd)
    body.down_sample1 = conv_block(bc * 4, bc * 4, 3, 2)
    body.conv3_for_down_sample1 = c3_block(bc * 8, bc * 8, bd)
    body.down_sample2 = conv_block(bc * 8, bc * 8, 3, 2)
    body.conv3_for_down_sample2 = c3_block(bc * 16, bc * 16, bd)
    body.yolo_head_P3 = nn.Conv2d(bc * 4, na * (5 + nc), (1, 1))
    body.yolo_head_P4 = nn.Conv2d(bc * 8, na * (5 + nc), (1, 1))
    body.yolo_head_P5 = nn.Conv2d(bc * 16, na * (5 + nc), (1, 1))
    return body


class _Builder:
    """Walks the parameter tree and records the fused conv graph."""

    def __init__(self, prog: Program):
        self.p = prog

    def cbs(self, name: str, m: nn.Module, x: View, out: Optional[View] = None, residual: Optional[View] = None
            ) -> View:
        conv = m.conv
        if hasattr(m, "bn"):
            w, b = torch_utils.fold_conv_bn(conv.weight, conv.bias, m.bn)
        else:                                   # already fused by fuse()
            w, b = conv.weight.detach(), conv.bias.detach()
        return self.p.conv(name, x, w, b, conv.stride[0], _lib.ACT_SILU, out=out, residual=residual)

    def c3(self, name: str, m: nn.Module, x: View, shortcut: bool, out: Optional[View] = None) -> View:
        """cv1 and cv2 read the same input, so they run as ONE conv with their output channels side by side:
        buffer = [last bottleneck out | cv2 out | cv1 out]; cv3 reads the first two slices (the reference's
        cat((m(cv1(x)), cv2(x)), 1)).  Saves one full read of x and one launch per C3."""
        h = m.cv1.conv.out_channels
        n = len(m.m)
        assert n >= 1
        w1, b1 = self._folded(m.cv1)
        w2, b2 = self._folded(m.cv2)
        if h * 2 < 128:
            # a 64-byte channel slice of a 192-byte pixel costs a whole 128-byte memory line per read: the cv1 branch
            # (read twice: bottleneck input and residual) gets a buffer of its own, [m out | cv2] stays line aligned
            buf = self.p.buffer(x.h, x.w, 2 * h)
            a = self.p.buffer(x.h, x.w, h)
            self.p.conv(f"{name}.cv2+cv1", x, torch.cat((w2, w1), 0), torch.cat((b2, b1), 0), 1, _lib.ACT_SILU,
                        out=buf.slice(h, h), out2=a)
        else:
            buf = self.p.buffer(x.h, x.w, 3 * h)
            self.p.conv(f"{name}.cv2+cv1", x, torch.cat((w2, w1), 0), torch.cat((b2, b1), 0), 1, _lib.ACT_SILU,
                        out=buf.slice(h, 2 * h))
            a = buf.slice(2 * h, h)
        for i, bot in enumerate(m.m):
            t = self.cbs(f"{name}.m.{i}.cv1", bot.cv1, a)
            a = self.cbs(f"{name}.m.{i}.cv2", bot.cv2, t, out=buf.slice(0, h) if i == n - 1 else None,
                         residual=a if shortcut else None)
        return self.cbs(f"{name}.cv3", m.cv3, buf.slice(0, 2 * h), out=out)

    @staticmethod
    def _folded(m: nn.Module):
        if hasattr(m, "bn"):
            return torch_utils.fold_conv_bn(m.conv.weight, m.conv.bias, m.bn)
        return m.conv.weight.detach().float(), m.conv.bias.detach().float()


class YoloV5(AnchorTaskModel):
    """``YoloV5(param)`` as in the reference; inputs/outputs of ``forward``/``predict`` are the
    reference's ``tensor_dict`` entries.  Images must be CUDA float32 NCHW in [0,1]."""

    DECODE_KIND = _lib.DECODE_V5

    def __init__(self, param: Yolo5Param, anchor_num: int = 3):
        super().__init__()
        self.param = param
        self.anchor_num = anchor_num
        self.body_output_size = 5 + param.class_num
        t = param.m_type.split("_")[-1]
        self.body_net = _tree(param.class_num, anchor_num, _WIDTH[t], _DEPTH[t])
        self.eval()

    def _compile(self, S: int) -> Program:
        body, bb = self.body_net, self.body_net.backbone
        P = Program(S, S)
        B = _Builder(P)
        c3ch = bb.dark3[0].conv.out_channels           # feat1 channels (bc*4)
        c4ch = bb.dark4[0].conv.out_channels           # feat2 channels (bc*8)
        s8, s16, s32 = S // 8, S // 16, S // 32
        # concat buffers of the neck exist up front so the backbone writes its features in place
        cat_p4 = P.buffer(s16, s16, 2 * c4ch, name="cat_p4")          # [up(p5) | feat2]
        cat_p3 = P.buffer(s8, s8, 2 * c3ch, name="cat_p3")            # [up(p4) | feat1]
        cat_n4 = P.buffer(s16, s16, 2 * c3ch, name="cat_n4")          # [down(p3) | p4]
        cat_n5 = P.buffer(s32, s32, 2 * c4ch, name="cat_n5")          # [down(p4) | p5]

        T = P.taps
        T["stem"] = x = B.cbs("stem", bb.stem.conv, P.focus())
        T["dark2.0"] = x = B.cbs("dark2.0", bb.dark2[0], x)
        T["dark2"] = x = B.c3("dark2.1", bb.dark2[1], x, True)
        T["dark3.0"] = d30 = B.cbs("dark3.0", bb.dark3[0], x)
        B.c3("dark3.1", bb.dark3[1], d30, True, out=cat_p3.slice(c3ch, c3ch))
        T["feat1"] = feat1 = cat_p3.slice(c3ch, c3ch)
        B.c3("dark4.1", bb.dark4[1], B.cbs("dark4.0", bb.dark4[0], feat1), True, out=cat_p4.slice(c4ch, c4ch))
        T["feat2"] = feat2 = cat_p4.slice(c4ch, c4ch)
        T["dark5.0"] = x = B.cbs("dark5.0", bb.dark5[0], feat2)
        spp = bb.dark5[1]
        hc = spp.cv1.conv.out_channels
        cat_spp = P.buffer(s32, s32, 4 * hc, name="cat_spp")
        B.cbs("spp.cv1", spp.cv1, x, out=cat_spp.slice(0, hc))
        P.maxpool5_cascade(cat_spp.slice(0, hc), cat_spp.slice(hc, hc), cat_spp.slice(2 * hc, hc),
                           cat_spp.slice(3 * hc, hc))
        T["spp"] = x = B.cbs("spp.cv2", spp.cv2, cat_spp)
        T["feat3"] = feat3 = B.c3("dark5.2", bb.dark5[2], x, False)

        T["p5_lateral"] = p5 = B.cbs("conv_for_feat3", body.conv_for_feat3, feat3, out=cat_n5.slice(c4ch, c4ch))
        P.upsample2x(p5, cat_p4.slice(0, c4ch))
        p4 = B.c3("conv3_for_up_sample1", body.conv3_for_up_sample1, cat_p4, False)
        p4 = B.cbs("conv_for_feat2", body.conv_for_feat2, p4, out=cat_n4.slice(c3ch, c3ch))
        P.upsample2x(p4, cat_p3.slice(0, c3ch))
        T["p3"] = p3 = B.c3("conv3_for_up_sample2", body.conv3_for_up_sample2, cat_p3, False)
        T["down_sample1"] = B.cbs("down_sample1", body.down_sample1, p3, out=cat_n4.slice(0, c3ch))
        p4 = B.c3("conv3_for_down_sample1", body.conv3_for_down_sample1, cat_n4, False)
        B.cbs("down_sample2", body.down_sample2, p4, out=cat_n5.slice(0, c4ch))
        p5 = B.c3("conv3_for_down_sample2", body.conv3_for_down_sample2, cat_n5, False)
        for name, head, t in (("P3", body.yolo_head_P3, p3), ("P4", body.yolo_head_P4, p4),
                              ("P5", body.yolo_head_P5, p5)):
            P.outputs[name] = P.conv(f"yolo_head_{name}", t, head.weight.detach(), head.bias.detach(), 1,
                                     _lib.ACT_NONE, out_f32=True)
        return P

    def fuse(self) -> None:
        """Fold every BatchNorm into its conv (YoloV5.py:171-176); ``state_dict`` then matches a fused
        reference model (conv.weight + conv.bias, no bn.*)."""
        self.fuse_conv_bn_holders()
