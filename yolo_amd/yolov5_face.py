"""YOLOv5 face (5 landmarks) and YOLOv5 pose (17 keypoints with confidences) on the HIP path.

Mirror luoyang/yolov5_face/YoloV5Face.py and luoyang/yolov5_pose/YoloV5Pose.py (``_forward``, ``forward``, ``predict``,
``fuse``) over the module tree of luoyang/yolov5_face/Yolov5FaceBody.py (= yolov5_pose/YoloV5PoseBody.py up to the
head width): StemBlock :54-69 instead of Focus, dark2 without a leading stride-2 conv :133-134, SPPF with parallel
k = 3, 5, 7 pools :106-125, and the unused ``down_sample3`` / ``conv3_for_down_sample3`` modules :207-208 that are
part of the ``state_dict``.  Heads decode as luoyang/yolov5_face/YoloV5FaceHead.py:25-77 (landmarks scaled by the
anchor) and luoyang/yolov5_pose/YoloV5PoseHead.py:25-81 (keypoints ``2t - 0.5 + grid``, one confidence each).
"""
import torch.nn as nn
import torch.nn.functional as F

from . import lib as _lib
from .engine import Program
from .graph import AnchorTaskModel
from .param import Yolo5FaceParam, Yolo5PoseParam
from .yolov5 import _DEPTH, _WIDTH, _Builder, c3_block, conv_block


def _tree(out_size: int, na: int, wid: float, dep: float) -> nn.Module:
    bc, bd = int(wid * 64), max(round(dep * 3), 1)
    body = nn.Module()
    bb = nn.Module()
    stem = nn.Module()
    stem.stem_1 = conv_block(3, bc, 3, 2)
    stem.stem_2a = conv_block(bc, bc // 2, 1)
    stem.stem_2b = conv_block(bc // 2, bc, 3, 2)
    stem.stem_3 = conv_block(bc * 2, bc, 1)
    bb.stem = stem
    bb.dark2 = nn.Sequential(c3_block(bc, bc * 2, bd))
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
    body.down_sample3 = conv_block(bc * 16, bc * 16, 3, 2)                # held, never run (as in the reference)
    body.conv3_for_down_sample3 = c3_block(bc * 32, bc * 32, bd)
    body.yolo_head_P3 = nn.Conv2d(bc * 4, na * out_size, (1, 1))
    body.yolo_head_P4 = nn.Conv2d(bc * 8, na * out_size, (1, 1))
    body.yolo_head_P5 = nn.Conv2d(bc * 16, na * out_size, (1, 1))
    return body


class _V5PointModel(AnchorTaskModel):
    """Shared graph of the two variants; subclasses fix the decode kind and the head width."""

    def _init_tree(self, param, anchor_num: int, out_size: int) -> None:
        self.param = param
        self.anchor_num = anchor_num
        self.body_output_size = out_size
        t = param.m_type.split("_")[-1]
        self.body_net = _tree(out_size, anchor_num, _WIDTH[t], _DEPTH[t])
        self.eval()

    def _compile(self, S: int) -> Program:
        body, bb = self.body_net, self.body_net.backbone
        P = Program(S, S)
        B = _Builder(P)
        bc = bb.stem.stem_1.conv.out_channels
        c3ch = bb.dark3[0].conv.out_channels
        c4ch = bb.dark4[0].conv.out_channels
        s4, s8, s16, s32 = S // 4, S // 8, S // 16, S // 32
        cat_p4 = P.buffer(s16, s16, 2 * c4ch, name="cat_p4")          # [up(p5) | feat2]
        cat_p3 = P.buffer(s8, s8, 2 * c3ch, name="cat_p3")            # [up(p4) | feat1]
        cat_n4 = P.buffer(s16, s16, 2 * c3ch, name="cat_n4")          # [down(p3) | p4]
        cat_n5 = P.buffer(s32, s32, 2 * c4ch, name="cat_n5")          # [down(p4) | p5]
        cat_stem = P.buffer(s4, s4, 2 * bc, name="cat_stem")          # [stem_2b | stem_2p]

        s1 = B.cbs("stem.stem_1", bb.stem.stem_1, P.image16(3))
        if (bc // 2) % 16:
            # m / x widths: stem_2a has 24 / 40 outputs; activations are stored in multiples of 16 channels, so it
            # computes the missing ones with zero weights and zero bias (SiLU(0) = 0) and stem_2b ignores them
            pad = -(bc // 2) % 16
            wa, ba = B._folded(bb.stem.stem_2a)
            wb, bb_ = B._folded(bb.stem.stem_2b)
            t = P.conv("stem.stem_2a", s1, F.pad(wa, [0, 0, 0, 0, 0, 0, 0, pad]), F.pad(ba, [0, pad]), 1, _lib.ACT_SILU)
            P.conv("stem.stem_2b", t, F.pad(wb, [0, 0, 0, 0, 0, pad]), bb_, 2, _lib.ACT_SILU, out=cat_stem.slice(0, bc))
        else:
            B.cbs("stem.stem_2b", bb.stem.stem_2b, B.cbs("stem.stem_2a", bb.stem.stem_2a, s1), out=cat_stem.slice(0, bc))
        P.maxpool2(s1, cat_stem.slice(bc, bc))
        x = B.cbs("stem.stem_3", bb.stem.stem_3, cat_stem)
        x = B.c3("dark2.0", bb.dark2[0], x, True)
        B.c3("dark3.1", bb.dark3[1], B.cbs("dark3.0", bb.dark3[0], x), True, out=cat_p3.slice(c3ch, c3ch))
        feat1 = cat_p3.slice(c3ch, c3ch)
        B.c3("dark4.1", bb.dark4[1], B.cbs("dark4.0", bb.dark4[0], feat1), True, out=cat_p4.slice(c4ch, c4ch))
        feat2 = cat_p4.slice(c4ch, c4ch)
        x = B.cbs("dark5.0", bb.dark5[0], feat2)
        spp = bb.dark5[1]
        hc = spp.cv1.conv.out_channels
        cat_spp = P.buffer(s32, s32, 4 * hc, name="cat_spp")
        B.cbs("spp.cv1", spp.cv1, x, out=cat_spp.slice(0, hc))
        P.maxpool5_cascade(cat_spp.slice(0, hc), cat_spp.slice(hc, hc), cat_spp.slice(2 * hc, hc), cat_spp.slice(3 * hc, hc),
                           kernel=3)
        x = B.cbs("spp.cv2", spp.cv2, cat_spp)
        feat3 = B.c3("dark5.2", bb.dark5[2], x, False)

        p5 = B.cbs("conv_for_feat3", body.conv_for_feat3, feat3, out=cat_n5.slice(c4ch, c4ch))
        P.upsample2x(p5, cat_p4.slice(0, c4ch))
        p4 = B.c3("conv3_for_up_sample1", body.conv3_for_up_sample1, cat_p4, False)
        p4 = B.cbs("conv_for_feat2", body.conv_for_feat2, p4, out=cat_n4.slice(c3ch, c3ch))
        P.upsample2x(p4, cat_p3.slice(0, c3ch))
        p3 = B.c3("conv3_for_up_sample2", body.conv3_for_up_sample2, cat_p3, False)
        B.cbs("down_sample1", body.down_sample1, p3, out=cat_n4.slice(0, c3ch))
        p4 = B.c3("conv3_for_down_sample1", body.conv3_for_down_sample1, cat_n4, False)
        B.cbs("down_sample2", body.down_sample2, p4, out=cat_n5.slice(0, c4ch))
        p5 = B.c3("conv3_for_down_sample2", body.conv3_for_down_sample2, cat_n5, False)
        for name, head, t in (("P3", body.yolo_head_P3, p3), ("P4", body.yolo_head_P4, p4), ("P5", body.yolo_head_P5, p5)):
            P.outputs[name] = P.conv(f"yolo_head_{name}", t, head.weight.detach(), head.bias.detach(), 1, _lib.ACT_NONE,
                                     out_f32=True)
        return P

    def fuse(self) -> None:
        """Fold every BatchNorm into its conv (YoloV5Face.py:214-219), the unused modules included."""
        self.fuse_conv_bn_holders()


class YoloV5Face(_V5PointModel):
    """``YoloV5Face(param)`` as in the reference: rows [cx, cy, w, h, obj, 5 x (lx, ly), cls]."""

    DECODE_KIND = _lib.DECODE_FACE
    POINT_NUM = 5
    POINT_CONF = False
    CLAMP_POINTS_IN_PREDICT = True        # YoloV5Face.py:141-144

    def __init__(self, param: Yolo5FaceParam, anchor_num: int = 3):
        super().__init__()
        self.POINT_NUM = param.point_num
        self._init_tree(param, anchor_num, 5 + 2 * param.point_num + param.class_num)


class YoloV5Pose(_V5PointModel):
    """``YoloV5Pose(param)`` as in the reference: rows [cx, cy, w, h, obj, 17 x (kx, ky), 17 x conf, cls]."""

    DECODE_KIND = _lib.DECODE_POSE
    POINT_NUM = 17
    POINT_CONF = True

    def __init__(self, param: Yolo5PoseParam, anchor_num: int = 3):
        super().__init__()
        self.POINT_NUM = param.point_num
        self._init_tree(param, anchor_num, 5 + 3 * param.point_num + param.class_num)
