"""YOLOv4 (CSPDarkNet-53 + SPP + PANet) on the HIP path, behind the reference's TaskLayer interface.

Mirrors luoyang/yolov4/YoloV4.py and the module tree of luoyang/yolov4/YoloV4BackBone.py (BasicConv :18-43 with
Mish, ResBlock :46-59, ResBlockBody :62-93, CSPDarkNet :96-127, conv2d :137-148 with LeakyReLU(0.1),
SpatialPyramidPooling :151-162 in the order [13, 9, 5, x], YoloV4BackBone.forward :233-262), so ``state_dict``
keys are identical.  BatchNorm uses torch's default eps 1e-5.  ``fuse()`` folds the backbone's BasicConv blocks
only, like the reference (YoloV4.py:216-221).
"""
import torch
import torch.nn as nn

from . import lib as _lib
from . import torch_utils
from .engine import Program, View
from .graph import AnchorTaskModel, Recorder, folded
from .param import Yolo4Param
from .yolov3 import _cbl, _seq


def _res_body(cin: int, cout: int, n: int, first: bool) -> nn.Module:
    m = nn.Module()
    m.downsample_conv = _cbl(cin, cout, 3, 2)
    h = cout if first else cout // 2

    def res(c, hidden):
        r = nn.Module()
        r.block = _seq(_cbl(c, hidden, 1), _cbl(hidden, c, 3))
        return r

    m.split_conv0, m.split_conv1 = _cbl(cout, h, 1), _cbl(cout, h, 1)
    blocks = [res(cout, cout // 2)] if first else [res(h, h) for _ in range(n)]
    m.blocks_conv = _seq(*blocks, _cbl(h, h, 1))
    m.concat_conv = _cbl(2 * h, cout, 1)
    return m


def _three(f, cin):
    return _seq(_cbl(cin, f[0], 1), _cbl(f[0], f[1], 3), _cbl(f[1], f[0], 1))


def _five(f, cin):
    return _seq(_cbl(cin, f[0], 1), _cbl(f[0], f[1], 3), _cbl(f[1], f[0], 1), _cbl(f[0], f[1], 3), _cbl(f[1], f[0], 1))


def _up(cin, cout):
    m = nn.Module()
    m.up_sample = _seq(_cbl(cin, cout, 1))
    return m


def _tree(param: Yolo4Param, out_ch: int) -> nn.Module:
    body = nn.Module()
    bb = nn.Module()
    bb.conv1 = _cbl(3, 32, 3)
    ch = [64, 128, 256, 512, 1024]
    stages, cin = [], 32
    for i, (n, c) in enumerate(zip(param.dark_blocks, ch)):
        stages.append(_res_body(cin, c, n, i == 0))
        cin = c
    bb.stages = nn.ModuleList(stages)
    body.backbone = bb
    body.conv1 = _three([512, 1024], 1024)
    body.conv2 = _three([512, 1024], 2048)
    body.up_sample1 = _up(512, 256)
    body.conv_for_P4 = _cbl(512, 256, 1)
    body.make_five_conv1 = _five([256, 512], 512)
    body.up_sample2 = _up(256, 128)
    body.conv_for_P3 = _cbl(256, 128, 1)
    body.make_five_conv2 = _five([128, 256], 256)
    body.yolo_head3 = _seq(_cbl(128, 256, 3), nn.Conv2d(256, out_ch, 1))
    body.down_sample1 = _cbl(128, 256, 3, 2)
    body.make_five_conv3 = _five([256, 512], 512)
    body.yolo_head2 = _seq(_cbl(256, 512, 3), nn.Conv2d(512, out_ch, 1))
    body.down_sample2 = _cbl(256, 512, 3, 2)
    body.make_five_conv4 = _five([512, 1024], 1024)
    body.yolo_head1 = _seq(_cbl(512, 1024, 3), nn.Conv2d(1024, out_ch, 1))
    return body


class YoloV4(AnchorTaskModel):
    """``YoloV4(param)`` as in the reference.  Images must be CUDA float32 NCHW in [0,1]."""

    DECODE_KIND = _lib.DECODE_V3

    def __init__(self, param: Yolo4Param):
        super().__init__()
        self.param = param
        self.body_output_size = 5 + param.class_num
        self.body_net = _tree(param, 3 * self.body_output_size)
        self.eval()

    def _compile(self, S: int) -> Program:
        body, bb = self.body_net, self.body_net.backbone
        P = Program(S, S)
        M = Recorder(P, _lib.ACT_MISH)       # backbone
        L = Recorder(P, _lib.ACT_LEAKY)      # neck and heads
        x = M.cba("conv1", bb.conv1, P.image16(3))
        feats = []
        for i, st in enumerate(bb.stages):
            n = f"stages.{i}"
            x = M.cba(f"{n}.downsample_conv", st.downsample_conv, x)
            # split_conv0 and split_conv1 read the same input: ONE conv computes both (one read of x, one launch), its
            # couts [0, h) = split_conv0 go to the second half of the concat buffer, [h, 2h) = split_conv1 to a buffer of
            # their own (the residual chain works on it); the chain's last conv fills the first half
            w0, b0, _ = folded(st.split_conv0)
            w1, b1, _ = folded(st.split_conv1)
            h = w0.shape[0]
            catb = P.buffer(x.h, x.w, 2 * h)
            x1 = P.buffer(x.h, x.w, h)
            P.conv(f"{n}.split_conv0+1", x, torch.cat((w0, w1), 0), torch.cat((b0, b1), 0), 1, _lib.ACT_MISH,
                   out=catb.slice(h, h), out2=x1)
            mods = list(st.blocks_conv.children())
            for j, r in enumerate(mods[:-1]):
                t = M.cba(f"{n}.blocks_conv.{j}.block.0", getattr(r.block, "0"), x1)
                x1 = M.cba(f"{n}.blocks_conv.{j}.block.1", getattr(r.block, "1"), t, residual=x1)
            M.cba(f"{n}.blocks_conv.{len(mods) - 1}", mods[-1], x1, out=catb.slice(0, h))
            x = M.cba(f"{n}.concat_conv", st.concat_conv, catb)
            feats.append(x)
        x2, x1, x0 = feats[2], feats[3], feats[4]

        def chain(name: str, seq: nn.Module, t: View, out=None) -> View:
            mods = list(seq.children())
            for k, m in enumerate(mods):
                t = L.cba(f"{name}.{k}", m, t, out=out if k == len(mods) - 1 else None)
            return t

        # SPP: cat([mp13, mp9, mp5, x]); the cascade gives mp5, mp9 = mp5(mp5), mp13 = mp5(mp9)
        c = 512
        spp = P.buffer(x0.h, x0.w, 4 * c, name="cat_spp")
        chain("conv1", body.conv1, x0, out=spp.slice(3 * c, c))
        P.maxpool5_cascade(spp.slice(3 * c, c), spp.slice(2 * c, c), spp.slice(c, c), spp.slice(0, c))
        p5 = chain("conv2", body.conv2, spp)
        up = P.upsample2x(L.cba("up_sample1", getattr(body.up_sample1.up_sample, "0"), p5))
        p4 = chain("make_five_conv1", body.make_five_conv1, P.cat([L.cba("conv_for_P4", body.conv_for_P4, x1), up]))
        up = P.upsample2x(L.cba("up_sample2", getattr(body.up_sample2.up_sample, "0"), p4))
        p3 = chain("make_five_conv2", body.make_five_conv2, P.cat([L.cba("conv_for_P3", body.conv_for_P3, x2), up]))
        p4 = chain("make_five_conv3", body.make_five_conv3, P.cat([L.cba("down_sample1", body.down_sample1, p3), p4]))
        p5 = chain("make_five_conv4", body.make_five_conv4, P.cat([L.cba("down_sample2", body.down_sample2, p4), p5]))
        for out_name, head, t in (("P3", body.yolo_head3, p3), ("P4", body.yolo_head2, p4), ("P5", body.yolo_head1, p5)):
            h = L.cba(f"yolo_head_{out_name}.0", getattr(head, "0"), t)
            P.outputs[out_name] = L.head(f"yolo_head_{out_name}.1", getattr(head, "1"), h)
        return P

    def fuse(self) -> None:
        """Fold BatchNorm into the backbone's BasicConv blocks (YoloV4.py:216-221); the neck keeps its bn keys."""
        for m in self.body_net.backbone.modules():
            if isinstance(getattr(m, "bn", None), nn.BatchNorm2d) and isinstance(getattr(m, "conv", None), nn.Conv2d):
                m.conv = torch_utils.fuse_conv_and_bn(m.conv, m.bn)
                delattr(m, "bn")
        self.invalidate()
