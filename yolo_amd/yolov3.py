"""YOLOv3 (DarkNet-53) on the HIP path, behind the reference's TaskLayer interface.

Mirrors luoyang/yolov3/YoloV3.py and the module trees of luoyang/yolov3/DarkNet53.py (BasicBlock :10-47,
_make_layer :50-61, DarkNet53 :64-107) and luoyang/yolov3/YoloV3BackBone.py (_conv :8-20, _make_last_layers :23-38,
forward :78-105), so ``state_dict`` keys are identical (``body_net.darknet53.conv1.0.weight`` ...).
BatchNorm uses torch's default eps 1e-5; the activation is LeakyReLU(0.1).  The reference defines no
``fuse`` for this model; BatchNorm is folded when the plan is compiled.
"""
import torch.nn as nn

from . import lib as _lib
from .engine import Program, View
from .graph import AnchorTaskModel, Recorder
from .param import Yolo3Param

EPS = 1e-5


def _seq(*mods) -> nn.Module:
    """Holder with numeric child names like nn.Sequential; ``None`` leaves a gap (an activation in the reference)."""
    m = nn.Module()
    for i, c in enumerate(mods):
        if c is not None:
            m.add_module(str(i), c)
    return m


def _cbl(cin: int, cout: int, k: int, s: int = 1) -> nn.Module:
    m = nn.Module()
    m.conv = nn.Conv2d(cin, cout, (k, k), (s, s), k // 2, bias=False)
    m.bn = nn.BatchNorm2d(cout)
    return m


def _basic_block(cin: int, planes) -> nn.Module:
    m = nn.Module()
    m.conv = _seq(nn.Conv2d(cin, planes[0], 1, bias=False), nn.BatchNorm2d(planes[0]), None,
                  nn.Conv2d(planes[0], planes[1], 3, padding=1, bias=False), nn.BatchNorm2d(planes[1]))
    return m


def _make_layer(planes, blocks: int) -> nn.Module:
    m = _cbl(planes[0], planes[1], 3, 2)
    for i in range(blocks):
        m.add_module(f"residual_{i}", _basic_block(planes[1], planes))
    return m


def _last_layers(filters, cin: int, cout: int) -> nn.Module:
    f0, f1 = filters
    return _seq(_cbl(cin, f0, 1), _cbl(f0, f1, 3), _cbl(f1, f0, 1), _cbl(f0, f1, 3), _cbl(f1, f0, 1), _cbl(f0, f1, 3),
                nn.Conv2d(f1, cout, 1))


def _tree(param: Yolo3Param, out_ch: int) -> nn.Module:
    body = nn.Module()
    dn = nn.Module()
    dn.conv1 = _seq(nn.Conv2d(3, 32, 3, padding=1, bias=False), nn.BatchNorm2d(32))
    filt = param.darknet53_output_filters
    cin = 32
    for i, (blocks, f) in enumerate(zip(param.dark_blocks, filt)):
        dn.add_module(f"layer_{i + 1}", _make_layer((cin, f), blocks))
        cin = f
    body.darknet53 = dn
    body.last_layer0 = _last_layers((512, 1024), filt[-1], out_ch)
    body.last_layer1_conv = _cbl(512, 256, 1)
    body.last_layer1 = _last_layers((256, 512), filt[-2] + 256, out_ch)
    body.last_layer2_conv = _cbl(256, 128, 1)
    body.last_layer2 = _last_layers((128, 256), filt[-3] + 128, out_ch)
    return body


class YoloV3(AnchorTaskModel):
    """``YoloV3(param)`` as in the reference.  Images must be CUDA float32 NCHW in [0,1]."""

    DECODE_KIND = _lib.DECODE_V3

    def __init__(self, param: Yolo3Param):
        super().__init__()
        self.param = param
        self.body_output_size = 5 + param.class_num
        self.body_net = _tree(param, 3 * self.body_output_size)
        self.eval()

    def _compile(self, S: int) -> Program:
        body, dn = self.body_net, self.body_net.darknet53
        P = Program(S, S)
        R = Recorder(P, _lib.ACT_LEAKY)
        x = R.cba("conv1", dn.conv1, P.image16(3), conv_name="0", bn_name="1")
        feats = []
        for i in range(1, 6):
            layer = getattr(dn, f"layer_{i}")
            x = R.cba(f"layer_{i}", layer, x)
            for j in range(self.param.dark_blocks[i - 1]):
                blk = getattr(layer, f"residual_{j}").conv
                t = R.cba(f"layer_{i}.residual_{j}.0", blk, x, conv_name="0", bn_name="1")
                x = R.cba(f"layer_{i}.residual_{j}.3", blk, t, conv_name="3", bn_name="4", residual=x)
            feats.append(x)
        x2, x1, x0 = feats[2], feats[3], feats[4]

        def last(name: str, seq: nn.Module, t: View, out_name: str) -> View:
            for k in range(5):
                t = R.cba(f"{name}.{k}", getattr(seq, str(k)), t)
            h = R.cba(f"{name}.5", getattr(seq, "5"), t)
            P.outputs[out_name] = R.head(f"{name}.6", getattr(seq, "6"), h)
            return t

        b0 = last("last_layer0", body.last_layer0, x0, "P5")
        u = P.upsample2x(R.cba("last_layer1_conv", body.last_layer1_conv, b0))
        b1 = last("last_layer1", body.last_layer1, P.cat([u, x1]), "P4")
        u = P.upsample2x(R.cba("last_layer2_conv", body.last_layer2_conv, b1))
        last("last_layer2", body.last_layer2, P.cat([u, x2]), "P3")
        return P
