"""YOLOv6 face (n/s: EfficientRep6 backbone, RepBiFPAN neck, 4 levels, 5 landmarks) on the HIP path, behind the
reference's TaskLayer interface.

Mirrors luoyang/yolov6_face/YoloV6Face.py (``forward`` :23-40, ``_forward`` :42-58, ``predict`` :60-85, ``fuse`` :129-138)
and the module tree of luoyang/yolov6_face/YoloV6FaceBody.py (ConvModule :98-134, RepVGGBlock :137-286, SPPF/CSPSPPF
:321-424, RepBlock :427-448, Transpose :475-492, EfficientRep6 :495-586, BiFusion :589-609, RepBiFPANNeck6 :612-716,
create_head :1070-1182, YoloV6FaceHead :1230-1455), so ``state_dict`` keys are identical before and after ``fuse()``.
Every BatchNorm runs with eps 1e-3 (YoloV6FaceBody.py:1479-1486).  RepVGG blocks are re-parameterised to a single
3x3 conv + ReLU when the plan is compiled; ConvTranspose2d(2,2) runs as a 1x1 conv + depth-to-space.
The three models also come as m/l: CSPBepBackboneP6 / CSPRepBiFPANNeckP6 of luoyang/yolov6/YoloV6Body.py
(:811-906, :909-1022) made of BepC3 (:719-739) over BottleRep pairs with a learned shortcut weight (:427-472), SimSPPF /
SPPF (:321-369) and a 17-bin DFL box head (:1322-1343); m uses RepVGG blocks + ReLU, l uses conv + SiLU (:1185-1188);
the face and pose bodies (luoyang/yolov6_face/YoloV6FaceBody.py, luoyang/yolov6_pose/YoloV6PoseBody.py) repeat these classes.
"""
import math
from typing import Any, Dict, List

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import lib as _lib
from . import torch_utils
from .engine import Program, View
from .graph import Recorder, TaskModel, conv_bn, folded
from .param import Yolo6FaceParam

_CFG = {"s": (0.33, 0.50), "n": (0.33, 0.25), "m": (0.60, 0.75), "l": (1.0, 1.0)}
_CSP_E = {"m": 2.0 / 3, "l": 0.5}            # create_body / create_neck, YoloV6Body.py:1042, :1063
DFL_BINS = 17                                 # reg_max 16 -> 17 logits per box side
_BACKBONE = ([1, 6, 12, 18, 6, 6], [64, 128, 256, 512, 768, 1024])
_NECK = ([12] * 6, [512, 256, 128, 256, 512, 1024])


def _cm(cin, cout, k, s=1):
    return conv_bn(cin, cout, k, s, eps=1e-3, momentum=0.03)


def _wrapped(cin, cout, k=3, s=1) -> nn.Module:
    """ConvBNReLU / ConvBNSiLU: a ConvModule under ``.block``."""
    m = nn.Module()
    m.block = _cm(cin, cout, k, s)
    return m


def rep_vgg(cin: int, cout: int, s: int = 1) -> nn.Module:
    m = nn.Module()
    m.is_rep_vgg = True
    m.stride = s
    m.rbr_identity = nn.BatchNorm2d(cin, eps=1e-3, momentum=0.03) if (cin == cout and s == 1) else None
    m.rbr_dense = _cm(cin, cout, 3, s)
    m.rbr_1x1 = _cm(cin, cout, 1, s)
    m.rbr_1x1.conv.padding = (0, 0)
    return m


def rep_vgg_equivalent(m: nn.Module):
    """(weight, bias) of the 3x3 conv equal to the block (get_equivalent_kernel_bias, YoloV6FaceBody.py:202-261, with
    the branch BatchNorms folded first as YoloV6Face.fuse does), or the deployed conv."""
    if getattr(m, "rbr_re_param", None) is not None:
        return m.rbr_re_param.weight.detach(), m.rbr_re_param.bias.detach()

    def branch(b):
        bn = getattr(b, "bn", None)
        if isinstance(bn, nn.BatchNorm2d):
            return torch_utils.fold_conv_bn(b.conv.weight, b.conv.bias, bn)
        return b.conv.weight.detach(), b.conv.bias.detach()
    w3, b3 = branch(m.rbr_dense)
    w1, b1 = branch(m.rbr_1x1)
    w, b = w3 + F.pad(w1, [1, 1, 1, 1]), b3 + b1
    if isinstance(m.rbr_identity, nn.BatchNorm2d):
        c = w.shape[1]
        eye = torch.zeros(c, c, 3, 3)
        eye[torch.arange(c), torch.arange(c), 1, 1] = 1.0
        wi, bi = torch_utils.fold_conv_bn(eye, None, m.rbr_identity)
        w, b = w + wi, b + bi
    return w, b


def _rep_block(cin, cout, n) -> nn.Module:
    m = nn.Module()
    m.conv1 = rep_vgg(cin, cout)
    m.block = nn.Sequential(*(rep_vgg(cout, cout) for _ in range(n - 1))) if n > 1 else None
    return m


def _csp_sppf(cin, cout) -> nn.Module:
    m = nn.Module()
    c_ = int(cout * 0.5)
    inner = nn.Module()
    inner.cv1, inner.cv2, inner.cv3, inner.cv4 = _wrapped(cin, c_, 1), _wrapped(cin, c_, 1), _wrapped(c_, c_, 3), _wrapped(c_, c_, 1)
    inner.cv5, inner.cv6, inner.cv7 = _wrapped(4 * c_, c_, 1), _wrapped(c_, c_, 3), _wrapped(2 * c_, cout, 1)
    m.cspsppf = inner
    return m


def _bifusion(cins, cout) -> nn.Module:
    m = nn.Module()
    m.cv1, m.cv2, m.cv3 = _wrapped(cins[0], cout, 1), _wrapped(cins[1], cout, 1), _wrapped(cout * 3, cout, 1)
    m.up_sample = nn.Module()
    m.up_sample.up_sample_transpose = nn.ConvTranspose2d(cout, cout, (2, 2), (2, 2), bias=True)
    m.down_sample = _wrapped(cout, cout, 3, 2)
    return m


def _bottle_rep(c: int, basic) -> nn.Module:
    """BottleRep(c, c, weight=True) (YoloV6Body.py:451-472): two basic blocks and x * alpha added to their output."""
    m = nn.Module()
    m.conv1, m.conv2 = basic(c, c), basic(c, c)
    m.alpha = nn.Parameter(torch.ones(1))
    return m


def _bep_c3(cin: int, cout: int, n: int, e: float, basic) -> nn.Module:
    """BepC3 (YoloV6Body.py:719-739) with its RepBlock of n // 2 BottleReps (:427-449)."""
    m = nn.Module()
    c_ = int(cout * e)
    m.cv1, m.cv2, m.cv3 = _wrapped(cin, c_, 1), _wrapped(cin, c_, 1), _wrapped(2 * c_, cout, 1)
    rb = nn.Module()
    rb.conv1 = _bottle_rep(c_, basic)
    n = n // 2
    rb.block = nn.Sequential(*(_bottle_rep(c_, basic) for _ in range(n - 1))) if n > 1 else None
    m.m = rb
    return m


def _sppf(cin: int, cout: int, attr: str) -> nn.Module:
    """SimSPPF (``spp_f``) / SPPF (``sppf``) around SPPFModule (YoloV6Body.py:321-369)."""
    m, inner = nn.Module(), nn.Module()
    c_ = cin // 2
    inner.cv1, inner.cv2 = _wrapped(cin, c_, 1), _wrapped(4 * c_, cout, 1)
    setattr(m, attr, inner)
    return m


def _tree(param) -> nn.Module:
    if param.m_type not in _CFG:
        raise NotImplementedError(f"YoloV6 m_type={param.m_type!r} is not built on the HIP path")
    csp = param.m_type in _CSP_E
    dm, wm = _CFG[param.m_type]
    rep = [(max(round(i * dm), 1) if i > 1 else i) for i in (_BACKBONE[0] + _NECK[0])]
    ch = [math.ceil(i * wm / 8) * 8 for i in (_BACKBONE[1] + _NECK[1])]
    if csp:
        e = _CSP_E[param.m_type]
        basic = (lambda a, b, s=1: rep_vgg(a, b, s)) if param.m_type == "m" else (lambda a, b, s=1: _wrapped(a, b, 3, s))

        def stage(cin, cout, n):
            return _bep_c3(cin, cout, n, e, basic)
    else:
        basic, stage = rep_vgg, _rep_block
    fusion = "bi_fusion" if csp else "Bi_fusion"          # attribute names of the two neck classes (:929 vs :624)
    body = nn.Module()
    bb = nn.Module()
    bb.stem = basic(3, ch[0], 2)
    for i in range(1, 6):
        mods = [basic(ch[i - 1], ch[i], 2), stage(ch[i], ch[i], rep[i])]
        if i == 5:
            if csp:
                mods.append(_sppf(ch[5], ch[5], "spp_f" if param.m_type == "m" else "sppf"))
            else:
                mods.append(_csp_sppf(ch[5], ch[5]))
        bb.add_module(f"ERBlock_{i + 1}", nn.Sequential(*mods))
    body.back_bone_network = bb
    nk = nn.Module()
    nk.reduce_layer0 = _wrapped(ch[5], ch[6], 1)
    setattr(nk, f"{fusion}_0", _bifusion([ch[4], ch[6]], ch[6]))
    nk.Rep_p5 = stage(ch[6], ch[6], rep[6])
    nk.reduce_layer1 = _wrapped(ch[6], ch[7], 1)
    setattr(nk, f"{fusion}_1", _bifusion([ch[3], ch[7]], ch[7]))
    nk.Rep_p4 = stage(ch[7], ch[7], rep[7])
    nk.reduce_layer2 = _wrapped(ch[7], ch[8], 1)
    setattr(nk, f"{fusion}_2", _bifusion([ch[2], ch[8]], ch[8]))
    nk.Rep_p3 = stage(ch[8], ch[8], rep[8])
    nk.down_sample_2 = _wrapped(ch[8], ch[8], 3, 2)
    nk.Rep_n4 = stage(ch[8] + ch[8], ch[9], rep[9])
    nk.down_sample_1 = _wrapped(ch[9], ch[9], 3, 2)
    nk.Rep_n5 = stage(ch[7] + ch[9], ch[10], rep[10])
    nk.down_sample_0 = _wrapped(ch[10], ch[10], 3, 2)
    nk.Rep_n6 = stage(ch[6] + ch[10], ch[11], rep[11])
    body.neck_network = nk
    hd = nn.Module()
    hc = [ch[8], ch[9], ch[10], ch[11]][:param.num_layers] if param.num_layers == 4 else [ch[6], ch[8], ch[10]]
    nc, P = param.class_num, getattr(param, "point_num", 0)
    hd.stems = nn.ModuleList(_wrapped(c, c, 1) for c in hc)
    hd.cls_conv_list = nn.ModuleList(_wrapped(c, c, 3) for c in hc)
    hd.reg_conv_list = nn.ModuleList(_wrapped(c, c, 3) for c in hc)
    if P:
        hd.reg_point_conv_list = nn.ModuleList(_wrapped(c, c, 3) for c in hc)
    ncp = nc + (P if getattr(param, "point_conf", False) else 0)     # pose: one confidence logit per keypoint after the classes
    hd.cls_predict_list = nn.ModuleList(nn.Conv2d(c, ncp, 1) for c in hc)
    hd.reg_predict_list = nn.ModuleList(nn.Conv2d(c, 4 * DFL_BINS if csp else 4, 1) for c in hc)
    if P:
        hd.reg_point_predict_list = nn.ModuleList(nn.Conv2d(c, 2 * P, 1) for c in hc)
    hd.proj = nn.Parameter(torch.linspace(0, 16, 17), requires_grad=False)
    for conv in hd.cls_predict_list:                 # YoloV6FaceHead.initialize_biases (:1289-1314)
        conv.bias.data.fill_(-math.log((1 - 1e-2) / 1e-2))
        conv.weight.data.fill_(0.0)
    for conv in hd.reg_predict_list:
        conv.bias.data.fill_(1.0)
        conv.weight.data.fill_(0.0)
    body.head = hd
    return body


class YoloV6Face(TaskModel):
    """``YoloV6Face(param)`` as in the reference.  Images must be CUDA float32 NCHW in [0,1].
    ``predicts`` rows: [x1,y1,x2,y2, 1.0, cls_conf, 5 x (x,y), cls_id]."""

    def __init__(self, param: Yolo6FaceParam):
        super().__init__()
        self.param = param
        self.point_num = getattr(param, "point_num", 0)
        self.point_conf = bool(getattr(param, "point_conf", False)) and self.point_num > 0
        self.yolo_body = _tree(param)
        self.box_channels = 4 * DFL_BINS if param.m_type in _CSP_E else 4        # raw box logits per cell
        self.eval()

    # ---- graph -----------------------------------------------------------------------------------
    @staticmethod
    def _rv(R: Recorder, name: str, m: nn.Module, x: View, **kw) -> View:
        """A basic block: RepVGG (one 3x3 conv + ReLU once re-parameterised) or, in the l model, ConvBNSiLU."""
        if getattr(m, "is_rep_vgg", False):
            w, b = rep_vgg_equivalent(m)
            return R.p.conv(name, x, w, b, m.stride, _lib.ACT_RELU, **kw)
        w, b, s = folded(m.block)
        return R.p.conv(name, x, w, b, s, _lib.ACT_SILU, **kw)

    def _rb(self, R: Recorder, name: str, m: nn.Module, x: View) -> View:
        if hasattr(m, "cv3"):
            return self._bep_c3(R, name, m, x)
        x = self._rv(R, f"{name}.conv1", m.conv1, x)
        if m.block is not None:
            for i, b in enumerate(m.block):
                x = self._rv(R, f"{name}.block.{i}", b, x)
        return x

    def _bep_c3(self, R: Recorder, name: str, m: nn.Module, x: View) -> View:
        """cv3(cat(m(cv1(x)), cv2(x))).  cv1 and cv2 read the same input and run as one conv writing side by side into
        [m out | cv2 out | cv1 out]; the last BottleRep writes the first slice and cv3 reads the first two."""
        act = _lib.ACT_SILU if self.param.m_type == "l" else _lib.ACT_RELU
        c_ = m.cv1.block.conv.out_channels
        (w1, b1, _), (w2, b2, _) = folded(m.cv1.block), folded(m.cv2.block)
        buf = R.p.buffer(x.h, x.w, 3 * c_)
        R.p.conv(f"{name}.cv2+cv1", x, torch.cat((w2, w1), 0), torch.cat((b2, b1), 0), 1, act, out=buf.slice(c_, 2 * c_))
        a = buf.slice(2 * c_, c_)
        bots = [(f"{name}.m.conv1", m.m.conv1)]
        if m.m.block is not None:
            bots += [(f"{name}.m.block.{i}", b) for i, b in enumerate(m.m.block)]
        for i, (bn, bot) in enumerate(bots):
            t = self._rv(R, f"{bn}.conv1", bot.conv1, a)
            a = self._rv(R, f"{bn}.conv2", bot.conv2, t, out=buf.slice(0, c_) if i == len(bots) - 1 else None,
                         residual=a, residual_scale=float(bot.alpha.detach()))
        return self._w(R, f"{name}.cv3", m.cv3, buf.slice(0, 2 * c_), act)

    @staticmethod
    def _w(R: Recorder, name: str, m: nn.Module, x: View, act: int, out=None) -> View:
        return R.cba(name, m.block, x, act=act, out=out)

    def _bifusion(self, R: Recorder, name: str, m: nn.Module, xs: List[View]) -> View:
        relu = _lib.ACT_RELU
        t = m.up_sample.up_sample_transpose
        x0 = R.p.conv_transpose2x(f"{name}.up_sample", xs[0], t.weight, t.bias)
        x1 = self._w(R, f"{name}.cv1", m.cv1, xs[1], relu)
        x2 = self._w(R, f"{name}.down_sample", m.down_sample, self._w(R, f"{name}.cv2", m.cv2, xs[2], relu), relu)
        return self._w(R, f"{name}.cv3", m.cv3, R.p.cat([x0, x1, x2]), relu)

    def _compile(self, S: int) -> Program:
        bb, nk, hd = self.yolo_body.back_bone_network, self.yolo_body.neck_network, self.yolo_body.head
        relu, silu = _lib.ACT_RELU, _lib.ACT_SILU
        csp = self.param.m_type in _CSP_E
        fusion = [getattr(nk, f"{'bi_fusion' if csp else 'Bi_fusion'}_{i}") for i in range(3)]
        P = Program(S, S)
        R = Recorder(P, relu)
        x = self._rv(R, "stem", bb.stem, P.image16(3))
        feats = []
        for i in range(2, 7):
            blk = getattr(bb, f"ERBlock_{i}")
            x = self._rb(R, f"ERBlock_{i}.1", blk[1], self._rv(R, f"ERBlock_{i}.0", blk[0], x))
            if i == 6 and csp:
                sp, a = (blk[2].spp_f, relu) if hasattr(blk[2], "spp_f") else (blk[2].sppf, silu)
                c_ = sp.cv1.block.conv.out_channels
                cat = P.buffer(x.h, x.w, 4 * c_, name="cat_sppf")
                self._w(R, "sppf.cv1", sp.cv1, x, a, out=cat.slice(0, c_))
                P.maxpool5_cascade(cat.slice(0, c_), cat.slice(c_, c_), cat.slice(2 * c_, c_), cat.slice(3 * c_, c_))
                x = self._w(R, "sppf.cv2", sp.cv2, cat, a)
            elif i == 6:
                sp = blk[2].cspsppf
                c_ = sp.cv4.block.conv.out_channels
                cat = P.buffer(x.h, x.w, 4 * c_, name="cat_sppf")
                t = self._w(R, "sppf.cv3", sp.cv3, self._w(R, "sppf.cv1", sp.cv1, x, relu), relu)
                self._w(R, "sppf.cv4", sp.cv4, t, relu, out=cat.slice(0, c_))
                P.maxpool5_cascade(cat.slice(0, c_), cat.slice(c_, c_), cat.slice(2 * c_, c_), cat.slice(3 * c_, c_))
                y0 = self._w(R, "sppf.cv2", sp.cv2, x, relu)
                y3 = self._w(R, "sppf.cv6", sp.cv6, self._w(R, "sppf.cv5", sp.cv5, cat, relu), relu)
                x = self._w(R, "sppf.cv7", sp.cv7, P.cat([y0, y3]), relu)
            feats.append(x)
        x4, x3, x2, x1, x0 = feats

        fpn0 = self._w(R, "reduce_layer0", nk.reduce_layer0, x0, relu)
        f0 = self._rb(R, "Rep_p5", nk.Rep_p5, self._bifusion(R, "bi_fusion_0", fusion[0], [fpn0, x1, x2]))
        fpn1 = self._w(R, "reduce_layer1", nk.reduce_layer1, f0, relu)
        f1 = self._rb(R, "Rep_p4", nk.Rep_p4, self._bifusion(R, "bi_fusion_1", fusion[1], [fpn1, x2, x3]))
        fpn2 = self._w(R, "reduce_layer2", nk.reduce_layer2, f1, relu)
        pan3 = self._rb(R, "Rep_p3", nk.Rep_p3, self._bifusion(R, "bi_fusion_2", fusion[2], [fpn2, x3, x4]))
        pan2 = self._rb(R, "Rep_n4", nk.Rep_n4, P.cat([self._w(R, "down_sample_2", nk.down_sample_2, pan3, relu), fpn2]))
        pan1 = self._rb(R, "Rep_n5", nk.Rep_n5, P.cat([self._w(R, "down_sample_1", nk.down_sample_1, pan2, relu), fpn1]))
        pan0 = self._rb(R, "Rep_n6", nk.Rep_n6, P.cat([self._w(R, "down_sample_0", nk.down_sample_0, pan1, relu), fpn0]))

        nc, K = self.param.class_num, 2 * self.point_num
        ncp = nc + (self.point_num if self.point_conf else 0)
        nb = self.box_channels
        for i, t in enumerate((pan3, pan2, pan1, pan0)[:self.param.num_layers]):
            # [ltrb (or 4 x 17 DFL logits) | landmarks | (kpt conf) | cls]
            raw = P.buffer(t.h, t.w, nb + K + ncp, torch.float32, name=f"raw{i}")
            s = self._w(R, f"head.stems.{i}", hd.stems[i], t, silu)
            R.head(f"head.reg_pred.{i}", hd.reg_predict_list[i],
                   self._w(R, f"head.reg_conv.{i}", hd.reg_conv_list[i], s, silu), out=raw.slice(0, nb))
            if K:
                R.head(f"head.point_pred.{i}", hd.reg_point_predict_list[i],
                       self._w(R, f"head.point_conv.{i}", hd.reg_point_conv_list[i], s, silu), out=raw.slice(nb, K))
            cp = hd.cls_predict_list[i]
            cw, cb = cp.weight.detach(), cp.bias.detach()
            if self.point_conf:
                # the reference's conv emits [classes | keypoint confidences] (YoloV6PoseBody.py:1412-1430); rows follow
                # the order of its output columns, confidences first, so the kernel can treat both as "scores"
                cw, cb = torch.cat((cw[nc:], cw[:nc]), 0), torch.cat((cb[nc:], cb[:nc]), 0)
            P.conv(f"head.cls_pred.{i}", self._w(R, f"head.cls_conv.{i}", hd.cls_conv_list[i], s, silu), cw, cb, 1,
                   _lib.ACT_NONE, out=raw.slice(nb + K, ncp), out_f32=True)
            P.outputs[f"L{i}"] = raw
        return P

    # ---- reference interface -------------------------------------------------------------------
    def raw_heads(self, images: torch.Tensor) -> List[torch.Tensor]:
        """Raw head tensors per level, float32 NHWC (B,h,w,4+2P+nc), channel order [ltrb | landmarks | cls]
        (pose: [ltrb | keypoints | keypoint confidences | cls]; detect m/l: [4 x 17 DFL logits | cls])."""
        out = self.program().run(images)
        return [out[f"L{i}"] for i in range(self.param.num_layers)]

    def _decode(self, raws, fmt: str, arena=None) -> torch.Tensor:
        dfl = self.box_channels != 4
        if dfl and not self.point_num:
            return torch_utils.decode_dfl(raws, self.param.strides, DFL_BINS, self.param.class_num, fmt, unit_conf=True,
                                          arena=arena)
        ncp = self.param.class_num + (self.point_num if self.point_conf else 0)
        return torch_utils.decode_ltrb_points(raws, self.param.strides, self.point_num, ncp, fmt,
                                              bins=DFL_BINS if dfl else 1, arena=arena)

    def _forward(self, images: torch.Tensor) -> List[torch.Tensor]:
        """[(B, A, 5+2P+nc)]: cxcywh, 1.0, landmarks, class scores (YoloV6Face.py:42-58)."""
        return [self._decode(self.raw_heads(images), "cxcywh")]

    def forward(self, tensor_dict: Dict[str, Any]) -> Dict[str, Any]:
        """Inference entries of YoloV6Face.py:23-40 (the anchor boxes used only by the loss are left out)."""
        raws = self.raw_heads(tensor_dict["batch_images"])
        pred = self._decode(raws, "xyxy")
        K = 2 * self.point_num
        B = pred.shape[0]
        if self.point_conf:
            tensor_dict["point_conf_tensor"] = pred[..., 5 + K:5 + K + self.point_num]
            K_all = K + self.point_num
        else:
            K_all = K
        tensor_dict["cls_score_tensor"] = pred[..., 5 + K_all:]
        tensor_dict["predict_xy_xy"] = pred[..., :4]
        if K:
            tensor_dict["point_predict_xy"] = pred[..., 5:5 + K]
        nb = self.box_channels
        tensor_dict["reg_distribute_tensor"] = torch.cat([r[..., :nb].reshape(B, -1, nb) for r in raws], 1)
        tensor_dict["num_anchors_list"] = [r.shape[1] * r.shape[2] for r in raws]
        tensor_dict["stride_tensor"] = torch.cat([torch.full((r.shape[1] * r.shape[2], 1), float(s), device=pred.device)
                                                  for r, s in zip(raws, self.param.strides)])
        tensor_dict["_pred_xyxy"] = pred
        return tensor_dict

    def decode_predict(self, raws, arena=None) -> torch.Tensor:
        return self._decode(raws, "xyxy", arena=arena)

    def nms_predict(self, pred: torch.Tensor, arena=None):
        rows, counts, _ = torch_utils.nms_batched(pred, self.param.img_size, "xyxy", self.param.conf_threshold,
                                                  self.param.nms_threshold, (3 if self.point_conf else 2) * self.point_num,
                                                  arena=arena)
        return rows, counts

    def predict(self, tensor_dict: Dict[str, Any]) -> Dict[str, Any]:
        tensor_dict = self.forward(tensor_dict)
        rows, counts = self.nms_predict(tensor_dict.pop("_pred_xyxy"))
        tensor_dict["predicts"] = torch_utils.rows_to_host(rows, counts)
        return tensor_dict

    def fuse(self):
        """Two passes like YoloV6Face.py:129-138: fold every ConvModule's BatchNorm, then deploy the RepVGG blocks
        (``rbr_re_param`` conv with bias replaces the three branches)."""
        self.fuse_conv_bn_holders()
        for m in self.modules():
            if getattr(m, "is_rep_vgg", False) and getattr(m, "rbr_re_param", None) is None:
                w, b = rep_vgg_equivalent(m)
                conv = nn.Conv2d(w.shape[1], w.shape[0], 3, m.stride, 1, bias=True).requires_grad_(False)
                conv.weight.copy_(w)
                conv.bias.copy_(b)
                m.rbr_re_param = conv
                del m.rbr_dense, m.rbr_1x1
                if hasattr(m, "rbr_identity"):
                    del m.rbr_identity
                m.rbr_identity = None
        self.invalidate()


class YoloV6(YoloV6Face):
    """``YoloV6(param)`` (detect, n/s/m/l): for n/s the graph of luoyang/yolov6/YoloV6Body.py is the face graph without the landmark
    branch of the head (five modules per level instead of seven, :1071-1165, :1245-1252); ``_forward`` /
    ``predict`` as luoyang/yolov6/YoloV6.py:41-83.  ``predicts`` rows: [x1,y1,x2,y2, 1.0, cls_conf, cls_id]."""

    def __init__(self, param):
        if getattr(param, "point_num", 0):
            raise ValueError("YoloV6 (detect) takes a Yolo6Param; use YoloV6Face for landmark heads")
        super().__init__(param)


class YoloV6Pose(YoloV6Face):
    """``YoloV6Pose(param)``: the face graph with one confidence per keypoint, which the reference takes from the
    class-prediction conv (``class_num + point_num`` outputs, luoyang/yolov6_pose/YoloV6PoseBody.py:1412-1430);
    ``_forward`` / ``predict`` as luoyang/yolov6_pose/YoloV6Pose.py:43-89.
    ``predicts`` rows: [x1,y1,x2,y2, 1.0, cls_conf, 17 x (x,y), 17 confidences, cls_id]."""

    def __init__(self, param):
        if not getattr(param, "point_conf", False):
            raise ValueError("YoloV6Pose takes a Yolo6PoseParam")
        super().__init__(param)
