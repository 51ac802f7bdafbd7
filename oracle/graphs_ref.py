"""TEST INFRASTRUCTURE ONLY -- PyTorch-CPU fp32 restatements of the reference's conv graphs other than
YOLOv5 (see v5_ref.py for that one).

Each ``*_body`` function is a functional interpreter over a ``state_dict`` with the reference's key
names and returns the raw head tensors in NCHW, as the reference's ``body_net`` does.  They are pinned
by tests/golden/<model>_*.npz, which hold outputs of the reference itself on the same seeded weights.
"""
from typing import Callable, Dict, List

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def silu(y):
    return y * torch.sigmoid(y)


def leaky(y):
    return F.leaky_relu(y, 0.1)


def mish(y):
    return y * torch.tanh(F.softplus(y))


def conv_bn_act(sd: SD, conv: str, bn: str, x: torch.Tensor, s: int, act: Callable, eps: float) -> torch.Tensor:
    """act(BN(conv(x))) with eval-mode BN, or act(conv(x)+bias) when the checkpoint is fused (no bn keys)."""
    w = sd[conv + ".weight"]
    k = w.shape[-1]
    y = F.conv2d(x, w, sd.get(conv + ".bias"), stride=s, padding=k // 2)
    if bn + ".weight" in sd:
        y = F.batch_norm(y, sd[bn + ".running_mean"], sd[bn + ".running_var"], sd[bn + ".weight"], sd[bn + ".bias"],
                         False, 0.0, eps)
    return act(y)


def _count(sd: SD, fmt: str) -> int:
    n = 0
    while fmt.format(n) in sd:
        n += 1
    return n


# ------------------------------------------------------------------------------------------------
# YOLOv8: luoyang/yolov8/YoloV8Body.py  Conv :26-63 (BN eps 1e-3, SiLU), Bottleneck :66-80, C2f :83-99,
# SPPF :102-114, Backbone :117-151, YoloV8Body.forward :288-308
# ------------------------------------------------------------------------------------------------
def _v8_conv(sd, p, x, s=1):
    return conv_bn_act(sd, p + ".conv", p + ".bn", x, s, silu, 1e-3)


def _v8_c2f(sd, p, x, shortcut):
    y = _v8_conv(sd, p + ".cv1", x)
    c = y.shape[1] // 2
    ys = [y[:, :c], y[:, c:]]
    for i in range(_count(sd, p + ".m.{}.cv1.conv.weight")):
        t = _v8_conv(sd, f"{p}.m.{i}.cv2", _v8_conv(sd, f"{p}.m.{i}.cv1", ys[-1]))
        ys.append(ys[-1] + t if shortcut else t)
    return _v8_conv(sd, p + ".cv2", torch.cat(ys, 1))


def _v8_sppf(sd, p, x):
    x = _v8_conv(sd, p + ".cv1", x)
    y1 = F.max_pool2d(x, 5, 1, 2)
    y2 = F.max_pool2d(y1, 5, 1, 2)
    return _v8_conv(sd, p + ".cv2", torch.cat((x, y1, y2, F.max_pool2d(y2, 5, 1, 2)), 1))


def v8_body(sd: SD, images: torch.Tensor) -> List[torch.Tensor]:
    """images (B,3,S,S) -> [(B, 64+nc, S/8, S/8), (.., S/16, ..), (.., S/32, ..)], channels [reg 64 | cls nc]."""
    bb, n = "body_net.backbone.", "body_net."
    x = _v8_conv(sd, bb + "stem", images, 2)
    x = _v8_c2f(sd, bb + "dark2.1", _v8_conv(sd, bb + "dark2.0", x, 2), True)
    feat1 = _v8_c2f(sd, bb + "dark3.1", _v8_conv(sd, bb + "dark3.0", x, 2), True)
    feat2 = _v8_c2f(sd, bb + "dark4.1", _v8_conv(sd, bb + "dark4.0", feat1, 2), True)
    x = _v8_c2f(sd, bb + "dark5.1", _v8_conv(sd, bb + "dark5.0", feat2, 2), True)
    feat3 = _v8_sppf(sd, bb + "dark5.2", x)
    up = lambda t: F.interpolate(t, scale_factor=2, mode="nearest")
    p4 = _v8_c2f(sd, n + "conv3_for_up_sample1", torch.cat((up(feat3), feat2), 1), False)
    p3 = _v8_c2f(sd, n + "conv3_for_up_sample2", torch.cat((up(p4), feat1), 1), False)
    p4 = _v8_c2f(sd, n + "conv3_for_down_sample1", torch.cat((_v8_conv(sd, n + "down_sample1", p3, 2), p4), 1), False)
    p5 = _v8_c2f(sd, n + "conv3_for_down_sample2", torch.cat((_v8_conv(sd, n + "down_sample2", p4, 2), feat3), 1), False)
    outs = []
    for i, t in enumerate((p3, p4, p5)):
        br = []
        for tag in ("cv2", "cv3"):
            a = _v8_conv(sd, f"{n}{tag}.{i}.1", _v8_conv(sd, f"{n}{tag}.{i}.0", t))
            br.append(F.conv2d(a, sd[f"{n}{tag}.{i}.2.weight"], sd[f"{n}{tag}.{i}.2.bias"]))
        outs.append(torch.cat(br, 1))
    return outs


# ------------------------------------------------------------------------------------------------
# YOLOv3: luoyang/yolov3/DarkNet53.py BasicBlock :10-47 (x + conv3x3(conv1x1(x)), BN eps 1e-5, LeakyReLU 0.1),
# _make_layer :50-61, DarkNet53.forward :90-107; luoyang/yolov3/YoloV3BackBone.py forward :78-105
# ------------------------------------------------------------------------------------------------
def _v3_cbl(sd, p, x, s=1):
    return conv_bn_act(sd, p + ".conv", p + ".bn", x, s, leaky, 1e-5)


def _v3_last(sd, p, x):
    for k in range(5):
        x = _v3_cbl(sd, f"{p}.{k}", x)
    out = F.conv2d(_v3_cbl(sd, f"{p}.5", x), sd[f"{p}.6.weight"], sd[f"{p}.6.bias"])
    return x, out


def v3_body(sd: SD, images: torch.Tensor) -> List[torch.Tensor]:
    """images (B,3,S,S) -> raw heads [stride 8, 16, 32], each (B, 3*(5+nc), h, w)."""
    d = "body_net.darknet53."
    x = conv_bn_act(sd, d + "conv1.0", d + "conv1.1", images, 1, leaky, 1e-5)
    feats = []
    for i in range(1, 6):
        lp = f"{d}layer_{i}"
        x = _v3_cbl(sd, lp, x, 2)
        for j in range(_count(sd, lp + ".residual_{}.conv.0.weight")):
            rp = f"{lp}.residual_{j}.conv."
            t = conv_bn_act(sd, rp + "0", rp + "1", x, 1, leaky, 1e-5)
            x = x + conv_bn_act(sd, rp + "3", rp + "4", t, 1, leaky, 1e-5)
        feats.append(x)
    x2, x1, x0 = feats[2], feats[3], feats[4]
    up = lambda t: F.interpolate(t, scale_factor=2, mode="nearest")
    n = "body_net."
    b0, out0 = _v3_last(sd, n + "last_layer0", x0)
    b1, out1 = _v3_last(sd, n + "last_layer1", torch.cat((up(_v3_cbl(sd, n + "last_layer1_conv", b0)), x1), 1))
    _, out2 = _v3_last(sd, n + "last_layer2", torch.cat((up(_v3_cbl(sd, n + "last_layer2_conv", b1)), x2), 1))
    return [out2, out1, out0]


# ------------------------------------------------------------------------------------------------
# YOLOv4: luoyang/yolov4/YoloV4BackBone.py BasicConv :18-43 (Mish), ResBlock :46-59, ResBlockBody :62-93,
# CSPDarkNet.forward :119-127, conv2d :137-148 (LeakyReLU 0.1), SPP :151-162 ([13,9,5,x]), forward :233-262;
# BN eps 1e-5; a backbone fused by YoloV4.fuse (:216-221) has conv.bias and no bn keys
# ------------------------------------------------------------------------------------------------
def _v4_basic(sd, p, x, s=1):
    return conv_bn_act(sd, p + ".conv", p + ".bn", x, s, mish, 1e-5)


def _v4_seq(sd, p, x, n):
    for k in range(n):
        x = _v3_cbl(sd, f"{p}.{k}", x)
    return x


def v4_body(sd: SD, images: torch.Tensor) -> List[torch.Tensor]:
    """images (B,3,S,S) -> raw heads [stride 8, 16, 32], each (B, 3*(5+nc), h, w)."""
    b = "body_net.backbone."
    x = _v4_basic(sd, b + "conv1", images)
    feats = []
    for i in range(5):
        sp = f"{b}stages.{i}."
        x = _v4_basic(sd, sp + "downsample_conv", x, 2)
        x0 = _v4_basic(sd, sp + "split_conv0", x)
        x1 = _v4_basic(sd, sp + "split_conv1", x)
        nres = _count(sd, sp + "blocks_conv.{}.block.0.conv.weight")
        for j in range(nres):
            rp = f"{sp}blocks_conv.{j}.block."
            x1 = x1 + _v4_basic(sd, rp + "1", _v4_basic(sd, rp + "0", x1))
        x1 = _v4_basic(sd, f"{sp}blocks_conv.{nres}", x1)
        x = _v4_basic(sd, sp + "concat_conv", torch.cat((x1, x0), 1))
        feats.append(x)
    x2, x1, x0 = feats[2], feats[3], feats[4]
    n = "body_net."
    up = lambda t: F.interpolate(t, scale_factor=2, mode="nearest")
    p5 = _v4_seq(sd, n + "conv1", x0, 3)
    p5 = torch.cat([F.max_pool2d(p5, k, 1, k // 2) for k in (13, 9, 5)] + [p5], 1)
    p5 = _v4_seq(sd, n + "conv2", p5, 3)
    p4 = torch.cat((_v3_cbl(sd, n + "conv_for_P4", x1), up(_v3_cbl(sd, n + "up_sample1.up_sample.0", p5))), 1)
    p4 = _v4_seq(sd, n + "make_five_conv1", p4, 5)
    p3 = torch.cat((_v3_cbl(sd, n + "conv_for_P3", x2), up(_v3_cbl(sd, n + "up_sample2.up_sample.0", p4))), 1)
    p3 = _v4_seq(sd, n + "make_five_conv2", p3, 5)
    p4 = _v4_seq(sd, n + "make_five_conv3", torch.cat((_v3_cbl(sd, n + "down_sample1", p3, 2), p4), 1), 5)
    p5 = _v4_seq(sd, n + "make_five_conv4", torch.cat((_v3_cbl(sd, n + "down_sample2", p4, 2), p5), 1), 5)
    outs = []
    for head, t in (("yolo_head3", p3), ("yolo_head2", p4), ("yolo_head1", p5)):
        h = _v3_cbl(sd, f"{n}{head}.0", t)
        outs.append(F.conv2d(h, sd[f"{n}{head}.1.weight"], sd[f"{n}{head}.1.bias"]))
    return outs


# ------------------------------------------------------------------------------------------------
# YOLOv7 pose: luoyang/yolov7_pose/YoloV7PoseBody.py Conv :24-48 (BN eps 1e-3, SiLU), MultiConcatBlock :51-73,
# TransitionBlock :85-100, Backbone.forward :131-148, SppCspC :175-192, RepConv.forward :247-254 (three branches,
# or the single rbr_reparam conv after fuse), YoloV7PoseBody.forward :479-509
# ------------------------------------------------------------------------------------------------
def _v7_conv(sd, p, x, s=1):
    return conv_bn_act(sd, p + ".conv", p + ".bn", x, s, silu, 1e-3)


def _v7_mcb(sd, p, x, ids):
    xs = [_v7_conv(sd, p + ".cv1", x), _v7_conv(sd, p + ".cv2", x)]
    for i in range(_count(sd, p + ".cv3.{}.conv.weight")):
        xs.append(_v7_conv(sd, f"{p}.cv3.{i}", xs[-1]))
    return _v7_conv(sd, p + ".cv4", torch.cat([xs[i] for i in ids], 1))


def _v7_transition(sd, p, x):
    x1 = _v7_conv(sd, p + ".cv1", F.max_pool2d(x, 2, 2))
    x2 = _v7_conv(sd, p + ".cv3", _v7_conv(sd, p + ".cv2", x), 2)
    return torch.cat((x2, x1), 1)


def _v7_rep(sd, p, x):
    if p + ".rbr_reparam.weight" in sd:
        return silu(F.conv2d(x, sd[p + ".rbr_reparam.weight"], sd[p + ".rbr_reparam.bias"], padding=1))
    if p + ".conv.weight" in sd:                         # the 'x' variant uses a plain Conv here
        return _v7_conv(sd, p, x)

    def bn(y, q):
        return F.batch_norm(y, sd[q + ".running_mean"], sd[q + ".running_var"], sd[q + ".weight"], sd[q + ".bias"],
                            False, 0.0, 1e-3)
    y = bn(F.conv2d(x, sd[p + ".rbr_dense.0.weight"], padding=1), p + ".rbr_dense.1")
    y = y + bn(F.conv2d(x, sd[p + ".rbr_1x1.0.weight"]), p + ".rbr_1x1.1")
    if p + ".rbr_identity.weight" in sd:
        y = y + bn(x, p + ".rbr_identity")
    return silu(y)


def v7pose_body(sd: SD, images: torch.Tensor) -> List[torch.Tensor]:
    """images (B,3,S,S) -> raw heads [stride 8, 16, 32], each (B, 3*(5+3P+nc), h, w).  The width (l: 32 base channels,
    x: 40) decides which branch outputs the concat blocks gather (YoloV7PoseBody.py:416-421)."""
    phi = {32: "l", 40: "x"}[sd["body_net.backbone.stem.0.conv.weight"].shape[0]]
    bids = {"l": [-1, -3, -5, -6], "x": [-1, -3, -5, -7, -8]}[phi]
    nids = {"l": [-1, -2, -3, -4, -5, -6], "x": [-1, -3, -5, -7, -8]}[phi]
    b, n = "body_net.backbone.", "body_net."
    x = _v7_conv(sd, b + "stem.2", _v7_conv(sd, b + "stem.1", _v7_conv(sd, b + "stem.0", images), 2))
    x = _v7_mcb(sd, b + "dark2.1", _v7_conv(sd, b + "dark2.0", x, 2), bids)
    feat1 = _v7_mcb(sd, b + "dark3.1", _v7_transition(sd, b + "dark3.0", x), bids)
    feat2 = _v7_mcb(sd, b + "dark4.1", _v7_transition(sd, b + "dark4.0", feat1), bids)
    feat3 = _v7_mcb(sd, b + "dark5.1", _v7_transition(sd, b + "dark5.0", feat2), bids)
    sp = n + "spp_csp_c"
    x1 = _v7_conv(sd, sp + ".cv4", _v7_conv(sd, sp + ".cv3", _v7_conv(sd, sp + ".cv1", feat3)))
    y1 = torch.cat([x1] + [F.max_pool2d(x1, k, 1, k // 2) for k in (5, 9, 13)], 1)
    y1 = _v7_conv(sd, sp + ".cv6", _v7_conv(sd, sp + ".cv5", y1))
    p5 = _v7_conv(sd, sp + ".cv7", torch.cat((y1, _v7_conv(sd, sp + ".cv2", feat3)), 1))
    up = lambda t: F.interpolate(t, scale_factor=2, mode="nearest")
    p4 = torch.cat((_v7_conv(sd, n + "conv_for_feat2", feat2), up(_v7_conv(sd, n + "conv_for_P5", p5))), 1)
    p4 = _v7_mcb(sd, n + "conv3_for_up_sample1", p4, nids)
    p3 = torch.cat((_v7_conv(sd, n + "conv_for_feat1", feat1), up(_v7_conv(sd, n + "conv_for_P4", p4))), 1)
    p3 = _v7_mcb(sd, n + "conv3_for_up_sample2", p3, nids)
    p4 = _v7_mcb(sd, n + "conv3_for_down_sample1", torch.cat((_v7_transition(sd, n + "down_sample1", p3), p4), 1), nids)
    p5 = _v7_mcb(sd, n + "conv3_for_down_sample2", torch.cat((_v7_transition(sd, n + "down_sample2", p4), p5), 1), nids)
    outs = []
    for rep, head, t in (("rep_conv_1", "yolo_head_P3", p3), ("rep_conv_2", "yolo_head_P4", p4),
                         ("rep_conv_3", "yolo_head_P5", p5)):
        outs.append(F.conv2d(_v7_rep(sd, n + rep, t), sd[n + head + ".weight"], sd[n + head + ".bias"]))
    return outs


# ------------------------------------------------------------------------------------------------
# YOLOv6 face (n/s): luoyang/yolov6_face/YoloV6FaceBody.py ConvModule :98-134, RepVGGBlock.forward :187-200 (ReLU over
# the sum of the 3x3, 1x1 and identity-BN branches, or the deployed rbr_re_param conv), CSPSPPFModule :372-396,
# RepBlock :427-448, Transpose :475-492, EfficientRep6.forward :572-586, BiFusion.forward :605-609,
# RepBiFPANNeck6.forward :688-716, YoloV6FaceHead.forward :1399-1437 (per-level convs); all BN eps 1e-3 (:1479-1486)
# ------------------------------------------------------------------------------------------------
def _v6_cm(sd, p, x, s, act, pad=None):
    w = sd[p + ".conv.weight"]
    k = w.shape[-1]
    y = F.conv2d(x, w, sd.get(p + ".conv.bias"), stride=s, padding=k // 2 if pad is None else pad)
    if p + ".bn.weight" in sd:
        y = F.batch_norm(y, sd[p + ".bn.running_mean"], sd[p + ".bn.running_var"], sd[p + ".bn.weight"],
                         sd[p + ".bn.bias"], False, 0.0, 1e-3)
    return y if act is None else act(y)


def _v6_rep(sd, p, x, s=1):
    if p + ".rbr_re_param.weight" in sd:
        return F.relu(F.conv2d(x, sd[p + ".rbr_re_param.weight"], sd[p + ".rbr_re_param.bias"], stride=s, padding=1))
    y = _v6_cm(sd, p + ".rbr_dense", x, s, None) + _v6_cm(sd, p + ".rbr_1x1", x, s, None, pad=0)
    q = p + ".rbr_identity"
    if q + ".weight" in sd:
        y = y + F.batch_norm(x, sd[q + ".running_mean"], sd[q + ".running_var"], sd[q + ".weight"], sd[q + ".bias"],
                             False, 0.0, 1e-3)
    return F.relu(y)


def _v6_is_rep(sd, p):
    return (p + ".rbr_dense.conv.weight") in sd or (p + ".rbr_re_param.weight") in sd


def _v6_repblock(sd, p, x):
    x = _v6_rep(sd, p + ".conv1", x)
    i = 0
    while _v6_is_rep(sd, f"{p}.block.{i}"):
        x = _v6_rep(sd, f"{p}.block.{i}", x)
        i += 1
    return x


def _v6_w(sd, p, x, s=1, act=F.relu):
    return _v6_cm(sd, p + ".block", x, s, act)


def _v6_bifusion(sd, p, xs):
    x0 = F.conv_transpose2d(xs[0], sd[p + ".up_sample.up_sample_transpose.weight"],
                            sd[p + ".up_sample.up_sample_transpose.bias"], stride=2)
    x1 = _v6_w(sd, p + ".cv1", xs[1])
    x2 = _v6_w(sd, p + ".down_sample", _v6_w(sd, p + ".cv2", xs[2]), 2)
    return _v6_w(sd, p + ".cv3", torch.cat((x0, x1, x2), 1))


# YOLOv6 detect m/l: luoyang/yolov6/YoloV6Body.py BottleRep.forward :468-472 (outputs + alpha * x), RepBlock :427-449
# (conv1 + block of BottleReps), BepC3.forward :738-739, SPPFModule.forward :333-337 under SimSPPF (``spp_f``, ReLU) or
# SPPF (``sppf``, SiLU), CSPBepBackboneP6.forward :892-906, CSPRepBiFPANNeckP6.forward :1000-1022; the basic block is
# RepVGGBlock (m) or ConvBNSiLU (l), and BepC3's own 1x1 convs follow it (ReLU / SiLU), :1185-1188, :730-736
def _v6_basic(sd, p, x, s=1):
    return _v6_rep(sd, p, x, s) if _v6_is_rep(sd, p) else _v6_w(sd, p, x, s, silu)


def _v6_bottle_rep(sd, p, x):
    return _v6_basic(sd, p + ".conv2", _v6_basic(sd, p + ".conv1", x)) + sd[p + ".alpha"] * x


def _v6_bepc3(sd, p, x, act):
    y = _v6_bottle_rep(sd, p + ".m.conv1", _v6_w(sd, p + ".cv1", x, 1, act))
    i = 0
    while f"{p}.m.block.{i}.alpha" in sd:
        y = _v6_bottle_rep(sd, f"{p}.m.block.{i}", y)
        i += 1
    return _v6_w(sd, p + ".cv3", torch.cat((y, _v6_w(sd, p + ".cv2", x, 1, act)), 1), 1, act)


def v6face_body(sd: SD, images: torch.Tensor) -> List[torch.Tensor]:
    """images (B,3,S,S) -> per level (stride 8, 16, 32, 64) raw head logits (B, 4+2P+nc, h, w) in the channel
    order [ltrb | landmarks | cls] (detect m/l: [4 x 17 DFL logits | cls])."""
    b, n, h = "yolo_body.back_bone_network.", "yolo_body.neck_network.", "yolo_body.head."
    csp = f"{b}ERBlock_2.1.cv3.block.conv.weight" in sd
    csp_act = F.relu if _v6_is_rep(sd, b + "stem") else silu

    def stage(sd, p, x):
        return _v6_bepc3(sd, p, x, csp_act) if csp else _v6_repblock(sd, p, x)

    def fusion(sd, p, xs):
        return _v6_bifusion(sd, p.replace("Bi_fusion", "bi_fusion") if csp else p, xs)
    x = _v6_basic(sd, b + "stem", images, 2)
    feats = []
    for i in range(2, 7):
        x = stage(sd, f"{b}ERBlock_{i}.1", _v6_basic(sd, f"{b}ERBlock_{i}.0", x, 2))
        if i == 6 and csp:
            sp, a = (f"{b}ERBlock_6.2.spp_f", F.relu) if f"{b}ERBlock_6.2.spp_f.cv1.block.conv.weight" in sd else \
                (f"{b}ERBlock_6.2.sppf", silu)
            x1 = _v6_w(sd, sp + ".cv1", x, 1, a)
            y1 = F.max_pool2d(x1, 5, 1, 2)
            y2 = F.max_pool2d(y1, 5, 1, 2)
            x = _v6_w(sd, sp + ".cv2", torch.cat((x1, y1, y2, F.max_pool2d(y2, 5, 1, 2)), 1), 1, a)
        elif i == 6:
            sp = f"{b}ERBlock_6.2.cspsppf"
            x1 = _v6_w(sd, sp + ".cv4", _v6_w(sd, sp + ".cv3", _v6_w(sd, sp + ".cv1", x)))
            y0 = _v6_w(sd, sp + ".cv2", x)
            y1 = F.max_pool2d(x1, 5, 1, 2)
            y2 = F.max_pool2d(y1, 5, 1, 2)
            y3 = _v6_w(sd, sp + ".cv6", _v6_w(sd, sp + ".cv5", torch.cat((x1, y1, y2, F.max_pool2d(y2, 5, 1, 2)), 1)))
            x = _v6_w(sd, sp + ".cv7", torch.cat((y0, y3), 1))
        feats.append(x)
    x4, x3, x2, x1, x0 = feats
    fpn0 = _v6_w(sd, n + "reduce_layer0", x0)
    f0 = stage(sd, n + "Rep_p5", fusion(sd, n + "Bi_fusion_0", [fpn0, x1, x2]))
    fpn1 = _v6_w(sd, n + "reduce_layer1", f0)
    f1 = stage(sd, n + "Rep_p4", fusion(sd, n + "Bi_fusion_1", [fpn1, x2, x3]))
    fpn2 = _v6_w(sd, n + "reduce_layer2", f1)
    pan3 = stage(sd, n + "Rep_p3", fusion(sd, n + "Bi_fusion_2", [fpn2, x3, x4]))
    pan2 = stage(sd, n + "Rep_n4", torch.cat((_v6_w(sd, n + "down_sample_2", pan3, 2), fpn2), 1))
    pan1 = stage(sd, n + "Rep_n5", torch.cat((_v6_w(sd, n + "down_sample_1", pan2, 2), fpn1), 1))
    pan0 = stage(sd, n + "Rep_n6", torch.cat((_v6_w(sd, n + "down_sample_0", pan1, 2), fpn0), 1))
    outs = []
    for i, t in enumerate((pan3, pan2, pan1, pan0)):
        if f"{h}stems.{i}.block.conv.weight" not in sd:
            break
        s_ = _v6_w(sd, f"{h}stems.{i}", t, 1, silu)

        def pred(conv_list, pred_list):
            f = _v6_w(sd, f"{h}{conv_list}.{i}", s_, 1, silu)
            return F.conv2d(f, sd[f"{h}{pred_list}.{i}.weight"], sd[f"{h}{pred_list}.{i}.bias"])
        parts = [pred("reg_conv_list", "reg_predict_list")]
        if f"{h}reg_point_predict_list.{i}.weight" in sd:          # the detect head (yolov6/YoloV6Body.py) has no landmarks
            parts.append(pred("reg_point_conv_list", "reg_point_predict_list"))
        parts.append(pred("cls_conv_list", "cls_predict_list"))
        outs.append(torch.cat(parts, 1))
    return outs


# ------------------------------------------------------------------------------------------------
# per-model glue: body graph, decode of its raw heads, NMS flavour of its predict()
# ------------------------------------------------------------------------------------------------
def v6pose_body(sd: SD, images: torch.Tensor) -> List[torch.Tensor]:
    """The v6 pose graph = the face graph; its class conv emits [classes | keypoint confidences]
    (luoyang/yolov6_pose/YoloV6PoseBody.py:1412-1430).  Returned per level as [ltrb | keypoints | confidences | classes],
    the column order of YoloV6Pose._forward (luoyang/yolov6_pose/YoloV6Pose.py:43-63)."""
    outs = []
    for i, r in enumerate(v6face_body(sd, images)):
        P = sd[f"yolo_body.head.reg_point_predict_list.{i}.weight"].shape[0] // 2
        k = sd[f"yolo_body.head.reg_predict_list.{i}.weight"].shape[0] + 2 * P
        nc = r.shape[1] - k - P
        outs.append(torch.cat((r[:, :k], r[:, k + nc:], r[:, k:k + nc]), 1))
    return outs


def model_ref(name: str):
    """dict(body=fn(sd, images)->raws NCHW ascending stride, decode=fn(raws_np, param, predict)->(B,A,C),
    nms=kwargs of oracle.postprocess.non_max_suppression used by the reference's predict())."""
    from oracle import decode_ref, v5_ref
    if name in ("v5s", "v5m"):
        return dict(body=v5_ref.body_forward,
                    decode=lambda raws, p, predict: decode_ref.decode_anchor(raws, p.anchors, p.strides, p.img_size, "v5"),
                    nms=dict(in_fmt="cxcywh"))
    if name in ("v8s", "v8m", "v8n"):
        return dict(body=v8_body,
                    decode=lambda raws, p, predict: decode_ref.decode_dfl(raws, p.strides, 16, p.class_num,
                                                                          "xyxy" if predict else "cxcywh"),
                    nms=dict(in_fmt="xyxy"))
    if name in ("v3", "v4"):
        return dict(body=v3_body if name == "v3" else v4_body,
                    decode=lambda raws, p, predict: decode_ref.decode_anchor(raws, p.anchors, p.strides, p.img_size, "v3"),
                    nms=dict(in_fmt="cxcywh"))
    if name == "v7pose":
        return dict(body=v7pose_body,
                    decode=lambda raws, p, predict: decode_ref.decode_anchor(raws, p.anchors, p.strides, p.img_size, "pose",
                                                                             point_num=p.point_num, clamp_points=predict),
                    nms=dict(in_fmt="cxcywh", point_num=17, point_conf=True))
    if name in ("v7", "v7x"):
        return dict(body=v7pose_body,
                    decode=lambda raws, p, predict: decode_ref.decode_anchor(raws, p.anchors, p.strides, p.img_size, "v5"),
                    nms=dict(in_fmt="cxcywh"))
    if name == "v7face":
        return dict(body=v7pose_body,
                    decode=lambda raws, p, predict: decode_ref.decode_anchor(raws, p.anchors, p.strides, p.img_size, "face",
                                                                             point_num=p.point_num, clamp_points=predict),
                    nms=dict(in_fmt="cxcywh", point_num=5, point_conf=False))
    if name in ("v5face", "v5facem"):
        return dict(body=v5_ref.face_body_forward,
                    decode=lambda raws, p, predict: decode_ref.decode_anchor(raws, p.anchors, p.strides, p.img_size, "face",
                                                                             point_num=p.point_num, clamp_points=predict),
                    nms=dict(in_fmt="cxcywh", point_num=5, point_conf=False))
    if name in ("v5pose", "v5posex"):
        return dict(body=v5_ref.face_body_forward,
                    decode=lambda raws, p, predict: decode_ref.decode_anchor(raws, p.anchors, p.strides, p.img_size, "pose",
                                                                             point_num=p.point_num, clamp_points=predict),
                    nms=dict(in_fmt="cxcywh", point_num=17, point_conf=True))
    if name in ("v6m", "v6l"):
        return dict(body=v6face_body,
                    decode=lambda raws, p, predict: decode_ref.decode_dfl(raws, p.strides, 17, p.class_num,
                                                                          "xyxy" if predict else "cxcywh", unit_conf=True),
                    nms=dict(in_fmt="xyxy"))
    if name == "v6":
        return dict(body=v6face_body,
                    decode=lambda raws, p, predict: decode_ref.decode_ltrb_points(raws, p.strides, 0, p.class_num,
                                                                                  "xyxy" if predict else "cxcywh"),
                    nms=dict(in_fmt="xyxy"))
    if name in ("v6pose", "v6posel"):
        bins = 17 if name == "v6posel" else 1
        return dict(body=v6pose_body,
                    decode=lambda raws, p, predict: decode_ref.decode_ltrb_points(raws, p.strides, p.point_num,
                                                                                  p.class_num + p.point_num,
                                                                                  "xyxy" if predict else "cxcywh", bins),
                    nms=dict(in_fmt="xyxy", point_num=17, point_conf=True))
    if name in ("v6face", "v6facen", "v6facem"):
        bins = 17 if name == "v6facem" else 1
        return dict(body=v6face_body,
                    decode=lambda raws, p, predict: decode_ref.decode_ltrb_points(raws, p.strides, p.point_num, p.class_num,
                                                                                  "xyxy" if predict else "cxcywh", bins),
                    nms=dict(in_fmt="xyxy", point_num=5, point_conf=False))
    raise KeyError(name)
