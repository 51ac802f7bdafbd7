"""TEST INFRASTRUCTURE ONLY -- regenerates tests/golden/*.npz by RUNNING THE REFERENCE.

Run in the build container only (needs /root/reference):  python -m oracle.gen_golden
The reference publishes no tests or fixtures, so every golden vector here is an output of the
reference's own code (luoyang/...) on seeded inputs; torchvision.ops is replaced by the stand-in of
oracle/ref_harness.py (torchvision is not installed, see there).  Fixtures are data only: seeds,
input tensors and the reference's outputs.
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_harness  # noqa: E402

ref_harness.install()

from luoyang.param.Param import Yolo6FaceParam, Yolo3Param, Yolo4Param, Yolo5Param, Yolo7FaceParam, Yolo7PoseParam, Yolo8Param  # noqa: E402
from luoyang.utils import torch_utils as ref_tu  # noqa: E402
from luoyang.yolov3.YoloHead import YoloHead  # noqa: E402
from luoyang.yolov3.YoloV3 import YoloV3  # noqa: E402
from luoyang.yolov4.YoloV4 import YoloV4  # noqa: E402
from luoyang.yolov5.YoloV5 import YoloV5  # noqa: E402
from luoyang.yolov6_face.YoloV6Face import YoloV6Face  # noqa: E402
from luoyang.yolov5.YoloV5Head import YoloV5Head  # noqa: E402
from luoyang.yolov7_face.YoloV7FaceHead import YoloV7FaceHead  # noqa: E402
from luoyang.yolov7_pose.YoloV7Pose import YoloV7Pose  # noqa: E402
from luoyang.yolov7_pose.YoloV7PoseHead import YoloV7PoseHead  # noqa: E402
from luoyang.yolov8.YoloV8 import YoloV8  # noqa: E402
from luoyang.yolov8.YoloV8Head import YoloV8Head  # noqa: E402

from yolo_amd import models as _models  # noqa: E402
from yolo_amd.init import seeded_images, seeded_state_dict  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def _rows(lst, width):
    """List[Optional[Tensor(K,W)]] -> (counts, concatenated rows)."""
    counts = np.array([0 if r is None else int(r.shape[0]) for r in lst], dtype=np.int64)
    rows = [np.asarray(r, dtype=np.float32).reshape(-1, width) for r in lst if r is not None]
    return counts, (np.concatenate(rows, 0) if rows else np.zeros((0, width), np.float32))


def _np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


# ------------------------------------------------------------------------------------------------
def gen_v5s():
    """YOLOv5s end to end at 64x64 (B=2) and 640x640 (B=1, subsampled rows)."""
    torch.set_num_threads(8)
    for size, batch, seed in ((64, 2, 11), (640, 1, 12)):
        p = Yolo5Param(m_type="s")
        p.img_size = size
        p.feature_size = [size // 8, size // 16, size // 32]
        p.darknet_csp_weight_path = None
        model = YoloV5(p).eval()
        sd = seeded_state_dict(model, seed=seed, mode="spread")
        model.load_state_dict(sd, strict=True)
        imgs = torch.from_numpy(seeded_images(batch, size, seed=seed))
        with torch.no_grad():
            out5, out4, out3 = model.body_net(imgs)
            pred = model._forward(imgs)[0]
            td = model.forward({"batch_images": imgs})
            res = model.predict({"batch_images": imgs})["predicts"]
        counts, rows = _rows(res, 7)
        d = dict(seed=seed, size=size, batch=batch, counts=counts, rows=rows,
                 conf_threshold=p.conf_threshold, nms_threshold=p.nms_threshold)
        if size == 64:
            d.update(raw_p3=_np(out3), raw_p4=_np(out4), raw_p5=_np(out5), pred=_np(pred),
                     head3=_np(td["head3_predicts"]), head4=_np(td["head4_predicts"]), head5=_np(td["head5_predicts"]))
        else:
            sel = np.arange(0, pred.shape[1], 37)
            d.update(pred_rows=sel, pred=_np(pred)[:, sel], raw_p5=_np(out5))
        np.savez_compressed(os.path.join(GOLD, f"v5s_{size}.npz"), **d)
        print("v5s", size, "kept", counts.tolist())
        if size == 64:
            # fused checkpoint of the same model: state_dict keys/values after the reference's fuse()
            model.fuse()
            fs = model.state_dict()
            keys = ["body_net.backbone.stem.conv.conv.weight", "body_net.backbone.stem.conv.conv.bias",
                    "body_net.conv3_for_up_sample1.m.0.cv2.conv.weight", "body_net.conv3_for_up_sample1.m.0.cv2.conv.bias"]
            with torch.no_grad():
                pred_f = model._forward(imgs)[0]
            np.savez_compressed(os.path.join(GOLD, "v5s_64_fused.npz"), n_keys=len(fs), pred=_np(pred_f),
                                **{k.replace(".", "__"): _np(fs[k])[:8] for k in keys})


# ------------------------------------------------------------------------------------------------
def _stub(cls, param, heads, out_size, body):
    m = cls.__new__(cls)
    nn.Module.__init__(m)
    m.param = param
    m.body_output_size = out_size
    m.yolo_head3, m.yolo_head4, m.yolo_head5 = heads
    m.body_net = body
    return m


def gen_decode():
    """Head decode + level merge of every head kind on N(0,2) logits, feature sizes 8/4/2 (img 64)."""
    rng = np.random.RandomState(5)
    size, fs, B = 64, [8, 4, 2], 2
    out = {}

    def raws(C):
        r = [torch.from_numpy((rng.standard_normal((B, 3 * C, f, f)) * 2.0).astype(np.float32)) for f in fs]
        return r   # [P3, P4, P5]

    # v5
    p = Yolo5Param(m_type="s"); p.img_size = size; p.feature_size = fs
    C = 25
    heads = [YoloV5Head(stride=s, yolo_output_size=C, input_size=(f, f), anchors=p.anchors[3 * i:3 * i + 3])
             for i, (s, f) in enumerate(zip(p.strides, fs))]
    r = raws(C)
    m = _stub(YoloV5, p, heads, C, lambda images: (r[2].clone(), r[1].clone(), r[0].clone()))
    with torch.no_grad():
        out["v5_pred"] = _np(m._forward(None)[0])
        out["v5_head3"] = _np(heads[0](r[0].clone()))
    for i in range(3):
        out[f"v5_raw{i}"] = _np(r[i])
    # v3
    p3 = Yolo3Param(); p3.img_size = size; p3.feature_size = fs
    heads = [YoloHead(stride=s, yolo_output_size=C, input_size=(f, f), anchors=p3.anchors[3 * i:3 * i + 3])
             for i, (s, f) in enumerate(zip(p3.strides, fs))]
    r3 = raws(C)
    m = _stub(YoloV3, p3, heads, C, lambda images: (r3[2].clone(), r3[1].clone(), r3[0].clone()))
    with torch.no_grad():
        out["v3_pred"] = _np(m._forward(None)[0])
    for i in range(3):
        out[f"v3_raw{i}"] = _np(r3[i])
    # pose (17 points, 1 class): _forward (no point clamp) and predict's merged tensor (points clamped)
    pp = Yolo7PoseParam(m_type="l"); pp.img_size = size; pp.feature_size = fs
    Cp = 5 + 3 * 17 + 1
    heads = [YoloV7PoseHead(stride=s, point_num=17, yolo_output_size=Cp, input_size=(f, f),
                            anchors=pp.anchors[3 * i:3 * i + 3]) for i, (s, f) in enumerate(zip(pp.strides, fs))]
    rp = raws(Cp)
    m = _stub(YoloV7Pose, pp, heads, Cp, lambda images: (rp[2].clone(), rp[1].clone(), rp[0].clone()))
    with torch.no_grad():
        out["pose_pred"] = _np(m._forward(None)[0])
        res = m.predict({"batch_images": None})["predicts"]
    out["pose_counts"], out["pose_rows"] = _rows(res, 6 + 3 * 17 + 1)
    out["pose_conf"], out["pose_nms"] = pp.conf_threshold, pp.nms_threshold
    for i in range(3):
        out[f"pose_raw{i}"] = _np(rp[i])
    # face (5 points, 1 class): head modules + the merge of the face models (same as pose: box and points x stride)
    pf = Yolo7FaceParam(m_type="l"); pf.img_size = size; pf.feature_size = fs
    Cf = 5 + 2 * 5 + 1
    heads = [YoloV7FaceHead(stride=s, point_num=5, yolo_output_size=Cf, input_size=(f, f),
                            anchors=pf.anchors[3 * i:3 * i + 3]) for i, (s, f) in enumerate(zip(pf.strides, fs))]
    rf = raws(Cf)
    with torch.no_grad():
        lv = [h(x.clone()) for h, x in zip(heads, rf)]
    for i in range(3):
        out[f"face_raw{i}"] = _np(rf[i])
        out[f"face_head{i}"] = _np(lv[i])
    # v8 DFL: _forward (cxcywh) and predict (xyxy + NMS)
    p8 = Yolo8Param(m_type="s"); p8.img_size = size; p8.feature_size = fs
    nc = 20
    m8 = YoloV8.__new__(YoloV8)
    nn.Module.__init__(m8)
    m8.param = p8
    m8.head = YoloV8Head(reg_max=16, stride=p8.strides, feature_size=fs, num_classes=nc, use_dfl=True)
    r8 = [torch.from_numpy((rng.standard_normal((B, 64 + nc, f, f)) * 2.0).astype(np.float32)) for f in fs]
    m8.body_net = lambda images: [x.clone() for x in r8]
    with torch.no_grad():
        out["v8_pred_cxcywh"] = _np(m8._forward(None)[0])
        td = m8._feed_forward({"batch_images": None})
        boxes = td["predict_scaled_bounding_boxes"] * td["stride_tensor"]
        conf, _ = torch.max(td["predict_scores"], dim=-1, keepdim=True)
        out["v8_pred_xyxy"] = _np(torch.cat([boxes, conf, td["predict_scores"]], dim=-1))
        res = m8.predict({"batch_images": None})["predicts"]
    out["v8_counts"], out["v8_rows"] = _rows(res, 7)
    for i in range(3):
        out[f"v8_raw{i}"] = _np(r8[i])
    np.savez_compressed(os.path.join(GOLD, "decode.npz"), **out)
    print("decode goldens:", {k: v.shape for k, v in out.items() if hasattr(v, "shape") and "pred" in k})


# ------------------------------------------------------------------------------------------------
from oracle.gen_golden_synth import synthetic_pred as _synthetic_pred  # noqa: E402


def gen_nms():
    rng = np.random.RandomState(7)
    out = {}
    cases = []

    def run(name, pred, img, fmt="cxcywh", conf=0.1, nms=0.3, P=0, point_conf=False):
        t = torch.from_numpy(pred)
        if P == 0:
            res = ref_tu.non_max_suppression(t, img_size=img, in_fmt=fmt, conf_threshold=conf, nms_threshold=nms)
            width = 7
        elif not point_conf:
            res = ref_tu.non_max_suppression_with_point(t, img_size=img, point_num=P, in_fmt=fmt,
                                                        conf_threshold=conf, nms_threshold=nms)
            width = 6 + 2 * P + 1
        else:
            res = ref_tu.non_max_suppression_with_point_conf(t, img_size=img, point_num=P, in_fmt=fmt,
                                                             conf_threshold=conf, nms_threshold=nms)
            width = 6 + 3 * P + 1
        counts, rows = _rows(res, width)
        out[f"{name}__pred"] = pred
        out[f"{name}__counts"] = counts
        out[f"{name}__rows"] = rows
        out[f"{name}__meta"] = np.array([img, 1.0 if fmt == "cxcywh" else 0.0, conf, nms, P, 1.0 if point_conf else 0.0],
                                        dtype=np.float64)
        cases.append(name)
        print("nms case", name, pred.shape, "kept", counts.tolist())

    # clustered, 20 classes (class ids up to 19 -> offsets up to 77824: coordinates quantised to 2^-7)
    run("clustered20", _synthetic_pred(rng, 3, 300, 20, 640.0), 640.0)
    # single class, xyxy input, higher thresholds
    run("single_xyxy", _synthetic_pred(rng, 2, 200, 1, 416.0, fmt="xyxy"), 416.0, fmt="xyxy", conf=0.3, nms=0.5)
    # the all-pass regime of a default-initialised detector: grid of anchor-sized boxes, near-equal scores
    g = 16
    ys, xs = np.meshgrid(np.arange(g), np.arange(g), indexing="ij")
    boxes = []
    for (aw, ah) in ((10, 13), (16, 30), (33, 23)):
        boxes.append(np.stack((xs.ravel() * 8 + 4.0, ys.ravel() * 8 + 4.0, np.full(g * g, aw), np.full(g * g, ah)), -1))
    box = np.concatenate(boxes, 0)[None]
    A = box.shape[1]
    obj = np.full((1, A, 1), 0.5) + rng.uniform(-1e-3, 1e-3, (1, A, 1))
    cls = np.full((1, A, 20), 0.5) + rng.uniform(-1e-3, 1e-3, (1, A, 20))
    run("allpass_grid", np.concatenate((box, obj, cls), -1).astype(np.float32), 128.0)
    # exact score ties (stable order decides) incl. duplicated boxes
    pt = _synthetic_pred(rng, 1, 96, 3, 320.0)
    pt[0, :, 4] = 0.5
    pt[0, :, 5:] = np.where(pt[0, :, 5:] > 0.5, 0.75, 0.25)
    pt[0, 48:] = pt[0, :48]
    run("ties", pt, 320.0)
    # IoU exactly at the threshold: (0,0,4,4)&(0,0,4,2) -> 0.5 with thr 0.5 (strict '>' keeps both);
    # (0,0,2,2)&(1,0,3,2) -> float 1/3 compared with double 1/3 (float is larger -> suppressed)
    def two(b1, b2, s1=0.9, s2=0.8):
        p = np.zeros((1, 2, 6), np.float32)
        p[0, 0, :4], p[0, 1, :4] = b1, b2
        p[0, :, 4] = (s1, s2)
        p[0, :, 5] = 1.0
        return p
    run("iou_eq_half", two((0, 0, 4, 4), (0, 0, 4, 2)), 64.0, fmt="xyxy", conf=0.1, nms=0.5)
    run("iou_third", two((0, 0, 2, 2), (1, 0, 3, 2)), 64.0, fmt="xyxy", conf=0.1, nms=1.0 / 3.0)
    # zero-area boxes (0/0 = NaN never suppresses) + identical boxes + boxes clamped at the border
    pz = np.zeros((1, 8, 7), np.float32)
    pz[0, :, :4] = [(5, 5, 5, 5), (5, 5, 5, 5), (5, 5, 9, 5), (10, 10, 20, 20), (10, 10, 20, 20), (-30, -30, 8, 8),
                    (60, 60, 90, 90), (63, 63, 200, 200)]
    pz[0, :, 4] = np.linspace(0.9, 0.5, 8)
    pz[0, :, 5] = 0.9
    pz[0, :, 6] = 0.1
    run("degenerate", pz, 64.0, fmt="xyxy", conf=0.1, nms=0.3)
    # images without any candidate -> None
    pn = _synthetic_pred(rng, 3, 64, 4, 320.0)
    pn[0, :, 4] = 1e-3
    pn[2, :, 4] = 1e-3
    run("none_images", pn, 320.0, conf=0.25)
    # landmark / keypoint variants
    run("face5", _synthetic_pred(rng, 2, 150, 1, 640.0, n_extra=10), 640.0, conf=0.3, P=5)
    run("pose17", _synthetic_pred(rng, 2, 150, 1, 640.0, n_extra=51), 640.0, conf=0.3, P=17, point_conf=True)
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(GOLD, "nms.npz"), **out)


# ------------------------------------------------------------------------------------------------
def _ref_models():
    """name -> (reference class, param factory, raw head order of body_net, row width, small size)"""
    return {
        "v8s": (YoloV8, lambda: Yolo8Param(m_type="s"), "asc", 7, 64),
        "v8m": (YoloV8, lambda: Yolo8Param(m_type="m"), "asc", 7, 64),
        "v8n": (YoloV8, lambda: Yolo8Param(m_type="n"), "asc", 7, 64),
        "v5m": (_lazy("luoyang.yolov5.YoloV5", "YoloV5"), lambda: _lazy("luoyang.param.Param", "Yolo5Param")(m_type="m"), "desc", 7, 64),
        "v3": (YoloV3, lambda: Yolo3Param(), "desc", 7, 64),
        "v4": (YoloV4, lambda: Yolo4Param(), "desc", 7, 64),
        "v7pose": (YoloV7Pose, lambda: Yolo7PoseParam(m_type="l"), "desc", 6 + 3 * 17 + 1, 64),
        "v6face": (YoloV6Face, lambda: Yolo6FaceParam(m_type="s"), "v6", 6 + 2 * 5 + 1, 128),
        "v6facen": (YoloV6Face, lambda: Yolo6FaceParam(m_type="n"), "v6", 6 + 2 * 5 + 1, 128),
        "v6": (_lazy("luoyang.yolov6.YoloV6", "YoloV6"), lambda: _lazy("luoyang.param.Param", "Yolo6Param")(m_type="s"), "v6", 7, 128),
        "v6m": (_lazy("luoyang.yolov6.YoloV6", "YoloV6"), lambda: _lazy("luoyang.param.Param", "Yolo6Param")(m_type="m"), "v6", 7, 128),
        "v6l": (_lazy("luoyang.yolov6.YoloV6", "YoloV6"), lambda: _lazy("luoyang.param.Param", "Yolo6Param")(m_type="l"), "v6", 7, 128),
        "v6facem": (YoloV6Face, lambda: Yolo6FaceParam(m_type="m"), "v6", 6 + 2 * 5 + 1, 128),
        "v6posel": (_lazy("luoyang.yolov6_pose.YoloV6Pose", "YoloV6Pose"), lambda: _lazy("luoyang.param.Param", "Yolo6PoseParam")(m_type="l"),
                    "v6", 6 + 3 * 17 + 1, 128),
        "v6pose": (_lazy("luoyang.yolov6_pose.YoloV6Pose", "YoloV6Pose"), lambda: _lazy("luoyang.param.Param", "Yolo6PoseParam")(m_type="s"),
                   "v6", 6 + 3 * 17 + 1, 128),
        "v7": (_lazy("luoyang.yolov7.YoloV7", "YoloV7"), lambda: _lazy("luoyang.param.Param", "Yolo7Param")(m_type="l"), "desc", 7, 64),
        "v7x": (_lazy("luoyang.yolov7.YoloV7", "YoloV7"), lambda: _lazy("luoyang.param.Param", "Yolo7Param")(m_type="x"), "desc", 7, 64),
        "v7face": (_lazy("luoyang.yolov7_face.YoloV7Face", "YoloV7Face"), lambda: _lazy("luoyang.param.Param", "Yolo7FaceParam")(m_type="l"),
                   "desc", 6 + 2 * 5 + 1, 64),
        "v5face": (_lazy("luoyang.yolov5_face.YoloV5Face", "YoloV5Face"), lambda: _lazy("luoyang.param.Param", "Yolo5FaceParam")(m_type="s"),
                   "desc", 6 + 2 * 5 + 1, 64),
        "v5facem": (_lazy("luoyang.yolov5_face.YoloV5Face", "YoloV5Face"), lambda: _lazy("luoyang.param.Param", "Yolo5FaceParam")(m_type="m"),
                    "desc", 6 + 2 * 5 + 1, 64),
        "v5posex": (_lazy("luoyang.yolov5_pose.YoloV5Pose", "YoloV5Pose"), lambda: _lazy("luoyang.param.Param", "Yolo5PoseParam")(m_type="x"),
                    "desc", 6 + 3 * 17 + 1, 64),
        "v5pose": (_lazy("luoyang.yolov5_pose.YoloV5Pose", "YoloV5Pose"), lambda: _lazy("luoyang.param.Param", "Yolo5PoseParam")(m_type="s"),
                   "desc", 6 + 3 * 17 + 1, 64),
    }


def _lazy(module, attr):
    import importlib
    return getattr(importlib.import_module(module), attr)


def _v6_raw_heads(model, imgs):
    """The per-level logits of the reference's YoloV6FaceHead (its own sub-modules, called in the order of
    YoloV6FaceBody.py:1403-1418), concatenated as [ltrb | landmarks | cls]."""
    hd = model.yolo_body.head
    outs = []
    for i, f in enumerate(model.yolo_body(imgs)):
        s = hd.stems[i](f)
        reg = hd.reg_predict_list[i](hd.reg_conv_list[i](s))
        cls = hd.cls_predict_list[i](hd.cls_conv_list[i](s))
        if hasattr(hd, "reg_point_predict_list"):
            pts = hd.reg_point_predict_list[i](hd.reg_point_conv_list[i](s))
            nc = hd.num_classes
            if cls.shape[1] > nc:          # pose: the class conv also emits the keypoint confidences, stored first here
                cls = torch.cat((cls[:, nc:], cls[:, :nc]), 1)
            outs.append(torch.cat((reg, pts, cls), 1))
        else:
            outs.append(torch.cat((reg, cls), 1))
    return outs


def gen_model(name):
    """One model end to end with the reference: small size (B=2, everything stored) and full size (B=1, a
    subsample of the decoded rows + the stride-32 raw head)."""
    torch.set_num_threads(8)
    cls, mk, order, width, small = _ref_models()[name]
    full = mk().img_size if name not in ("v3", "v4") else 608
    for size, batch, seed in ((small, 2, 31), (full, 1, 32)):
        p = mk()
        p.img_size = size
        p.feature_size = [size // s for s in p.strides]
        for attr in ("darknet_csp_weight_path", "darknet53_weight_path", "pretrain_path"):
            if hasattr(p, attr):
                setattr(p, attr, None)
        model = cls(p).eval()
        sd = seeded_state_dict(model, seed=seed, mode="spread", **_models.init_options(name))
        model.load_state_dict(sd, strict=True)
        imgs = torch.from_numpy(seeded_images(batch, size, seed=seed))
        with torch.no_grad():
            if order == "v6":
                raws = _v6_raw_heads(model, imgs)
            else:
                raws = list(model.body_net(imgs))
            if order == "desc":
                raws = raws[::-1]
            pred = model._forward(imgs)[0]
            try:
                res = model.predict({"batch_images": imgs})["predicts"]
            except AttributeError as e:
                # luoyang/yolov7_face/YoloV7Face.py defines `_forward` twice (:58, :71) and no `_feed_forward`, so its
                # predict() (:130-166) cannot run.  Its remaining lines are applied by hand to the `_forward` tensor
                # (boxes already scaled and clamped there): landmark clamp :153-156, then the reference's own
                # torch_utils.non_max_suppression_with_point :157-161.
                print(name, "reference predict() unavailable:", repr(e))
                import luoyang.utils.torch_utils as ref_tu
                q = pred.clone()
                k1 = 5 + 2 * p.point_num
                q[..., 5:k1] = torch.clamp(q[..., 5:k1], min=0, max=p.img_size - 1)
                res = [None if r is None else r.numpy()
                       for r in ref_tu.non_max_suppression_with_point(prediction=q, point_num=p.point_num,
                                                                      conf_threshold=p.conf_threshold,
                                                                      nms_threshold=p.nms_threshold, img_size=p.img_size)]
        counts, rows = _rows(res, width)
        d = dict(seed=seed, size=size, batch=batch, counts=counts, rows=rows, n_keys=len(sd),
                 conf_threshold=p.conf_threshold, nms_threshold=p.nms_threshold)
        if size == small:
            d.update(pred=_np(pred), **{f"raw{i}": _np(r) for i, r in enumerate(raws)})
            try:                       # the reference's own fuse(): checkpoint layout + outputs after it
                model.fuse()
                fs = model.state_dict()
                with torch.no_grad():
                    d.update(fused_n_keys=len(fs), fused_pred=_np(model._forward(imgs)[0]),
                             fused_bias_keys=np.array(sorted(k for k in fs if k.endswith(".bias"))[:6]))
            except (NotImplementedError, TypeError) as e:
                print(name, "reference fuse() unavailable:", repr(e))
        else:
            sel = np.arange(0, pred.shape[1], 37)
            d.update(pred_rows=sel, pred=_np(pred)[:, sel], raw_last=_np(raws[-1]))
        np.savez_compressed(os.path.join(GOLD, f"{name}_{'small' if size == small else 'full'}.npz"), **d)
        print(name, size, "keys", len(sd), "kept", counts.tolist(), "pred", tuple(pred.shape))


def _metric_inputs(rng, B, img, nc, n_gt, n_fp):
    """ground truths (normalised cx,cy,w,h,cls, zero-padded to a common length) and NMS-style prediction rows around them"""
    preds, targets = [], []
    for b in range(B):
        g = int(rng.randint(0, n_gt + 1))
        wh = rng.uniform(0.05, 0.4, (g, 2))
        c = rng.uniform(0.2, 0.8, (g, 2))
        cls = rng.randint(0, nc, (g, 1)).astype(np.float64)
        t = np.concatenate((c, wh, cls), 1).astype(np.float32)
        rows = []
        for k in range(g):
            for _ in range(int(rng.randint(0, 4))):                     # 0-3 detections per object, jittered
                j = rng.normal(0, 0.02, 4)
                x, y, w, h = (t[k, 0] + j[0]) * img, (t[k, 1] + j[1]) * img, (t[k, 2] + j[2]) * img, (t[k, 3] + j[3]) * img
                lab = t[k, 4] if rng.uniform() < 0.85 else float(rng.randint(0, nc))
                rows.append([x - w / 2, y - h / 2, x + w / 2, y + h / 2, rng.uniform(0.3, 1), rng.uniform(0.3, 1), lab])
        for _ in range(int(rng.randint(0, n_fp + 1))):
            x1, y1 = rng.uniform(0, img * 0.7, 2)
            rows.append([x1, y1, x1 + rng.uniform(10, 150), y1 + rng.uniform(10, 150), rng.uniform(0.1, 0.6), rng.uniform(0.1, 0.9),
                         float(rng.randint(0, nc))])
        if rows and b % 5 == 0 and g:                                   # exact duplicates: IoU ties go to the first row
            rows.append(list(rows[0]))
        p = np.array(rows, np.float32).reshape(-1, 7)
        if len(rows):
            p = p[np.argsort(-(p[:, 4] * p[:, 5]), kind="stable")]
        preds.append(p if len(rows) else None)
        pad = np.zeros((n_gt, 5), np.float32)
        pad[:g] = t
        targets.append(pad if b % 7 != 3 else None)                      # some images carry no target tensor at all
    return preds, targets


def gen_metric():
    """DetectMetric of the reference (metric_feed_forward over two batches, then summary) on synthetic detections."""
    import importlib
    DM = importlib.import_module("luoyang.model.metric.DetectMetric").DetectMetric
    rng = np.random.RandomState(23)
    out = {}
    cases = [("voc20", 16, 640, 20, 12, 10, 0.5), ("few_classes", 8, 416, 3, 30, 40, 0.5), ("strict_iou", 8, 640, 5, 8, 5, 0.75)]
    for name, B, img, nc, n_gt, n_fp, thr in cases:
        names = [f"c{i}" for i in range(nc)]
        m = DM(image_size=(img, img), class_names=names, predict_name="predicts", target_name="batch_targets", ap_iou_threshold=thr)
        summary = {}
        tps = []
        for batch in range(2):
            preds, targets = _metric_inputs(rng, B, img, nc, n_gt, n_fp)
            for b in range(B):
                out[f"{name}__b{batch}_pred{b}"] = np.zeros((0, 7), np.float32) if preds[b] is None else preds[b]
                out[f"{name}__b{batch}_tgt{b}"] = np.zeros((0, 5), np.float32) if targets[b] is None else targets[b].copy()
            out[f"{name}__b{batch}_pred_none"] = np.array([p is None for p in preds])
            out[f"{name}__b{batch}_tgt_none"] = np.array([t is None for t in targets])
            n0 = len(summary.get("sample_metrics", []))
            m.metric_feed_forward({"predicts": preds, "batch_targets": [None if t is None else torch.from_numpy(t.copy()) for t in targets]},
                                  summary)
            new = summary["sample_metrics"][n0:]
            out[f"{name}__b{batch}_tp"] = np.concatenate([x[0] for x in new]) if new else np.zeros(0)
        res = m.summary(summary)
        out[f"{name}__meta"] = np.array([B, img, nc, thr], np.float64)
        out[f"{name}__mean"] = np.array([res["m_recall"], res["m_precision"], res["m_ap"]], np.float64)
        out[f"{name}__per_class"] = np.array([[res[n]["precision"], res[n]["recall"], res[n]["f1_score"]] for n in names], np.float64)
    out["cases"] = np.array([c[0] for c in cases])
    out["numpy_version"] = np.array(np.__version__)
    np.savez_compressed(os.path.join(GOLD, "metric.npz"), **out)
    print("metric goldens:", [c[0] for c in cases], {k: out[k] for k in out if k.endswith("__mean")})


def gen_preprocess():
    """ResizeImage(is_train=False) + ImageNorm of the reference on random uint8 images (Pillow does the resize), and
    RestoreSize on random boxes.  The reference's transformer module imports cv2/imgaug at the top (absent here), so
    the three classes are loaded from its source with those two imports stubbed out."""
    import importlib.util, types, PIL
    from PIL import Image
    for name in ("cv2", "imgaug", "imgaug.augmenters"):
        sys.modules.setdefault(name, types.ModuleType(name))
    spec = importlib.util.spec_from_file_location("ref_transformer", "/root/reference/luoyang/transformer/transformer.py")
    T = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(T)
    rng = np.random.RandomState(17)
    out = {"pillow_version": np.array(PIL.__version__)}
    cases = [(37, 53, 64, 64), (120, 90, 64, 64), (64, 64, 64, 64), (30, 100, 96, 96), (333, 500, 160, 160), (50, 40, 128, 96)]
    for i, (h, w, th, tw) in enumerate(cases):
        img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        # smooth half of the cases a little so that the fixtures are not pure noise
        if i % 2:
            img = np.array(Image.fromarray(img).resize((w, h), Image.BILINEAR).resize((w // 2 + 1, h // 2 + 1)).resize((w, h)))
        d = {"image": Image.fromarray(img)}
        T.ResizeImage(target_height=th, target_width=tw, is_train=False).transformer(d)
        T.ImageNorm().transformer(d)
        out[f"img{i}"] = img
        out[f"out{i}"] = d["image"]
        out[f"meta{i}"] = np.array([th, tw, d["ori_height"], d["ori_width"], d["resize_w"], d["resize_h"], d["dw"], d["dh"]],
                                   np.int64)
        out[f"ratio{i}"] = np.array(d["resize_ratio"], np.float64)
        pred = np.concatenate((rng.uniform(-5, max(th, tw) + 5, (9, 4)), rng.uniform(0, 1, (9, 2)),
                               rng.randint(0, 20, (9, 1))), 1).astype(np.float32)
        d["predict"] = pred
        T.RestoreSize().transformer(d)
        out[f"pred{i}"] = pred
        out[f"boxes{i}"] = np.asarray(d["predict_boxs"], np.int32)
        out[f"labels{i}"] = np.asarray(d["predict_label"], np.int32)
    out["n"] = np.array(len(cases))
    np.savez_compressed(os.path.join(GOLD, "preprocess.npz"), **out)
    print("preprocess goldens:", len(cases), "cases, Pillow", PIL.__version__)


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    which = sys.argv[1:] or ["nms", "decode", "v5s"]
    if "nms" in which:
        gen_nms()
    if "decode" in which:
        gen_decode()
    if "v5s" in which:
        gen_v5s()
    if "preprocess" in which:
        gen_preprocess()
    if "metric" in which:
        gen_metric()
    for name in _ref_models():
        if name in which:
            gen_model(name)
