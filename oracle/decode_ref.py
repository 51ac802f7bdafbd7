"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference head decodes.

Every function takes the raw head tensors exactly as the reference's body emits them
(float32, NCHW) and returns the merged ``(B, A, C)`` tensor of ``_forward`` (cxcywh, input
pixels).  float32 arithmetic, same operation order as the reference.

Follows:
  v5 / v7 detect : luoyang/yolov5/YoloV5Head.py:23-60 + luoyang/yolov5/YoloV5.py:60-69
  v3 / v4        : luoyang/yolov3/YoloHead.py:23-65   + luoyang/yolov3/YoloV3.py:129-139
  pose           : luoyang/yolov7_pose/YoloV7PoseHead.py:25-81 + YoloV7Pose.py:143-159
  face           : luoyang/yolov7_face/YoloV7FaceHead.py:25-77
  v8 (DFL)       : luoyang/yolov8/YoloV8Head.py:28-60, YoloV8Body.py:204-223,
                   luoyang/utils/torch_utils.py:373-394, luoyang/yolov8/YoloV8.py:48-57,118-124
"""
from typing import List, Sequence, Tuple

import numpy as np

F32 = np.float32


def _sigmoid(x: np.ndarray) -> np.ndarray:
    x = x.astype(F32)
    return (F32(1.0) / (F32(1.0) + np.exp(-x, dtype=F32))).astype(F32)


def decode_anchor_level(raw: np.ndarray, anchors: Sequence[Tuple[int, int]], stride: int, kind: str,
                        point_num: int = 0) -> np.ndarray:
    """One level: raw (B, na*C, H, W) -> (B, na, H, W, C) in GRID units (what the head module returns)."""
    B, ch, H, W = raw.shape
    na = len(anchors)
    C = ch // na
    t = raw.reshape(B, na, C, H, W).transpose(0, 1, 3, 4, 2).astype(F32).copy()
    gx = np.arange(W, dtype=F32).reshape(1, 1, 1, W)
    gy = np.arange(H, dtype=F32).reshape(1, 1, H, 1)
    anc = (np.asarray(anchors, dtype=np.float64) / stride).astype(F32)   # int64 / int -> float32 in torch
    aw = anc[:, 0].reshape(1, na, 1, 1)
    ah = anc[:, 1].reshape(1, na, 1, 1)
    out = np.empty_like(t)
    P = point_num
    if kind == "v3":
        sx, sy = _sigmoid(t[..., 0]), _sigmoid(t[..., 1])
        out[..., 0] = sx + gx
        out[..., 1] = sy + gy
        out[..., 2] = np.exp(t[..., 2], dtype=F32) * aw
        out[..., 3] = np.exp(t[..., 3], dtype=F32) * ah
        out[..., 4:] = _sigmoid(t[..., 4:])
        return out
    if kind == "v5":
        s = _sigmoid(t)
        tail = s[..., 4:]
    elif kind == "pose":
        s = t.copy()
        s[..., :5] = _sigmoid(t[..., :5])
        s[..., 5 + 2 * P:] = _sigmoid(t[..., 5 + 2 * P:])       # kpt conf and classes
        pts = s[..., 5:5 + 2 * P]
        pts[..., 0::2] = (F32(2.0) * pts[..., 0::2] - F32(0.5)) + gx[..., None]
        pts[..., 1::2] = (F32(2.0) * pts[..., 1::2] - F32(0.5)) + gy[..., None]
        tail = s[..., 4:]
    elif kind == "face":
        s = t.copy()
        s[..., :5] = _sigmoid(t[..., :5])
        s[..., 5 + 2 * P:] = _sigmoid(t[..., 5 + 2 * P:])
        pts = s[..., 5:5 + 2 * P]
        pts[..., 0::2] = pts[..., 0::2] * aw[..., None] + gx[..., None]
        pts[..., 1::2] = pts[..., 1::2] * ah[..., None] + gy[..., None]
        tail = s[..., 4:]
    else:
        raise ValueError(kind)
    out[..., 0] = F32(2.0) * s[..., 0] - F32(0.5) + gx
    out[..., 1] = F32(2.0) * s[..., 1] - F32(0.5) + gy
    w2 = F32(2.0) * s[..., 2]
    h2 = F32(2.0) * s[..., 3]
    out[..., 2] = (w2 * w2) * aw
    out[..., 3] = (h2 * h2) * ah
    out[..., 4:] = tail
    return out


def merge_anchor_levels(levels: List[np.ndarray], strides: Sequence[int], img_size: int, kind: str,
                        point_num: int = 0, clamp_points: bool = False) -> np.ndarray:
    """Level outputs (grid units) -> (B, A, C): box (and point) columns x stride, flatten
    anchor-major, concat [s8, s16, s32], clamp cx,cy,w,h to [0, img-1]."""
    outs = []
    P = point_num
    for lv, s in zip(levels, strides):
        p = lv.copy()
        p[..., :4] = p[..., :4] * F32(s)
        if kind in ("pose", "face"):
            p[..., 5:5 + 2 * P] = p[..., 5:5 + 2 * P] * F32(s)
        B = p.shape[0]
        outs.append(p.reshape(B, -1, p.shape[-1]))
    o = np.concatenate(outs, axis=1)
    hi = F32(img_size - 1)
    o[..., :4] = np.clip(o[..., :4], F32(0), hi)
    if clamp_points and kind in ("pose", "face"):
        o[..., 5:5 + 2 * P] = np.clip(o[..., 5:5 + 2 * P], F32(0), hi)
    return o


def decode_anchor(raws: List[np.ndarray], anchors: Sequence[Tuple[int, int]], strides: Sequence[int],
                  img_size: int, kind: str, point_num: int = 0, clamp_points: bool = False) -> np.ndarray:
    """raws ordered [s8, s16, s32]; anchors is the flat 9-entry list of the Param object."""
    na = len(anchors) // len(raws)
    lv = [decode_anchor_level(r, anchors[i * na:(i + 1) * na], s, kind, point_num)
          for i, (r, s) in enumerate(zip(raws, strides))]
    return merge_anchor_levels(lv, strides, img_size, kind, point_num, clamp_points)


def decode_dfl(raws: List[np.ndarray], strides: Sequence[int], reg_max: int, num_classes: int,
               out_fmt: str, unit_conf: bool = False) -> np.ndarray:
    """v8: raws[i] (B, 4*reg_max+nc, H, W) -> (B, A, 4+1+nc): [box, max-class conf, class scores];
    box is xyxy (predict form) or cxcywh (_forward form), input pixels.
    v6 m/l (luoyang/yolov6/YoloV6Body.py:1322-1343 and :1345-1380, luoyang/yolov6/YoloV6.py:41-74): the same with
    reg_max = 17 bins and ``unit_conf``: the confidence column is the constant 1."""
    B = raws[0].shape[0]
    flat = np.concatenate([r.reshape(B, r.shape[1], -1) for r in raws], axis=2).astype(F32)
    dist, cls = flat[:, :4 * reg_max], flat[:, 4 * reg_max:]
    cls = cls.transpose(0, 2, 1)
    dist = dist.transpose(0, 2, 1)
    pts, strd = [], []
    for r, s in zip(raws, strides):
        H, W = r.shape[2], r.shape[3]
        sx = np.arange(W, dtype=F32) + F32(0.5)
        sy = np.arange(H, dtype=F32) + F32(0.5)
        yy, xx = np.meshgrid(sy, sx, indexing="ij")
        pts.append(np.stack((xx, yy), -1).reshape(-1, 2))
        strd.append(np.full((H * W, 1), s, dtype=F32))
    pts = np.concatenate(pts, 0)
    strd = np.concatenate(strd, 0)
    A = pts.shape[0]
    d = dist.reshape(B, A, 4, reg_max)
    d = d - d.max(axis=3, keepdims=True)
    e = np.exp(d, dtype=F32)
    sm = e / e.sum(axis=3, keepdims=True, dtype=F32)
    ltrb = (sm * np.arange(reg_max, dtype=F32)).sum(axis=3, dtype=F32)
    x1y1 = pts - ltrb[..., :2]
    x2y2 = pts + ltrb[..., 2:]
    box = np.concatenate((x1y1, x2y2), -1) * strd
    if out_fmt == "cxcywh":
        cxy = (box[..., :2] + box[..., 2:]) / F32(2.0)
        wh = box[..., 2:] - box[..., :2]
        box = np.concatenate((cxy, wh), -1)
    sc = _sigmoid(cls)
    conf = np.ones_like(sc[..., :1]) if unit_conf else sc.max(axis=-1, keepdims=True)
    return np.concatenate((box, conf, sc), -1).astype(F32)


def _dfl_expectation(d: np.ndarray) -> np.ndarray:
    """softmax over the last axis projected on 0..bins-1 (F.softmax(..).matmul(proj), YoloV6FaceBody.py:1363-1372)."""
    d = d - d.max(axis=-1, keepdims=True)
    e = np.exp(d, dtype=F32)
    sm = e / e.sum(axis=-1, keepdims=True, dtype=F32)
    return (sm * np.arange(d.shape[-1], dtype=F32)).sum(axis=-1, dtype=F32)


def decode_ltrb_points(raws: List[np.ndarray], strides: Sequence[int], point_num: int, num_classes: int,
                       out_fmt: str, bins: int = 1) -> np.ndarray:
    """v6 face without DFL (luoyang/yolov6_face/YoloV6FaceBody.py:1354-1397, :1439-1455; assembled by
    luoyang/yolov6_face/YoloV6Face.py:42-58 / :66-74): raws[i] (B, 4+2P+nc, H, W) in the order [ltrb | landmarks | cls]
    -> (B, A, 5+2P+nc) = [box, 1.0, landmarks, sigmoid(cls)]; box xyxy (predict) or cxcywh (_forward), pixels."""
    outs = []
    for raw, s in zip(raws, strides):
        B, ch, H, W = raw.shape
        t = raw.astype(F32).reshape(B, ch, H * W).transpose(0, 2, 1)
        if bins > 1:         # m/l: the four distances are DFL expectations of 4 x bins logits
            t = np.concatenate((_dfl_expectation(t[..., :4 * bins].reshape(B, H * W, 4, bins)), t[..., 4 * bins:]), -1)
        fs = F32(s)
        ys, xs = np.meshgrid(np.arange(H, dtype=F32), np.arange(W, dtype=F32), indexing="ij")
        ax = (((xs.reshape(-1) + F32(0.5)) * fs) / fs).astype(F32)
        ay = (((ys.reshape(-1) + F32(0.5)) * fs) / fs).astype(F32)
        x1, y1 = (ax - t[..., 0]) * fs, (ay - t[..., 1]) * fs
        x2, y2 = (ax + t[..., 2]) * fs, (ay + t[..., 3]) * fs
        o = np.empty((B, H * W, 5 + 2 * point_num + num_classes), F32)
        if out_fmt == "cxcywh":
            o[..., 0], o[..., 1], o[..., 2], o[..., 3] = (x1 + x2) / F32(2), (y1 + y2) / F32(2), x2 - x1, y2 - y1
        else:
            o[..., 0], o[..., 1], o[..., 2], o[..., 3] = x1, y1, x2, y2
        o[..., 4] = 1.0
        for k in range(2 * point_num):
            o[..., 5 + k] = (t[..., 4 + k] + (ay if k & 1 else ax)) * fs
        o[..., 5 + 2 * point_num:] = _sigmoid(t[..., 4 + 2 * point_num:])
        outs.append(o)
    return np.concatenate(outs, 1)
