"""CPU restatement of the reference's detection metric (test infrastructure; the product path never imports it).

Follows luoyang/model/metric/DetectMetric.py: `_get_batch_statistics` (:35-105, ground truths claim predictions in their
stored order), `_metric_per_class` (:129-172), `_metric_` (:174-199), `summary` (:201-242), with
luoyang/utils/anchor_utils.py `bbox_iou` (:25-44, the +1 convention, float32) and luoyang/utils/metric_utils.py
`compute_ap` (:4-20, area under the monotone precision envelope).
"""
import numpy as np


def iou_plus1(boxes: np.ndarray, gt: np.ndarray) -> np.ndarray:
    lt = np.maximum(boxes[:, :2], gt[None, :2])
    rb = np.minimum(boxes[:, 2:4], gt[None, 2:4])
    wh = np.maximum(rb - lt + 1, 0)
    inter = wh[:, 0] * wh[:, 1]
    pa = (boxes[:, 2] - boxes[:, 0] + 1) * (boxes[:, 3] - boxes[:, 1] + 1)
    ga = (gt[2] - gt[0] + 1) * (gt[3] - gt[1] + 1)
    return inter / (pa + ga - inter)


def match_image(pred, target, image_size, iou_threshold):
    """pred (K,W) float32 rows; target (G,5) float32 normalised cx,cy,w,h,cls or None -> (tp float64 (K,), labels or None)"""
    tp = np.zeros(pred.shape[0])
    if target is None:
        return tp, None
    t = np.array(target, dtype=np.float32, copy=True)
    t = t[np.sum(t, axis=-1) != 0]
    t[:, 0] *= image_size[0]
    t[:, 2] *= image_size[0]
    t[:, 1] *= image_size[1]
    t[:, 3] *= image_size[1]
    gt_boxes = np.concatenate((t[:, :2] - 0.5 * t[:, 2:4], t[:, :2] + 0.5 * t[:, 2:4]), -1)
    labels = pred[:, -1]
    taken = np.zeros(pred.shape[0], bool)
    for g in range(t.shape[0]):
        cand = np.nonzero((labels == t[g, 4]) & ~taken)[0]
        if cand.size == 0:
            continue
        iou = iou_plus1(pred[cand, :4], gt_boxes[g])
        k = int(np.argmax(iou))
        if iou[k] >= iou_threshold:
            taken[cand[k]] = True
            tp[cand[k]] = 1
    return tp, t[:, 4]


def batch_statistics(predict, targets, image_size, iou_threshold):
    metrics, labels = [], []
    for p, t in zip(predict, targets):
        if p is None:
            continue
        tp, lab = match_image(p, t, image_size, iou_threshold)
        metrics.append((tp, p[:, -1], p[:, 4] * p[:, 5]))
        if lab is not None:
            labels.append(lab)
    return metrics, labels


def average_precision(recall, precision):
    r = np.concatenate(([0.0], recall, [1.0]))
    q = np.concatenate(([0.0], precision, [0.0]))
    q = np.maximum.accumulate(q[::-1])[::-1]
    i = np.nonzero(r[1:] != r[:-1])[0]
    return np.sum((r[i + 1] - r[i]) * q[i + 1])


def summarise(sample_metrics, target_batches, class_names):
    if len(sample_metrics) == 0:
        out = {"m_recall": 0, "m_precision": 0, "m_ap": 0}
        for n in class_names:
            out[n] = {"precision": 0, "recall": 0, "f1_score": 0}
        return out
    tp, lab, conf = [np.concatenate(x, 0) for x in zip(*sample_metrics)]
    target_cls = np.array([int(v) for batch in target_batches for sample in batch for v in sample])
    ps, rs = [], []
    for c in range(len(class_names)):
        n_gt = (target_cls == c).astype(np.float32).sum()
        n_pr = (lab == c).astype(np.float32).sum()
        if n_pr == 0 or n_gt == 0:
            ps.append(0)
            rs.append(0)
            continue
        ctp = tp[lab == c]
        tpc, fpc = ctp.sum(), (1 - ctp).sum()
        rs.append(tpc / (n_gt + 1e-16))
        ps.append(tpc / (tpc + fpc))
    ps, rs = np.array(ps), np.array(rs)
    f1 = 2 * ps * rs / (ps + rs + 1e-16)
    order = conf.argsort()[::-1]
    stp = tp[order]
    tpc, fpc = stp.cumsum(), (1 - stp).cumsum()
    rc, pc = tpc / len(target_cls), tpc / (tpc + fpc)
    out = {"m_recall": rc[-1], "m_precision": pc[-1], "m_ap": average_precision(rc, pc)}
    for i, n in enumerate(class_names):
        out[n] = {"precision": round(ps[i], 4), "recall": round(rs[i], 4), "f1_score": round(f1[i], 4)}
    return out
