"""Detection metric with the reference's interface (luoyang/model/metric/DetectMetric.py).

``DetectMetric.metric_feed_forward(tensor_dict, summary)`` accumulates per-prediction (true positive, label, confidence)
triples, ``summary(summary)`` turns them into mean recall / precision / AP and per-class precision / recall / F1.  The
matching of ground truths to predictions (the O(ground truths x predictions) Python loops of :35-105) runs on the
device for the whole batch in one launch (`yamd_detect_match`); the final curves are a few numpy lines on the host.

``tensor_dict[predict_name]`` is what ``TaskLayer.predict`` returns (list of ``None`` | ndarray (K, W)) or, to stay on
the device, the ``(rows, counts)`` pair of ``torch_utils.nms_batched`` / ``TaskModel.predict_device``.
"""
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import lib as _lib

LIB = _lib.LIB


def match_batch(rows: torch.Tensor, counts: torch.Tensor, targets: torch.Tensor, target_counts: Optional[torch.Tensor],
                image_size: Tuple[int, int], iou_threshold: float) -> torch.Tensor:
    """rows (B,M,W) fp32, counts (B,) int32, targets (B,T,5) fp32 normalised cx,cy,w,h,cls -> (B,M) uint8 true-positive flags"""
    _lib.require_gpu(rows, torch.float32, "rows")
    _lib.require_gpu(counts, torch.int32, "counts")
    B, M, W = rows.shape
    T = int(targets.shape[1]) if targets is not None and targets.numel() else 0
    if T:
        _lib.require_gpu(targets, torch.float32, "targets")
        if targets.shape[0] != B or targets.shape[2] != 5:
            raise _lib.YamdError(f"targets must be (B, T, 5), got {tuple(targets.shape)}")
    if target_counts is not None:
        _lib.require_gpu(target_counts, torch.int32, "target_counts")
    tp = torch.empty((B, M), dtype=torch.uint8, device=rows.device)
    _lib.check(LIB.yamd_detect_match(rows.data_ptr(), counts.data_ptr(), B, M, W, targets.data_ptr() if T else None,
                                     target_counts.data_ptr() if (T and target_counts is not None) else None, T,
                                     float(image_size[0]), float(image_size[1]), float(np.float32(iou_threshold)),
                                     tp.data_ptr(), _lib.current_stream()), "detect_match")
    return tp


def _average_precision(recall: np.ndarray, precision: np.ndarray) -> float:
    r = np.concatenate(([0.0], recall, [1.0]))
    q = np.concatenate(([0.0], precision, [0.0]))
    for i in range(q.size - 1, 0, -1):
        q[i - 1] = max(q[i - 1], q[i])
    step = np.where(r[1:] != r[:-1])[0]
    return np.sum((r[step + 1] - r[step]) * q[step + 1])


class DetectMetric(object):

    def __init__(self, image_size: Tuple[int, int], class_names: List[str], predict_name: str, target_name: str,
                 name: str = "detect", conf_threshold: float = 0.5, ap_iou_threshold: float = 0.5, nms_threshold: float = 0.5,
                 device="cuda"):
        self.name = name
        self.image_size = image_size
        self.class_names = class_names
        self.class_ids = list(range(len(class_names)))
        self.predict_name = predict_name
        self.target_name = target_name
        self.conf_threshold = conf_threshold
        self.ap_iou_threshold = ap_iou_threshold
        self.nms_threshold = nms_threshold
        self.device = torch.device(device)

    # -- packing ------------------------------------------------------------------------------------------------
    def _rows(self, predict) -> Tuple[torch.Tensor, torch.Tensor, List[Optional[np.ndarray]]]:
        if isinstance(predict, (tuple,)) and len(predict) == 2 and torch.is_tensor(predict[0]):
            rows, counts = predict
            c = counts.cpu().tolist()
            host = rows.cpu().numpy()
            return rows, counts, [host[b, :k] if k else None for b, k in enumerate(c)]
        host = [None if p is None else np.ascontiguousarray(p, dtype=np.float32) for p in predict]
        W = next((p.shape[1] for p in host if p is not None), 7)
        M = max([p.shape[0] for p in host if p is not None] + [1])
        packed = np.zeros((len(host), M, W), np.float32)
        counts = np.zeros(len(host), np.int32)
        for b, p in enumerate(host):
            if p is not None:
                packed[b, :p.shape[0]] = p
                counts[b] = p.shape[0]
        return torch.from_numpy(packed).to(self.device), torch.from_numpy(counts).to(self.device), host

    def _targets(self, targets: Sequence[Optional[torch.Tensor]]):
        host = [None if t is None else t.detach().cpu().numpy().astype(np.float32).reshape(-1, 5) for t in targets]
        T = max([t.shape[0] for t in host if t is not None] + [0])
        packed = np.zeros((len(host), max(T, 1), 5), np.float32)
        counts = np.zeros(len(host), np.int32)
        for b, t in enumerate(host):
            if t is not None:
                packed[b, :t.shape[0]] = t
                counts[b] = t.shape[0]
        return torch.from_numpy(packed).to(self.device), torch.from_numpy(counts).to(self.device), host, T

    # -- reference interface ----------------------------------------------------------------------------------------
    def metric_feed_forward(self, tensor_dict: Dict[str, Any], summary: Dict[str, Any]):
        rows, counts, host_pred = self._rows(tensor_dict[self.predict_name])
        tg, tcount, host_t, T = self._targets(tensor_dict[self.target_name])
        if T:
            tp = match_batch(rows, counts, tg, tcount, self.image_size, self.ap_iou_threshold).cpu().numpy()
        else:
            tp = np.zeros(rows.shape[:2], np.uint8)
        batch_metrics, labels = [], []
        for b, p in enumerate(host_pred):
            if p is None:
                continue                                  # as in the reference: such an image adds no labels either
            batch_metrics.append((tp[b, :p.shape[0]].astype(np.float64), p[:, -1], p[:, 4] * p[:, 5]))
            t = host_t[b]
            if t is not None:
                labels.append(t[np.sum(t, axis=-1) != 0][:, -1])
        summary["sample_metrics"] = summary.get("sample_metrics", []) + batch_metrics
        summary["target"] = summary.get("target", []) + [labels]

    def summary(self, summary: Dict[str, Any]) -> Dict[str, Any]:
        samples = summary["sample_metrics"]
        if len(samples) == 0:
            out: Dict[str, Any] = {"m_recall": 0, "m_precision": 0, "m_ap": 0}
            for n in self.class_names:
                out[n] = {"precision": 0, "recall": 0, "f1_score": 0}
            return out
        tp = np.concatenate([s[0] for s in samples], 0)
        lab = np.concatenate([s[1] for s in samples], 0)
        conf = np.concatenate([s[2] for s in samples], 0)
        target_cls = np.array([int(v) for batch in summary["target"] for sample in batch for v in sample])
        p, r = [], []
        for c in self.class_ids:
            n_gt = (target_cls == c).astype(np.float32).sum()
            n_pr = (lab == c).astype(np.float32).sum()
            if n_pr == 0 or n_gt == 0:
                p.append(0)
                r.append(0)
                continue
            ctp = tp[lab == c]
            hit, miss = ctp.sum(), (1 - ctp).sum()
            r.append(hit / (n_gt + 1e-16))
            p.append(hit / (hit + miss))
        p, r = np.array(p), np.array(r)
        f1 = 2 * p * r / (p + r + 1e-16)
        order = conf.argsort()[::-1]
        stp = tp[order]
        hit, miss = stp.cumsum(), (1 - stp).cumsum()
        recall_curve, precision_curve = hit / len(target_cls), hit / (hit + miss)
        out = {"m_recall": recall_curve[-1], "m_precision": precision_curve[-1],
               "m_ap": _average_precision(recall_curve, precision_curve)}
        for i, n in enumerate(self.class_names):
            out[n] = {"precision": round(p[i], 4), "recall": round(r[i], 4), "f1_score": round(f1[i], 4)}
        return out
