"""CPU: the detection-metric restatement (oracle/metric_ref.py) and the host half of yolo_amd.metric against fixtures
produced by the reference's DetectMetric (metric_feed_forward over two batches + summary; numpy version in the fixture)."""
import os

import numpy as np
import pytest

from oracle import metric_ref as M

GOLD = os.path.join(os.path.dirname(__file__), "golden", "metric.npz")


def load_case(g, name):
    B, img, nc, thr = g[f"{name}__meta"]
    B, img, nc = int(B), int(img), int(nc)
    batches = []
    for k in range(2):
        pn, tn = g[f"{name}__b{k}_pred_none"], g[f"{name}__b{k}_tgt_none"]
        preds = [None if pn[b] else g[f"{name}__b{k}_pred{b}"] for b in range(B)]
        tgts = [None if tn[b] else g[f"{name}__b{k}_tgt{b}"] for b in range(B)]
        batches.append((preds, tgts, g[f"{name}__b{k}_tp"]))
    return B, img, nc, float(thr), batches


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def test_oracle_matches_reference(gold):
    for name in [str(c) for c in gold["cases"]]:
        B, img, nc, thr, batches = load_case(gold, name)
        samples, targets = [], []
        for preds, tgts, want_tp in batches:
            m, lab = M.batch_statistics(preds, tgts, (img, img), thr)
            got = np.concatenate([x[0] for x in m]) if m else np.zeros(0)
            assert np.array_equal(got, want_tp), name
            samples += m
            targets.append(lab)
        res = M.summarise(samples, targets, [f"c{i}" for i in range(nc)])
        assert np.array_equal(np.array([res["m_recall"], res["m_precision"], res["m_ap"]]), gold[f"{name}__mean"]), name
        per = np.array([[res[f"c{i}"][k] for k in ("precision", "recall", "f1_score")] for i in range(nc)], np.float64)
        assert np.array_equal(per, gold[f"{name}__per_class"]), name


def test_host_summary_matches_reference(gold):
    """summary() of the product class from the reference's true-positive flags (no device involved)"""
    from yolo_amd.metric import DetectMetric
    for name in [str(c) for c in gold["cases"]]:
        B, img, nc, thr, batches = load_case(gold, name)
        names = [f"c{i}" for i in range(nc)]
        dm = DetectMetric((img, img), names, "predicts", "batch_targets", ap_iou_threshold=thr)
        summary = {"sample_metrics": [], "target": []}
        for preds, tgts, want_tp in batches:
            o, labels = 0, []
            for p, t in zip(preds, tgts):
                if p is None:
                    continue
                summary["sample_metrics"].append((want_tp[o:o + p.shape[0]], p[:, -1], p[:, 4] * p[:, 5]))
                o += p.shape[0]
                if t is not None:
                    labels.append(t[np.sum(t, axis=-1) != 0][:, -1])
            summary["target"].append(labels)
        res = dm.summary(summary)
        assert np.array_equal(np.array([res["m_recall"], res["m_precision"], res["m_ap"]]), gold[f"{name}__mean"]), name
        per = np.array([[res[n][k] for k in ("precision", "recall", "f1_score")] for n in names], np.float64)
        assert np.array_equal(per, gold[f"{name}__per_class"]), name
    empty = DetectMetric((64, 64), ["a"], "p", "t").summary({"sample_metrics": [], "target": []})
    assert empty == {"m_recall": 0, "m_precision": 0, "m_ap": 0, "a": {"precision": 0, "recall": 0, "f1_score": 0}}
