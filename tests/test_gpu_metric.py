"""GPU: yamd_detect_match behind yolo_amd.metric.DetectMetric against the reference-generated fixtures (true-positive
flags and summary values identical), the device-resident (rows, counts) input, and a large random batch against the
CPU restatement."""
import os

import numpy as np
import pytest
import torch

from test_oracle_metric import GOLD, load_case

pytestmark = pytest.mark.gpu


def test_detect_metric_matches_reference_fixtures():
    from yolo_amd.metric import DetectMetric
    g = np.load(GOLD)
    for name in [str(c) for c in g["cases"]]:
        B, img, nc, thr, batches = load_case(g, name)
        names = [f"c{i}" for i in range(nc)]
        dm = DetectMetric((img, img), names, "predicts", "batch_targets", ap_iou_threshold=thr)
        summary = {}
        for preds, tgts, want_tp in batches:
            n0 = len(summary.get("sample_metrics", []))
            dm.metric_feed_forward({"predicts": preds, "batch_targets": [None if t is None else torch.from_numpy(t) for t in tgts]},
                                   summary)
            new = summary["sample_metrics"][n0:]
            got = np.concatenate([x[0] for x in new]) if new else np.zeros(0)
            assert np.array_equal(got, want_tp), name
        res = dm.summary(summary)
        assert np.array_equal(np.array([res["m_recall"], res["m_precision"], res["m_ap"]]), g[f"{name}__mean"]), name
        per = np.array([[res[n][k] for k in ("precision", "recall", "f1_score")] for n in names], np.float64)
        assert np.array_equal(per, g[f"{name}__per_class"]), name


def test_detect_match_device_rows_large_batch():
    """64 images x up to 3000 predictions x 60 ground truths, straight from device tensors"""
    from oracle import metric_ref as M
    from yolo_amd.metric import DetectMetric, match_batch
    rng = np.random.RandomState(2)
    B, K, G, nc, img = 64, 3000, 60, 6, 640
    rows = np.zeros((B, K, 7), np.float32)
    counts = rng.randint(0, K + 1, B).astype(np.int32)
    counts[0], counts[1] = 0, K
    tg = np.zeros((B, G, 5), np.float32)
    for b in range(B):
        g = rng.randint(0, G + 1)
        tg[b, :g, :2] = rng.uniform(0.1, 0.9, (g, 2))
        tg[b, :g, 2:4] = rng.uniform(0.05, 0.3, (g, 2))
        tg[b, :g, 4] = rng.randint(0, nc, g)
        k = counts[b]
        src = rng.randint(0, max(g, 1), k)
        c = tg[b, src, :2] * img + rng.normal(0, 8, (k, 2))
        wh = tg[b, src, 2:4] * img * rng.uniform(0.8, 1.25, (k, 2))
        rows[b, :k, :2] = c - wh / 2
        rows[b, :k, 2:4] = c + wh / 2
        rows[b, :k, 4:6] = rng.uniform(0.2, 1, (k, 2))
        rows[b, :k, 6] = np.where(rng.uniform(size=k) < 0.8, tg[b, src, 4], rng.randint(0, nc, k))
    d_rows, d_counts = torch.from_numpy(rows).cuda(), torch.from_numpy(counts).cuda()
    tp = match_batch(d_rows, d_counts, torch.from_numpy(tg).cuda(), None, (img, img), 0.5).cpu().numpy()
    for b in range(B):
        want, _ = M.match_image(rows[b, :counts[b]], tg[b], (img, img), np.float32(0.5))
        assert np.array_equal(tp[b, :counts[b]].astype(np.float64), want), f"image {b}"
    # the same through the class, predictions given as the device pair
    dm = DetectMetric((img, img), [f"c{i}" for i in range(nc)], "predicts", "batch_targets")
    summary = {}
    dm.metric_feed_forward({"predicts": (d_rows, d_counts), "batch_targets": [torch.from_numpy(t) for t in tg]}, summary)
    preds = [rows[b, :counts[b]] if counts[b] else None for b in range(B)]
    m, lab = M.batch_statistics(preds, list(tg), (img, img), np.float32(0.5))
    want = M.summarise(m, [lab], dm.class_names)
    got = dm.summary(summary)
    assert got["m_ap"] == want["m_ap"] and got["m_recall"] == want["m_recall"] and got["m_precision"] == want["m_precision"]
    assert all(got[n] == want[n] for n in dm.class_names)


def test_detect_match_rejects_bad_shapes():
    from yolo_amd import lib as L
    from yolo_amd.metric import match_batch
    rows = torch.zeros((2, 4, 7), device="cuda")
    counts = torch.zeros(2, dtype=torch.int32, device="cuda")
    with pytest.raises(L.YamdError):
        match_batch(rows, counts, torch.zeros((3, 2, 5), device="cuda"), None, (64, 64), 0.5)
    with pytest.raises(L.YamdError):
        match_batch(rows.cpu(), counts, torch.zeros((2, 2, 5), device="cuda"), None, (64, 64), 0.5)
