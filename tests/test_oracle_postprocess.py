"""The C oracle (oracle/yolo_oracle.c) against the golden vectors produced by the reference's own
non_max_suppression* code (tests/golden/nms.npz, made by oracle/gen_golden.py).  Bit-exact."""
import numpy as np
import pytest

from conftest import load_golden, split_rows
from oracle import postprocess as pp

G = load_golden("nms.npz")
CASES = [str(c) for c in G["cases"]]


def run_case(name, fn):
    pred = G[f"{name}__pred"]
    img, cx, conf, nms, P, pc = G[f"{name}__meta"].tolist()
    got = fn(pred, img, "cxcywh" if cx else "xyxy", conf, nms, int(P), bool(pc))
    want = split_rows(G[f"{name}__counts"], G[f"{name}__rows"])
    assert len(got) == len(want)
    for b, (g, w) in enumerate(zip(got, want)):
        if w is None:
            assert g is None, f"{name}[{b}]: expected None"
            continue
        assert g is not None and g.shape == w.shape, f"{name}[{b}]: {None if g is None else g.shape} vs {w.shape}"
        assert np.array_equal(g.view(np.uint32), w.view(np.uint32)), f"{name}[{b}] rows differ"


@pytest.mark.parametrize("name", CASES)
def test_c_oracle_matches_reference(name):
    run_case(name, lambda pred, img, fmt, conf, nms, P, pc: pp.non_max_suppression(
        pred, img, fmt, conf, nms, point_num=P, point_conf=pc))


def test_reference_cases_cover_edge_semantics():
    # strict '>' at IoU == threshold keeps both; float IoU 1/3 vs double 1/3 suppresses (CPU kernel semantics)
    assert G["iou_eq_half__counts"].tolist() == [2]
    assert G["iou_third__counts"].tolist() == [1]
    assert G["none_images__counts"].tolist()[0] == 0 and G["none_images__counts"].tolist()[2] == 0


def test_nms_stable_ties_and_order():
    boxes = np.array([[0, 0, 10, 10], [100, 100, 110, 110], [0, 0, 10, 10], [200, 200, 210, 210]], np.float32)
    scores = np.array([0.5, 0.5, 0.5, 0.9], np.float32)
    keep = pp.nms(boxes, scores, 0.5)
    assert keep.tolist() == [3, 0, 1]          # ties in index order; duplicate box 2 suppressed by 0


def test_nms_empty_and_single():
    assert pp.nms(np.zeros((0, 4), np.float32), np.zeros((0,), np.float32), 0.5).tolist() == []
    assert pp.nms(np.array([[1, 1, 2, 2]], np.float32), np.array([0.3], np.float32), 0.5).tolist() == [0]
