"""numpy decode restatement (oracle/decode_ref.py) against the reference's head modules and merge
code (tests/golden/decode.npz).  Tolerance 1e-5 relative: only sigmoid/exp ulp differences remain."""
import numpy as np

from conftest import load_golden
from oracle import decode_ref as D
from yolo_amd import param as P

G = load_golden("decode.npz")
RTOL, ATOL = 1e-5, 1e-5


def close(a, b):
    np.testing.assert_allclose(a, b, rtol=RTOL, atol=ATOL)


def test_v5():
    p = P.Yolo5Param(m_type="s")
    raws = [G[f"v5_raw{i}"] for i in range(3)]
    close(D.decode_anchor(raws, p.anchors, p.strides, 64, "v5"), G["v5_pred"])
    close(D.decode_anchor_level(raws[0], p.anchors[:3], 8, "v5"), G["v5_head3"])


def test_v3():
    p = P.Yolo3Param()
    close(D.decode_anchor([G[f"v3_raw{i}"] for i in range(3)], p.anchors, p.strides, 64, "v3"), G["v3_pred"])


def test_pose():
    p = P.Yolo7PoseParam()
    close(D.decode_anchor([G[f"pose_raw{i}"] for i in range(3)], p.anchors, p.strides, 64, "pose", 17), G["pose_pred"])


def test_face_heads():
    p = P.Yolo7FaceParam()
    for i in range(3):
        close(D.decode_anchor_level(G[f"face_raw{i}"], p.anchors[3 * i:3 * i + 3], p.strides[i], "face", 5),
              G[f"face_head{i}"])


def test_v8_dfl():
    raws = [G[f"v8_raw{i}"] for i in range(3)]
    close(D.decode_dfl(raws, [8, 16, 32], 16, 20, "cxcywh"), G["v8_pred_cxcywh"])
    close(D.decode_dfl(raws, [8, 16, 32], 16, 20, "xyxy"), G["v8_pred_xyxy"])


def test_pose_predict_rows_via_oracle_nms():
    """decode (points clamped) + C NMS reproduces the reference's pose predict() rows."""
    from conftest import split_rows
    from oracle import postprocess as pp
    p = P.Yolo7PoseParam()
    pred = D.decode_anchor([G[f"pose_raw{i}"] for i in range(3)], p.anchors, p.strides, 64, "pose", 17,
                           clamp_points=True)
    got = pp.non_max_suppression(pred, 64, "cxcywh", float(G["pose_conf"]), float(G["pose_nms"]), point_num=17,
                                 point_conf=True)
    want = split_rows(G["pose_counts"], G["pose_rows"])
    for g, w in zip(got, want):
        assert (g is None) == (w is None)
        if w is not None:
            assert g.shape == w.shape
            np.testing.assert_allclose(g, w, rtol=RTOL, atol=1e-4)
