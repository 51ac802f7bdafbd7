"""CPU: the pre/post-processing restatement (oracle/preprocess_ref.py) against fixtures produced by the reference's
ResizeImage(is_train=False) + ImageNorm + RestoreSize (luoyang/transformer/transformer.py:134-295) running on Pillow
(version recorded in the fixture).  Bit-exact: the resize is integer arithmetic and the normalisation one division."""
import os

import numpy as np
import pytest

from oracle import preprocess_ref as P

GOLD = os.path.join(os.path.dirname(__file__), "golden", "preprocess.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def _cases(g):
    return range(int(g["n"]))


def test_letterbox_matches_reference(gold):
    for i in _cases(gold):
        th, tw, oh, ow, rw, rh, dw, dh = (int(v) for v in gold[f"meta{i}"])
        out, meta = P.letterbox(gold[f"img{i}"], th, tw)
        assert (meta["ori_height"], meta["ori_width"], meta["resize_w"], meta["resize_h"], meta["dw"], meta["dh"]) == \
               (oh, ow, rw, rh, dw, dh)
        assert meta["resize_ratio"] == float(gold[f"ratio{i}"])
        want = gold[f"out{i}"]
        assert out.dtype == want.dtype == np.float32 and out.shape == want.shape
        assert np.array_equal(out, want), f"case {i}: max diff {np.abs(out - want).max()}"


def test_restore_size_matches_reference(gold):
    for i in _cases(gold):
        th, tw, oh, ow, rw, rh, dw, dh = (int(v) for v in gold[f"meta{i}"])
        meta = dict(ori_height=oh, ori_width=ow, resize_ratio=float(gold[f"ratio{i}"]), dw=dw, dh=dh)
        boxes, labels = P.restore_size(gold[f"pred{i}"], meta)
        assert np.array_equal(boxes, gold[f"boxes{i}"])
        assert np.array_equal(labels, gold[f"labels{i}"])


def test_host_restore_size_matches_reference(gold):
    from yolo_amd.transformer import RestoreSize
    for i in _cases(gold):
        th, tw, oh, ow, rw, rh, dw, dh = (int(v) for v in gold[f"meta{i}"])
        d = dict(ori_height=oh, ori_width=ow, resize_ratio=float(gold[f"ratio{i}"]), dw=dw, dh=dh, predict=gold[f"pred{i}"])
        RestoreSize().transformer(d)
        assert np.array_equal(d["predict_boxs"], gold[f"boxes{i}"])
        assert np.array_equal(d["predict_label"], gold[f"labels{i}"])
    d = dict(predict=None)
    RestoreSize().transformer(d)
    assert d["predict_boxs"] == [] and d["predict_label"] == []
