"""PyTorch-CPU fp32 restatement of the v5 conv graph (oracle/v5_ref.py) against the reference model
run on the same seeded weights and images (tests/golden/v5s_64.npz, v5s_640.npz).
Tolerance: max-abs 1e-4 (rel 1e-4) on raw head logits and the decoded tensor."""
import numpy as np
import torch

from conftest import load_golden, split_rows
from oracle import decode_ref, postprocess, v5_ref
from yolo_amd.init import seeded_images, seeded_state_dict
from yolo_amd.param import Yolo5Param
from yolo_amd.yolov5 import YoloV5


def _setup(name):
    g = load_golden(name)
    size, batch, seed = int(g["size"]), int(g["batch"]), int(g["seed"])
    p = Yolo5Param(m_type="s").set_img_size(size)
    model = YoloV5(p)
    sd = seeded_state_dict(model, seed=seed, mode="spread")
    imgs = torch.from_numpy(seeded_images(batch, size, seed=seed))
    return g, p, model, sd, imgs


def test_v5s_64_raw_heads_and_forward():
    g, p, model, sd, imgs = _setup("v5s_64.npz")
    with torch.no_grad():
        raws = [t.numpy() for t in v5_ref.body_forward(sd, imgs)]
    for got, key in zip(raws, ("raw_p3", "raw_p4", "raw_p5")):
        np.testing.assert_allclose(got, g[key], rtol=1e-4, atol=1e-4)
    pred = decode_ref.decode_anchor(raws, p.anchors, p.strides, p.img_size, "v5")
    np.testing.assert_allclose(pred, g["pred"], rtol=1e-4, atol=1e-4)
    for i, key in enumerate(("head3", "head4", "head5")):
        lv = decode_ref.decode_anchor_level(raws[i], p.anchors[3 * i:3 * i + 3], p.strides[i], "v5")
        np.testing.assert_allclose(lv, g[key], rtol=1e-4, atol=1e-4)


def test_v5s_64_predict_rows():
    """fed the reference's own decoded tensor, the oracle NMS returns the reference's predict() rows bit for bit"""
    g = load_golden("v5s_64.npz")
    got = postprocess.non_max_suppression(g["pred"], 64, "cxcywh", float(g["conf_threshold"]), float(g["nms_threshold"]))
    want = split_rows(g["counts"], g["rows"])
    for a, b in zip(got, want):
        assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_v5s_640_subsampled_rows():
    g, p, model, sd, imgs = _setup("v5s_640.npz")
    torch.set_num_threads(8)
    with torch.no_grad():
        raws = [t.numpy() for t in v5_ref.body_forward(sd, imgs)]
    np.testing.assert_allclose(raws[2], g["raw_p5"], rtol=1e-4, atol=2e-4)
    pred = decode_ref.decode_anchor(raws, p.anchors, p.strides, p.img_size, "v5")
    np.testing.assert_allclose(pred[:, g["pred_rows"]], g["pred"], rtol=1e-4, atol=2e-3)


def test_fused_state_dict_matches_reference_fuse():
    """YoloV5.fuse() yields the reference's fused checkpoint layout and values; the oracle accepts it."""
    g, p, model, sd, imgs = _setup("v5s_64.npz")
    f = load_golden("v5s_64_fused.npz")
    model.load_state_dict(sd, strict=True)
    model.fuse()
    fs = model.state_dict()
    assert len(fs) == int(f["n_keys"])
    for key in ("body_net.backbone.stem.conv.conv.weight", "body_net.backbone.stem.conv.conv.bias",
                "body_net.conv3_for_up_sample1.m.0.cv2.conv.weight", "body_net.conv3_for_up_sample1.m.0.cv2.conv.bias"):
        np.testing.assert_allclose(fs[key].numpy()[:8], f[key.replace(".", "__")], rtol=1e-5, atol=1e-6)
    with torch.no_grad():
        raws = [t.numpy() for t in v5_ref.body_forward(fs, imgs)]
    pred = decode_ref.decode_anchor(raws, p.anchors, p.strides, p.img_size, "v5")
    np.testing.assert_allclose(pred, f["pred"], rtol=1e-4, atol=1e-4)
