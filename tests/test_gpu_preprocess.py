"""GPU: yamd_letterbox_u8 (bicubic resize + grey canvas + /255, CHW) bit-exact against the reference-generated
fixtures and against the CPU restatement on more sizes (up- and down-scaling, extreme aspect ratios, row stride)."""
import ctypes
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden", "preprocess.npz")


def test_letterbox_matches_reference_fixtures():
    from yolo_amd.transformer import letterbox_batch
    g = np.load(GOLD)
    for i in range(int(g["n"])):
        th, tw, oh, ow, rw, rh, dw, dh = (int(v) for v in g[f"meta{i}"])
        out, metas = letterbox_batch([g[f"img{i}"]], th, tw)
        m = metas[0]
        assert (m["ori_height"], m["ori_width"], m["resize_w"], m["resize_h"], m["dw"], m["dh"]) == (oh, ow, rw, rh, dw, dh)
        assert m["resize_ratio"] == float(g[f"ratio{i}"])
        got = out[0].cpu().numpy()
        want = g[f"out{i}"]
        assert np.array_equal(got, want), f"case {i}: {np.count_nonzero(got != want)} differ, max {np.abs(got - want).max()}"


@pytest.mark.parametrize("h,w,th,tw", [(480, 640, 640, 640), (1080, 1920, 640, 640), (17, 400, 320, 320), (700, 9, 256, 256),
                                       (640, 640, 640, 640), (33, 47, 640, 640), (375, 500, 416, 416)])
def test_letterbox_matches_oracle(h, w, th, tw):
    from oracle import preprocess_ref as P
    from yolo_amd.transformer import letterbox_batch
    rng = np.random.RandomState(h * 31 + w)
    img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
    want, meta = P.letterbox(img, th, tw)
    out, metas = letterbox_batch([img], th, tw)
    for k in ("ori_height", "ori_width", "resize_w", "resize_h", "dw", "dh", "resize_ratio"):
        assert metas[0][k] == meta[k], k
    got = out[0].cpu().numpy()
    assert np.array_equal(got, want), f"{np.count_nonzero(got != want)} differ, max {np.abs(got - want).max()}"


def test_letterbox_batch_ragged_and_row_stride():
    from oracle import preprocess_ref as P
    from yolo_amd import lib as L
    from yolo_amd.transformer import letterbox_batch
    rng = np.random.RandomState(5)
    imgs = [rng.randint(0, 256, s).astype(np.uint8) for s in ((90, 160, 3), (200, 120, 3), (64, 64, 3))]
    out, metas = letterbox_batch(imgs, 128, 128)
    for i, im in enumerate(imgs):
        assert np.array_equal(out[i].cpu().numpy(), P.letterbox(im, 128, 128)[0])
    # a crop of a wider image: row stride larger than 3*w
    big = torch.from_numpy(rng.randint(0, 256, (100, 300, 3)).astype(np.uint8)).cuda()
    crop = big[10:80, 20:220]                                   # 70 x 200 view, row stride 900 bytes
    o = torch.empty((3, 96, 96), dtype=torch.float32, device="cuda")
    ws = torch.empty(L.LIB.yamd_letterbox_workspace_bytes(70, 200, 96, 96), dtype=torch.uint8, device="cuda")
    geom = (ctypes.c_int * 4)()
    ratio = ctypes.c_double()
    L.check(L.LIB.yamd_letterbox_u8(crop.data_ptr(), 70, 200, 900, 96, 96, o.data_ptr(), geom, ctypes.byref(ratio),
                                    ws.data_ptr(), ws.numel(), L.current_stream()), "letterbox")
    assert np.array_equal(o.cpu().numpy(), P.letterbox(crop.cpu().numpy(), 96, 96)[0])


def test_letterbox_rejects_bad_arguments():
    from yolo_amd import lib as L
    assert L.LIB.yamd_letterbox_workspace_bytes(0, 10, 64, 64) == 0
    o = torch.empty((3, 64, 64), dtype=torch.float32, device="cuda")
    img = torch.zeros((10, 10, 3), dtype=torch.uint8, device="cuda")
    ws = torch.empty(16, dtype=torch.uint8, device="cuda")      # too small
    geom = (ctypes.c_int * 4)()
    ratio = ctypes.c_double()
    rc = L.LIB.yamd_letterbox_u8(img.data_ptr(), 10, 10, 30, 64, 64, o.data_ptr(), geom, ctypes.byref(ratio), ws.data_ptr(),
                                 16, L.current_stream())
    assert rc != 0 and L.LIB.yamd_last_error()
