"""GPU parity of decode and NMS through the C ABI.

decode: <= 1e-5 relative (+1e-5 abs) against the reference's golden tensors and the numpy oracle on
identical fp32 logits (exp/sigmoid ulp differences only).
NMS: rows and kept source indices bit-identical to the golden rows of the reference and to the C
oracle, including order; images without candidates give count 0 (the reference's None).
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, split_rows

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-5, 1e-5


def _nhwc(raw, dev):
    return torch.from_numpy(np.ascontiguousarray(raw.transpose(0, 2, 3, 1))).to(dev)


def test_decode_anchor_kinds_vs_golden(gpu_device):
    from yolo_amd import lib as L, param as P, torch_utils as tu
    G = load_golden("decode.npz")
    p5, p3, pp, pf = P.Yolo5Param(m_type="s"), P.Yolo3Param(), P.Yolo7PoseParam(), P.Yolo7FaceParam()
    raws = [_nhwc(G[f"v5_raw{i}"], gpu_device) for i in range(3)]
    got = tu.decode_anchor_levels(raws, p5.anchors, p5.strides, 64, L.DECODE_V5).cpu().numpy()
    np.testing.assert_allclose(got, G["v5_pred"], rtol=RTOL, atol=ATOL)
    h3 = tu.decode_anchor_head(raws[0], p5.anchors[:3], 8, L.DECODE_V5).cpu().numpy()
    np.testing.assert_allclose(h3, G["v5_head3"], rtol=RTOL, atol=ATOL)
    raws = [_nhwc(G[f"v3_raw{i}"], gpu_device) for i in range(3)]
    got = tu.decode_anchor_levels(raws, p3.anchors, p3.strides, 64, L.DECODE_V3).cpu().numpy()
    np.testing.assert_allclose(got, G["v3_pred"], rtol=RTOL, atol=ATOL)
    raws = [_nhwc(G[f"pose_raw{i}"], gpu_device) for i in range(3)]
    got = tu.decode_anchor_levels(raws, pp.anchors, pp.strides, 64, L.DECODE_POSE, 17).cpu().numpy()
    np.testing.assert_allclose(got, G["pose_pred"], rtol=RTOL, atol=ATOL)
    for i in range(3):
        r = _nhwc(G[f"face_raw{i}"], gpu_device)
        got = tu.decode_anchor_head(r, pf.anchors[3 * i:3 * i + 3], pf.strides[i], L.DECODE_FACE, 5).cpu().numpy()
        np.testing.assert_allclose(got, G[f"face_head{i}"], rtol=RTOL, atol=ATOL)


def test_decode_dfl_17_bins_unit_conf(gpu_device):
    """The YOLOv6 m/l form: 17 bins per side, four levels, confidence column 1.0; against the CPU restatement."""
    from oracle import decode_ref as D
    from yolo_amd import torch_utils as tu
    rng = np.random.RandomState(5)
    nc, strides, sizes = 20, [8, 16, 32, 64], [(16, 24), (8, 12), (4, 6), (2, 3)]
    raws = [(rng.standard_normal((3, 68 + nc, h, w)) * 2.5).astype(np.float32) for h, w in sizes]
    dev = [torch.from_numpy(np.ascontiguousarray(r.transpose(0, 2, 3, 1))).to(gpu_device) for r in raws]
    for fmt in ("cxcywh", "xyxy"):
        want = D.decode_dfl(raws, strides, 17, nc, fmt, unit_conf=True)
        got = tu.decode_dfl(dev, strides, 17, nc, fmt, unit_conf=True).cpu().numpy()
        assert (got[..., 4] == 1.0).all()
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=2e-4)


@pytest.mark.parametrize("P,nc", [(5, 1), (17, 18)])
def test_decode_dfl_points_vs_oracle(gpu_device, P, nc):
    """m/l face / pose heads: 17-bin DFL box sides next to the landmark columns; bins = 1 stays the plain ltrb decode."""
    from oracle import decode_ref as D
    from yolo_amd import torch_utils as tu
    rng = np.random.RandomState(9 + P)
    strides, sizes = [8, 16, 32, 64], [(16, 24), (8, 12), (4, 6), (2, 3)]
    for bins in (17, 1):
        raws = [(rng.standard_normal((3, 4 * bins + 2 * P + nc, h, w)) * 2.0).astype(np.float32) for h, w in sizes]
        dev = [_nhwc(r, gpu_device) for r in raws]
        for fmt in ("cxcywh", "xyxy"):
            want = D.decode_ltrb_points(raws, strides, P, nc, fmt, bins)
            got = tu.decode_ltrb_points(dev, strides, P, nc, fmt, bins=bins).cpu().numpy()
            np.testing.assert_allclose(got, want, rtol=1e-5, atol=2e-4)


def test_decode_dfl_vs_golden(gpu_device):
    from yolo_amd import torch_utils as tu
    G = load_golden("decode.npz")
    raws = [_nhwc(G[f"v8_raw{i}"], gpu_device) for i in range(3)]
    got = tu.decode_dfl(raws, [8, 16, 32], 16, 20, "cxcywh").cpu().numpy()
    np.testing.assert_allclose(got, G["v8_pred_cxcywh"], rtol=RTOL, atol=2e-5)
    got = tu.decode_dfl(raws, [8, 16, 32], 16, 20, "xyxy").cpu().numpy()
    np.testing.assert_allclose(got, G["v8_pred_xyxy"], rtol=RTOL, atol=2e-5)


def test_decode_ragged_sizes_vs_oracle(gpu_device):
    """odd feature sizes (19x19, 5x3): unaligned rows, partial tiles."""
    from oracle import decode_ref as D
    from yolo_amd import lib as L, torch_utils as tu
    rng = np.random.RandomState(3)
    anchors = [(10, 13), (16, 30), (33, 23)]
    for (h, w, C, kind, name, P) in ((19, 19, 25, L.DECODE_V5, "v5", 0), (5, 3, 85, L.DECODE_V3, "v3", 0),
                                     (7, 9, 57, L.DECODE_POSE, "pose", 17), (3, 11, 16, L.DECODE_FACE, "face", 5)):
        raw = (rng.standard_normal((2, 3 * C, h, w)) * 2).astype(np.float32)
        want = D.decode_anchor_level(raw, anchors, 16, name, P)
        got = tu.decode_anchor_head(_nhwc(raw, gpu_device), anchors, 16, kind, P).cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=RTOL, atol=ATOL)


def _check_rows(got_rows, got_counts, want, got_index=None, want_index=None):
    counts = got_counts.cpu().tolist()
    rows = got_rows.cpu().numpy()
    idx = None if got_index is None else got_index.cpu().numpy()
    assert len(counts) == len(want)
    for b, w in enumerate(want):
        if w is None:
            assert counts[b] == 0, f"image {b}: expected no detections, got {counts[b]}"
            continue
        assert counts[b] == w.shape[0], f"image {b}: kept {counts[b]} vs {w.shape[0]}"
        g = rows[b, :counts[b]]
        assert np.array_equal(g.view(np.uint32), w.view(np.uint32)), f"image {b}: rows differ"
        if want_index is not None:
            assert np.array_equal(idx[b, :counts[b]].astype(np.int64), want_index[b]), f"image {b}: kept indices differ"


def test_nms_vs_reference_golden(gpu_device):
    from oracle import postprocess as pp
    from yolo_amd import torch_utils as tu
    G = load_golden("nms.npz")
    for name in [str(c) for c in G["cases"]]:
        pred = G[f"{name}__pred"]
        img, cx, conf, nms, P, pc = G[f"{name}__meta"].tolist()
        n_extra = int(P) * (3 if pc else 2)
        fmt = "cxcywh" if cx else "xyxy"
        rows, counts, index = tu.nms_batched(torch.from_numpy(pred).to(gpu_device), img, fmt, conf, nms, n_extra,
                                             want_index=True)
        want = split_rows(G[f"{name}__counts"], G[f"{name}__rows"])
        _, want_idx = pp.non_max_suppression(pred, img, fmt, conf, nms, point_num=int(P), point_conf=bool(pc),
                                             return_index=True)
        _check_rows(rows, counts, want, index, want_idx)


def test_nms_public_signatures(gpu_device):
    from yolo_amd import torch_utils as tu
    G = load_golden("nms.npz")
    pred = torch.from_numpy(G["none_images__pred"]).to(gpu_device)
    out = tu.non_max_suppression(pred, img_size=320.0, conf_threshold=0.25, nms_threshold=0.3)
    want = split_rows(G["none_images__counts"], G["none_images__rows"])
    assert out[0] is None and out[2] is None
    assert np.array_equal(out[1].cpu().numpy(), want[1])
    pred = torch.from_numpy(G["face5__pred"]).to(gpu_device)
    out = tu.non_max_suppression_with_point(pred, 640.0, 5, conf_threshold=0.3, nms_threshold=0.3)
    assert out[0].shape[1] == 17
    pred = torch.from_numpy(G["pose17__pred"]).to(gpu_device)
    out = tu.non_max_suppression_with_point_conf(pred, 640.0, 17, conf_threshold=0.3, nms_threshold=0.3)
    assert out[0].shape[1] == 58


@pytest.mark.parametrize("B,A,nc,clusters,conf", [(4, 2500, 20, 40, 0.1), (2, 6000, 3, 25, 0.05), (3, 1000, 80, 10, 0.2)])
def test_nms_random_vs_c_oracle(gpu_device, B, A, nc, clusters, conf):
    """larger seeded sets: multi-tile sweeps, segments > 256 boxes, many classes"""
    from oracle import postprocess as pp
    from oracle.gen_golden_synth import synthetic_pred
    from yolo_amd import torch_utils as tu
    rng = np.random.RandomState(B * 7 + A)
    pred = synthetic_pred(rng, B, A, nc, 640.0, clusters=clusters)
    want, want_idx = pp.non_max_suppression(pred, 640.0, "cxcywh", conf, 0.3, return_index=True)
    rows, counts, index = tu.nms_batched(torch.from_numpy(pred).to(gpu_device), 640.0, "cxcywh", conf, 0.3, 0,
                                         want_index=True)
    _check_rows(rows, counts, want, index, want_idx)


@pytest.mark.parametrize("nc,thr", [(1, 0.3), (3, 0.45), (2, 0.0)])
def test_nms_xyxy_with_inverted_and_empty_boxes(gpu_device, nc, thr):
    """xyxy rows as an anchor-free head with untrained weights emits them: a third of the boxes have x2 < x1 or
    y2 < y1 (negative "areas", unions <= 0), some are empty, some lie outside the image and clamp to its border; the
    overlap test must still be the reference's division for them.  Several tiles per bucket."""
    from oracle import postprocess as pp
    from yolo_amd import torch_utils as tu
    rng = np.random.RandomState(11 + nc)
    A = 1500
    c = rng.uniform(-20, 340, (2, A, 2))
    half = rng.uniform(4, 60, (2, A, 2))
    x1y1, x2y2 = c - half, c + half
    flip = rng.uniform(size=(2, A)) < 0.33
    x1y1[flip], x2y2[flip] = x2y2[flip], x1y1[flip].copy()
    empty = rng.uniform(size=(2, A)) < 0.05
    x2y2[empty] = x1y1[empty]
    obj = rng.uniform(0.2, 1.0, (2, A, 1))
    cls = rng.uniform(0.3, 1.0, (2, A, nc))
    pred = np.concatenate((x1y1, x2y2, obj, cls), -1).astype(np.float32)
    want, want_idx = pp.non_max_suppression(pred, 320.0, "xyxy", 0.1, thr, return_index=True)
    rows, counts, index = tu.nms_batched(torch.from_numpy(pred).to(gpu_device), 320.0, "xyxy", 0.1, thr, 0, want_index=True)
    _check_rows(rows, counts, want, index, want_idx)


def test_nms_allpass_single_class_big_segment(gpu_device):
    """every anchor passes and falls into ONE class: the longest sequential sweep (segment = A boxes)"""
    from oracle import postprocess as pp
    from yolo_amd import torch_utils as tu
    g = 40
    ys, xs = np.meshgrid(np.arange(g), np.arange(g), indexing="ij")
    rng = np.random.RandomState(1)
    box = np.concatenate([np.stack((xs.ravel() * 8 + 4.0, ys.ravel() * 8 + 4.0, np.full(g * g, aw), np.full(g * g, ah)), -1)
                          for aw, ah in ((10, 13), (16, 30), (33, 23))], 0)[None]
    A = box.shape[1]
    obj = 0.5 + rng.uniform(-1e-3, 1e-3, (1, A, 1))
    cls = np.concatenate((np.full((1, A, 1), 0.6), np.full((1, A, 3), 0.4)), -1)
    pred = np.concatenate((box, obj, cls), -1).astype(np.float32)
    want, want_idx = pp.non_max_suppression(pred, 320.0, "cxcywh", 0.1, 0.3, return_index=True)
    rows, counts, index = tu.nms_batched(torch.from_numpy(pred).to(gpu_device), 320.0, "cxcywh", 0.1, 0.3, 0,
                                         want_index=True)
    _check_rows(rows, counts, want, index, want_idx)


def test_nms_max_out_truncates_but_counts(gpu_device):
    from yolo_amd import torch_utils as tu
    G = load_golden("nms.npz")
    pred = torch.from_numpy(G["clustered20__pred"]).to(gpu_device)
    rows, counts, _ = tu.nms_batched(pred, 640.0, "cxcywh", 0.1, 0.3, 0, max_out=50)
    want = split_rows(G["clustered20__counts"], G["clustered20__rows"])
    assert counts.cpu().tolist() == [w.shape[0] for w in want]
    for b, w in enumerate(want):
        assert np.array_equal(rows[b].cpu().numpy(), w[:50])


def test_rows_to_host_packs_like_reference_list(gpu_device):
    from yolo_amd import torch_utils as tu
    G = load_golden("nms.npz")
    pred = torch.from_numpy(G["none_images__pred"]).to(gpu_device)
    rows, counts, _ = tu.nms_batched(pred, 320.0, "cxcywh", 0.25, 0.3, 0)
    got = tu.rows_to_host(rows, counts)
    want = split_rows(G["none_images__counts"], G["none_images__rows"])
    assert got[0] is None and got[2] is None and np.array_equal(got[1], want[1])


def test_decode_ltrb_points_vs_oracle(gpu_device):
    """anchor-free ltrb + landmark decode (v6 face) on N(0,2) logits, four levels incl. an odd-sized one, against
    the numpy oracle (itself pinned to the reference model by tests/golden/v6face_*.npz)"""
    from oracle import decode_ref as D
    from yolo_amd import torch_utils as tu
    rng = np.random.RandomState(9)
    sizes, strides, P, nc = [(16, 16), (8, 8), (5, 3), (2, 2)], [8, 16, 32, 64], 5, 1
    raws = [(rng.standard_normal((3, 4 + 2 * P + nc, h, w)) * 2).astype(np.float32) for h, w in sizes]
    for fmt in ("xyxy", "cxcywh"):
        want = D.decode_ltrb_points(raws, strides, P, nc, fmt)
        got = tu.decode_ltrb_points([_nhwc(r, gpu_device) for r in raws], strides, P, nc, fmt).cpu().numpy()
        assert got.shape == want.shape
        np.testing.assert_allclose(got, want, rtol=RTOL, atol=ATOL)


def _nms_case(gpu_device, pred, img, conf=0.1, thr=0.3):
    from oracle import postprocess as pp
    from yolo_amd import torch_utils as tu
    want, want_idx = pp.non_max_suppression(pred, img, "cxcywh", conf, thr, return_index=True)
    rows, counts, index = tu.nms_batched(torch.from_numpy(pred).to(gpu_device), img, "cxcywh", conf, thr, 0, want_index=True)
    _check_rows(rows, counts, want, index, want_idx)
    return counts.cpu().tolist()


def _one_class_pred(box, score):
    """(A,4) cxcywh boxes + (A,) objectness -> (1, A, 5+2) with every row in class 0"""
    A = box.shape[0]
    cls = np.concatenate((np.full((A, 1), 0.9), np.full((A, 1), 0.1)), -1)
    return np.concatenate((box, score[:, None], cls), -1).astype(np.float32)[None]


def test_nms_two_level_lattice_bucket(gpu_device):
    """a bucket of 8000 boxes (two anchor planes of one class), sparse overlaps"""
    rng = np.random.RandomState(3)
    planes = []
    for g, st, (aw, ah) in ((80, 8, (16, 30)), (40, 16, (30, 61))):
        ys, xs = np.meshgrid(np.arange(g), np.arange(g), indexing="ij")
        s = rng.uniform(0.6, 1.6, (g * g, 2))
        planes.append(np.stack((xs.ravel() * st + st / 2 + rng.uniform(-2, 2, g * g), ys.ravel() * st + st / 2 + rng.uniform(-2, 2, g * g),
                                aw * s[:, 0], ah * s[:, 1]), -1))
    box = np.concatenate(planes, 0)
    score = 0.5 + rng.uniform(-0.2, 0.2, box.shape[0])
    kept = _nms_case(gpu_device, _one_class_pred(box, score), 640.0)
    assert 1000 < kept[0] < 7000


def test_nms_smooth_scores_long_dependency_chains(gpu_device):
    """scores fall smoothly across the lattice: every decision depends on a long chain of earlier ones"""
    g = 64
    ys, xs = np.meshgrid(np.arange(g), np.arange(g), indexing="ij")
    box = np.stack((xs.ravel() * 8 + 4.0, ys.ravel() * 8 + 4.0, np.full(g * g, 14.0), np.full(g * g, 14.0)), -1)
    score = 0.9 - 0.5 * (xs.ravel() + 0.37 * ys.ravel()) / (1.37 * g)
    _nms_case(gpu_device, _one_class_pred(box, score), 512.0)


def test_nms_chain_of_3000(gpu_device):
    """3000 boxes in a row, each overlapping only its neighbours, scores strictly falling: every other box survives"""
    n = 3000
    box = np.stack((np.arange(n) * 1.0 + 10.0, np.full(n, 50.0), np.full(n, 2.0), np.full(n, 20.0)), -1)
    score = np.linspace(0.95, 0.2, n)
    kept = _nms_case(gpu_device, _one_class_pred(box, score), 4096.0)
    assert kept[0] == n // 2


def test_nms_dense_cluster_in_sparse_background(gpu_device):
    """700 near-identical boxes plus a sparse background in the same bucket"""
    rng = np.random.RandomState(9)
    dense = np.stack((100 + rng.uniform(-1, 1, 700), 100 + rng.uniform(-1, 1, 700), 40 + rng.uniform(-1, 1, 700),
                      40 + rng.uniform(-1, 1, 700)), -1)
    sparse = np.stack((rng.uniform(0, 640, 900), rng.uniform(0, 640, 900), rng.uniform(5, 40, 900), rng.uniform(5, 40, 900)), -1)
    box = np.concatenate((dense, sparse), 0)
    score = rng.uniform(0.2, 0.9, box.shape[0])
    _nms_case(gpu_device, _one_class_pred(box, score), 640.0)


def test_nms_bucket_larger_than_lds_sort_capacity(gpu_device):
    rng = np.random.RandomState(11)
    n = 9000
    box = np.stack((rng.uniform(0, 640, n), rng.uniform(0, 640, n), rng.uniform(4, 30, n), rng.uniform(4, 30, n)), -1)
    score = rng.uniform(0.2, 0.9, n)
    _nms_case(gpu_device, _one_class_pred(box, score), 640.0)


@pytest.mark.parametrize("n", [1000, 1030, 1100, 1500, 2047, 2100, 5800])
def test_nms_group_counts_around_a_round_of_waves(gpu_device, n):
    """One bucket of 16, 17, 18, 24, 32, 33 and 91 groups of 64 boxes: the sweep deals one group per wave and round, shares
    the groups of the last partial round among the waves left over, keeps the boxes of a wave's first groups in registers,
    the next ones in LDS and reads the rest from memory; every split must give the plain greedy result."""
    rng = np.random.RandomState(n)
    box = np.stack((rng.uniform(0, 640, n), rng.uniform(0, 640, n), rng.uniform(20, 120, n), rng.uniform(20, 120, n)), -1)
    score = rng.uniform(0.2, 0.9, n)
    kept = _nms_case(gpu_device, _one_class_pred(box, score), 640.0)
    assert 0 < kept[0] < n


def test_nms_more_oversized_buckets_than_compute_units(gpu_device):
    """257+ images whose single bucket (8300 boxes) is beyond the LDS sort: with more such buckets than compute units they
    run in the two-per-compute-unit instantiation of the sweep, otherwise in the other one; both against the C oracle
    (every image holds the same boxes, so one oracle run checks them all)."""
    from oracle import postprocess as pp
    from yolo_amd import torch_utils as tu
    rng = np.random.RandomState(5)
    n = 8300
    box = np.stack((rng.uniform(0, 640, n), rng.uniform(0, 640, n), rng.uniform(4, 40, n), rng.uniform(4, 40, n)), -1)
    one = _one_class_pred(box, rng.uniform(0.2, 0.9, n))
    want, want_idx = pp.non_max_suppression(one, 640.0, "cxcywh", 0.1, 0.3, return_index=True)
    cus = torch.cuda.get_device_properties(gpu_device).multi_processor_count
    for B in (3, cus + 4):
        pred = torch.from_numpy(one).to(gpu_device).expand(B, -1, -1).contiguous()
        rows, counts, index = tu.nms_batched(pred, 640.0, "cxcywh", 0.1, 0.3, 0, want_index=True)
        k = want[0].shape[0]
        assert counts.cpu().tolist() == [k] * B
        r, ix = rows.cpu().numpy(), index.cpu().numpy()
        for b in (0, B // 2, B - 1):
            assert np.array_equal(r[b, :k].view(np.uint32), want[0].view(np.uint32))
            assert np.array_equal(ix[b, :k].astype(np.int64), want_idx[0])
        assert (r[:, :k] == r[0, :k]).all()


@pytest.mark.parametrize("kind_name,C,P,sizes,conf,scale,bias", [
    ("v5", 25, 0, [(16, 24), (8, 12), (4, 6)], 0.1, 2.0, 0.0),        # most anchors pass
    ("v5", 25, 0, [(80, 80), (40, 40), (20, 20)], 0.1, 1.0, 0.0),     # the benchmark's level sizes
    ("v5", 25, 0, [(19, 19), (10, 10), (5, 3)], 0.3, 2.0, -3.0),      # few pass; ragged sizes
    ("v5", 6, 0, [(8, 8), (4, 4), (2, 2)], 0.999, 1.0, -9.0),         # nothing passes anywhere
    ("v3", 85, 0, [(13, 13), (7, 7), (4, 4)], 0.2, 1.5, -1.0),
    ("pose", 57, 17, [(12, 20), (6, 10), (3, 5)], 0.3, 2.0, -1.0),
    ("face", 16, 5, [(16, 16), (8, 8), (4, 4)], 0.3, 2.0, -1.0),
])
def test_decode_filter_nms_in_one_call_equals_the_two_steps(gpu_device, kind_name, C, P, sizes, conf, scale, bias):
    """yamd_decode_nms_anchor (the decode kernel filters its own rows; rows that fail are never written) against
    yamd_decode_anchor_levels + yamd_nms_batched: counts, rows and kept source indices bit for bit, and against the CPU
    oracle on the decoded tensor; the rows it did write into the scratch tensor are the decoded ones."""
    from oracle import postprocess as pp
    from yolo_amd import lib as L, torch_utils as tu
    kind = {"v5": L.DECODE_V5, "v3": L.DECODE_V3, "pose": L.DECODE_POSE, "face": L.DECODE_FACE}[kind_name]
    rng = np.random.RandomState(C * 7 + len(sizes) + sizes[0][0])
    anchors = [(10, 13), (16, 30), (33, 23), (30, 61), (62, 45), (59, 119), (116, 90), (156, 198), (373, 326)]
    strides, img = [8, 16, 32], float(sizes[0][0] * 8)
    B = 3
    raws_np = []
    for h, w in sizes:
        r = (rng.standard_normal((B, 3 * C, h, w)) * scale).astype(np.float32)
        r[:, 4::C] += bias                                      # objectness logit of every anchor
        raws_np.append(r)
    raws_np[0][1] -= 40.0 if bias < 0 else 0.0                  # (one image with no candidate on the finest level)
    raws = [_nhwc(r, gpu_device) for r in raws_np]
    n_extra = {"pose": 3 * P, "face": 2 * P}.get(kind_name, 0)
    clamp = kind_name == "pose"
    pred = tu.decode_anchor_levels(raws, anchors, strides, img, kind, P, clamp)
    rows0, counts0, index0 = tu.nms_batched(pred, img, "cxcywh", conf, 0.3, n_extra, want_index=True)
    arena = tu.Arena()
    arena.get("decoded", pred.shape, torch.float32, gpu_device).fill_(float("nan"))      # rows that fail must stay untouched
    rows1, counts1, index1 = tu.decode_nms_anchor_levels(raws, anchors, strides, img, kind, P, clamp, conf, 0.3,
                                                         want_index=True, arena=arena)
    torch.cuda.synchronize()
    assert torch.equal(counts0, counts1)
    c = counts0.cpu().tolist()
    for b in range(B):
        assert torch.equal(rows0[b, :c[b]].view(torch.int32), rows1[b, :c[b]].view(torch.int32))
        assert torch.equal(index0[b, :c[b]], index1[b, :c[b]])
    want = pp.non_max_suppression(pred.cpu().numpy(), img, "cxcywh", conf, 0.3, point_num=P, point_conf=kind_name == "pose")
    _check_rows(rows1, counts1, want)
    # the scratch tensor: exactly the rows that pass the confidence filter were written, with the decoded values
    scratch = arena.get("decoded", pred.shape, torch.float32, gpu_device)
    nc = C - 5 - n_extra
    passed = (pred[..., 4] * pred[..., 5 + n_extra:].max(-1).values) >= torch.tensor(conf, dtype=torch.float32)
    assert torch.equal(scratch[passed].view(torch.int32), pred[passed].view(torch.int32))
    assert torch.isnan(scratch[~passed]).all()
    if conf > 0.99:
        assert sum(c) == 0 and not passed.any()
    else:
        assert sum(c) > 0 and nc >= 1
