"""yamd_nms_batched_static (the entry point that allocates nothing: its scratch is a static device array) against
yamd_nms_batched with a caller-owned workspace, through the C ABI: rows, kept indices and counts identical on two
calls in a row, the second with a smaller input; sizes beyond the static array are refused."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(L, fn_static, pred, img, conf, iou, n_extra=0):
    B, A, C = pred.shape
    dev = pred.device
    W = 7 + n_extra
    st = L.current_stream()
    out = []
    for static in (False, True):
        rows = torch.full((B, A, W), -7.0, dtype=torch.float32, device=dev)
        index = torch.full((B, A), -7, dtype=torch.int32, device=dev)
        counts = torch.full((B,), -7, dtype=torch.int32, device=dev)
        if static:
            L.check(fn_static(pred.data_ptr(), B, A, C, 1, n_extra, img, conf, iou, rows.data_ptr(), index.data_ptr(), A,
                              counts.data_ptr(), st), "nms_batched_static")
        else:
            nbytes = L.LIB.yamd_nms_workspace_bytes(B, A, C, n_extra)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            L.check(L.LIB.yamd_nms_batched(pred.data_ptr(), B, A, C, 1, n_extra, img, conf, iou, rows.data_ptr(),
                                           index.data_ptr(), A, counts.data_ptr(), ws.data_ptr(), nbytes, st), "nms_batched")
        torch.cuda.synchronize()
        out.append((rows.cpu().numpy(), index.cpu().numpy(), counts.cpu().numpy()))
    return out


def test_static_scratch_nms_equals_workspace_nms(gpu_device):
    from oracle import postprocess as pp
    from oracle.gen_golden_synth import synthetic_pred
    from yolo_amd import lib as L
    torch.cuda.set_device(gpu_device)
    fn = L.LIB.yamd_nms_batched_static
    cap = L.LIB.yamd_nms_static_capacity_bytes()
    assert cap >= L.LIB.yamd_nms_workspace_bytes(4, 25200, 25, 0)
    # two calls in a row on the same static array, the second smaller (stale contents of the first must not leak)
    for seed, (B, A, nc, clusters) in enumerate([(4, 6000, 20, 40), (2, 900, 3, 10)]):
        pred_h = synthetic_pred(np.random.RandomState(70 + seed), B, A, nc, 640.0, clusters=clusters)
        pred = torch.from_numpy(pred_h).to(gpu_device)
        (r0, i0, c0), (r1, i1, c1) = _run(L, fn, pred, 640.0, 0.1, 0.3)
        assert np.array_equal(c0, c1)
        assert c0.sum() > 0
        for b in range(B):
            k = int(c0[b])
            assert np.array_equal(r0[b, :k].view(np.uint32), r1[b, :k].view(np.uint32))
            assert np.array_equal(i0[b, :k], i1[b, :k])
        want = pp.non_max_suppression(pred_h, 640.0, "cxcywh", 0.1, 0.3)
        for b, w in enumerate(want):
            k = 0 if w is None else w.shape[0]
            assert int(c1[b]) == k
            if k:
                assert np.array_equal(r1[b, :k].view(np.uint32), w.view(np.uint32))


def test_static_scratch_nms_refuses_oversized_input(gpu_device):
    from yolo_amd import lib as L
    torch.cuda.set_device(gpu_device)
    B, A, C = 64, 25200, 25
    assert L.LIB.yamd_nms_workspace_bytes(B, A, C, 0) > L.LIB.yamd_nms_static_capacity_bytes()
    dummy = torch.zeros(16, dtype=torch.float32, device=gpu_device)     # never read: the size check comes first
    rc = L.LIB.yamd_nms_batched_static(dummy.data_ptr(), B, A, C, 1, 0, 640.0, 0.1, 0.3, dummy.data_ptr(), None, A,
                                       dummy.data_ptr(), L.current_stream())
    assert rc == -3
    assert b"static" in L.LIB.yamd_last_error()


def test_static_scratch_nms_allocates_nothing(gpu_device):
    """The property the entry point exists for: after one warm-up call, a call changes neither the free device memory
    the driver reports nor torch's allocation counter.  yamd_init resolves the static array on every visible device and
    fails unless its address is 256-byte aligned there."""
    from oracle.gen_golden_synth import synthetic_pred
    from yolo_amd import lib as L
    for d in range(torch.cuda.device_count()):
        with torch.cuda.device(d):
            assert L.LIB.yamd_init() == 0, L.LIB.yamd_last_error()
    torch.cuda.set_device(gpu_device)
    B, A, nc = 3, 4000, 20
    pred = torch.from_numpy(synthetic_pred(np.random.RandomState(91), B, A, nc, 640.0, clusters=30)).to(gpu_device)
    rows = torch.empty((B, A, 7), dtype=torch.float32, device=gpu_device)
    index = torch.empty((B, A), dtype=torch.int32, device=gpu_device)
    counts = torch.empty((B,), dtype=torch.int32, device=gpu_device)
    st = L.current_stream()

    def call():
        L.check(L.LIB.yamd_nms_batched_static(pred.data_ptr(), B, A, 5 + nc, 1, 0, 640.0, 0.1, 0.3, rows.data_ptr(),
                                              index.data_ptr(), A, counts.data_ptr(), st), "nms_batched_static")
        torch.cuda.synchronize()

    call()                                                     # warm-up: code objects loaded, attributes raised
    free0 = torch.cuda.mem_get_info(gpu_device)[0]
    n0 = torch.cuda.memory_stats(gpu_device)["allocation.all.allocated"]
    for _ in range(3):
        call()
    assert torch.cuda.mem_get_info(gpu_device)[0] == free0, "free device memory changed across static-scratch calls"
    assert torch.cuda.memory_stats(gpu_device)["allocation.all.allocated"] == n0
    assert int(counts.sum()) > 0
