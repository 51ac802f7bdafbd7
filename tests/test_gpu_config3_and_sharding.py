"""YOLOv8s at the batch sizes of its benchmark configuration as one GPU sees them (1024 images on one GPU, and the
512 / 256 image slices of the 2- and 4-GPU splits), and the worker-process predictor of yolo_amd.parallel on a real
GPU: two workers on cuda:0 stand in for two devices."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("batch", [1024, 512, 256])
def test_v8s_benchmark_batches_on_one_gpu(gpu_device, batch):
    """Size-independent properties of the whole path at 640x640: the position in the batch does not matter (16 distinct
    images replicated give bit-identical rows everywhere), rows come in descending score order, and the kept rows of
    two images equal the C oracle's NMS of the same decoded tensor bit for bit."""
    from oracle import graphs_ref, postprocess as pp
    from yolo_amd import models
    from yolo_amd.init import seeded_images, seeded_state_dict
    p, m = models.create("v8s")
    m.load_state_dict(seeded_state_dict(m, seed=41, mode="spread", **models.init_options("v8s")))
    S = p.img_size
    base = torch.from_numpy(seeded_images(16, S, seed=43)).to(gpu_device)
    imgs = base.repeat(batch // 16, 1, 1, 1)
    rows, counts = m.predict_device(imgs)
    pred = m._arena.get("decoded", (batch, 8400, 25), torch.float32, gpu_device)
    assert rows.shape[0] == batch
    counts = counts.cpu().numpy()
    assert (counts.reshape(batch // 16, 16) == counts[:16]).all()
    r0, r1 = rows[:16].cpu().numpy(), rows[batch - 16:].cpu().numpy()
    assert counts[:16].sum() > 0
    for b in range(16):
        k = counts[b]
        assert np.array_equal(r0[b, :k].view(np.uint32), r1[b, :k].view(np.uint32))
        assert (np.diff(r0[b, :k, 4] * r0[b, :k, 5]) <= 0).all(), "rows must be in descending score order"
    ref = graphs_ref.model_ref("v8s")
    want = pp.non_max_suppression(pred[:2].cpu().numpy(), S, conf_threshold=p.conf_threshold,
                                  nms_threshold=p.nms_threshold, **ref["nms"])
    for b in range(2):
        k = 0 if want[b] is None else want[b].shape[0]
        assert counts[b] == k
        if k:
            assert np.array_equal(r0[b, :k].view(np.uint32), want[b].view(np.uint32))
    del m, imgs, rows, pred
    torch.cuda.empty_cache()


def _v5s_64_factory():
    """Module-level so that worker processes can import it."""
    from yolo_amd import models
    from yolo_amd.init import seeded_state_dict
    p, m = models.create("v5s", img_size=64)
    m.load_state_dict(seeded_state_dict(m, seed=11, mode="spread"))
    return m


def test_sharded_predictor_matches_single_process(gpu_device):
    from yolo_amd import parallel
    from yolo_amd.init import seeded_images
    imgs = seeded_images(7, 64, seed=12)
    m = _v5s_64_factory()
    want = m.predict({"batch_images": torch.from_numpy(imgs).to(gpu_device)})["predicts"]
    with parallel.ShardedPredictor(parallel.ModelRunner(_v5s_64_factory), ["cuda:0", "cuda:0"]) as sp:
        got = sp.predict(imgs)
        got2 = sp.predict(imgs[:1])
    assert len(got) == 7 and sum(w is not None for w in want) > 0
    for g, w in zip(got + got2, want + want[:1]):
        assert (g is None and w is None) or np.array_equal(g.view(np.uint32), w.view(np.uint32))
