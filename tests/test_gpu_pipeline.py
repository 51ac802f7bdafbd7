"""yolo_amd.pipeline: running NMS on a second stream and handing rows to the host one step late must not change a
single bit of the detections."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_two_stream_steps_equal_synchronous_predict(gpu_device):
    from yolo_amd import models, torch_utils as tu
    from yolo_amd.init import seeded_images, seeded_state_dict
    from yolo_amd.pipeline import HostSink, TwoStreamStep
    torch.cuda.set_device(gpu_device)
    p, m = models.create("v5s", img_size=128)
    m.load_state_dict(seeded_state_dict(m, seed=9, mode="spread", **models.init_options("v5s")))
    batches = [torch.from_numpy(seeded_images(3, 128, seed=60 + i)).to(gpu_device) for i in range(5)]
    want = []
    for x in batches:
        rows, counts = m.predict_device(x)
        want.append(tu.rows_to_host(rows, counts))
    rows0, _ = m.predict_device(batches[0])
    sink = HostSink(3, rows0.shape[1], rows0.shape[2], gpu_device)
    step = TwoStreamStep(m, gpu_device)
    got = []

    def collect():
        got.append([None if r is None else r.copy() for r in sink.finished()])

    for i, x in enumerate(batches):
        step(x, sink)                      # completes step i-1's hand-over
        if i > 0:
            collect()
    sink.drain()
    collect()
    assert len(got) == len(want)
    assert sum(0 if r is None else len(r) for w in want for r in w) > 0
    for g, w in zip(got, want):
        for a, b in zip(g, w):
            assert (a is None) == (b is None)
            if a is not None:
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
