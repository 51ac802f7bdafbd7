"""CPU tests of the multi-GPU path (yolo_amd.parallel): batch slices, order-preserving gather with None slots, the
worker-process predictor and the one-rank-per-GPU form under torch.distributed (gloo, world size 2).  The C oracle
NMS stands in for the per-GPU work."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from yolo_amd import parallel

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sharding_helpers as H      # noqa: E402


@pytest.mark.parametrize("batch,shards", [(1024, 8), (1024, 4), (7, 2), (3, 8), (0, 2), (256, 1)])
def test_shard_slices_partition_the_batch(batch, shards):
    sl = parallel.shard_slices(batch, shards)
    assert len(sl) == shards
    assert sl[0].start == 0 and sl[-1].stop == batch
    sizes = [s.stop - s.start for s in sl]
    assert all(a.stop == b.start for a, b in zip(sl, sl[1:]))
    assert max(sizes) - min(sizes) <= 1 and sum(sizes) == batch
    assert parallel.rank_slice(batch, shards - 1, shards) == sl[-1]


def _same(got, want):
    assert len(got) == len(want)
    for g, w in zip(got, want):
        if w is None:
            assert g is None
        else:
            assert g is not None and np.array_equal(g.view(np.uint32), w.view(np.uint32))


def test_worker_processes_shard_and_gather_in_image_order():
    pred = H.make_pred(batch=7)
    want = H.oracle_nms(pred)
    assert want[1] is None and want[4] is None and sum(w is not None for w in want) == 5
    with parallel.ShardedPredictor(H.oracle_runner_factory, ["cpu:0", "cpu:1", "cpu:2"]) as sp:
        _same(sp.predict(pred), want)
        _same(sp.predict(pred[:2]), want[:2])          # fewer images than workers: one worker gets an empty slice
        _same(sp.predict(pred[3:]), want[3:])
        # the producer writes straight into the workers' shared blocks; a larger batch makes the blocks grow
        big = H.make_pred(batch=11, seed=5)
        views = sp.input_views(big.shape[0], big.shape[1:], big.dtype)
        assert [v.shape[0] for v in views] == [4, 4, 3]
        for v, sl in zip(views, parallel.shard_slices(big.shape[0], 3)):
            v[...] = big[sl]
        _same(sp.predict_views(), H.oracle_nms(big))
        _same(sp.predict(pred), want)                  # and back to the smaller one in the grown blocks
        paths = [b.path for b in sp._blocks]
        assert all(os.path.exists(q) for q in paths)
    assert not any(os.path.exists(q) for q in paths), "shared blocks must be unlinked on close"


def test_worker_start_failure_is_reported():
    with pytest.raises(RuntimeError, match="no such device"):
        parallel.ShardedPredictor(H.failing_runner_factory, ["cuda:99"], start_timeout=120)


def test_two_gloo_ranks_shard_a_batch_and_rank0_gathers():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "tests", "run_shard_gloo.py"), "7"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("SHARD_OK=")]
    assert len(line) == 1, out.stdout
    assert "SHARD_OK=1" in line[0] and "none_slots=2" in line[0] and "sizes=[4, 3]" in line[0]
