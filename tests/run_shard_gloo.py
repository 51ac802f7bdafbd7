"""Launched by tests/test_sharding.py under torch.distributed.run with 2 gloo ranks on the CPU: every rank takes its
contiguous slice of one (B, A, C) tensor, runs the stand-in worker (C oracle NMS) on it, rank 0 gathers the per-image
lists and compares them with the unsharded result, None slots included."""
import os
import sys

import numpy as np
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

from sharding_helpers import make_pred, oracle_nms      # noqa: E402
from yolo_amd import parallel                            # noqa: E402


def main() -> int:
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    pred = make_pred(batch=batch)
    mine = pred[parallel.rank_slice(batch, rank, world)]
    local = oracle_nms(mine)
    assert len(local) == mine.shape[0]
    merged = parallel.gather_predicts_to_rank0(local)
    ok = True
    if rank == 0:
        want = oracle_nms(pred)
        ok = len(merged) == len(want)
        nones = 0
        for g, w in zip(merged, want):
            if w is None:
                nones += 1
                ok = ok and g is None
            else:
                ok = ok and g is not None and np.array_equal(g.view(np.uint32), w.view(np.uint32))
        print(f"SHARD_OK={int(ok)} images={len(want)} none_slots={nones} world={world} "
              f"sizes={[s.stop - s.start for s in parallel.shard_slices(batch, world)]}", flush=True)
    else:
        assert merged is None
    dist.barrier()
    dist.destroy_process_group()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
