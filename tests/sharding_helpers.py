"""Stand-in shard workers for the CPU tests of yolo_amd.parallel: the C oracle NMS takes the place of the GPU path
(the sharding logic never looks inside a worker).  Importable by spawned worker processes."""
import numpy as np

IMG, CONF, IOU = 640.0, 0.1, 0.3


def make_pred(batch: int = 7, anchors: int = 400, nc: int = 6, seed: int = 3) -> np.ndarray:
    """(B, A, 5+nc) decoded rows; images 1 and 4 have no candidate (objectness 0) -> the reference returns None."""
    from oracle.gen_golden_synth import synthetic_pred
    pred = synthetic_pred(np.random.RandomState(seed), batch, anchors, nc, IMG, clusters=12)
    for b in (1, 4):
        if b < batch:
            pred[b, :, 4] = 0.0
    return pred


def oracle_nms(pred: np.ndarray):
    from oracle import postprocess as pp
    if pred.shape[0] == 0:
        return []
    return pp.non_max_suppression(pred, IMG, "cxcywh", CONF, IOU)


def oracle_runner_factory(device: str):
    """runner_factory of ShardedPredictor: 'device' is only a label here."""
    return oracle_nms


def failing_runner_factory(device: str):
    raise RuntimeError(f"no such device {device}")
