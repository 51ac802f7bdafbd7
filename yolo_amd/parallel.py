"""Batch sharding over the GPUs of one node.

Every image goes through conv graph, decode and NMS independently of the others (NMS is per image,
luoyang/utils/torch_utils.py:126), so a batch splits into contiguous slices, one per GPU, with no exchange between
them: weights are replicated, each GPU runs its own plan on its own stream and hands its kept rows to the host with
one packed copy, and the host concatenates the per-image results in image order.  No collective is involved.

Two ways to run the shards:

* ``ShardedPredictor`` -- one worker *process* per device, created once and fed through queues (the reference's
  producer/consumer demo has the same shape: example/multiprocess_example.py:131-177, one session per OS process).
  Create it before the parent process touches a GPU.
* one rank per GPU under ``torch.distributed.run``: ``rank_slice`` gives a rank its images, ``gather_predicts_to_rank0``
  collects the per-image lists on rank 0 over the host backend (gloo), see bench.py.
"""
import traceback
from typing import Any, Callable, List, Optional, Sequence

import numpy as np


def shard_slices(batch: int, shards: int) -> List[slice]:
    """Contiguous slices of ``range(batch)``, sizes differing by at most one, empty slices when batch < shards."""
    if batch < 0 or shards <= 0:
        raise ValueError(f"cannot shard {batch} images {shards} ways")
    base, extra = divmod(batch, shards)
    out, start = [], 0
    for r in range(shards):
        n = base + (1 if r < extra else 0)
        out.append(slice(start, start + n))
        start += n
    return out


def rank_slice(batch: int, rank: int, world: int) -> slice:
    return shard_slices(batch, world)[rank]


def gather_predicts(parts: Sequence[Sequence[Optional[np.ndarray]]]) -> List[Optional[np.ndarray]]:
    """Per-shard ``predicts`` lists (``None`` = image without detections) -> one list in image order."""
    out: List[Optional[np.ndarray]] = []
    for part in parts:
        out.extend(part)
    return out


def gather_predicts_to_rank0(local: Sequence[Optional[np.ndarray]], group=None) -> Optional[List[Optional[np.ndarray]]]:
    """Inside an initialised ``torch.distributed`` job whose ranks hold consecutive batch slices: rank 0 receives the
    whole batch's list (image order = rank order), the other ranks ``None``.  Host data only; use a gloo group."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    bucket = [None] * world if rank == 0 else None
    dist.gather_object(list(local), bucket, dst=0, group=group)
    return gather_predicts(bucket) if rank == 0 else None


class ModelRunner:
    """Default per-device worker body: builds the model from ``model_factory()`` on its GPU and maps a float32
    (n,3,S,S) host array to the reference's ``predicts`` list through ``TaskLayer.predict``."""

    def __init__(self, model_factory: Callable[[], Any]):
        self.model_factory = model_factory

    def __call__(self, device: str):
        import torch
        dev = torch.device(device)
        torch.cuda.set_device(dev)
        model = self.model_factory()

        def run(images: np.ndarray) -> List[Optional[np.ndarray]]:
            if images.shape[0] == 0:
                return []
            x = torch.from_numpy(np.ascontiguousarray(images, dtype=np.float32)).to(dev)
            return model.predict({"batch_images": x})["predicts"]

        return run


def _worker(device: str, runner_factory, in_q, out_q) -> None:
    try:
        run = runner_factory(device)
        out_q.put(("ready", None))
    except BaseException:       # noqa: BLE001  (reported to the parent, which raises)
        out_q.put(("error", traceback.format_exc()))
        return
    while True:
        job = in_q.get()
        if job is None:
            return
        ticket, images = job
        try:
            out_q.put(("done", (ticket, run(images))))
        except BaseException:   # noqa: BLE001
            out_q.put(("error", traceback.format_exc()))
            return


class ShardedPredictor:
    """One worker process per device; ``predict(images)`` splits the batch into contiguous slices, runs them
    concurrently and returns the per-image list in the original order.

    ``runner_factory(device) -> callable(images ndarray) -> List[Optional[ndarray]]`` is executed inside each worker
    (it must be picklable: a module-level function, a ``functools.partial`` of one, or a ``ModelRunner``).
    ``devices`` are strings such as ``"cuda:0"``.  The workers are started with the ``spawn`` method, i.e. as fresh
    interpreters: construct the predictor before the calling process initialises a GPU."""

    def __init__(self, runner_factory, devices: Sequence[str], start_timeout: float = 600.0):
        import multiprocessing as mp
        if not devices:
            raise ValueError("no devices")
        ctx = mp.get_context("spawn")
        self.devices = list(devices)
        self._in = [ctx.Queue() for _ in self.devices]
        self._out = [ctx.Queue() for _ in self.devices]
        self._procs = [ctx.Process(target=_worker, args=(d, runner_factory, qi, qo), daemon=True)
                       for d, qi, qo in zip(self.devices, self._in, self._out)]
        for p in self._procs:
            p.start()
        self._ticket = 0
        try:
            for d, q in zip(self.devices, self._out):
                kind, payload = q.get(timeout=start_timeout)
                if kind != "ready":
                    raise RuntimeError(f"worker for {d} failed to start:\n{payload}")
        except BaseException:
            self.close()
            raise

    def predict(self, images: np.ndarray, timeout: float = 600.0) -> List[Optional[np.ndarray]]:
        images = np.asarray(images)
        slices = shard_slices(images.shape[0], len(self.devices))
        self._ticket += 1
        for q, sl in zip(self._in, slices):
            q.put((self._ticket, images[sl]))
        parts = []
        for d, q in zip(self.devices, self._out):
            kind, payload = q.get(timeout=timeout)
            if kind != "done":
                self.close()
                raise RuntimeError(f"worker for {d} failed:\n{payload}")
            ticket, part = payload
            if ticket != self._ticket:
                raise RuntimeError(f"worker for {d} answered request {ticket}, expected {self._ticket}")
            parts.append(part)
        return gather_predicts(parts)

    def close(self) -> None:
        for q, p in zip(self._in, self._procs):
            if p.is_alive():
                try:
                    q.put(None)
                except Exception:   # noqa: BLE001
                    pass
        for p in self._procs:
            p.join(timeout=30)
            if p.is_alive():
                p.terminate()
        self._procs = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def predict_sharded(model_factory: Callable[[], Any], images: np.ndarray, devices: Sequence[str]
                    ) -> List[Optional[np.ndarray]]:
    """One-shot form: start a worker per device, run one batch, shut down.  Serving code keeps a
    ``ShardedPredictor`` alive instead (model compilation and weight upload happen once per worker)."""
    with ShardedPredictor(ModelRunner(model_factory), devices) as sp:
        return sp.predict(images)
