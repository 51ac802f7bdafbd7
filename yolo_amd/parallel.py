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


class SharedBatch:
    """A block of host memory shared between the parent and one worker through a file in /dev/shm (plain ``mmap`` on
    both sides; no pickling, no pipe).  The parent writes a slice of images into it, the worker reads them from the
    same pages into a page-locked staging buffer of its own, from which the copy to the device is one DMA transfer.
    (Registering the shared mapping itself with the GPU runtime, hipHostRegister, would save that hop; the runtime
    of ROCm 7.2 refused copies from a mapping registered that way with hipErrorInvalidValue.)"""

    def __init__(self, path: str, nbytes: int, create: bool):
        import mmap
        import os
        self.path, self.nbytes = path, int(nbytes)
        fd = os.open(path, (os.O_CREAT | os.O_EXCL | os.O_RDWR) if create else os.O_RDWR, 0o600)
        try:
            if create:
                os.ftruncate(fd, self.nbytes)
            self._map = mmap.mmap(fd, self.nbytes)
        finally:
            os.close(fd)
        self.owner = create

    @staticmethod
    def new_path(tag: str) -> str:
        import os
        import tempfile
        import uuid
        root = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
        return os.path.join(root, f"yolo_amd_{os.getpid()}_{tag}_{uuid.uuid4().hex[:12]}")

    def array(self, shape, dtype=np.float32) -> np.ndarray:
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        if n > self.nbytes:
            raise ValueError(f"{tuple(shape)} {np.dtype(dtype)} needs {n} bytes, the block holds {self.nbytes}")
        return np.frombuffer(self._map, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def close(self) -> None:
        import os
        try:
            self._map.close()
        except (BufferError, ValueError):       # a numpy view is still alive: the pages go with the last reference
            pass
        if self.owner:
            try:
                os.unlink(self.path)
            except FileNotFoundError:
                pass
            self.owner = False


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

        staging = {}

        def run(images: np.ndarray) -> List[Optional[np.ndarray]]:
            if images.shape[0] == 0:
                return []
            # shared block -> page-locked staging buffer (kept between calls) -> device
            src = torch.from_numpy(np.ascontiguousarray(images, dtype=np.float32))
            buf = staging.get("pinned")
            if buf is None or buf.numel() < src.numel():
                buf = torch.empty(src.numel(), dtype=torch.float32).pin_memory()
                staging["pinned"] = buf
            host = buf[:src.numel()].view(src.shape)
            host.copy_(src)
            x = host.to(dev, non_blocking=True)
            return model.predict({"batch_images": x})["predicts"]

        return run


def _worker(device: str, runner_factory, in_q, out_q) -> None:
    try:
        run = runner_factory(device)
        out_q.put(("ready", None))
    except BaseException:       # noqa: BLE001  (reported to the parent, which raises)
        out_q.put(("error", traceback.format_exc()))
        return
    block: Optional[SharedBatch] = None
    while True:
        job = in_q.get()
        if job is None:
            if block is not None:
                block.close()
            return
        ticket, path, nbytes, shape, dtype = job
        try:
            if block is None or block.path != path:
                if block is not None:
                    block.close()
                block = SharedBatch(path, nbytes, create=False)
            images = block.array(shape, dtype)
            result = run(images)
            del images
            out_q.put(("done", (ticket, result)))
        except BaseException:   # noqa: BLE001
            out_q.put(("error", traceback.format_exc()))
            return


class ShardedPredictor:
    """One worker process per device; ``predict(images)`` splits the batch into contiguous slices, runs them
    concurrently and returns the per-image list in the original order.

    Images reach a worker through a block of shared host memory (``SharedBatch``: the parent writes the slice, the
    worker copies it through its own page-locked buffer to its GPU), not through the queue; the queues carry only
    the job header and the kept rows.  ``input_views(n, shape)`` hands out the blocks themselves, for a producer that
    can write its images where the workers read them (no copy in the parent at all); ``predict_views()`` then runs them.

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
        self._blocks: List[Optional[SharedBatch]] = [None] * len(self.devices)
        self._retired: List[SharedBatch] = []
        self._views = None
        try:
            for d, q in zip(self.devices, self._out):
                kind, payload = q.get(timeout=start_timeout)
                if kind != "ready":
                    raise RuntimeError(f"worker for {d} failed to start:\n{payload}")
        except BaseException:
            self.close()
            raise

    def input_views(self, batch: int, image_shape: Sequence[int], dtype=np.float32) -> List[np.ndarray]:
        """Per worker, a writable ``(n_r, *image_shape)`` array in that worker's shared block, ``n_r`` the size of its
        contiguous slice of a batch of ``batch`` images (rank order = image order).  Fill them, then ``predict_views()``."""
        slices = shard_slices(batch, len(self.devices))
        views = []
        for r, sl in enumerate(slices):
            shape = (sl.stop - sl.start,) + tuple(int(v) for v in image_shape)
            need = int(np.prod(shape)) * np.dtype(dtype).itemsize
            blk = self._blocks[r]
            if blk is None or blk.nbytes < need:
                if blk is not None:
                    self._retired.append(blk)      # unlinked once the worker has moved to the new block
                blk = SharedBatch(SharedBatch.new_path(f"w{r}"), max(need, 1), create=True)
                self._blocks[r] = blk
            views.append(blk.array(shape, dtype))
        self._views = (views, dtype)
        return views

    def predict_views(self, timeout: float = 600.0) -> List[Optional[np.ndarray]]:
        if self._views is None:
            raise RuntimeError("input_views() has not been called")
        views, dtype = self._views
        self._ticket += 1
        for q, blk, v in zip(self._in, self._blocks, views):
            q.put((self._ticket, blk.path, blk.nbytes, v.shape, np.dtype(dtype).str))
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
        for blk in self._retired:
            blk.close()
        self._retired = []
        return gather_predicts(parts)

    def predict(self, images: np.ndarray, timeout: float = 600.0) -> List[Optional[np.ndarray]]:
        images = np.asarray(images)
        views = self.input_views(images.shape[0], images.shape[1:], images.dtype)
        for v, sl in zip(views, shard_slices(images.shape[0], len(self.devices))):
            np.copyto(v, images[sl])
        return self.predict_views(timeout)

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
        self._views = None
        for blk in list(self._retired) + [b for b in self._blocks if b is not None]:
            blk.close()
        self._retired, self._blocks = [], [None] * len(self.devices)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001
            pass


def predict_sharded(model_factory: Callable[[], Any], images: np.ndarray, devices: Sequence[str]
                    ) -> List[Optional[np.ndarray]]:
    """One-shot form: start a worker per device, run one batch, shut down.  Serving code keeps a
    ``ShardedPredictor`` alive instead (model compilation and weight upload happen once per worker)."""
    with ShardedPredictor(ModelRunner(model_factory), devices) as sp:
        return sp.predict(images)
