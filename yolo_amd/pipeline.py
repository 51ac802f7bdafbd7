"""Throughput plumbing around ``TaskModel.predict_device`` for callers that process a stream of batches: nothing here
changes a result, it only decides on which stream work runs and when the host waits.

``HostSink`` hands the kept rows of a step to the host one step late, so that the copies run under the next step's
kernels.  ``TwoStreamStep`` additionally runs a step's NMS on a second stream while the launch stream already computes
the next step's conv graph: the NMS sweep is latency- and ALU-bound, the convs are memory- and MFMA-bound, and the two
fill each other's gaps.  The reference's ``predict`` is synchronous per batch (luoyang/yolov5/YoloV5.py:112-127); this is
what a serving loop puts around it.
"""
import torch

from . import lib as L
from .torch_utils import Arena


class HostSink:
    """Device->host hand-over of the detections, pipelined one step deep.

    Per step the kept rows are packed back to back on the device (compute stream); a second stream then
    copies the (B+1) offsets and, once the host knows the total, the kept rows into pinned memory.  The host
    only waits for step i's offsets after it has enqueued step i+1's kernels, so the copies and the host-side
    launch work overlap the next step's compute.  Two device/host buffer sets alternate; ``drain`` completes
    the last step.  Every step's rows are on the host when ``drain`` returns."""

    def __init__(self, batch, a, width, device):
        self.L = L
        self.width = width
        self.copy_stream = torch.cuda.Stream(device=device)
        self.slots = []
        for _ in range(2):
            self.slots.append(dict(
                compact=torch.empty((batch * a, width), dtype=torch.float32, device=device),
                offsets=torch.empty((batch + 1,), dtype=torch.int64, device=device),
                h_offsets=torch.empty((batch + 1,), dtype=torch.int64).pin_memory(),
                h_rows=torch.empty((batch * a, width), dtype=torch.float32).pin_memory(),
                packed=torch.cuda.Event(), offsets_done=torch.cuda.Event(), rows_done=None))
        self.n = 0
        self.pending = None
        self.last_total = 0
        self.h_offsets = self.slots[0]["h_offsets"]
        self.last_slot = None

    def submit(self, rows, counts):
        """Enqueue packing + the offsets copy of this step, then complete the previous step's row copy."""
        L = self.L
        slot = self.slots[self.n % 2]
        self.n += 1
        cur = torch.cuda.current_stream()
        if slot["rows_done"] is not None:
            cur.wait_event(slot["rows_done"])          # the copy that last read this slot's device buffer
        B, max_out, W = rows.shape
        L.check(L.LIB.yamd_compact_rows(rows.data_ptr(), counts.data_ptr(), B, max_out, W, slot["compact"].data_ptr(),
                                        slot["offsets"].data_ptr(), L.current_stream()), "compact_rows")
        slot["packed"].record(cur)
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(slot["packed"])
            slot["h_offsets"].copy_(slot["offsets"], non_blocking=True)
            slot["offsets_done"].record(self.copy_stream)
        prev, self.pending = self.pending, slot
        if prev is not None:
            self._finish(prev)

    def _finish(self, slot):
        slot["offsets_done"].synchronize()
        total = int(slot["h_offsets"][-1])
        with torch.cuda.stream(self.copy_stream):
            if total:
                slot["h_rows"][:total].copy_(slot["compact"][:total], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
            slot["rows_done"] = ev
        self.last_total = total
        self.h_offsets = slot["h_offsets"]
        self.last_slot = slot

    def finished(self):
        """Detections of the step whose hand-over was completed last (the step before the latest ``submit``, or the
        latest one after ``drain``): the reference's per-image list, ``None`` for an image without detections.  Waits
        for that step's row copy; the arrays are views of pinned buffers that the step after next overwrites."""
        slot = self.last_slot
        slot["rows_done"].synchronize()
        off = slot["h_offsets"].numpy()
        rows = slot["h_rows"].numpy()
        return [None if off[b + 1] == off[b] else rows[off[b]:off[b + 1]] for b in range(len(off) - 1)]

    def drain(self):
        if self.pending is not None:
            self._finish(self.pending)
            self.pending = None
        self.copy_stream.synchronize()
        return self.last_total


class TwoStreamStep:
    """One step with its NMS on a second stream: conv graph + decode of step i+1 (launch stream) run while NMS, row
    packing and the copies of step i are still in flight.  Two decode / NMS arenas alternate, guarded by events."""

    def __init__(self, model, device):
        self.model = model
        self.arenas = [Arena(), Arena()]
        self.free = [None, None]
        self.post = torch.cuda.Stream(device=device)
        self.n = 0

    def __call__(self, images, sink):
        k = self.n % 2
        self.n += 1
        m, arena = self.model, self.arenas[k]
        cur = torch.cuda.current_stream()
        raws = m.raw_heads(images)
        if self.free[k] is not None:
            cur.wait_event(self.free[k])          # the NMS that last read this arena's decoded tensor
        pred = m.decode_predict(raws, arena)
        ready = torch.cuda.Event()
        ready.record(cur)
        with torch.cuda.stream(self.post):
            self.post.wait_event(ready)
            rows, counts = m.nms_predict(pred, arena)
            sink.submit(rows, counts)
            done = torch.cuda.Event()
            done.record(self.post)
            self.free[k] = done
