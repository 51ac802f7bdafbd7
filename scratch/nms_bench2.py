"""NMS stage timing on a headline-like candidate mix (per image: buckets of 8000, 6400, 6400, 1600, 1600, 400, 400, 400)."""
import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd import torch_utils as tu
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rng = np.random.RandomState(0)
planes = [(80, 8, 10, 13, 0), (80, 8, 16, 30, 1), (80, 8, 33, 23, 2), (40, 16, 30, 61, 2), (40, 16, 62, 45, 3), (40, 16, 59, 119, 4),
          (20, 32, 116, 90, 5), (20, 32, 156, 198, 6), (20, 32, 373, 326, 7)]
rows = []
for g, st, aw, ah, c in planes:
    ys, xs = np.meshgrid(np.arange(g), np.arange(g), indexing="ij")
    n = g * g
    s = rng.uniform(0.8, 1.25, (n, 2))
    box = np.stack((xs.ravel() * st + st / 2 + rng.uniform(-1, 1, n), ys.ravel() * st + st / 2 + rng.uniform(-1, 1, n), aw * s[:, 0], ah * s[:, 1]), -1)
    obj = 0.5 + rng.uniform(-5e-3, 5e-3, (n, 1))
    cls = np.full((n, 20), 0.4); cls[:, c] = 0.6
    rows.append(np.concatenate((box, obj, cls), -1))
pred = np.concatenate(rows, 0).astype(np.float32)[None]
pred = torch.from_numpy(pred).cuda().repeat(B, 1, 1).contiguous()
A = pred.shape[1]
def run(tag):
    for _ in range(2):
        rows_, counts, _ = tu.nms_batched(pred, 640.0, "cxcywh", 0.1, 0.3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        rows_, counts, _ = tu.nms_batched(pred, 640.0, "cxcywh", 0.1, 0.3)
    torch.cuda.synchronize()
    print("%-28s %.3f ms per call, kept %d" % (tag, (time.perf_counter() - t0) / 5 * 1e3, int(counts[0])), flush=True)
only = sys.argv[2] if len(sys.argv) > 2 else None
for tag, dbg in (("graph full", 0), ("sweep full", 4), ("sorts only", 1), ("graph: lists only", 8), ("graph: lists+rounds", 16)):
    if only is not None and str(dbg) != only:
        continue
    os.environ["YAMD_NMS_DBG"] = str(dbg)
    run(tag)
