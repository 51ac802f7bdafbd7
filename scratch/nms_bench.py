import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd import torch_utils as tu
B = 256
g = 80
ys, xs = np.meshgrid(np.arange(g), np.arange(g), indexing="ij")
rng = np.random.RandomState(0)
box = np.stack((xs.ravel() * 8 + 4.0, ys.ravel() * 8 + 4.0, np.full(g * g, 16.0), np.full(g * g, 30.0)), -1)[None]
A = box.shape[1]
obj = 0.5 + rng.uniform(-1e-3, 1e-3, (1, A, 1))
cls = np.concatenate((np.full((1, A, 1), 0.6), np.full((1, A, 3), 0.4)), -1)
pred = np.concatenate((box, obj, cls), -1).astype(np.float32)
pred = torch.from_numpy(pred).cuda().repeat(B, 1, 1).contiguous()
for _ in range(2):
    rows, counts, _ = tu.nms_batched(pred, 640.0, "cxcywh", 0.1, 0.3)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    rows, counts, _ = tu.nms_batched(pred, 640.0, "cxcywh", 0.1, 0.3)
torch.cuda.synchronize()
print("n=%d per image, B=%d: %.3f ms per call, kept %d" % (A, B, (time.perf_counter() - t0) / 5 * 1e3, int(counts[0])))
