"""Prints, per model, the matched-row fraction of predict() against the reference golden rows (the quantity
tests/test_gpu_models.py::test_small_vs_reference_golden thresholds).  python scratch/match_fractions.py"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from conftest import load_golden, split_rows
import test_gpu_models as T
from test_gpu_v5_e2e import _match_rows
from yolo_amd.init import seeded_images
dev = torch.device("cuda:0")
for name in T.NAMES:
    g = load_golden(f"{name}_small.npz")
    size, seed = int(g["size"]), int(g["seed"])
    p, m, _ = T._model(name, size, seed)
    imgs = torch.from_numpy(seeded_images(int(g["batch"]), size, seed=seed)).to(dev)
    out = m.predict({"batch_images": imgs})["predicts"]
    want = split_rows(g["counts"], g["rows"])
    fr, cnt = [], []
    for a, b in zip(out, want):
        if b is None or b.shape[0] == 0:
            continue
        if name in ("v6face", "v6", "v6pose", "v6facen", "v6m", "v6l", "v6facem", "v6posel"):
            fr.append(T._match_rows_abs(a, b, 3.0))
        else:
            fr.append(_match_rows(a, b))
        cnt.append(abs(a.shape[0] - b.shape[0]) / max(1, b.shape[0]))
    print(f"{name:10s} match {min(fr) if fr else 1.0:.3f}  count diff {max(cnt) if cnt else 0:.3f}  rows {[0 if b is None else b.shape[0] for b in want]}", flush=True)
# v5s golden
g = load_golden("v5s_64.npz")
