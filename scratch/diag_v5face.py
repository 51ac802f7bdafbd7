"""Where the decoded boxes of a model at full size leave the fp32 reference: per column, and with the conv error taken out
(the oracle decode applied to the GPU's own raw logits).  python scratch/diag_v5face.py [name]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import load_golden
from test_gpu_models import _model
from oracle import graphs_ref
from yolo_amd.init import seeded_images
from test_gpu_models import POINTS
names = sys.argv[1:] or ["v5face"]
dev = torch.device("cuda:0")
for name in names:
    print("==", name)
    g = load_golden(f"{name}_full.npz")
    size, seed = int(g["size"]), int(g["seed"])
    p, m, sd = _model(name, size, seed)
    imgs_np = seeded_images(2, size, seed=seed)
    imgs = torch.from_numpy(imgs_np).to(dev)
    raws = [r.cpu().numpy().transpose(0, 3, 1, 2) for r in m.raw_heads(imgs)]
    ref = graphs_ref.model_ref(name)
    torch.set_num_threads(16)
    with torch.no_grad():
        want_raws = [t.numpy() for t in ref["body"](sd, torch.from_numpy(imgs_np))]
    for i, (a, b) in enumerate(zip(raws, want_raws)):
        d = (a - b).astype(np.float64)
        print(f"raw{i}: rel L2 {np.linalg.norm(d) / np.linalg.norm(b):.4e}  rms want {b.std():.3f}  rms err {d.std():.4f}  max err {np.abs(d).max():.3f}")
    pred = m._forward(imgs)[0].cpu().numpy()
    want = ref["decode"](want_raws, p, False)
    own = ref["decode"](raws, p, False)           # the oracle decode of the GPU's own logits
    for tag, w in (("vs fp32 graph", want), ("vs oracle decode of the GPU logits", own)):
        bad = np.abs(pred[..., :4] - w[..., :4]) > 0.5 + 0.01 * np.abs(w[..., :4])
        print(tag, "box mismatch per column", bad.reshape(-1, 4).mean(0), "overall", bad.mean())
        A = pred.shape[1]
        n3 = 3 * (size // 8) ** 2; n4 = 3 * (size // 16) ** 2
        for lvl, (s0, s1) in enumerate(((0, n3), (n3, n3 + n4), (n3 + n4, A))):
            print(f"   level {lvl}: {bad[:, s0:s1].mean():.4f}   |d| q99.8 {np.quantile(np.abs(pred[:, s0:s1, :4] - w[:, s0:s1, :4]), 0.998):.3f} px")
    P = POINTS.get(name, 0)
    def need(gv, wv):
        d = np.abs(gv - wv) - 0.5
        k = np.where(d > 0, d / np.maximum(np.abs(wv), 1e-6), 0.0)
        return float(np.quantile(k, 0.998))
    print(f"   needed relative allowance (0.5 px + k |want| for 99.8 %): xy {need(pred[..., :2], want[..., :2]):.4f}  wh {need(pred[..., 2:4], want[..., 2:4]):.4f}"
          + (f"  pts {need(pred[..., 5:5 + 2 * P], want[..., 5:5 + 2 * P]):.4f}" if P else ""))
    k1 = 5 + 2 * P
    ds = np.abs(np.concatenate((pred[..., 4:5], pred[..., k1:]), -1) - np.concatenate((want[..., 4:5], want[..., k1:]), -1))
    print(f"   score |d| q99.8 {np.quantile(ds, 0.998):.4f} max {ds.max():.4f}")
