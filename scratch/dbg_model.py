import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from yolo_amd import models
from yolo_amd.init import seeded_images, seeded_state_dict
name = sys.argv[1]
g = np.load(f"tests/golden/{name}_small.npz")
size, seed = int(g["size"]), int(g["seed"])
p, m = models.create(name, size)
m.load_state_dict(seeded_state_dict(m, seed=seed, mode="spread", **models.init_options(name)))
imgs = torch.from_numpy(seeded_images(int(g["batch"]), size, seed=seed)).cuda()
raws = [r.cpu().numpy().transpose(0, 3, 1, 2) for r in m.raw_heads(imgs)]
for i, r in enumerate(raws):
    w = g[f"raw{i}"]
    print(i, "rel", np.linalg.norm(r - w) / np.linalg.norm(w), "std", w.std(), "maxabs", np.abs(r - w).max())
pred = m._forward(imgs)[0].cpu().numpy()
w = g["pred"]
d = np.abs(pred - w)
for c in range(6):
    print("col", c, "max", d[..., c].max(), "mean", d[..., c].mean(), "want range", w[..., c].min(), w[..., c].max())
tol = 0.5 + 0.01 * np.abs(w[..., :4]); tol[..., 2:4] = 0.5 + 0.05 * np.abs(w[..., 2:4])
bad = np.argwhere(d[..., :4] > tol)
print(len(bad)); print(bad[:20])
for b, a, c in bad[:10]:
    print(pred[b, a, :5], w[b, a, :5])
if w.shape[-1] > 30:
    P = 17
    dk = d[..., 5:5 + 2 * P]
    print("kp max", dk.max(), "mean", dk.mean(), "frac>1px", (dk > 1).mean(), "frac>3px", (dk > 3).mean())
    ds = d[..., 5 + 2 * P:]
    print("kpconf/cls max", ds.max(), "frac>2e-2", (ds > 2e-2).mean())
    lv = np.cumsum([0] + [3 * (size // s) ** 2 for s in p.strides])
    for i in range(3):
        print("level", i, "kp max", dk[:, lv[i]:lv[i + 1]].max(), "mean", dk[:, lv[i]:lv[i + 1]].mean())
