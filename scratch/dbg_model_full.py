import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from yolo_amd import models
from yolo_amd.init import seeded_images, seeded_state_dict
from oracle import graphs_ref
name = sys.argv[1]
g = np.load(f"tests/golden/{name}_full.npz")
size, seed = int(g["size"]), int(g["seed"])
p, m = models.create(name, size)
sd = seeded_state_dict(m, seed=seed, mode="spread", **models.init_options(name))
m.load_state_dict(sd)
imgs_np = seeded_images(1, size, seed=seed)
imgs = torch.from_numpy(imgs_np).cuda()
raws = [r.cpu().numpy().transpose(0, 3, 1, 2) for r in m.raw_heads(imgs)]
ref = graphs_ref.model_ref(name)
torch.set_num_threads(16)
with torch.no_grad():
    want_raws = [t.numpy() for t in ref["body"](sd, torch.from_numpy(imgs_np))]
for i, (r, w) in enumerate(zip(raws, want_raws)):
    print(i, "rel", np.linalg.norm(r - w) / np.linalg.norm(w), "std", w.std(), "maxabs", np.abs(r - w).max(), "absmax", np.abs(w).max())
pred = m._forward(imgs)[0].cpu().numpy()
w = ref["decode"](want_raws, p, False)
d = np.abs(pred - w)
tol = 0.5 + 0.01 * np.abs(w[..., :4]); tol[..., 2:4] = 0.5 + 0.05 * np.abs(w[..., 2:4])
for c in range(4):
    print("col", c, "bad frac", (d[..., c] > tol[..., c]).mean(), "max", d[..., c].max())
lv = np.cumsum([0] + [3 * (size // s) ** 2 for s in p.strides])
for i in range(3):
    sl = slice(lv[i], lv[i + 1])
    print("level", i, "bad frac", (d[:, sl, :4] > tol[:, sl]).mean())
