"""Class histogram of the candidates of the headline workload (CPU oracle, a few images)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bench
from oracle import graphs_ref
from yolo_amd.init import seeded_images
p, m, sd = bench.build_model("v5s", "default")
ref = graphs_ref.model_ref("v5s")
imgs = torch.from_numpy(seeded_images(4, p.img_size, seed=0))
with torch.no_grad():
    raws = [t.numpy() for t in ref["body"](sd, imgs)]
pred = ref["decode"](raws, p, True)
for b in range(pred.shape[0]):
    r = pred[b]
    cls = r[:, 5:].argmax(1)
    sc = r[:, 4] * r[:, 5:].max(1)
    ok = sc >= p.conf_threshold
    h = np.bincount(cls[ok], minlength=20)
    print(b, ok.sum(), sorted(h.tolist(), reverse=True))
