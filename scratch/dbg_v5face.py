import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from yolo_amd import models
from yolo_amd.init import seeded_state_dict, seeded_images
from oracle import graphs_ref
def rel(a, b): return float(np.linalg.norm(a - b) / np.linalg.norm(b))
for name in sys.argv[1:]:
    p, m = models.create(name, 640)
    sd = seeded_state_dict(m, seed=32, mode="spread", **models.init_options(name))
    m.load_state_dict(sd)
    x = seeded_images(1, 640, seed=32)
    raws = [r.cpu().numpy().transpose(0, 3, 1, 2) for r in m.raw_heads(torch.from_numpy(x).cuda())]
    torch.set_num_threads(16)
    with torch.no_grad():
        want = [t.numpy() for t in graphs_ref.model_ref(name)["body"](sd, torch.from_numpy(x))]
    for i, (g, w) in enumerate(zip(raws, want)):
        print(name, i, "rel", rel(g, w), "std", float(w.std()), "max abs err", float(np.abs(g - w).max()), "mean abs err", float(np.abs(g - w).mean()))
