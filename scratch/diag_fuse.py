"""Raw-logit error of a model at full size (the full golden's seed) with the fused forms / the tuning table switched off
one at a time: which kernel the error comes from.  python scratch/diag_fuse.py [name]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import load_golden
from test_gpu_models import _model
from oracle import graphs_ref
from yolo_amd import engine, tuning
from yolo_amd.init import seeded_images
name = sys.argv[1] if len(sys.argv) > 1 else "v5face"
dev = torch.device("cuda:0")
g = load_golden(f"{name}_full.npz")
size, seed = int(g["size"]), int(g["seed"])
imgs_np = seeded_images(2, size, seed=seed)
imgs = torch.from_numpy(imgs_np).to(dev)
ref = graphs_ref.model_ref(name)
torch.set_num_threads(16)
want = None
def run(tag):
    global want
    p, m, sd = _model(name, size, seed)
    if want is None:
        with torch.no_grad():
            want = [t.numpy() for t in ref["body"](sd, torch.from_numpy(imgs_np))]
    raws = [r.cpu().numpy().transpose(0, 3, 1, 2) for r in m.raw_heads(imgs)]
    kinds = {}
    for op in m.program().ops:
        kinds[op[0]] = kinds.get(op[0], 0) + 1
    print(f"{tag:28s}", " ".join(f"{np.linalg.norm((a - b).astype(np.float64)) / np.linalg.norm(b):.3e}" for a, b in zip(raws, want)), kinds, flush=True)
run("default")
for k in list(engine.FUSE):
    engine.FUSE[k] = False
    run(f"FUSE[{k}] off")
    engine.FUSE[k] = True
for k in engine.FUSE: engine.FUSE[k] = False
run("all fused forms off")
tuning.enabled = False
run("... and tuning table off")
for k in engine.FUSE: engine.FUSE[k] = True
run("tuning table off only")
