"""Runs only the conv program of a model (no decode / NMS), `reps` times at one batch size, and writes the recorded op
list with its algorithmic flops / bytes: the subject of the per-layer counter passes (scratch/layer_counters.sh).
python3 scratch/run_program.py <model> <batch> <reps> <ops.json>"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd import models
from yolo_amd.init import seeded_state_dict

name, B, reps, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
p, m = models.create(name)
m.load_state_dict(seeded_state_dict(m, seed=0, mode="torch_default"))
dev = torch.device("cuda:0")
x = torch.rand(B, 3, p.img_size, p.img_size, device=dev)
prog = m.program()
prog.run(x)
for _ in range(reps - 1):
    prog.run(x)
torch.cuda.synchronize()
ops = []
for op in prog.ops:
    fl, by = prog.op_cost(op, B)
    kind = op[0]
    name_ = getattr(op[1], "name", kind) if kind in ("conv", "conv_cat2", "stem", "stem2", "bneck", "c3") else kind
    desc = ""
    if kind in ("conv", "conv_cat2"):
        d = op[1].desc
        o = op[3] if kind == "conv" else op[6]
        desc = f"k{d.ksize}s{d.stride} {d.cin}>{d.cout} @{o.h}x{o.w} tiling={op[1].tiling}"
    if kind == "c3":
        desc = f"C3 block (1x1 x2, 1x1 + 3x3 + shortcut, 1x1) {op[1].cin}>{op[1].ch}>{op[1].cout} @{op[3].h}x{op[3].w}"
    if kind == "bneck":
        desc = f"1x1 + 3x3 + shortcut {op[1].cin}>{op[1].cmid}>{op[1].cout} @{op[3].h}x{op[3].w}"
    launches = 1
    ops.append(dict(kind=kind, name=name_, desc=desc, flops=fl, bytes=by))
json.dump(dict(model=name, batch=B, reps=reps, ops=ops, conv_launches=prog.conv_launch_signature()), open(out, "w"))
