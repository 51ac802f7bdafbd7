"""CPU simulation of the sparse-graph NMS idea on the headline workload: adjacency sizes, fixpoint rounds, task counts."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bench
from oracle import graphs_ref
from yolo_amd.init import seeded_images

def spread10(v):
    v = v & 0x3ff
    v = (v | (v << 8)) & 0x00ff00ff
    v = (v | (v << 4)) & 0x0f0f0f0f
    v = (v | (v << 2)) & 0x33333333
    v = (v | (v << 1)) & 0x55555555
    return v

def analyse(boxes, scores, thr, coff=0.0, hi=639.0, SG=16):
    n = len(boxes)
    order = np.argsort(-scores, kind="stable")
    boxes = boxes[order]                      # rank order
    x1, y1, x2, y2 = boxes.T
    area = (x2 - x1) * (y2 - y1)
    # dense IoU (n<=8000 -> 64M floats ok in chunks)
    sup_lists = [None] * n
    maxlen = 0; tot = 0
    cnt = np.zeros(n, np.int64)
    adj = []
    for j0 in range(0, n, 1000):
        j1 = min(n, j0 + 1000)
        xx1 = np.maximum(x1[None, :], x1[j0:j1, None]); yy1 = np.maximum(y1[None, :], y1[j0:j1, None])
        xx2 = np.minimum(x2[None, :], x2[j0:j1, None]); yy2 = np.minimum(y2[None, :], y2[j0:j1, None])
        inter = np.maximum(0, xx2 - xx1) * np.maximum(0, yy2 - yy1)
        with np.errstate(invalid="ignore", divide="ignore"):
            ovr = inter / (area[None, :] + area[j0:j1, None] - inter)
        m = ovr > thr
        for j in range(j0, j1):
            s = np.nonzero(m[j - j0, :j])[0]      # higher-score suppressor candidates
            adj.append(s)
            cnt[j] = len(s)
    # synchronous fixpoint
    status = np.zeros(n, np.int8)
    rounds = 0
    und_hist = []
    while (status == 0).any():
        rounds += 1
        new = status.copy()
        for j in np.nonzero(status == 0)[0]:
            st = status[adj[j]]
            if (st == 1).any(): new[j] = 2
            elif (st == 2).all(): new[j] = 1
        status = new
        und_hist.append(int((status == 0).sum()))
    kept = int((status == 1).sum())
    # greedy check
    keep = np.ones(n, bool)
    for j in range(n):
        if keep[adj[j]].any() if len(adj[j]) else False:
            keep[j] = False
    assert (keep == (status == 1)).all()
    # morton subgroups
    qs = 1024.0 / (hi + 1)
    qx = np.clip(((x1 + x2 - 2 * coff) * 0.5 * qs), 0, 1023).astype(np.uint32); qy = np.clip(((y1 + y2 - 2 * coff) * 0.5 * qs), 0, 1023).astype(np.uint32)
    mk = spread10(qx) | (spread10(qy) << 1)
    mo = np.argsort(-(mk.astype(np.int64) * 65536 + np.arange(n)), kind="stable")   # descending like the kernel
    mb = boxes[mo]
    nsg = (n + SG - 1) // SG
    bb = np.zeros((nsg, 4), np.float32)
    for g in range(nsg):
        c = mb[g * SG:(g + 1) * SG]
        bb[g] = (c[:, 0].min(), c[:, 1].min(), c[:, 2].max(), c[:, 3].max())
    ov = (bb[:, None, 2] > bb[None, :, 0]) & (bb[:, None, 0] < bb[None, :, 2]) & (bb[:, None, 3] > bb[None, :, 1]) & (bb[:, None, 1] < bb[None, :, 3])
    tasks = int(np.triu(ov).sum())
    return dict(n=n, kept=kept, adj_mean=float(cnt.mean()), adj_max=int(cnt.max()), rounds=rounds, und=und_hist[:12], nsg=nsg, tasks=tasks)

which = sys.argv[1] if len(sys.argv) > 1 else "default"
p, m, sd = bench.build_model("v5s", which)
if which == "spread":
    C = 5 + p.class_num
    for k in sd:
        if k.startswith("body_net.yolo_head_P") and k.endswith(".bias"):
            bias = sd[k].view(3, C); bias[:, 4] = -5.0
ref = graphs_ref.model_ref("v5s")
imgs = torch.from_numpy(seeded_images(1, p.img_size, seed=0))
with torch.no_grad():
    raws = [t.numpy() for t in ref["body"](sd, imgs)]
pred = ref["decode"](raws, p, True)[0]
cx, cy, w, h = pred[:, 0], pred[:, 1], pred[:, 2], pred[:, 3]
b = np.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], 1).clip(0, 639).astype(np.float32)
cls = pred[:, 5:].argmax(1); sc = pred[:, 4] * pred[:, 5:].max(1)
ok = sc >= p.conf_threshold
for c in np.unique(cls[ok]):
    sel = ok & (cls == c)
    print(c, analyse(b[sel] + np.float32(c * 4096.0), sc[sel], np.float32(p.nms_threshold), coff=c * 4096.0))
