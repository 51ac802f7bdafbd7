"""CPU model of the NMS sweep's work on the headline input (v5s, 640, seeded torch-default weights, one image): how many
(kept box, group) steps, bounding-box tests and matrix columns a bucket costs, and what regrouping the undecided boxes
every M tiles, or more size classes in the grouping, would save.  Counts only; the kept set is checked against the plain
greedy loop.  python scratch/nms_sweep_sim.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from oracle import graphs_ref

def spread10(v):
    v = v & 0x3ff
    v = (v | (v << 8)) & 0x00ff00ff
    v = (v | (v << 4)) & 0x0f0f0f0f
    v = (v | (v << 2)) & 0x33333333
    v = (v | (v << 1)) & 0x55555555
    return v

def cells(b, hi, classes):
    qs = 1024.0 / (hi + 1.0)
    qx = np.clip((b[:, 0] + b[:, 2]) * 0.5 * qs, 0, 1023).astype(np.int64) >> 4
    qy = np.clip((b[:, 1] + b[:, 3]) * 0.5 * qs, 0, 1023).astype(np.int64) >> 4
    side2 = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]) * qs * qs
    if classes == 2:
        sc = (side2 > 96.0 * 96.0).astype(np.int64)
    else:   # four classes at sides 48, 96, 192 (in 1/1024 of the image)
        sc = (side2 > 48.0 ** 2).astype(np.int64) + (side2 > 96.0 ** 2) + (side2 > 192.0 ** 2)
    return spread10(qx) | (spread10(qy) << 1) | (sc << 12)

def iou_gt(k, bx, thr):
    w = np.maximum(0, np.minimum(k[2], bx[:, 2]) - np.maximum(k[0], bx[:, 0]))
    h = np.maximum(0, np.minimum(k[3], bx[:, 3]) - np.maximum(k[1], bx[:, 1]))
    inter = w * h
    uni = (k[2] - k[0]) * (k[3] - k[1]) + (bx[:, 2] - bx[:, 0]) * (bx[:, 3] - bx[:, 1]) - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        return inter / uni > thr

def sweep(box, thr, hi, regroup_every=0, classes=2, regroup_at=()):
    """box: (n,4) xyxy in descending score order.  Returns (kept mask, counts)."""
    n = box.shape[0]
    removed = np.zeros(n, bool)
    kept = np.zeros(n, bool)
    cell = cells(box, hi, classes)
    def group(members):          # members: ranks still in play, any order -> list of groups (arrays of ranks) in Morton order
        order = members[np.argsort(cell[members], kind="stable")]
        return [order[i:i + 64] for i in range(0, len(order), 64)]
    groups = group(np.arange(n))
    nt = (n + 63) // 64
    cnt = dict(c_steps=0, bb_tests=0, a_cols=0, lanes_used=0, regroups=0)
    for t in range(nt):
        if (regroup_every and t and t % regroup_every == 0) or t in regroup_at:
            alive_later = np.nonzero(~removed & (np.arange(n) >= t * 64))[0]
            groups = group(alive_later)
            cnt["regroups"] += 1
        r0, r1 = t * 64, min(n, t * 64 + 64)
        cnt["a_cols"] += int((~removed[r0:r1]).sum())
        kl = []
        for r in range(r0, r1):
            if removed[r]:
                continue
            kept[r] = True
            kl.append(r)
            later = np.arange(r + 1, r1)
            if len(later):
                removed[later[iou_gt(box[r], box[later], thr) & ~removed[later]]] = True
        if r1 >= n or not kl:
            continue
        kb = box[kl]
        for g in groups:
            if g.max() < r1:
                continue
            cnt["bb_tests"] += 1
            gb = box[g]
            bb = (gb[:, 0].min(), gb[:, 1].min(), gb[:, 2].max(), gb[:, 3].max())
            rel = (kb[:, 2] > bb[0]) & (kb[:, 0] < bb[2]) & (kb[:, 3] > bb[1]) & (kb[:, 1] < bb[3])
            if not rel.any():
                continue
            todo = (g >= r1) & ~removed[g]
            if not todo.any():
                continue
            cnt["c_steps"] += int(rel.sum())
            cnt["lanes_used"] += int(rel.sum()) * int(todo.sum())
            for k in kb[rel]:
                s = iou_gt(k, gb, thr) & todo
                removed[g[s]] = True
    return kept, cnt

def main():
    p, model, _ = bench.build_model("v5s", "torch_default")
    imgs = bench.device_images.__wrapped__(1, 640, 0, torch.device("cpu")) if hasattr(bench.device_images, "__wrapped__") else None
    g = torch.Generator(); g.manual_seed(1000003)
    imgs = torch.empty((1, 3, 640, 640)).uniform_(0.0, 1.0, generator=g)
    ref = graphs_ref.model_ref("v5s")
    sd = {k: v.float() for k, v in model.state_dict().items()}
    with torch.no_grad():
        raws = [t.numpy() for t in ref["body"](sd, imgs)]
    pred = ref["decode"](raws, p, True)[0]
    nc = p.class_num
    cls = pred[:, 5:5 + nc].argmax(1)
    score = pred[:, 4] * pred[np.arange(pred.shape[0]), 5 + cls]
    ok = score >= p.conf_threshold
    hi = 639.0
    x1 = np.clip(pred[:, 0] - 0.5 * pred[:, 2], 0, hi); y1 = np.clip(pred[:, 1] - 0.5 * pred[:, 3], 0, hi)
    x2 = np.clip(pred[:, 0] + 0.5 * pred[:, 2], 0, hi); y2 = np.clip(pred[:, 1] + 0.5 * pred[:, 3], 0, hi)
    allb = np.stack((x1, y1, x2, y2), 1).astype(np.float32)
    print("candidates", int(ok.sum()), "classes with boxes", np.unique(cls[ok]).size, "nms thr", p.nms_threshold)
    tot = {}
    for c in np.unique(cls[ok])[:3]:
        idx = np.nonzero(ok & (cls == c))[0]
        order = idx[np.argsort(-score[idx], kind="stable")]
        box = allb[order]
        base = None
        nt = (len(order) + 63) // 64
        variants = [("as built", {}), ("regroup / 8 tiles", dict(regroup_every=8)), ("regroup / 4 tiles", dict(regroup_every=4)),
                    ("regroup / 2 tiles", dict(regroup_every=2)), ("4 size classes", dict(classes=4)),
                    ("4 classes + regroup / 4", dict(classes=4, regroup_every=4))]
        if os.environ.get("SIM_SCHEDULES"):
            variants = [("as built", {}), ("every 8", dict(regroup_every=8)), ("every 16", dict(regroup_every=16)),
                        ("at 1/4, 1/2, 3/4", dict(regroup_at=(nt // 4, nt // 2, 3 * nt // 4))),
                        ("at 1/8, 1/4, 1/2", dict(regroup_at=(nt // 8, nt // 4, nt // 2))),
                        ("at 1/6, 1/3, 2/3", dict(regroup_at=(nt // 6, nt // 3, 2 * nt // 3))),
                        ("at 1/3, 2/3", dict(regroup_at=(nt // 3, 2 * nt // 3))), ("at 1/2", dict(regroup_at=(nt // 2,)))]
        for tag, kw in variants:
            kept, cnt = sweep(box, float(p.nms_threshold), hi, **kw)
            if base is None:
                base = kept
            assert np.array_equal(kept, base)
            print(f"class {c} n={len(order)} kept={int(kept.sum())} {tag:26s} C steps {cnt['c_steps']:6d}  bb tests {cnt['bb_tests']:5d}  "
                  f"A columns {cnt['a_cols']:5d}  undecided lanes per step {cnt['lanes_used'] / max(cnt['c_steps'], 1):5.1f}", flush=True)

if __name__ == "__main__":
    main()
