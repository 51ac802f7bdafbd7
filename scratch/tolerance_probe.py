"""How much of the end-to-end bf16 tolerances each model really needs: for every model, against the reference's small
golden and the fp32 oracle graph at full size, the raw-logit relative L2 and, per column family, the smallest extra
allowance (percent of the value / fraction of the row's stride or anchor) for which fewer than 0.2 % of the values
miss 0.5 px + 1 %.  python scratch/tolerance_probe.py [names...] > table"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from conftest import load_golden
from test_gpu_models import NAMES, POINTS, _model, _row_stride
from test_gpu_v5_e2e import _rel_l2
from oracle import graphs_ref
from yolo_amd.init import seeded_images

dev = torch.device("cuda:0")
names = sys.argv[1:] or ["v5s"] + NAMES


def need(got, want, base_pct, scale=None, q=0.998):
    """smallest k such that |d| <= 0.5 + base_pct |want| + k * unit for a fraction q of the values; unit = |want| (percent
    form, scale None) or the per-row scale"""
    d = np.abs(got - want) - 0.5 - base_pct * np.abs(want)
    unit = np.abs(want) if scale is None else np.broadcast_to(scale, want.shape)
    k = np.where(d > 0, d / np.maximum(unit, 1e-6), 0.0)
    return float(np.quantile(k, q))


def report(tag, name, p, size, raws, want_raws, pred, want_pred, rs):
    l2 = max(_rel_l2(g, w) for g, w in zip(raws, want_raws))
    out = [f"{name:9s} {tag:5s} rawL2 {l2:.4f}"]
    xy = need(pred[..., :2], want_pred[..., :2], 0.01)
    wh = need(pred[..., 2:4], want_pred[..., 2:4], 0.01)
    out.append(f"xy +{100 * xy:.2f}% wh +{100 * wh:.2f}%")
    if rs is not None:
        sc = np.asarray(rs, np.float32)[None, :, None]
        out.append(f"box/stride {need(pred[..., :4], want_pred[..., :4], 0.01, sc):.4f}")
    P = POINTS.get(name, 0)
    if P:
        k0, k1 = 5, 5 + 2 * P
        out.append(f"pts +{100 * need(pred[..., k0:k1], want_pred[..., k0:k1], 0.01):.2f}%")
        if rs is not None:
            out.append(f"pts/stride {need(pred[..., k0:k1], want_pred[..., k0:k1], 0.01, sc):.4f}")
        keep = np.r_[4:5, k1:pred.shape[-1]]
    else:
        keep = np.r_[4:pred.shape[-1]]
    ds = np.abs(pred[..., keep] - want_pred[..., keep])
    out.append(f"score q99.8 {np.quantile(ds, 0.998):.4f}")
    print("  ".join(out), flush=True)


for name in names:
    gname = "v5s_64.npz" if name == "v5s" else f"{name}_small.npz"
    g = load_golden(gname)
    size, seed = (64, int(g["seed"])) if name == "v5s" else (int(g["size"]), int(g["seed"]))
    p, m, sd = _model(name, size, seed)
    imgs = torch.from_numpy(seeded_images(int(g["batch"]), size, seed=seed)).to(dev)
    raws = [r.cpu().numpy().transpose(0, 3, 1, 2) for r in m.raw_heads(imgs)]
    want_raws = [g[k] for k in (("raw_p3", "raw_p4", "raw_p5") if name == "v5s" else [f"raw{i}" for i in range(len(raws))])]
    pred = m._forward(imgs)[0].cpu().numpy()
    report("small", name, p, size, raws, want_raws, pred, g["pred"], _row_stride(name, p, size))
    # full size against the fp32 oracle graph
    p, m, sd = _model(name, 0, seed)
    size = p.img_size
    imgs_np = seeded_images(2, size, seed=seed)
    raws = [r.cpu().numpy().transpose(0, 3, 1, 2) for r in m.raw_heads(torch.from_numpy(imgs_np).to(dev))]
    ref = graphs_ref.model_ref(name)
    torch.set_num_threads(16)
    with torch.no_grad():
        want_raws = [t.numpy() for t in ref["body"](sd, torch.from_numpy(imgs_np))]
    pred = m._forward(torch.from_numpy(imgs_np).to(dev))[0].cpu().numpy()
    report("full", name, p, size, raws, want_raws, pred, ref["decode"](want_raws, p, False), _row_stride(name, p, size))
