"""TEST INFRASTRUCTURE ONLY -- seeded synthetic (B, A, C) prediction tensors with clustered, overlapping
boxes (the input family of the NMS golden vectors); importable without the reference."""
import numpy as np


def synthetic_pred(rng, B, A, nc, img, n_extra=0, clusters=12, fmt="cxcywh", obj_bias=0.0):
    cen = rng.uniform(0.1 * img, 0.9 * img, (B, clusters, 2))
    siz = rng.uniform(0.05 * img, 0.4 * img, (B, clusters, 2))
    which = rng.randint(0, clusters, (B, A))
    bi = np.arange(B)[:, None]
    cxy = cen[bi, which] + rng.standard_normal((B, A, 2)) * 0.03 * img
    wh = siz[bi, which] * np.exp(rng.standard_normal((B, A, 2)) * 0.15)
    obj = 1.0 / (1.0 + np.exp(-(rng.standard_normal((B, A, 1)) * 2.0 + obj_bias)))
    cls_logit = rng.standard_normal((B, A, nc)) * 2.0
    fav = rng.randint(0, nc, (B, clusters))[bi, which]
    cls_logit[bi, np.arange(A)[None, :], fav] += 3.0
    cls = 1.0 / (1.0 + np.exp(-cls_logit))
    if fmt == "xyxy":
        box = np.concatenate((cxy - wh / 2, cxy + wh / 2), -1)
    else:
        box = np.concatenate((cxy, wh), -1)
    extra = rng.uniform(-0.1 * img, 1.1 * img, (B, A, n_extra))
    return np.concatenate((box, obj, extra, cls), -1).astype(np.float32)
