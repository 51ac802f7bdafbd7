"""Stage-by-stage relative error of the v5 face graph at 640 (named concat buffers vs the fp32 oracle)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, torch.nn.functional as F
from yolo_amd import models
from yolo_amd.init import seeded_state_dict, seeded_images
from oracle import v5_ref as R
def rel(a, b): return float(np.linalg.norm(a - b) / np.linalg.norm(b))
name = "v5face"
p, m = models.create(name, 640)
sd = seeded_state_dict(m, seed=32, mode="spread", **models.init_options(name))
m.load_state_dict(sd)
x = seeded_images(1, 640, seed=32)
prog = m.program()
prog.run(torch.from_numpy(x).cuda())
torch.cuda.synchronize()
bufs = prog._buffers_for(1, torch.device("cuda", 0))
def get(n, lo, c):
    return bufs[n][..., lo:lo + c].float().cpu().numpy().transpose(0, 3, 1, 2)
torch.set_num_threads(16)
img = torch.from_numpy(x)
bb = "body_net.backbone."
with torch.no_grad():
    s1 = R._conv_block(sd, bb + "stem.stem_1", img, 2)
    s2 = R._conv_block(sd, bb + "stem.stem_2b", R._conv_block(sd, bb + "stem.stem_2a", s1), 2)
    mp = F.max_pool2d(s1, 2, 2, ceil_mode=True)
    bc = s2.shape[1]
    print("stem_2b", rel(get("cat_stem", 0, bc), s2.numpy()), "std", float(s2.std()), "mean", float(s2.mean()))
    print("stem_2p", rel(get("cat_stem", bc, bc), mp.numpy()), "std", float(mp.std()), "mean", float(mp.mean()))
    xx = R._conv_block(sd, bb + "stem.stem_3", torch.cat((s2, mp), 1))
    xx = R._c3(sd, bb + "dark2.0", xx, True)
    f1 = R._c3(sd, bb + "dark3.1", R._conv_block(sd, bb + "dark3.0", xx, 2), True)
    c3 = f1.shape[1]
    print("feat1", rel(get("cat_p3", c3, c3), f1.numpy()), "std", float(f1.std()), "mean", float(f1.mean()), "absmax", float(f1.abs().max()))
    f2 = R._c3(sd, bb + "dark4.1", R._conv_block(sd, bb + "dark4.0", f1, 2), True)
    c4 = f2.shape[1]
    print("feat2", rel(get("cat_p4", c4, c4), f2.numpy()), "std", float(f2.std()), "mean", float(f2.mean()), "absmax", float(f2.abs().max()))
    d5 = R._conv_block(sd, bb + "dark5.0", f2, 2)
    sp1 = R._conv_block(sd, bb + "dark5.1.cv1", d5)
    hc = sp1.shape[1]
    print("spp.cv1", rel(get("cat_spp", 0, hc), sp1.numpy()), "std", float(sp1.std()), "absmax", float(sp1.abs().max()))
    for i, k in enumerate((3, 5, 7)):
        pk = F.max_pool2d(sp1, k, 1, k // 2)
        print("pool", k, rel(get("cat_spp", (i + 1) * hc, hc), pk.numpy()))
