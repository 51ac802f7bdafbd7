"""Tiling sweep of single conv layers in one process: every built tiling of every listed shape, interleaved rounds,
median of the rounds.  python scratch/conv_sweep.py [shape-set] [B]   (shape-set: v3 | v5s | v7 | all)"""
import ctypes, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd import lib as L, tuning

SETS = {
    "v3": [(3, 1, 32, 64, 304), (3, 1, 64, 128, 152), (3, 1, 128, 256, 76), (3, 1, 256, 512, 38), (3, 1, 512, 1024, 19),
           (3, 2, 32, 64, 608), (3, 2, 64, 128, 304), (3, 2, 128, 256, 152), (3, 2, 256, 512, 76), (3, 2, 512, 1024, 38),
           (1, 1, 64, 32, 304), (1, 1, 128, 64, 152), (1, 1, 256, 128, 76), (1, 1, 512, 256, 38), (1, 1, 1024, 512, 19)],
    "v5s": [(3, 1, 64, 64, 80), (3, 1, 128, 128, 40), (3, 1, 256, 256, 20), (3, 2, 64, 128, 160), (3, 2, 128, 256, 80),
            (3, 2, 256, 512, 40), (1, 1, 128, 128, 80), (1, 1, 256, 256, 40), (1, 1, 512, 512, 20), (1, 1, 256, 128, 40)],
    "v7": [(3, 1, 64, 64, 320), (3, 1, 64, 64, 160), (3, 1, 128, 128, 80), (3, 1, 256, 256, 40), (3, 1, 256, 256, 20)],
}
which = sys.argv[1] if len(sys.argv) > 1 else "v3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
shapes = sum(SETS.values(), []) if which == "all" else SETS[which]
dev = torch.device("cuda")
st = L.current_stream()
for (k, s, cin, cout, H) in shapes:
    Ho = (H + 2 * (k // 2) - k) // s + 1
    # keep the activation tensors of one launch near 1 GB at most
    Bs = max(8, min(B, int(1.0e9 / ((H * H * cin + Ho * Ho * cout) * 2))))
    x = torch.randn(Bs, H, H, cin, device=dev).to(torch.bfloat16)
    y = torch.empty(Bs, Ho, Ho, cout, device=dev, dtype=torch.bfloat16)
    w = torch.randn(cout, cin, k, k) * 0.05
    b = torch.zeros(cout)
    cands = []
    for t in tuning.candidates(k, s, cin, cout):
        d = L.ConvDesc()
        L.check(L.LIB.yamd_conv_desc_init(ctypes.byref(d), k, s, cin, cout, 1, 0))
        if L.LIB.yamd_conv_desc_tune(ctypes.byref(d), *t) != 0:
            continue
        wp = torch.empty(d.packed_w_elems, dtype=torch.int16); bp = torch.empty(d.cout_pad)
        L.check(L.LIB.yamd_conv_pack_weights(ctypes.byref(d), w.data_ptr(), cin, b.data_ptr(), wp.data_ptr(), bp.data_ptr()))
        wd, bd = wp.to(dev), bp.to(dev)
        def run(d=d, wd=wd, bd=bd):
            return L.LIB.yamd_conv2d_bias_act(ctypes.byref(d), x.data_ptr(), cin, wd.data_ptr(), bd.data_ptr(), None, 0,
                                              y.data_ptr(), cout, Bs, H, H, st)
        if run() != 0:          # e.g. the tile does not fit LDS at this image size
            continue
        cands.append((t, run, []))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for r in range(rounds):
        for t, run, ts in cands:
            run()
            e0.record()
            for _ in range(3):
                run()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 3)
    fl = 2.0 * Bs * Ho * Ho * cout * cin * k * k
    res = sorted((statistics.median(ts), t) for t, run, ts in cands)
    print(f"k{k}s{s} {cin}>{cout} @{Ho} B{Bs}: " + "  ".join(f"{t}={ms:.3f}ms/{fl/ms/1e9:.0f}TF" for ms, t in res[:6]), flush=True)
    best1 = [r for r in res if r[1][4] == 1][0]
    best2 = [r for r in res if r[1][4] == 2][:1]
    if best2:
        print(f"      best 1-stage {best1[1]} {fl/best1[0]/1e9:.0f} TF   best 2-stage {best2[0][1]} {fl/best2[0][0]/1e9:.0f} TF", flush=True)
    del x, y
