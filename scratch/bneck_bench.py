"""Fused bottleneck against the two tuned conv launches it replaces, on the layer shapes of the models: python scratch/bneck_bench.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd import lib as L, tuning

dev = torch.device("cuda:0")
st = L.current_stream()

def layer(k, cin, cout, H, act):
    d = L.ConvDesc()
    L.check(L.LIB.yamd_conv_desc_init(ctypes.byref(d), k, 1, cin, cout, act, 0))
    t = tuning.lookup(k, 1, cin, cout, H, H)
    if t is not None:
        L.check(L.LIB.yamd_conv_desc_tune(ctypes.byref(d), *t))
    w = torch.randn(cout, cin, k, k) / (k * cin ** 0.5)
    b = torch.randn(cout) * 0.1
    wp = torch.empty(d.packed_w_elems, dtype=torch.int16); bp = torch.empty(d.cout_pad)
    L.check(L.LIB.yamd_conv_pack_weights(ctypes.byref(d), w.data_ptr(), cin, b.data_ptr(), wp.data_ptr(), bp.data_ptr()))
    return d, wp.to(dev), bp.to(dev), w, b, t

def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for cin, cmid, H, B, act in ((32, 32, 160, 256, 1), (64, 64, 80, 256, 1), (128, 128, 40, 256, 1), (64, 32, 304, 64, 2), (128, 64, 152, 128, 2),
                             (128, 128, 80, 256, 1), (64, 64, 160, 256, 1)):
    d1, w1, b1, w1f, b1f, t1 = layer(1, cin, cmid, H, act)
    d2, w2, b2, w2f, b2f, t2 = layer(3, cmid, cin, H, act)
    x = torch.randn(B, H, H, cin, device=dev).to(torch.bfloat16)
    t = torch.empty(B, H, H, cmid, device=dev, dtype=torch.bfloat16)
    y = torch.empty(B, H, H, cin, device=dev, dtype=torch.bfloat16)
    def unfused():
        L.check(L.LIB.yamd_conv2d_bias_act(ctypes.byref(d1), x.data_ptr(), cin, w1.data_ptr(), b1.data_ptr(), None, 0, t.data_ptr(), cmid, B, H, H, st))
        L.check(L.LIB.yamd_conv2d_bias_act(ctypes.byref(d2), t.data_ptr(), cmid, w2.data_ptr(), b2.data_ptr(), x.data_ptr(), cin, y.data_ptr(), cin, B, H, H, st))
    n = L.LIB.yamd_bottleneck_packed_elems(cin, cmid, cin)
    wp = torch.empty(n, dtype=torch.int16); bp = torch.empty(cmid + cin)
    L.check(L.LIB.yamd_bottleneck_pack_weights(cin, cmid, cin, w1f.data_ptr(), b1f.data_ptr(), w2f.data_ptr(), b2f.data_ptr(), wp.data_ptr(), bp.data_ptr()))
    wd, bd = wp.to(dev), bp.to(dev)
    y2 = torch.empty_like(y)
    def fused():
        L.check(L.LIB.yamd_bottleneck(cin, cmid, cin, act, x.data_ptr(), cin, wd.data_ptr(), bd.data_ptr(), 1, y2.data_ptr(), cin, B, H, H, st))
    tu, tf = timed(unfused), timed(fused)
    gf = 2.0 * B * H * H * (cin * cmid + 9 * cmid * cin) / 1e6
    print(f"{cin:4d}>{cmid:3d}>{cin:3d} @{H}x{H} B={B}: unfused {tu:7.1f} us (tilings {t1} {t2})  fused {tf:7.1f} us ({gf / tf:6.1f} TFLOP/s, {2 * x.numel() * 2 / tf / 1e3:5.0f} GB/s of the two passes)  "
          f"same {bool(torch.equal(y, y2))}", flush=True)
