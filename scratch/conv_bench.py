"""Single-layer conv microbenchmark: python scratch/conv_bench.py k s cin cout H [B] [reps] [mrep nrep ck nwaves stages]
env RES=1: with a residual input; ACT=<YAMD_ACT_*> (default 1 = SiLU)"""
import ctypes, sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd import lib as L
k, s, cin, cout, H = [int(v) for v in sys.argv[1:6]]
B = int(sys.argv[6]) if len(sys.argv) > 6 else 256
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 20
d = L.ConvDesc()
L.check(L.LIB.yamd_conv_desc_init(ctypes.byref(d), k, s, cin, cout, int(os.environ.get('ACT', '1')), 0))
if len(sys.argv) > 11:
    t = [int(v) for v in sys.argv[8:13]]
    L.check(L.LIB.yamd_conv_desc_tune(ctypes.byref(d), *(t + [1] * (5 - len(t)))))
w = torch.randn(cout, cin, k, k) * 0.05
b = torch.zeros(cout)
wp = torch.empty(d.packed_w_elems, dtype=torch.int16); bp = torch.empty(d.cout_pad)
L.check(L.LIB.yamd_conv_pack_weights(ctypes.byref(d), w.data_ptr(), cin, b.data_ptr(), wp.data_ptr(), bp.data_ptr()))
dev = torch.device('cuda')
x = torch.randn(B, H, H, cin, device=dev).to(torch.bfloat16)
Ho = (H + 2 * (k // 2) - k) // s + 1
y = torch.empty(B, Ho, Ho, cout, device=dev, dtype=torch.bfloat16)
wd, bd = wp.to(dev), bp.to(dev)
res = torch.randn(B, Ho, Ho, cout, device=dev).to(torch.bfloat16) if os.environ.get('RES') else None
st = L.current_stream()
def run():
    L.check(L.LIB.yamd_conv2d_bias_act(ctypes.byref(d), x.data_ptr(), cin, wd.data_ptr(), bd.data_ptr(), None if res is None else res.data_ptr(), cout,
                                       y.data_ptr(), cout, B, H, H, st))
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
fl = 2.0 * B * Ho * Ho * cout * cin * k * k
by = B * (H * H * cin + Ho * Ho * cout) * 2.0
print(("res " if res is not None else "    ") + f"k{k}s{s} {cin}>{cout} @{Ho} B{B} nrep{d.nrep} ck{d.ck} mrep{d.mrep} nw{d.nwaves} st{d.stages}: {ms:.4f} ms  {fl/ms/1e9:.1f} TFLOP/s  {by/ms/1e6:.1f} GB/s")
