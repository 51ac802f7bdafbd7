"""Phase timing of one conv config: python scratch/conv_trace.py k s cin cout H tw mrep nrep ck"""
import ctypes, os, subprocess, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yolo_amd import lib as L
k, s, cin, cout, H, tw, mrep, nrep, ck = [int(v) for v in sys.argv[1:10]]
nw, stages, B = [int(v) for v in sys.argv[10:13]] if len(sys.argv) > 12 else (4, 1, 256)
so = os.path.join(ROOT, "scratch", f"libconv_trace_{k}{s}_{tw}_{mrep}{nrep}_{ck}" + (f"_{nw}_{stages}" if (nw, stages) != (4, 1) else "") + os.environ.get("TRACE_TAG", "") + ".so")
lib = ctypes.CDLL(so)
d = L.ConvDesc()
L.check(L.LIB.yamd_conv_desc_init(ctypes.byref(d), k, s, cin, cout, 1, 0))
if (d.mrep, d.nrep, d.ck) != (mrep, nrep, ck):
    L.check(L.LIB.yamd_conv_desc_tune(ctypes.byref(d), min(mrep, 2), nrep, ck, 4, 1))
w = torch.randn(cout, cin, k, k) * 0.05
wp = torch.empty(d.packed_w_elems, dtype=torch.int16); bp = torch.empty(d.cout_pad)
L.check(L.LIB.yamd_conv_pack_weights(ctypes.byref(d), w.data_ptr(), cin, torch.zeros(cout).data_ptr(), wp.data_ptr(), bp.data_ptr()))
dev = torch.device("cuda")
x = torch.randn(B, H, H, cin, device=dev).to(torch.bfloat16)
Ho = (H + 2 * (k // 2) - k) // s + 1
y = torch.empty(B, Ho, Ho, cout, device=dev, dtype=torch.bfloat16)
wd, bd = wp.to(dev), bp.to(dev)
trace = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
grid = ctypes.c_int(0)
lib.conv_trace_run.argtypes = [ctypes.c_void_p] * 1 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 6 + [ctypes.c_void_p, ctypes.c_void_p]
for _ in range(int(os.environ.get("TRACE_RUNS", "3"))):
    trace.zero_()
    rc = lib.conv_trace_run(x.data_ptr(), cin, wd.data_ptr(), bd.data_ptr(), y.data_ptr(), cout, B, H, H, cin, cout, trace.data_ptr(), ctypes.byref(grid))
    assert rc == 0, L.LIB.yamd_last_error()
torch.cuda.synchronize()
t = trace.cpu().numpy()[: grid.value * 8].reshape(-1, 8).astype(np.int64)
t = t[t[:, 4] > 0]
d_ = np.diff(t[:, :5], axis=1).astype(np.float64)
names = ["prologue", "first stage", "main loop", "epilogue"]
print(f"k{k}s{s} {cin}>{cout}@{Ho} blocks {len(t)}  total span {(t[:,4].max()-t[:,0].min())} ticks")
for i, n in enumerate(names):
    print(f"  {n:12s} mean {d_[:, i].mean():9.0f}  median {np.median(d_[:, i]):9.0f}  p90 {np.percentile(d_[:, i], 90):9.0f}")
print(f"  block total  mean {(t[:,4]-t[:,0]).mean():9.0f}")
span = float(t[:, 4].max() - t[:, 0].min())
print(f"  mean concurrency (sum of block durations / span) {(t[:,4]-t[:,0]).sum() / span:7.1f} workgroups; blocks per slot {len(t) / ((t[:,4]-t[:,0]).sum() / span):6.1f}")
if stages == 2:
    print(f"  last epilogue (stamp 3 -> end) median {np.median(t[:,4]-t[:,3]):8.0f}  p90 {np.percentile(t[:,4]-t[:,3], 90):8.0f}")
    order = np.argsort(t[:, 0]); st = t[order, 0]
    print("  start times of the first 16 workgroups (ticks after the first):", (st[:16] - st[0]).tolist())
    print("  starts, percentiles over all blocks:", [int(np.percentile(st - st[0], q)) for q in (10, 25, 50, 75, 90)])
if t[:, 7].min() > 0:
    if stages == 1:
        print(f"  chunk 1: issue+MFMA {np.median(t[:,6]-t[:,5]):8.0f}   barrier+write+barrier {np.median(t[:,7]-t[:,6]):8.0f}")
    else:
        print(f"  chunk 1 (pipelined): {np.median(t[:,7]-t[:,5]):8.0f}   first chunk's loads landed after {np.median(t[:,6]-t[:,1]):8.0f}")
