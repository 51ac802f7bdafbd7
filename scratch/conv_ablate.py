"""Where one conv configuration spends its time: the product kernel against profiling builds with parts of the chunk loop
removed.  `build` (on the CPU box) compiles the four libraries; `run` (on the GPU) times them.
  python scratch/conv_ablate.py build|run k s cin cout H tw mrep nrep ck nw [B]"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mode = sys.argv[1]
k, s, cin, cout, H, tw, mrep, nrep, ck, nw = [int(v) for v in sys.argv[2:12]]
B = int(sys.argv[12]) if len(sys.argv) > 12 else 256
stages = int(sys.argv[13]) if len(sys.argv) > 13 else 1
tag = f"{k}{s}_{tw}_{mrep}{nrep}_{ck}_{nw}_{stages}"
VARS = {"product": None, "no staging (barriers kept)": 1, "no MFMA": 2, "no staging, no barriers": 3}
if len(sys.argv) > 13 and int(sys.argv[13]) == 2: VARS = {"product": None, "no MFMA": 2}
if os.environ.get("ABLATE_EPI"): VARS = {"product": None, "no global stores": 4}
if os.environ.get("ABLATE_TAG"):
    VARS = {"product": None}; tag += "_" + os.environ["ABLATE_TAG"]
if os.environ.get("SKEW_BUILD") or os.environ.get("CONV_SKEW"):
    VARS = {"product": None}; tag += "_skew"
def so(v): return os.path.join(ROOT, "scratch", f"libconv_ablate_{tag}_{v or 0}.so")
if mode == "build":
    for name, v in VARS.items():
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-shared"] + ([f"-I{os.environ['ABLATE_INC']}"] if os.environ.get("ABLATE_INC") else []) + [f"-I{ROOT}/yolo_amd/csrc", f"-I{ROOT}/include",
               f"-DTR_KS={k}", f"-DTR_S={s}", f"-DTR_TW={tw}", f"-DTR_MREP={mrep}", f"-DTR_NREP={nrep}", f"-DTR_CK={ck}", f"-DTR_NW={nw}", f"-DTR_STAGES={stages}"]
        if v: cmd.append(f"-DYAMD_CONV_ABLATE={v}")
        if os.environ.get("SKEW_BUILD"): cmd.append("-DYAMD_CONV_SKEW")
        cmd += [f"{ROOT}/scratch/conv_ablate.hip", f"{ROOT}/yolo_amd/csrc/yamd_core.hip", "-o", so(v)]
        subprocess.check_call(cmd)
    sys.exit(0)
import torch
from yolo_amd import lib as L
d = L.ConvDesc()
L.check(L.LIB.yamd_conv_desc_init(ctypes.byref(d), k, s, cin, cout, 1, 0))
L.check(L.LIB.yamd_conv_desc_tune(ctypes.byref(d), min(mrep, 2), nrep, ck, 4, 1))  # packing depends on nrep and ck only
w = torch.randn(cout, cin, k, k) * 0.05
wp = torch.empty(d.packed_w_elems, dtype=torch.int16); bp = torch.empty(d.cout_pad)
L.check(L.LIB.yamd_conv_pack_weights(ctypes.byref(d), w.data_ptr(), cin, torch.zeros(cout).data_ptr(), wp.data_ptr(), bp.data_ptr()))
dev = torch.device("cuda")
x = torch.randn(B, H, H, cin, device=dev).to(torch.bfloat16)
Ho = (H + 2 * (k // 2) - k) // s + 1
y = torch.empty(B, Ho, Ho, cout, device=dev, dtype=torch.bfloat16)
hh, ww = (H * H // tw, tw) if k == 1 and (H * H) % tw == 0 else (H, H)      # 1x1: the pixel list as a tw-wide strip
wd, bd = wp.to(dev), bp.to(dev)
fl = 2.0 * B * Ho * Ho * cout * cin * k * k
if os.environ.get("ABLATE_ZEROS"):
    x.zero_(); wd.zero_()
    print("all-zero operands")
print(f"k{k}s{s} {cin}>{cout} @{Ho} B{B} tiling ({mrep},{nrep},{ck},{nw},{stages}) tw {tw}")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, v in VARS.items():
    lib = ctypes.CDLL(so(v))
    lib.conv_ablate_run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 6
    def run():
        rc = lib.conv_ablate_run(x.data_ptr(), cin, wd.data_ptr(), bd.data_ptr(), y.data_ptr(), cout, B, hh, ww, cin, cout)
        assert rc == 0
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"  {name:28s} {ms:8.4f} ms   ({fl / ms / 1e9:7.0f} TFLOP/s equivalent)", flush=True)
