"""Phase timing of nms_segment_kernel on the headline workload (v5s, 640, batch 256, seeded random weights).

  python scratch/nms_phases.py --build     (here: compiles yamd_nms.hip with -DYAMD_NMS_PHASES and links it with the
                                            library's other objects into scratch/_prof/libyolo_amd_prof.so)
  python scratch/nms_phases.py [--batch 256]   (on the GPU box: loads that library in place of the product one)

Prints, per bucket (one workgroup), the shader-clock ticks thread 0 spent between the marks of the kernel.
"""
import argparse
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROF_DIR = os.path.join(ROOT, "scratch", "_prof")
PROF_LIB = os.path.join(PROF_DIR, "libyolo_amd_prof.so")
NAMES = ["sort", "boxes+sorted write", "grouping", "sweep set-up + A(0)", "B: wave 0 resolve", "B: barrier",
         "C + A(next): wave 0", "C + A: wait for slowest wave"]


def build(level=1):
    from yolo_amd import build as B
    B.build_library(verbose=False)
    os.makedirs(PROF_DIR, exist_ok=True)
    o = os.path.join(PROF_DIR, "yamd_nms.o")
    subprocess.check_call([B._hipcc()] + B.COMMON + ["-ffp-contract=off", f"-DYAMD_NMS_PHASES={level}", "-c",
                                                     os.path.join(B.CSRC, "yamd_nms.hip"), "-o", o])
    objs = [os.path.join(B.OBJ, s.replace(".hip", ".o")) for s, _ in B.SOURCES if s != "yamd_nms.hip"] + [o]
    subprocess.check_call([B._hipcc(), "-shared", "-fPIC", "--offload-arch=gfx950", "-o", PROF_LIB] + objs)
    print(PROF_LIB)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--counters", action="store_true", help="with --build: count the sweep's work per wave too (slow)")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--model", default="v5s")
    args = ap.parse_args()
    if args.build:
        return build(2 if args.counters else 1)
    import torch
    import bench
    import yolo_amd.lib as lib
    from yolo_amd import models  # noqa: F401  (every module that keeps its own reference to the library is loaded now)
    models.create(args.model)
    old = lib.LIB
    # swap the library object the host code calls through for the profiling build
    prof = ctypes.CDLL(PROF_LIB)
    for name in lib.EXPORTS:
        fn = getattr(prof, name)
        ref = getattr(old, name)
        fn.argtypes, fn.restype = ref.argtypes, ref.restype
    lib.LIB = prof
    for modname, mod in list(sys.modules.items()):
        if modname.startswith("yolo_amd") and getattr(mod, "LIB", None) is old:
            mod.LIB = prof
    prof.yamd_debug_nms_phases.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    lib.check(prof.yamd_init(), "init")
    p, model, _ = bench.build_model(args.model, "torch_default")
    model = model.to(dev).eval()
    images = bench.device_images(args.batch, p.img_size if hasattr(p, "img_size") else 640, 0, dev)
    raws = model.raw_heads(images)
    pred = model.decode_predict(raws, model._arena)
    out = (ctypes.c_ulonglong * 16)()
    for rep in range(3):
        torch.cuda.synchronize()
        prof.yamd_debug_nms_phases(out, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        model.nms_predict(pred, model._arena)
        e1.record()
        torch.cuda.synchronize()
        prof.yamd_debug_nms_phases(out, 0)
        nb = max(int(out[8]), 1)
        tot = sum(int(out[i]) for i in range(8))
        print(f"rep {rep}: nms {e0.elapsed_time(e1):.3f} ms, {nb} buckets, {out[9] / nb:.1f} boxes and {out[10] / nb:.1f} kept per bucket, "
              f"{tot / nb:.0f} ticks per bucket")
        for i, nm in enumerate(NAMES):
            print(f"    {nm:32s} {out[i] / nb:10.0f} ticks  {100.0 * out[i] / max(tot, 1):5.1f} %")
        print(f"    per bucket: {out[11] / nb:.0f} (tile, group) bounding-box tests, {out[12] / nb:.0f} with kept boxes to apply, "
              f"{out[13] / nb:.0f} (kept box, group) steps, {out[14] / nb:.0f} of them reach the division")


if __name__ == "__main__":
    main()
