"""Builds yolo_amd/libyolo_amd.so from yolo_amd/csrc/*.hip with hipcc for gfx950.

The library is built in-tree so the same file is what tests, bench.py and smoke() load; there is
no JIT and no fallback: if the .so is missing, ``yolo_amd.lib`` raises.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libyolo_amd.so")

# (source, extra flags).  decode/nms follow the reference's fp32 operation order exactly, so the
# compiler must not fuse a*b+c into an fma there.
SOURCES = [
    ("yamd_core.hip", []),
    ("yamd_conv.hip", []),
    ("yamd_conv_k1.hip", []),
    ("yamd_conv_k1b.hip", []),
    ("yamd_conv_k3s1.hip", []),
    ("yamd_conv_k3s1b.hip", []),
    ("yamd_conv_k3s2.hip", []),
    ("yamd_conv_k3s2b.hip", []),
    ("yamd_decode.hip", ["-ffp-contract=off"]),
    ("yamd_nms.hip", ["-ffp-contract=off"]),
    ("yamd_preprocess.hip", ["-ffp-contract=off"]),
    ("yamd_metric.hip", ["-ffp-contract=off"]),
    ("yamd_stem.hip", []),
    ("yamd_bneck.hip", []),
    ("yamd_c3.hip", []),
    ("yamd_microbench.hip", []),
    ("yamd_plan.hip", []),
]
COMMON = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wall", "-Wno-unused-function"]
HEADERS = [os.path.join(CSRC, "yamd_common.h"), os.path.join(CSRC, "yamd_conv_kernel.h"),
           os.path.join(CSRC, "yamd_conv_variants.inc"),
           os.path.join(HERE, "..", "include", "yolo_amd.h")]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    objs = []
    for src, extra in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + HEADERS):
            jobs.append([hipcc] + COMMON + extra + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print("[yolo_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
    print(LIB)
