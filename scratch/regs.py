"""Register / scratch usage of every kernel in one source file: python scratch/regs.py yolo_amd/csrc/yamd_conv_k3s1b.hip [filter]"""
import re, subprocess, sys, os
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = ["-ffp-contract=off"] if any(k in src for k in ("decode", "nms", "preprocess", "metric")) else []
r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function",
                    "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/regs_tmp.o"] + extra + sys.argv[3:], capture_output=True, text=True)
txt = r.stderr
if r.returncode:
    print(txt[-3000:]); sys.exit(1)
for b in re.split(r"remark: Function Name: ", txt)[1:]:
    name = b.split()[0]
    g = lambda k: re.search(re.escape(k) + r": (\d+)", b).group(1)
    m = re.search(r"conv_igemm_kernelI(.*?)EEEv", name)
    short = m.group(1).replace("ELi", " ").replace("ELb", " b").replace("Li", "").replace("n1", "-1") if m else name[:70]
    if flt and flt not in short and flt not in name:
        continue
    print("%-46s VGPR %3s AGPR %3s spill %3s scratch %4s occ %s SGPR %s" % (short, g("VGPRs"), g("AGPRs"), g("VGPRs Spill"),
          g("ScratchSize [bytes/lane]"), g("Occupancy [waves/SIMD]"), g("TotalSGPRs")))
