"""A steady load for a few seconds: python scratch/load_loop.py mfma|v3|v5s|copy seconds"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd import lib as L, models
from yolo_amd.init import seeded_state_dict
kind, secs = sys.argv[1], float(sys.argv[2])
dev = torch.device("cuda:0")
st = L.current_stream()
if kind == "mfma":
    ops = (torch.rand(4096, device=dev) * 2 - 1).to(torch.bfloat16)
    sink = torch.empty(512 * 256, dtype=torch.float32, device=dev)
    fn = lambda: L.check(L.LIB.yamd_microbench_mfma_bf16(ops.data_ptr(), sink.data_ptr(), 512, 4096, st))
elif kind == "copy":
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device=dev).random_(0, 255); b = torch.empty_like(a)
    fn = lambda: L.check(L.LIB.yamd_microbench_copy(a.data_ptr(), b.data_ptr(), n, 2, st))
else:
    p, m = models.create(kind)
    m.load_state_dict(seeded_state_dict(m, seed=0, mode="torch_default"))
    x = torch.rand(256, 3, p.img_size, p.img_size, device=dev)
    prog = m.program()
    fn = lambda: prog.run(x)
fn(); torch.cuda.synchronize()
t0 = time.time(); n = 0
while time.time() - t0 < secs:
    for _ in range(5): fn()
    torch.cuda.synchronize(); n += 5
print(kind, "iterations", n, "ms per iteration", (time.time() - t0) / n * 1e3)
