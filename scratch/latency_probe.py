"""Where a batch-1 (or batch-N) predict spends its time: python scratch/latency_probe.py [model] [batch]"""
import sys, os, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd import models, torch_utils as tu, lib as L
from yolo_amd.init import seeded_state_dict
name = sys.argv[1] if len(sys.argv) > 1 else "v5s"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
p, m = models.create(name)
m.load_state_dict(seeded_state_dict(m, seed=0, mode="spread", **models.init_options(name)))
x = torch.rand(B, 3, p.img_size, p.img_size, device=dev)
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return statistics.median(ts) * 1e3
def host_only(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    return statistics.median(ts) * 1e3
prog = m.program()
print(name, "batch", B, "ops", len(prog.ops))
print("conv program: wall %.3f ms, host enqueue %.3f ms" % (timeit(lambda: prog.run(x)), host_only(lambda: prog.run(x))))
prog.use_c_plan = False
print("conv program, per-op python calls: wall %.3f ms, host enqueue %.3f ms" % (timeit(lambda: prog.run(x)), host_only(lambda: prog.run(x))))
prog.use_c_plan = True
raws = m.raw_heads(x)
print("decode: wall %.3f ms" % timeit(lambda: m.decode_predict(raws, m._arena)))
pred = m.decode_predict(raws, m._arena)
print("nms: wall %.3f ms, host %.3f" % (timeit(lambda: m.nms_predict(pred, m._arena)), host_only(lambda: m.nms_predict(pred, m._arena))))
print("predict_device: wall %.3f ms, host %.3f" % (timeit(lambda: m.predict_device(x)), host_only(lambda: m.predict_device(x))))
rows, counts = m.predict_device(x)
print("rows_to_host: wall %.3f ms" % timeit(lambda: tu.rows_to_host(rows, counts)))
print("predict: wall %.3f ms" % timeit(lambda: m.predict({"batch_images": x})))
