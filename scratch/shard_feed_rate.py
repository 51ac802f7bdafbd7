"""Host -> worker -> GPU rate of yolo_amd.parallel.ShardedPredictor's image feed (no model: the worker copies its slice
to the device and answers), for one worker on cuda:0 and an fp32 batch of the benchmark's size.
python scratch/shard_feed_rate.py [batch] [size]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def copy_only_runner(device: str):
    import torch
    dev = torch.device(device)
    torch.cuda.set_device(dev)
    staging = {}

    def run(images):
        if images.shape[0] == 0:
            return []
        src = torch.from_numpy(images)
        buf = staging.get("pinned")
        if buf is None or buf.numel() < src.numel():
            buf = torch.empty(src.numel(), dtype=torch.float32).pin_memory()
            staging["pinned"] = buf
        host = buf[:src.numel()].view(src.shape)
        t0 = time.perf_counter()
        host.copy_(src)
        t1 = time.perf_counter()
        x = host.to(dev, non_blocking=True)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        return [np.array([t1 - t0, t2 - t1, float(x[0, 0, 0, 0])], dtype=np.float64)]

    return run


if __name__ == "__main__":
    from yolo_amd import parallel
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 640
    imgs = np.random.RandomState(0).rand(B, 3, S, S).astype(np.float32)
    gb = imgs.nbytes / 1e9
    with parallel.ShardedPredictor(copy_only_runner, ["cuda:0"]) as sp:
        for mode in ("predict (parent copies into the shared block)", "predict_views (producer wrote the block itself)"):
            ts = []
            for rep in range(4):
                t0 = time.perf_counter()
                if mode.startswith("predict ("):
                    out = sp.predict(imgs)
                else:
                    if rep == 0:
                        views = sp.input_views(B, imgs.shape[1:], imgs.dtype)
                        for v in views:
                            v[...] = imgs
                    out = sp.predict_views()
                ts.append(time.perf_counter() - t0)
            t = min(ts[1:])
            print(f"{mode}: {gb:.2f} GB in {t * 1e3:.1f} ms = {gb / t:.1f} GB/s end to end; inside the worker: shared block -> pinned "
                  f"{out[0][0] * 1e3:.1f} ms ({gb / out[0][0]:.1f} GB/s), pinned -> device {out[0][1] * 1e3:.1f} ms ({gb / out[0][1]:.1f} GB/s)")
