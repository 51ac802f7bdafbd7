#!/usr/bin/env python3
"""Headline benchmark: images/s of the YOLOv5s 640x640 inference hot path (conv graph -> decode ->
confidence filter + per-class NMS -> detections on the host), batch 256 per MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU; each rank runs its own 256-image
   shard, there is no data-path collective; timing = max over ranks between two barriers.)

One step = one pass of the whole path over one batch that is already resident in HBM: 62 fused
conv kernels (bf16 MFMA), head decode, NMS, then the compacted detections are copied to the host
(counts + the kept rows), which is what the reference's ``predict`` returns.  The device->host copy of
step i runs on a second stream under step i+1's kernels; all K steps' detections are on the host
before the clock stops.
Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

METRIC = "images/sec (whole node) YOLOv5s/v8s 640² incl. decode+NMS; % MFMA/HBM roofline"
IMG = 640
PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA peak of one MI355X
PEAK_HBM_GBS = 8000.0         # HBM3E peak of one MI355X


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--model", default="v5s", help="model family (yolo_amd.models.registry); v5s is the headline config")
    ap.add_argument("--batch", type=int, default=0, help="images per GPU (default: the benchmark configuration of the model)")
    ap.add_argument("--weights", default="torch_default", choices=["torch_default", "spread"],
                    help="seeded random-init scheme (torch_default = the reference's YoloV5(param) distribution)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-images", type=int, default=64)
    ap.add_argument("--detail", default="", help="write the per-op timing table to this file")
    ap.add_argument("--backend", default="nccl", help="process-group backend for the barrier (gloo for CPU rehearsal)")
    ap.add_argument("--dry", action="store_true", help="no GPU work: rehearse the multi-rank timing logic only")
    return ap.parse_args()


def dist_setup(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if not args.dry:
            torch.cuda.set_device(local)
        dist.init_process_group(backend=args.backend, rank=rank, world_size=world)
    return world, rank, local


def barrier(world, dry):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    if not dry:
        torch.cuda.synchronize()


def max_over_ranks(value: float, world: int, dry: bool) -> float:
    if world == 1:
        return value
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64)
    if not dry and dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# per-GPU batch of the benchmark configurations (v8s: 1024 images over 8 GPUs)
DEFAULT_BATCH = {"v5s": 256, "v8s": 128, "v7pose": 512, "v6face": 512, "v3": 256, "v4": 256,
                 # variants outside the benchmark configurations: the batch of the nearest configuration
                 "v5face": 256, "v5pose": 256, "v7": 256, "v7face": 256, "v6": 256, "v6pose": 256,
                 "v5m": 128, "v8m": 64, "v8n": 128, "v6facen": 512, "v6m": 64, "v6l": 32, "v6facem": 64, "v6posel": 32, "v7x": 64, "v5facem": 128, "v5posex": 32}
LABEL = {"v5s": "YOLOv5s VOC-20 detect", "v8s": "YOLOv8s VOC-20 detect", "v7pose": "YOLOv7-l pose (17 keypoints)",
         "v6face": "YOLOv6-s face (5 landmarks, 4 levels)", "v3": "YOLOv3 VOC-20 detect", "v4": "YOLOv4 VOC-20 detect",
         "v5face": "YOLOv5s face (5 landmarks)", "v5pose": "YOLOv5s pose (17 keypoints)", "v7": "YOLOv7-l VOC-20 detect",
         "v7face": "YOLOv7-l face (5 landmarks)", "v6": "YOLOv6-s VOC-20 detect (4 levels)",
         "v6pose": "YOLOv6-s pose (17 keypoints, 4 levels)", "v5m": "YOLOv5m VOC-20 detect", "v8m": "YOLOv8m VOC-20 detect",
         "v8n": "YOLOv8n VOC-20 detect", "v6facen": "YOLOv6-n face (5 landmarks, 4 levels)",
         "v6m": "YOLOv6-m VOC-20 detect (4 levels, DFL)", "v6l": "YOLOv6-l VOC-20 detect (4 levels, DFL)", "v6facem": "YOLOv6-m face (5 landmarks, 4 levels, DFL)",
         "v6posel": "YOLOv6-l pose (17 keypoints, 4 levels, DFL)", "v7x": "YOLOv7-x VOC-20 detect", "v5facem": "YOLOv5m face (5 landmarks)",
         "v5posex": "YOLOv5x pose (17 keypoints)"}


def build_model(name: str, weights: str):
    from yolo_amd import models
    from yolo_amd.init import seeded_state_dict
    p, m = models.create(name)
    opts = models.init_options(name) if weights == "spread" else {}
    sd = seeded_state_dict(m, seed=0, mode=weights, **opts)
    m.load_state_dict(sd)
    return p, m, sd


class HostSink:
    """Device->host hand-over of the detections, pipelined one step deep.

    Per step the kept rows are packed back to back on the device (compute stream); a second stream then
    copies the (B+1) offsets and, once the host knows the total, the kept rows into pinned memory.  The host
    only waits for step i's offsets after it has enqueued step i+1's kernels, so the copies and the host-side
    launch work overlap the next step's compute.  Two device/host buffer sets alternate; ``drain`` completes
    the last step.  Every step's rows are on the host when ``drain`` returns."""

    def __init__(self, batch, a, width, device):
        from yolo_amd import lib as L
        self.L = L
        self.width = width
        self.copy_stream = torch.cuda.Stream(device=device)
        self.slots = []
        for _ in range(2):
            self.slots.append(dict(
                compact=torch.empty((batch * a, width), dtype=torch.float32, device=device),
                offsets=torch.empty((batch + 1,), dtype=torch.int64, device=device),
                h_offsets=torch.empty((batch + 1,), dtype=torch.int64).pin_memory(),
                h_rows=torch.empty((batch * a, width), dtype=torch.float32).pin_memory(),
                packed=torch.cuda.Event(), offsets_done=torch.cuda.Event(), rows_done=None, keep=None))
        self.n = 0
        self.pending = None
        self.last_total = 0
        self.h_offsets = self.slots[0]["h_offsets"]

    def submit(self, rows, counts):
        """Enqueue packing + the offsets copy of this step, then complete the previous step's row copy."""
        L = self.L
        slot = self.slots[self.n % 2]
        self.n += 1
        cur = torch.cuda.current_stream()
        if slot["rows_done"] is not None:
            cur.wait_event(slot["rows_done"])          # the copy that last read this slot's device buffer
        B, max_out, W = rows.shape
        L.check(L.LIB.yamd_compact_rows(rows.data_ptr(), counts.data_ptr(), B, max_out, W, slot["compact"].data_ptr(),
                                        slot["offsets"].data_ptr(), L.current_stream()), "compact_rows")
        slot["packed"].record(cur)
        slot["keep"] = (rows, counts)                  # keep the producers' memory alive until packed
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(slot["packed"])
            slot["h_offsets"].copy_(slot["offsets"], non_blocking=True)
            slot["offsets_done"].record(self.copy_stream)
        prev, self.pending = self.pending, slot
        if prev is not None:
            self._finish(prev)

    def _finish(self, slot):
        slot["offsets_done"].synchronize()
        total = int(slot["h_offsets"][-1])
        with torch.cuda.stream(self.copy_stream):
            if total:
                slot["h_rows"][:total].copy_(slot["compact"][:total], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
            slot["rows_done"] = ev
        slot["keep"] = None
        self.last_total = total
        self.h_offsets = slot["h_offsets"]

    def drain(self):
        if self.pending is not None:
            self._finish(self.pending)
            self.pending = None
        self.copy_stream.synchronize()
        return self.last_total


def step(model, images, sink):
    rows, counts = model.predict_device(images)
    sink.submit(rows, counts)


def stage_profile(model, p, images, reps=3):
    """Per-kernel timing with events on the launch stream (torch's current stream), after the timed region."""
    from yolo_amd import lib as L
    from yolo_amd import torch_utils as tu
    prog = model.program()
    B = images.shape[0]
    agg = {}
    for _ in range(reps):
        t = []
        prog.run(images, timings=t)
        torch.cuda.synchronize()
        for kind, name, e0, e1, fl, by in t:
            a = agg.setdefault((kind, name), [0.0, fl, by, 0])
            a[0] += e0.elapsed_time(e1)
            a[3] += 1
    table = [dict(kind=k, name=n, ms=v[0] / v[3], flops=v[1], bytes=v[2]) for (k, n), v in agg.items()]
    conv_ms = sum(r["ms"] for r in table if r["kind"] == "conv")
    conv_fl = sum(r["flops"] for r in table if r["kind"] == "conv")
    n_conv = sum(1 for r in table if r["kind"] == "conv")
    # decode and NMS stages
    raws = model.raw_heads(images)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    dec_ms = nms_ms = 0.0
    for _ in range(reps):
        ev[0].record()
        pred = model.decode_predict(raws)
        ev[1].record()
        rows, counts = model.nms_predict(pred)
        ev[2].record()
        torch.cuda.synchronize()
        dec_ms += ev[0].elapsed_time(ev[1]) / reps
        nms_ms += ev[1].elapsed_time(ev[2]) / reps
    dec_bytes = float(sum(r.numel() for r in raws) + pred.numel()) * 4.0      # raw logits read + decoded rows written
    return dict(table=table, conv_ms=conv_ms, conv_flops=conv_fl, n_conv=n_conv, decode_ms=dec_ms, decode_bytes=dec_bytes,
                nms_ms=nms_ms, other_ms=sum(r["ms"] for r in table if r["kind"] != "conv"))


def realistic_regime(args, p, model, images, sink_factory, steps=5):
    """Second NMS regime for the record: the same model with well-spread random weights and an objectness bias of
    -5 on every head (what a trained detector looks like to NMS: O(10^2) candidates per image instead of all 25 200).
    Same step, same timing rule (rank-local; reported next to the headline, never as ``value``)."""
    from yolo_amd import models
    from yolo_amd.init import seeded_state_dict
    sd = seeded_state_dict(model, seed=0, mode="spread", **models.init_options(args.model))
    C = 5 + p.class_num
    for k in sd:
        if k.startswith("body_net.yolo_head_P") and k.endswith(".bias"):
            sd[k] = sd[k].clone()
            sd[k][4::C] -= 5.0
    model.load_state_dict(sd)
    sink = sink_factory()
    for _ in range(2):
        step(model, images, sink)
    sink.drain()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(model, images, sink)
    sink.drain()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    pred = model.decode_predict(model.raw_heads(images[:8]))
    cand = ((pred[..., 4] * pred[..., -p.class_num:].max(-1).values) >= p.conf_threshold).sum(1).float().mean().item()
    counts = np.diff(sink.h_offsets.numpy())
    return {"images_per_s": round(images.shape[0] / dt, 1), "ms_per_step": round(dt * 1e3, 3),
            "candidates_per_image": round(cand, 1), "kept_per_image": round(float(counts.mean()), 1),
            "weights": "spread random init, objectness bias -5 (seed 0)"}


def measured_conv_traffic(model_name: str, batch: int, n_conv: int):
    """HBM bytes per conv launch from the committed PMC passes (profiles/*_conv_hbm_traffic.json, written by
    scratch/pmc_conv.py from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same command): counters cannot be
    read from inside the process, so the latest committed measurement of the same workload is reported."""
    import glob
    if model_name != "v5s" or batch != 256:
        return None, None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_conv_hbm_traffic.json")))
    for f in reversed(files):
        with open(f) as fh:
            d = json.load(fh)
        if d.get("convs_per_pass") == n_conv:
            return float(d["bytes_per_launch"]), os.path.relpath(f, ROOT)
    return None, None


def cpu_baseline(name, sd, p, n_images):
    """The CPU oracle (torch fp32 graph + numpy decode + C NMS) timed on the host cores: reported baseline."""
    from oracle import graphs_ref, postprocess
    from yolo_amd.init import seeded_images
    threads = max(1, min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16"))))
    torch.set_num_threads(threads)
    imgs = torch.from_numpy(seeded_images(n_images, p.img_size, seed=0))
    ref = graphs_ref.model_ref(name)

    def once(x):
        with torch.no_grad():
            raws = [t.numpy() for t in ref["body"](sd, x)]
        pred = ref["decode"](raws, p, True)
        return postprocess.non_max_suppression(pred, p.img_size, conf_threshold=p.conf_threshold,
                                               nms_threshold=p.nms_threshold, **ref["nms"])

    once(imgs[:1])
    t0 = time.perf_counter()
    once(imgs)
    dt = time.perf_counter() - t0
    return dict(value=round(n_images / dt, 3), unit="images/s", cores=threads, kind="port",
                sample=f"{n_images} images {p.img_size}x{p.img_size} in one batch, 1 timed pass after a 1-image warm-up; torch-CPU fp32 conv "
                       f"graph + numpy decode + C NMS of oracle/ ({dt:.2f} s)")


def main():
    args = parse()
    world, rank, local = dist_setup(args)
    if args.dry:
        # rehearsal of the rank logic without a GPU: every rank "works" rank-dependent time
        barrier(world, True)
        t0 = time.perf_counter()
        time.sleep(0.05 * (rank + 1))
        el = max_over_ranks(time.perf_counter() - t0, world, True)
        if rank == 0:
            print(json.dumps({"dry": True, "n_gpus": world, "elapsed_max": el}))
        return
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    from yolo_amd.init import seeded_images
    p, model, sd = build_model(args.model, args.weights)
    B = args.batch or DEFAULT_BATCH[args.model]
    IMG = p.img_size
    # synthetic uniform[0,1) images, seeded per global image index; resident in HBM before timing starts
    base = seeded_images(min(B, 32), IMG, seed=0, start=rank * B)
    images = torch.from_numpy(base).to(dev).repeat((B + base.shape[0] - 1) // base.shape[0], 1, 1, 1)[:B].contiguous()
    rows0, _ = model.predict_device(images[:1])
    A, width = rows0.shape[1], rows0.shape[2]
    sink = HostSink(B, A, width, dev)

    for _ in range(args.warmup):
        step(model, images, sink)
    sink.drain()
    barrier(world, False)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(model, images, sink)
    sink.drain()                      # the last step's detections are on the host before the clock stops
    barrier(world, False)
    elapsed = max_over_ranks(time.perf_counter() - t0, world, False)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        counts = np.diff(sink.h_offsets.numpy())
        prof = stage_profile(model, p, images)
        conv_tflops = prof["conv_flops"] / (prof["conv_ms"] * 1e-3) / 1e12
        # candidates per image (score >= conf) on the first images, for the record
        pred = model.decode_predict(model.raw_heads(images[:8]))
        ncls = p.class_num
        cand = ((pred[..., 4] * pred[..., -ncls:].max(-1).values) >= p.conf_threshold).sum(1).float().mean().item()
        traffic, traffic_src = measured_conv_traffic(args.model, B, prof["n_conv"])
        out = {
            "metric": METRIC, "value": round(value, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "%s, %dx%d, batch %d per GPU, conv graph bf16 (fp32 accumulate) + "
                                   "fp32 decode + per-class NMS + D2H of kept rows; random-init weights (%s, seed 0)"
                                   % (LABEL[args.model], IMG, IMG, B, args.weights),
                       "batch_per_gpu": B, "parallelism": "batch-sharded x%d, no collective" % world,
                       "conf_threshold": p.conf_threshold, "nms_threshold": p.nms_threshold,
                       "candidates_per_image": round(cand, 1), "kept_per_image": round(float(counts.mean()), 1),
                       "kept_max": int(counts.max())},
            "roofline": {"bound": "mfma", "achieved": round(conv_tflops, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(conv_tflops / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                         "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)", "traffic_source": traffic_src,
                         "kernel": "conv_igemm_kernel + stem_conv_kernel (all %d fused conv launches of one step)" % prof["n_conv"],
                         "flops_per_launch": prof["conv_flops"] / prof["n_conv"],
                         "avg_launch_ms": round(prof["conv_ms"] / prof["n_conv"], 4)},
            "stages_ms": {"conv": round(prof["conv_ms"], 3), "layout_pool_upsample": round(prof["other_ms"], 3),
                          "decode": round(prof["decode_ms"], 3), "nms": round(prof["nms_ms"], 3)},
            "decode_hbm": {"achieved_GBs": round(prof["decode_bytes"] / (prof["decode_ms"] * 1e-3) / 1e9, 1),
                           "peak_GBs": PEAK_HBM_GBS,
                           "frac": round(prof["decode_bytes"] / (prof["decode_ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.model, sd, p, args.cpu_images)
        if world == 1 and args.model == "v5s" and not args.no_cpu_baseline:
            out["nms_regime_realistic"] = realistic_regime(args, p, model, images, lambda: HostSink(B, A, width, dev))
        if args.detail:
            with open(args.detail, "w") as f:
                for r in sorted(prof["table"], key=lambda r: -r["ms"]):
                    tf = r["flops"] / (r["ms"] * 1e-3) / 1e12 if r["flops"] else 0.0
                    gb = r["bytes"] / (r["ms"] * 1e-3) / 1e9
                    f.write(f"{r['kind']:6s} {r['name']:34s} {r['ms']:8.4f} ms {tf:8.1f} TFLOP/s {gb:8.1f} GB/s(alg)\n")
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
