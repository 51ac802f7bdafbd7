#!/usr/bin/env python3
"""Headline benchmark: images/s of the YOLO inference hot path (conv graph -> decode -> confidence filter +
per-class NMS -> detections on the host) on MI355X.

  python bench.py --gpus N --steps K --warmup W [--model v5s|v8s|...] [--scaling weak|strong] [--global-batch B]

Default workload: YOLOv5s 640x640, 256 images per GPU (weak scaling).  ``--model v8s`` defaults to the other
benchmark configuration: 1024 images in total, split into contiguous slices over the GPUs (strong scaling).

N > 1: one rank per GPU.  Either the caller starts the ranks (torch.distributed.run sets RANK / LOCAL_RANK /
WORLD_SIZE), or -- when WORLD_SIZE is not set -- this script starts N child processes itself before it touches a
GPU and relays rank 0's JSON line.  Ranks never exchange data on the data path; the process group only provides the
barriers on both sides of the timed region and the MAX over the ranks' times.

One step = one pass of the whole path over one batch that is already resident in HBM: the fused conv kernels
(bf16 MFMA), head decode, NMS, then the kept rows are copied to the host (counts + rows), which is what the
reference's ``predict`` returns.  The copy of step i runs on a second stream under step i+1's kernels; all K steps'
detections are on the host before the clock stops.
Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

METRIC = "images/sec (whole node) YOLOv5s/v8s 640² incl. decode+NMS; % MFMA/HBM roofline"
PEAK_BF16_TFLOPS = 2500.0     # nominal dense bf16 MFMA peak of one MI355X
PEAK_HBM_GBS = 8000.0         # nominal HBM3E peak of one MI355X

# per-GPU batch of the weak-scaling configurations
DEFAULT_BATCH = {"v5s": 256, "v8s": 128, "v7pose": 512, "v6face": 512, "v3": 256, "v4": 256,
                 # variants outside the benchmark configurations: the batch of the nearest configuration
                 "v5face": 256, "v5pose": 256, "v7": 256, "v7face": 256, "v6": 256, "v6pose": 256,
                 "v5m": 128, "v8m": 64, "v8n": 128, "v6facen": 512, "v6m": 64, "v6l": 32, "v6facem": 64, "v6posel": 32,
                 "v7x": 64, "v5facem": 128, "v5posex": 32}
# configurations defined by a total batch that is split over the GPUs
DEFAULT_GLOBAL_BATCH = {"v8s": 1024}
LABEL = {"v5s": "YOLOv5s VOC-20 detect", "v8s": "YOLOv8s VOC-20 detect", "v7pose": "YOLOv7-l pose (17 keypoints)",
         "v6face": "YOLOv6-s face (5 landmarks, 4 levels)", "v3": "YOLOv3 VOC-20 detect", "v4": "YOLOv4 VOC-20 detect",
         "v5face": "YOLOv5s face (5 landmarks)", "v5pose": "YOLOv5s pose (17 keypoints)", "v7": "YOLOv7-l VOC-20 detect",
         "v7face": "YOLOv7-l face (5 landmarks)", "v6": "YOLOv6-s VOC-20 detect (4 levels)",
         "v6pose": "YOLOv6-s pose (17 keypoints, 4 levels)", "v5m": "YOLOv5m VOC-20 detect", "v8m": "YOLOv8m VOC-20 detect",
         "v8n": "YOLOv8n VOC-20 detect", "v6facen": "YOLOv6-n face (5 landmarks, 4 levels)",
         "v6m": "YOLOv6-m VOC-20 detect (4 levels, DFL)", "v6l": "YOLOv6-l VOC-20 detect (4 levels, DFL)",
         "v6facem": "YOLOv6-m face (5 landmarks, 4 levels, DFL)", "v6posel": "YOLOv6-l pose (17 keypoints, 4 levels, DFL)",
         "v7x": "YOLOv7-x VOC-20 detect", "v5facem": "YOLOv5m face (5 landmarks)", "v5posex": "YOLOv5x pose (17 keypoints)"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--model", default="v5s", help="model family (yolo_amd.models.registry); v5s is the headline config")
    ap.add_argument("--scaling", default="", choices=["", "weak", "strong"],
                    help="weak: --batch images on every GPU; strong: --global-batch images split over the GPUs "
                         "(default: strong for v8s, weak otherwise)")
    ap.add_argument("--batch", type=int, default=0, help="weak scaling: images per GPU")
    ap.add_argument("--global-batch", type=int, default=0, help="strong scaling: images in total")
    ap.add_argument("--weights", default="torch_default", choices=["torch_default", "spread"],
                    help="seeded random-init scheme (torch_default = the reference's YoloV5(param) distribution)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the second NMS regime and the peak microbenchmarks (profiling runs: every launch of the "
                         "process then belongs to a full-batch step)")
    ap.add_argument("--realistic", action="store_true", help="with --no-extras: still time the second NMS regime")
    ap.add_argument("--two-stream", action="store_true", help="NMS of step i on a second stream under step i+1's convs")
    ap.add_argument("--detail", default="", help="write the per-op timing table to this file")
    ap.add_argument("--backend", default="gloo", choices=["gloo", "nccl"],
                    help="process-group backend for the two barriers and the MAX over the ranks' times.  gloo (default) keeps "
                         "them on the host: the data path has no collective, so RCCL never has to come up; nccl = RCCL")
    ap.add_argument("--rank-timeout", type=float, default=900.0, help="seconds a rank waits for the others (rendezvous, barriers)")
    ap.add_argument("--dry", action="store_true", help="no GPU work: rehearse the multi-rank protocol only")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------
# ranks
def spawn_ranks(n: int) -> int:
    """WORLD_SIZE is not set and --gpus N > 1: start one child per GPU (fresh processes; this parent has made no GPU
    call), relay rank 0's stdout and every rank's stderr, and fail as soon as any rank fails: the others are then ended
    instead of waiting in a barrier for a rank that will never arrive."""
    import socket
    import tempfile
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs, errs = [], []
    out0 = tempfile.TemporaryFile(mode="w+")
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        errs.append(tempfile.TemporaryFile(mode="w+"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL, stderr=errs[-1], text=True))
    rcs = [None] * n
    failed = None
    while any(rc is None for rc in rcs):
        for r, p in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = p.poll()
                if rcs[r] not in (None, 0) and failed is None:
                    failed = r
        if failed is not None:
            for r, p in enumerate(procs):
                if rcs[r] is None:
                    p.terminate()
            for r, p in enumerate(procs):
                if rcs[r] is None:
                    try:
                        rcs[r] = p.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        rcs[r] = p.wait()
            break
        time.sleep(0.05)
    out0.seek(0)
    sys.stdout.write(out0.read())
    sys.stdout.flush()
    for r, f in enumerate(errs):
        f.seek(0)
        text = f.read()
        if text.strip():
            sys.stderr.write("".join("[rank %d] %s\n" % (r, line) for line in text.splitlines()))
    if failed is not None:
        sys.stderr.write("rank %d exited with code %s; the other ranks were ended\n" % (failed, rcs[failed]))
        return abs(rcs[failed]) or 1
    return 0


def dist_setup(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if not args.dry:
            local %= max(1, torch.cuda.device_count())      # more ranks than GPUs (a rehearsal on one GPU): ranks share devices
            torch.cuda.set_device(local)
        import datetime
        dist.init_process_group(backend=args.backend, rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=args.rank_timeout))
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


def plan_batches(args, world: int):
    """(scaling, images of every rank, global batch)."""
    from yolo_amd.parallel import shard_slices
    scaling = args.scaling or ("strong" if args.model in DEFAULT_GLOBAL_BATCH and not args.batch else "weak")
    if scaling == "strong":
        total = args.global_batch or DEFAULT_GLOBAL_BATCH.get(args.model, DEFAULT_BATCH[args.model] * 8)
        sizes = [s.stop - s.start for s in shard_slices(total, world)]
        starts = [s.start for s in shard_slices(total, world)]
    else:
        per = args.batch or DEFAULT_BATCH[args.model]
        sizes = [per] * world
        starts = [r * per for r in range(world)]
        total = per * world
    return scaling, sizes, starts, total


# ---------------------------------------------------------------------------------------------------------------
def build_model(name: str, weights: str):
    from yolo_amd import models
    from yolo_amd.init import seeded_state_dict
    p, m = models.create(name)
    opts = models.init_options(name) if weights == "spread" else {}
    sd = seeded_state_dict(m, seed=0, mode=weights, **opts)
    m.load_state_dict(sd)
    return p, m, sd


def device_images(n: int, size: int, first_index: int, dev) -> torch.Tensor:
    """n distinct uniform[0,1) images generated on the device; image i depends only on its global index."""
    out = torch.empty((n, 3, size, size), dtype=torch.float32, device=dev)
    g = torch.Generator(device=dev)
    for i in range(n):
        g.manual_seed(1000003 + first_index + i)
        out[i].uniform_(0.0, 1.0, generator=g)
    return out


from yolo_amd.pipeline import HostSink, TwoStreamStep  # noqa: E402


def step(model, images, sink):
    """One step on the launch stream: the whole path for one batch, then the hand-over of its rows."""
    rows, counts = model.predict_device(images)
    sink.submit(rows, counts)


def stage_profile(model, images, reps=3):
    """Per-op timing with events on the launch stream (torch's current stream): the conv program replayed on its
    own after the timed region (nothing else on the GPU), then decode and NMS."""
    prog = model.program()
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
    raws = model.raw_heads(images)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    dec_ms = nms_ms = 0.0
    for _ in range(reps):
        ev[0].record()
        pred = model.decode_predict(raws, model._arena)
        ev[1].record()
        model.nms_predict(pred, model._arena)
        ev[2].record()
        torch.cuda.synchronize()
        dec_ms += ev[0].elapsed_time(ev[1]) / reps
        nms_ms += ev[1].elapsed_time(ev[2]) / reps
    dec_bytes = float(sum(r.numel() for r in raws) + pred.numel()) * 4.0      # raw logits read + decoded rows written
    # what the step itself runs after the conv program: for the anchor families one call in which the decode kernel
    # filters its own rows (decode + filter + NMS), else the two calls timed above
    post_ms = dec_ms + nms_ms
    if hasattr(model, "decode_nms_predict"):
        post_ms = 0.0
        for _ in range(reps):
            ev[0].record()
            model.decode_nms_predict(raws, model._arena)
            ev[1].record()
            torch.cuda.synchronize()
            post_ms += ev[0].elapsed_time(ev[1]) / reps
        pred = model.decode_predict(raws, model._arena)       # (the fused call leaves the decoded tensor incomplete)
    return dict(table=table, pred=pred, post_ms=post_ms,
                conv_ms=sum(r["ms"] for r in table if r["kind"] == "conv"),
                conv_flops=sum(r["flops"] for r in table if r["kind"] == "conv"),
                conv_bytes=sum(r["bytes"] for r in table if r["kind"] == "conv"),
                n_conv=sum(1 for r in table if r["kind"] == "conv"), decode_ms=dec_ms, decode_bytes=dec_bytes,
                nms_ms=nms_ms, other_ms=sum(r["ms"] for r in table if r["kind"] != "conv"))


def candidate_stats(p, pred):
    """Candidates per image and the candidate pairs inside (image, class) buckets, n(n-1)/2 each: the IoU tests a dense
    per-class suppression mask would evaluate, the unit the NMS stage is reported in."""
    nc = p.class_num
    scores, cls = pred[..., -nc:].max(-1)
    passed = (pred[..., 4] * scores) >= p.conf_threshold
    B = pred.shape[0]
    key = (torch.arange(B, device=pred.device)[:, None] * nc + cls)[passed]
    n = torch.bincount(key, minlength=B * nc).double()
    return float(passed.sum(1).float().mean().item()), float((n * (n - 1) / 2).sum().item())


def confidence_bias_keys(sd):
    """The head-conv biases that set how many anchors pass the confidence filter, as (key, step): ``bias[4::step]`` is
    the objectness logit of every anchor of an anchor-based head (v3 / v4 / v5 / v7, step = columns per anchor); step 0
    = every element (the class-score conv of the anchor-free heads of v6 / v8)."""
    import re
    out = []
    for k in sd:
        if re.fullmatch(r"body_net\.(yolo_head_P\d|last_layer\d\.6|yolo_head\d\.1)\.bias", k):
            out.append((k, sd[k].numel() // 3))
        elif re.fullmatch(r"body_net\.cv3\.\d\.2\.bias|yolo_body\.head\.cls_predict_list\.\d\.bias", k):
            out.append((k, 0))
    return out


def realistic_regime(args, p, model, images, sink, steps=5, target=300.0):
    """Second NMS regime for the record: the same model with well-spread random weights and a negative bias on the
    confidence logits of every head (what a trained detector looks like to NMS: O(10^2) candidates per image instead of
    all anchors, or none at all).  The bias is the one of +6 ... -8 whose candidate count per image is nearest ``target``
    (the single-class face / pose heads at threshold 0.3 need a positive one to have candidates at all).
    Same step, same timing rule (rank-local; reported next to the headline, never as ``value``)."""
    from yolo_amd import models
    from yolo_amd.init import seeded_state_dict
    base = seeded_state_dict(model, seed=0, mode="spread", **models.init_options(args.model))
    keys = confidence_bias_keys(base)
    if not keys:
        return {"skipped": "no confidence bias found in this model's state_dict"}

    def load(shift):
        sd = dict(base)
        for k, stepc in keys:
            sd[k] = base[k].clone()
            if stepc:
                sd[k][4::stepc] -= shift
            else:
                sd[k] -= shift
        model.load_state_dict(sd)

    sub = images[:min(8, images.shape[0])]

    def count(shift):
        load(shift)
        return candidate_stats(p, model.decode_predict(model.raw_heads(sub)))[0]

    def miss(cand):
        return abs(np.log(max(cand, 1e-3) / target))

    grid = (-6.0, -4.0, -3.0, -2.0, -1.0, 0.0, 1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 8.0)
    counts = [count(sh) for sh in grid]
    best = min(zip(grid, counts), key=lambda t: miss(t[1]))
    # the count falls with the shift; where two grid points straddle the target (single-class heads go from every
    # anchor to none within one unit), halve the interval a few times
    for lo, hi, clo, chi in zip(grid[:-1], grid[1:], counts[:-1], counts[1:]):
        if clo >= target >= chi and miss(best[1]) > 0.3:
            for _ in range(6):
                mid = 0.5 * (lo + hi)
                cm = count(mid)
                if miss(cm) < miss(best[1]):
                    best = (mid, cm)
                if cm >= target:
                    lo = mid
                else:
                    hi = mid
            break
    load(best[0])
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
    cand, _ = candidate_stats(p, model.decode_predict(model.raw_heads(images)))
    counts = np.diff(sink.h_offsets.numpy())
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    pred = model.decode_predict(model.raw_heads(images), model._arena)
    ev[0].record()
    model.nms_predict(pred, model._arena)
    ev[1].record()
    torch.cuda.synchronize()
    return {"images_per_s": round(images.shape[0] / dt, 1), "ms_per_step": round(dt * 1e3, 3),
            "candidates_per_image": round(cand, 1), "kept_per_image": round(float(counts.mean()), 1),
            "nms_ms": round(ev[0].elapsed_time(ev[1]), 3),
            "weights": "spread random init, confidence-logit bias %+g (seed 0)" % -best[0]}


def two_stream_rate(model, images, sink, dev, steps):
    """The same K steps with each step's NMS on a second stream (yolo_amd.pipeline.TwoStreamStep), same timing rule.
    Reported next to the headline, never as ``value``: under it the conv kernels share the device with the NMS kernels, so
    their durations no longer measure the conv kernels alone."""
    ts = TwoStreamStep(model, dev)
    for _ in range(4):
        ts(images, sink)
    sink.drain()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ts(images, sink)
    sink.drain()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"images_per_s": round(images.shape[0] / dt, 1), "ms_per_step": round(dt * 1e3, 3),
            "what": "NMS, row packing and copies of step i on a second stream under the conv graph of step i+1"}


def small_batch_latency(model, p, dev, batches=(1, 16), reps=20):
    """Host-visible latency of ``predict`` (H2D of the images not included: they are resident; D2H of the detections
    included) at small batch sizes: median over ``reps`` synchronous calls after 3 warm-ups."""
    out = {}
    for B in batches:
        imgs = device_images(B, p.img_size, 10_000, dev)
        for _ in range(3):
            model.predict({"batch_images": imgs})
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model.predict({"batch_images": imgs})
            ts.append(time.perf_counter() - t0)
        ms = statistics.median(ts) * 1e3
        out["batch%d" % B] = {"ms": round(ms, 3), "images_per_s": round(B / ms * 1e3, 1)}
    return out


def measured_peaks(dev):
    """The two denominators measured on this device, beside the nominal ones: a register-operand bf16 MFMA loop on
    random data (2 wavefronts per SIMD) and a 16-byte streaming copy of 1 GiB (read + write bytes)."""
    from yolo_amd import lib as L
    st = L.current_stream()
    ops = (torch.rand(4096, device=dev) * 2 - 1).to(torch.bfloat16)
    blocks, iters = 512, 4096
    sink = torch.empty(blocks * 256, dtype=torch.float32, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed(fn, warm, reps):
        for _ in range(warm):
            fn()
        ts = []
        for _ in range(reps):
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return statistics.median(ts)

    ms = timed(lambda: L.check(L.LIB.yamd_microbench_mfma_bf16(ops.data_ptr(), sink.data_ptr(), blocks, iters, st)), 30, 15)
    mfma = blocks * 4 * 4 * iters * 32768.0 / (ms * 1e-3) / 1e12
    nbytes = 1 << 30
    src = torch.empty(nbytes, dtype=torch.uint8, device=dev).random_(0, 255)
    dst = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    variants = {v: timed(lambda v=v: L.check(L.LIB.yamd_microbench_copy(src.data_ptr(), dst.data_ptr(), nbytes, v, st)), 3, 7)
                for v in range(4)}
    best = min(variants, key=variants.get)
    return {"mfma_bf16_TFLOPs": round(mfma, 1), "mfma_loop": "v_mfma_f32_32x32x16_bf16, register operands, random data, "
            "8 wavefronts per CU, %.2f ms per launch" % ms,
            "hbm_copy_GBs": round(2.0 * nbytes / (variants[best] * 1e-3) / 1e9, 1),
            "hbm_copy": "1 GiB copy with 16-byte accesses, read + write bytes, best of four variants (%s)" % ", ".join(
                "%s%d in flight: %.0f GB/s" % ("non-temporal " if v >= 2 else "", 4 if v % 2 == 0 else 8,
                                                2.0 * nbytes / (t * 1e-3) / 1e9) for v, t in variants.items())}


def sustained_clock(load, dev, min_ms: float = 30.0):
    """Shader clock (GHz) the device holds while ``load()`` keeps the launch stream busy: a one-wavefront probe on a
    second stream samples the shader clock counter against the constant-rate counter (calibrated here against the
    probe's own event time).  Median over the samples taken while the load was running."""
    from yolo_amd import lib as L
    n, naps = 2000, 3
    samples = torch.zeros(2 * n, dtype=torch.int64, device=dev)
    side = torch.cuda.Stream(device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    m0, m1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    load()
    torch.cuda.synchronize()
    m0.record()
    load()
    m1.record()
    torch.cuda.synchronize()
    load_ms = m0.elapsed_time(m1)
    reps = max(2, int(np.ceil(min_ms * 1.6 / max(load_ms, 1e-3))))     # the probe takes about min_ms; the load outlasts it
    load()
    with torch.cuda.stream(side):
        e0.record(side)
        L.check(L.LIB.yamd_microbench_clock_probe(samples.data_ptr(), n, naps, int(side.cuda_stream)))
        e1.record(side)
    for _ in range(reps):
        load()
    torch.cuda.synchronize()
    s = samples.cpu().numpy().reshape(n, 2).astype(np.float64)
    probe_ms = e0.elapsed_time(e1)
    tick_hz = (s[-1, 1] - s[0, 1]) / (probe_ms * 1e-3)            # rate of the constant counter, measured
    dc, dr = np.diff(s[:, 0]), np.diff(s[:, 1])
    ghz = dc / np.maximum(dr, 1.0) * tick_hz / 1e9
    keep = ghz[n // 10: -n // 10]
    return {"GHz": round(float(np.median(keep)), 3), "p10": round(float(np.percentile(keep, 10)), 3),
            "p90": round(float(np.percentile(keep, 90)), 3), "probe_ms": round(probe_ms, 2),
            "constant_counter_MHz": round(tick_hz / 1e6, 2), "load_repetitions": reps + 1,
            "load_ms_each": round(load_ms, 3)}


def measured_conv_traffic(model_name: str, batch: int, signature):
    """HBM bytes per conv launch from the committed PMC passes (profiles/*_conv_hbm_traffic.json, written by
    scratch/layer_counters.sh from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs over the conv program of this same
    workload): counters cannot be read from inside the process, so the latest committed measurement is reported --
    but only one whose list of conv launches (layer, shape, tiling of every launch) equals the running plan's.  A file
    measured on other kernels or tilings is refused and ``traffic`` stays null (the reason goes to ``traffic_source``)."""
    import glob
    if model_name != "v5s" or batch != 256:
        return None, None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_conv_hbm_traffic.json")))      # names sort by round
    reason = "no profiles/*_conv_hbm_traffic.json"
    for f in reversed(files):
        with open(f) as fh:
            d = json.load(fh)
        if d.get("model") != model_name or d.get("batch") != batch:
            continue
        if d.get("conv_launches") == list(signature):
            return float(d["bytes_per_launch"]), os.path.relpath(f, ROOT)
        reason = ("refused %s: its conv launch list differs from the running plan's (re-measure with "
                  "scratch/layer_counters.sh)" % os.path.relpath(f, ROOT))
        break
    return None, reason


def h2d_rate(B: int, size: int, dev, ms_per_step: float, reps: int = 3):
    """Host -> device feed of one rank's batch, outside the timed region (the timed steps start from images resident
    in HBM): a page-locked host buffer copied to the device on the launch stream, timed with events, best of ``reps``.
    Two forms: the float32 (B,3,S,S) tensor the reference's ``predict`` / ``YoloPredict.predict`` boundary hands over,
    and the uint8 (B,S,S,3) frames the letterbox kernel of this library takes instead (a quarter of the bytes)."""
    out = {}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for key, shape, dtype in (("fp32_nchw", (B, 3, size, size), torch.float32), ("uint8_hwc", (B, size, size, 3), torch.uint8)):
        host = torch.empty(shape, dtype=dtype).pin_memory()
        host.view(torch.uint8).view(-1)[::4096] = 1          # touch every page
        devbuf = torch.empty(shape, dtype=dtype, device=dev)
        best = float("inf")
        for _ in range(reps + 1):
            e0.record()
            devbuf.copy_(host, non_blocking=True)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        nbytes = host.numel() * host.element_size()
        out[key] = {"bytes": nbytes, "ms": round(best, 3), "GBs": round(nbytes / (best * 1e-3) / 1e9, 2),
                    "ms_over_step_ms": round(best / ms_per_step, 3)}
        del host, devbuf
    out["what"] = ("one rank's batch from page-locked host memory to the device, copy alone (events on the launch stream, "
                   "best of %d); not part of `value`, whose steps start from HBM-resident images.  ms_over_step_ms > 1 means "
                   "the link, not the GPU, would bound a pipeline fed in that form" % reps)
    return out


def cpu_baseline(name, sd, p):
    """The CPU oracle (torch fp32 graph + numpy decode + C NMS) timed on the host cores at batch 1 and batch 16:
    median of 5 passes after 2 warm-ups each.  ``value`` is the batch-16 rate."""
    from oracle import graphs_ref, postprocess
    from yolo_amd.init import seeded_images
    threads = max(1, min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16"))))
    torch.set_num_threads(threads)
    imgs = torch.from_numpy(seeded_images(16, p.img_size, seed=0))
    ref = graphs_ref.model_ref(name)

    def once(x):
        with torch.no_grad():
            raws = [t.numpy() for t in ref["body"](sd, x)]
        pred = ref["decode"](raws, p, True)
        return postprocess.non_max_suppression(pred, p.img_size, conf_threshold=p.conf_threshold,
                                               nms_threshold=p.nms_threshold, **ref["nms"])

    def median_s(x, warm=2, reps=5):
        for _ in range(warm):
            once(x)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            once(x)
            ts.append(time.perf_counter() - t0)
        return statistics.median(ts)

    t1, t16 = median_s(imgs[:1]), median_s(imgs)
    return dict(value=round(16 / t16, 3), unit="images/s", cores=threads, kind="port", os_cpu_count=os.cpu_count(),
                batch1_ms=round(t1 * 1e3, 1), batch1_images_per_s=round(1 / t1, 3),
                sample=f"{p.img_size}x{p.img_size} images, batch 16 (value) and batch 1, median of 5 passes after 2 "
                       f"warm-ups each, {threads} torch threads (OMP_NUM_THREADS; os.cpu_count() = {os.cpu_count()}); "
                       f"torch-CPU fp32 conv graph + numpy decode + C NMS of oracle/")


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    if os.environ.get("YAMD_BENCH_FAIL_RANK") == os.environ.get("RANK", "0"):      # test hook: this rank dies at start-up
        print("rank %s: induced start-up failure" % os.environ.get("RANK", "0"), file=sys.stderr)
        sys.exit(3)
    world, rank, local = dist_setup(args)
    scaling, sizes, starts, total_images = plan_batches(args, world)
    if args.dry:
        # rehearsal of the rank protocol without a GPU: every rank "works" for a rank-dependent time on its slice
        barrier(world, True)
        t0 = time.perf_counter()
        time.sleep(0.05 * (rank + 1))
        el = max_over_ranks(time.perf_counter() - t0, world, True)
        if rank == 0:
            print(json.dumps({"dry": True, "n_gpus": world, "elapsed_max": el, "scaling": scaling, "images_per_rank": sizes,
                              "first_image_of_rank": starts, "global_batch": total_images}))
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()
        return
    if args.gpus != world and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    p, model, sd = build_model(args.model, args.weights)
    B = sizes[rank]
    IMG = p.img_size
    # the CPU baseline runs first (rank 0, one GPU only): the GPU phase then is the tail of the process
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        cpu = cpu_baseline(args.model, sd, p)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    from yolo_amd import lib as L
    L.check(L.LIB.yamd_init(), "yamd_init")
    images = device_images(B, IMG, starts[rank], dev)          # resident in HBM before timing starts
    rows0, _ = model.predict_device(images)                    # full batch: sizes every buffer
    A, width = rows0.shape[1], rows0.shape[2]
    sink = HostSink(B, A, width, dev)
    prog = model.program()
    if args.two_stream:
        two_stream = TwoStreamStep(model, dev)
        run_step = lambda: two_stream(images, sink)      # noqa: E731
    else:
        run_step = lambda: step(model, images, sink)     # noqa: E731

    for _ in range(args.warmup):
        run_step()
    sink.drain()
    spans = []
    prog.span_sink = spans          # conv launches bracketed by events on the launch stream, inside the timed region
    barrier(world, False)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step()
    sink.drain()                      # the last step's detections are on the host before the clock stops
    barrier(world, False)
    elapsed = max_over_ranks(time.perf_counter() - t0, world, False)
    prog.span_sink = None

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_images * args.steps / elapsed
        counts = np.diff(sink.h_offsets.numpy())
        conv_ms_pipe = sum(a.elapsed_time(b) for a, b in spans) / args.steps
        prof = stage_profile(model, images)
        n_conv = prof["n_conv"]
        conv_tflops = prof["conv_flops"] / (conv_ms_pipe * 1e-3) / 1e12
        conv_tflops_iso = prof["conv_flops"] / (prof["conv_ms"] * 1e-3) / 1e12
        cand, pairs = candidate_stats(p, prof["pred"])
        traffic, traffic_src = measured_conv_traffic(args.model, B, prog.conv_launch_signature())
        out = {
            "metric": METRIC, "value": round(value, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "%s, %dx%d, %s, conv graph bf16 (fp32 accumulate) + fp32 decode + per-class NMS + "
                                   "D2H of kept rows; random-init weights (%s, seed 0); every image distinct, "
                                   "uniform[0,1) seeded by its global index"
                                   % (LABEL[args.model], IMG, IMG,
                                      ("batch %d per GPU" % B) if scaling == "weak" else
                                      ("global batch %d split %s over %d GPU(s)" % (total_images, sizes, world)),
                                      args.weights),
                       "global_batch": total_images, "images_per_rank": sizes,
                       "parallelism": "batch-sharded x%d, no collective" % world,
                       "conf_threshold": p.conf_threshold, "nms_threshold": p.nms_threshold,
                       "candidates_per_image": round(cand, 1), "kept_per_image": round(float(counts.mean()), 1),
                       "kept_max": int(counts.max())},
            "roofline": {"bound": "mfma", "achieved": round(conv_tflops, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(conv_tflops / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                         "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)", "traffic_source": traffic_src,
                         "kernel": "conv_igemm_kernel + bneck_kernel + stem_conv2_kernel / stem_conv_kernel (all %d fused conv launches of one step)" % n_conv,
                         "how": "achieved = conv FLOPs of one step (2 x MACs of every conv launch) / conv time of one step; "
                                "conv time = HIP events on the launch stream around every run of consecutive conv launches "
                                "inside the timed region, summed over the %d steps / %d (in the pipeline: the previous step's "
                                "D2H copy runs underneath).  The rocprofv3 kernel trace of this command gives the same sum per kernel "
                                "(conv_igemm + bneck + stem rows of profiles/*_kernel_stats.csv / program runs) except for the first kernel of "
                                "a step, the fused stem, which runs longer under the profiler: the previous step's row copy is a "
                                "blit kernel beside it there" % (args.steps, args.steps),
                         "flops_per_launch": prof["conv_flops"] / n_conv,
                         "avg_launch_ms": round(conv_ms_pipe / n_conv, 4),
                         "conv_ms_in_pipeline": round(conv_ms_pipe, 3),
                         "conv_ms_isolated": round(prof["conv_ms"], 3),
                         "achieved_isolated": round(conv_tflops_iso, 2)},
            # the same launches against the other roof: every conv's input, output, residual and weights counted once
            "conv_hbm": {"algorithmic_GB_per_step": round(prof["conv_bytes"] / 1e9, 2),
                         "achieved_GBs": round(prof["conv_bytes"] / (conv_ms_pipe * 1e-3) / 1e9, 1), "peak_GBs": PEAK_HBM_GBS,
                         "frac": round(prof["conv_bytes"] / (conv_ms_pipe * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)},
            "stages_ms": {"conv": round(prof["conv_ms"], 3), "layout_pool_upsample": round(prof["other_ms"], 3),
                          "decode": round(prof["decode_ms"], 3), "nms": round(prof["nms_ms"], 3),
                          "decode_filter_nms_as_run": round(prof["post_ms"], 3),
                          "note": "decode / nms: the two stages as separate calls (what decode_hbm and nms below are computed "
                                  "from); decode_filter_nms_as_run: the post-processing the timed steps run, for anchor heads one "
                                  "call whose decode kernel filters its own rows and writes only the survivors"},
            "decode_hbm": {"achieved_GBs": round(prof["decode_bytes"] / (prof["decode_ms"] * 1e-3) / 1e9, 1),
                           "peak_GBs": PEAK_HBM_GBS,
                           "frac": round(prof["decode_bytes"] / (prof["decode_ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)},
            "nms": {"ms": round(prof["nms_ms"], 3), "share_of_step": round(prof["nms_ms"] / ms_per_step, 4),
                    "candidate_pairs_per_step": pairs,
                    "candidate_pairs_per_s": round(pairs / (prof["nms_ms"] * 1e-3), 1),
                    "pairs": "sum over (image, class) buckets of n(n-1)/2 candidates: the IoU tests of a dense mask"},
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
        out["h2d"] = h2d_rate(B, IMG, dev, ms_per_step)
        if not args.no_extras:
            peaks = measured_peaks(dev)
            out["measured_peaks"] = peaks
            # the shader clock the device holds under its power limit, for the two loads the roofline compares
            ops = (torch.rand(4096, device=dev) * 2 - 1).to(torch.bfloat16)
            msink = torch.empty(512 * 256, dtype=torch.float32, device=dev)
            clk_conv = sustained_clock(lambda: prog.run(images), dev)
            clk_mfma = sustained_clock(lambda: L.check(L.LIB.yamd_microbench_mfma_bf16(ops.data_ptr(), msink.data_ptr(), 512, 4096,
                                                                                        L.current_stream())), dev)
            peaks["shader_clock"] = {"conv_program": clk_conv, "mfma_loop": clk_mfma, "nominal_GHz": 2.4,
                                     "note": "one-wavefront probe on a second stream while the load runs; the device runs "
                                             "at its socket power limit under both loads, below the 2.4 GHz the nominal "
                                             "MFMA peak assumes"}
            out["roofline"]["peak_at_sustained_clock"] = round(PEAK_BF16_TFLOPS * clk_conv["GHz"] / 2.4, 1)
            out["roofline"]["frac_at_sustained_clock"] = round(conv_tflops / (PEAK_BF16_TFLOPS * clk_conv["GHz"] / 2.4), 4)
            out["roofline"]["frac_of_measured_mfma_loop"] = round(conv_tflops / peaks["mfma_bf16_TFLOPs"], 4)
            out["decode_hbm"]["frac_of_measured_copy"] = round(out["decode_hbm"]["achieved_GBs"] / peaks["hbm_copy_GBs"], 4)
            out["conv_hbm"]["frac_of_measured_copy"] = round(out["conv_hbm"]["achieved_GBs"] / peaks["hbm_copy_GBs"], 4)
            if world == 1 and not args.two_stream:
                out["two_stream"] = two_stream_rate(model, images, sink, dev, args.steps)
            if world == 1:      # (loads other weights: keep it behind everything that needs the headline's)
                out["nms_regime_realistic"] = realistic_regime(args, p, model, images, sink)
            if world == 1:
                del images, sink
                out["latency_small_batch"] = small_batch_latency(model, p, dev)
        if args.no_extras and args.realistic and world == 1:
            out["nms_regime_realistic"] = realistic_regime(args, p, model, images, sink)
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
