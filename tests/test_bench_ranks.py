"""world_size-2 rehearsals (gloo, CPU) of bench.py's multi-rank protocol: one process per rank, barrier on both sides
of the timed region, MAX over ranks, one JSON line from rank 0 -- started by torch.distributed.run and started by
bench.py itself (``--gpus 2`` without WORLD_SIZE), in weak and in strong scaling."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _one_json_line(out):
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_bench_two_ranks_gloo():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--dry", "--backend", "gloo"]
    rec = _one_json_line(subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env, cwd=ROOT))
    assert rec["n_gpus"] == 2
    # rank 1 sleeps 0.10 s, rank 0 0.05 s: the reported time is the slowest rank's
    assert rec["elapsed_max"] >= 0.095
    assert rec["scaling"] == "weak" and rec["images_per_rank"] == [256, 256] and rec["first_image_of_rank"] == [0, 256]


def test_bench_starts_its_own_ranks_and_splits_a_global_batch():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry",
           "--backend", "gloo", "--model", "v8s"]
    rec = _one_json_line(subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env, cwd=ROOT))
    assert rec["n_gpus"] == 2 and rec["elapsed_max"] >= 0.095
    assert rec["scaling"] == "strong" and rec["global_batch"] == 1024 and rec["images_per_rank"] == [512, 512]
    cmd += ["--global-batch", "1023"]
    rec = _one_json_line(subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env, cwd=ROOT))
    assert rec["images_per_rank"] == [512, 511] and rec["first_image_of_rank"] == [0, 512]


def test_bench_ends_all_ranks_when_one_fails():
    """A rank that dies before the first barrier must not leave the others waiting: bench.py's own launcher ends them
    and returns the failing rank's code, with that rank's stderr relayed."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["YAMD_BENCH_FAIL_RANK"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert out.returncode == 3, (out.returncode, out.stderr[-1500:])
    assert "[rank 1]" in out.stderr and "rank 1 exited with code 3" in out.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("extra,scaling,per_rank", [
    (["--batch", "8"], "weak", [8, 8]),
    (["--model", "v8s", "--scaling", "strong", "--global-batch", "12"], "strong", [6, 6]),
])
def test_bench_two_ranks_on_the_gpu(gpu_device, extra, scaling, per_rank):
    """The real multi-rank path end to end: bench.py starts two ranks (they share the one GPU of the test box), each
    builds its plan, runs full steps on its slice, host barriers (gloo) bracket the timed region and rank 0 prints the
    line.  What the first run on a multi-GPU node exercises, at a small batch."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-extras"] + extra
    rec = _one_json_line(subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT))
    assert rec["n_gpus"] == 2 and rec["scaling"] == scaling
    assert rec["config"]["images_per_rank"] == per_rank and rec["config"]["global_batch"] == sum(per_rank)
    assert rec["value"] > 0 and rec["ms_per_step"] > 0
    assert rec["roofline"]["achieved"] > 0 and rec["h2d"]["fp32_nchw"]["GBs"] > 0
