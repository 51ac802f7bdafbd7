"""world_size-2 rehearsal (gloo, CPU) of bench.py's multi-rank protocol: one process per rank, barrier on
both sides of the timed region, MAX over ranks, one JSON line from rank 0."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def test_bench_two_ranks_gloo():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--dry", "--backend", "gloo"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2
    # rank 1 sleeps 0.10 s, rank 0 0.05 s: the reported time is the slowest rank's
    assert rec["elapsed_max"] >= 0.095
