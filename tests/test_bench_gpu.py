"""bench.py end to end on the GPU box: the one-line JSON contract, and the multi-rank protocol rehearsed with two
ranks on the single GPU (gloo instead of RCCL, which refuses two ranks on one device)."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"}


def _run(cmd):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, f"stdout must carry exactly one line, got {len(lines)}"
    return json.loads(lines[0])


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_single_rank_line(hip_lib):
    d = _run([sys.executable, "bench.py", "--workload", "1080p", "--steps", "16", "--warmup", "2", "--no-cpu-baseline",
              "--no-secondary"])
    assert REQUIRED <= set(d) and d["n_gpus"] == 1 and d["steps"] == 16 and d["config"]["workload"].startswith("cornell-1080p")
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert d["value"] > 1000 and d["rays_per_frame"] > 1920 * 1080


@pytest.mark.parametrize("halo", ["redundant", "exchange"])
def test_two_rank_rehearsal_counts_the_same_rays(hip_lib, halo):
    one = _run([sys.executable, "bench.py", "--workload", "1080p", "--steps", "6", "--warmup", "1", "--no-cpu-baseline",
                "--no-secondary", "--prewarm-seconds", "0"])
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                "127.0.0.1", "--master-port", str(_port()), "bench.py", "--gpus", "2", "--workload", "1080p", "--steps", "6",
                "--warmup", "1", "--rehearse-on-one-gpu", "--halo", halo, "--no-secondary", "--prewarm-seconds", "0"])
    assert two["n_gpus"] == 2 and halo in two["config"]["parallelism"]
    # the frame sequence is deterministic (RNG seeded by pixel + frame) and both runs render the same frame numbers
    assert two["rays_per_frame"] == one["rays_per_frame"]
