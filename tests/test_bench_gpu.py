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


def test_default_line_carries_the_stress_configuration_and_both_cpu_baselines(hip_lib):
    """`python bench.py` as the driver runs it (short): the headline is BASELINE configs[2]; `also` carries configs[1] and —
    round 4 — configs[4] (1,152,000 triangles) with per-kernel times, the traversal's committed PMC figures and a CPU baseline
    of its own from a bounded pixel sample; `cpu_baseline` of the headline is still there; K0 + K1 + K2 are one kernel."""
    d = _run([sys.executable, "bench.py", "--steps", "16", "--warmup", "2"])
    assert REQUIRED | {"cpu_baseline", "also"} <= set(d) and d["config"]["workload"].startswith("cornell-4k")
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 1 and d["cpu_baseline"]["cores"] >= 1
    assert "k_gbuffer_pathtrace" in d["kernels"] and "k_pathtrace" not in d["kernels"]
    inst = d["also"]["instanced-4k-1spp-8seg-5atrous"]
    assert inst["triangles"] == 1152000 and inst["max_segments"] == 8 and 0.5 < inst["ms_per_step"] < 50
    assert inst["kernels"]["k_gbuffer_pathtrace"]["avg_us"] > 100 and inst["traversal"]["committed_measurement"]
    cb = inst["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "Mray/s" and 0 < cb["value"] < 1 and "brute force" in cb["sample"]
    assert "cornell-1080p-1spp-4seg-5atrous" in d["also"]


def test_balanced_strips_render_the_same_frames(hip_lib):
    """--balance (the ranks gather their own frame times and re-cut the strips, strips.balanced_splits) and --splits: the job
    reports its rows, they are a valid division of the frame, and the frames are the ones equal strips render (ray count).
    --emulate-balance runs the same procedure with the strips of an N-rank job one after the other on this GPU."""
    base = [sys.executable, "bench.py", "--workload", "1080p", "--steps", "6", "--warmup", "1", "--no-cpu-baseline", "--no-secondary",
            "--prewarm-seconds", "0"]
    one = _run(base)
    run2 = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
            "--master-port", str(_port())] + base[1:] + ["--gpus", "2", "--rehearse-on-one-gpu"]
    two = _run(run2 + ["--balance", "2"])
    rows = two["strip_rows"]
    assert rows[0] == 0 and rows[-1] == 1080 and len(rows) == 3 and 15 <= rows[1] <= 1065
    assert two["rays_per_frame"] == one["rays_per_frame"]
    fixed = _run(run2 + ["--splits", "0,700,1080"])
    assert fixed["strip_rows"] == [0, 700, 1080] and fixed["rays_per_frame"] == one["rays_per_frame"]
    em = _run(base + ["--emulate-balance", "3:1"])
    assert em["emulated_balance"] == 3 and len(em["rounds"]) == 2
    for r in em["rounds"]:
        assert r["strip_rows"][0] == 0 and r["strip_rows"][-1] == 1080 and len(r["ms_per_strip"]) == 3 and r["slowest"] == max(r["ms_per_strip"])
    assert em["rounds"][0]["strip_rows"] == [0, 360, 720, 1080]


def test_one_rank_over_rccl(hip_lib):
    """what a one-GPU box can show of the RCCL path: bench.py under torchrun with ONE rank and --force-dist initialises the
    nccl (= RCCL) process group on the device, binds torch-owned planes, runs the barrier / all_reduce timing protocol and,
    with --balance, the balancing procedure's all_gather — same frames as the plain run"""
    base = ["bench.py", "--workload", "1080p", "--steps", "6", "--warmup", "1", "--no-cpu-baseline", "--no-secondary", "--prewarm-seconds", "0"]
    one = _run([sys.executable] + base)
    rccl = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                 "--master-port", str(_port())] + base + ["--gpus", "1", "--force-dist", "--balance", "1"])
    assert rccl["n_gpus"] == 1 and rccl["rays_per_frame"] == one["rays_per_frame"] and rccl["strip_rows"] is None


def test_multi_rank_line_carries_every_secondary_leg(hip_lib):
    """several ranks (rehearsed over gloo on this GPU): the legs behind the headline measurement — two frames in flight, the
    frame left distributed, the float gather, the other halo mode, a moving camera — are all in the line, and the watchdog that
    guards them (tests/test_host_logic.py::test_bench_watchdog) left no note"""
    full = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                 "--master-port", str(_port()), "bench.py", "--gpus", "2", "--workload", "1080p", "--steps", "6", "--warmup", "1",
                 "--rehearse-on-one-gpu", "--prewarm-seconds", "0"])
    assert REQUIRED <= set(full) and full["n_gpus"] == 2
    assert "_watchdog" not in full["also"] and "_error" not in full["also"]
    assert {"two_frames_in_flight", "without_output_gather", "with_f32_gather", "halo_exchange", "moving_camera"} <= set(full["also"])


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


@pytest.mark.parametrize("mode,flags,in_flight", [("redundant", 0, 1), ("exchange", 0, 1), ("redundant", 0x100, 1), ("redundant", 0x81, 1),
                                                  ("exchange", 0x100, 1), ("redundant", 0, 2), ("exchange", 0x100, 2), ("redundant", 0x981, 2)])
def test_gloo_ranks_on_one_gpu_reproduce_the_single_context_frames(hip_lib, tmp_path, mode, flags, in_flight):
    """the Python host's multi-rank path on DEVICE memory — halo rows, the history bands bounded by the reprojection
    reach, and (extension flags 0x100 variance, 0x80 disocclusion) the previous frame's id / moment bands — with three
    ranks on GPU 0 and gloo as the carrier, vertical camera moves in the script: every rank's rows of every frame equal
    the single-context frames bit for bit, and only bands travelled (not whole frames).  in_flight 2: app.PipelinedBackend on
    every rank (the previous frame, and with the flags its id / moment planes, rest in the rank's other context)"""
    import numpy as np
    W, H, keys = 144, 150, ",E,J,QA,,E"
    args = [str(tmp_path), mode, hex(flags), keys, str(W), str(H)]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    worker = os.path.join(ROOT, "tests", "strip_worker.py")
    out = subprocess.run([sys.executable, worker] + args, cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr",
                          "127.0.0.1", "--master-port", str(_port()), worker] + args + ["", str(in_flight)],
                         cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    ref = np.load(tmp_path / "w1_r0.npz")
    parts = [np.load(tmp_path / f"w3_r{r}.npz") for r in range(3)]
    n_frames = len(keys.split(","))
    for f in range(n_frames):
        got = np.concatenate([p[f"arr_{f}"] for p in parts], axis=0)
        assert got.tobytes() == ref[f"arr_{f}"].tobytes(), (mode, hex(flags), f)
    assert sum(int(p["rays"][0]) for p in parts) == int(ref["rays"][0])
    sent = sum(int(p["sent"][0]) for p in parts)
    moved_frames = 3   # E, QA, E
    planes = 1 + (flags & 0x100 != 0) * 1.25 + (flags & 0x180 != 0 and not flags & 0x100) * 0.25
    assert 0 < sent < moved_frames * 3 * H * W * 16 * planes, "bands, not a whole frame to every rank"


@pytest.mark.parametrize("mode,present", [("redundant", "rgba8"), ("exchange", "f32")])
def test_gloo_ranks_on_one_gpu_assemble_the_presented_frame(hip_lib, tmp_path, mode, present):
    """main.cpp:1338-1361 on strips, on DEVICE memory: three ranks on GPU 0 (gloo as the carrier), the gather issued on
    its own stream behind each frame; rank 0's assembled frame — float strips, or rtpt_present's B8G8R8A8 conversion —
    equals the single-context frame (converted by the oracle's restatement of the blit) bit for bit, every frame"""
    import numpy as np
    sys.path.insert(0, ROOT)
    from oracle import oracle as O
    W, H, keys = 144, 150, ",E,J,,Q"
    args = [str(tmp_path), mode, "0", keys, str(W), str(H), present]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    worker = os.path.join(ROOT, "tests", "strip_worker.py")
    out = subprocess.run([sys.executable, worker] + args, cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr",
                          "127.0.0.1", "--master-port", str(_port()), worker] + args,
                         cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    ref = np.load(tmp_path / "w1_r0.npz")
    root = np.load(tmp_path / "w3_r0.npz")
    for f in range(len(keys.split(","))):
        want = ref[f"arr_{f}"]
        if present == "rgba8":
            assert ref[f"shown_{f}"].tobytes() == O.present_bgra8(want).tobytes(), "k_present vs the oracle's blit"
            want = O.present_bgra8(want)
        assert root[f"shown_{f}"].tobytes() == want.tobytes(), (mode, present, f)
