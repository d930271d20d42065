#!/usr/bin/env python3
"""bench.py — whole-frame throughput of the hot path on N MI355X (one process per GPU).

A *step* is one `drawScene()` (main.cpp:1090-1113) of the headless host mirror: G-buffer, temporal
gradient, 1-spp path trace, N a-trous iterations (last one fused with reprojection + blend),
history hand-over — on synthetic input (the Cornell box, scripted static camera), every buffer
resident in HBM before the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload 4k|1080p] [--halo exchange|redundant]

N > 1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py
--gpus N ...`; the frame is split into N row strips (strong scaling, BASELINE.json configs[3]).

Prints ONE JSON line (rank 0): metric Mray/s over the whole frame time, ms per frame, the HBM
roofline of the a-trous kernel measured live with HIP events, and the CPU oracle timed beside it.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # BASELINE.json configs[2]/[3]: Cornell box 3840x2160, 1 spp, 4 bounces, 5-level a-trous
    "4k": dict(width=3840, height=2160, max_segments=4, iterations=5),
    # BASELINE.json configs[1]
    "1080p": dict(width=1920, height=1080, max_segments=4, iterations=5),
    # the reference's own constants (main.cpp:52-55, raytrace.comp.glsl:204)
    "reference": dict(width=1000, height=800, max_segments=32, iterations=9),
    # BASELINE.json configs[4]: 10x10x10 lattice of 6x6-tessellated Cornell boxes = 1,152,000 triangles,
    # 8 segments (divergent-traversal stress; BVH path, direct filter kernel)
    "instanced": dict(width=3840, height=2160, max_segments=8, iterations=5, instanced=True),
}
PREWARM_SECONDS = 0.3
_INSTANCED = {}  # the configs[4] scene, built once per process
# VALU issue, measured on this part (scripts/micro/valu_rate.hip -> profiles/r03_valu_issue_micro.txt, wall clock at 8 waves
# per SIMD): the fastest wave64 VALU instruction occupies a SIMD for 1.06 ns (v_xor_b32; v_mul_f32 1.12, a dependent
# v_fma_f32 1.16), a transcendental (v_rcp / v_sqrt / v_rsq / v_exp) for 3.42 ns.  1,024 SIMDs.
VALU_NS, TRANS_NS, N_SIMDS = 1.06, 3.42, 1024
HOST_SECONDS = [0.0]
LOCAL_SECONDS = [0.0]   # this rank's own K steps (its stream drained, before the ranks meet at the barrier): what --balance reads
HIST_BYTES = [0]
PRESENT_BYTES = [0]
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
# ALGORITHMIC bytes per pixel per launch: what the operator a launch computes must read and write in
# the survey's formats (colour 16, depth 4, id 4).  One a-trous iteration k < N: 16+4+4 read + 16 write = 40; the final
# one 72.  k_atrous_chain computes CHAIN_LEVELS iterations in one launch: as an operator it still reads the input image
# + guides once and writes one image, so its algorithmic bytes are 40 B/px as well — the intermediate image is not part
# of the fused operator's interface (round 2 priced the chain at CHAIN_LEVELS x 40, "the passes it replaces", which is a
# speed-up figure, not a roofline fraction: it is kept as `replaced_pass_bytes` / `frac_vs_separate_passes_at_peak`).
CHAIN_LEVELS = 2
BYTES_PER_PX = {"k_atrous": 40, "k_atrous_final": 72, "k_atrous_chain": 40, "k_gradient": 36, "k_gbuffer": 24,
                "k_gbuffer_gradient": 24 + 36, "k_pathtrace": 16, "k_present": 16 + 4,
                # K0 + K1 + K2 in one launch: G-buffer planes + gradient out (K1's inputs stay in registers), traced image out
                "k_gbuffer_pathtrace": 24 + 16 + 16}
# bytes the kernel as built MUST move per pixel (rgbd cells: depth rides in alpha, so 16 + 4 read and 16 written; the
# chain reads its input once and writes its last level once).  Scenes without an id-pair table (> 63 triangles) run the
# per-pixel-normal variant, which stages a 16-byte (normal, self weight) cell instead of the 4-byte id (+12 B/px; its
# final pass still reads the id for the reprojection: +16).
REQUIRED_PER_PX = {"k_atrous": 36, "k_atrous_final": 68, "k_atrous_chain": 36}
REQUIRED_PER_PX_NRM = {"k_atrous": 48, "k_atrous_final": 84, "k_atrous_chain": 48}


def _scene_kwargs(wl):
    """make_app's scene arguments for a workload: the OBJ as it is, or BASELINE configs[4]'s lattice of tessellated instances
    (built once per process)"""
    if not wl.get("instanced"):
        return {}
    if "scene" not in _INSTANCED:
        from real_time_path_tracing_with_spatiotemporal_filtering_amd import abi, scenes
        from real_time_path_tracing_with_spatiotemporal_filtering_amd.app import DEFAULT_SCENE
        xyz, idx = abi.load_obj(DEFAULT_SCENE)
        _INSTANCED["scene"] = scenes.instanced_cornell(xyz, idx)
    vx, ti, xf, cam, zfar = _INSTANCED["scene"]
    return dict(mesh=(vx, ti), instance_xforms=xf, cameraOrigin=cam, z_far=zfar, lightPos=(1.0, float(cam[1]), float(cam[2]) - 8.0))


def run_gpu(wl, args, rank, world, steps, warmup, torch, dist, collect_kernels=True, in_flight=1, halo=None, present="default",
            camera_keys=None):
    """one measured run; halo / present / camera_keys override the command line (the `also` legs of a multi-rank run)"""
    halo = halo or args.halo
    present = args.present if present == "default" else present
    camera_keys = args.camera_keys if camera_keys is None else camera_keys
    from real_time_path_tracing_with_spatiotemporal_filtering_amd.app import make_app
    extra = _scene_kwargs(wl)
    app = make_app(wl["width"], wl["height"], max_segments=wl["max_segments"], iterations=wl["iterations"],
                   rank=rank, world=world, mode=halo, flags=args.flags,
                   torch_planes=(dist is not None), frames_in_flight=in_flight, present=present,
                   splits=getattr(args, "strip_rows", ()) if world > 1 else (), **extra)
    keys_of = (lambda f: (camera_keys[f % len(camera_keys)],)) if camera_keys else (lambda f: ())
    present_bytes = [0]
    frame_no = [0]
    hist_bytes = [0]

    def draw():
        app.drawScene(keys_of(frame_no[0]))
        frame_no[0] += 1
        hist_bytes[0] += getattr(app, "history_bytes_sent", 0)
        present_bytes[0] += getattr(app, "present_bytes_sent", 0)
    ctxs = [b.ctx for b in app.backend.be] if in_flight == 2 else [app.backend.ctx]
    ctx = ctxs[0]
    collect_kernels = collect_kernels and in_flight == 1  # overlapping frames stretch every kernel's duration

    def fence():
        for c in ctxs:
            c.sync()
        app.present_sync()   # gathers posted on the present stream
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # device wake-up, untimed and before the W warm-up steps: a cold MI355X runs its first ~50 ms of kernels
    # 20 % slower (in-process A/B: the first 400 frames of a 270-row strip take 0.212 ms each, every later batch
    # 0.173 ms), which a 36 ms timed region at 8 ranks would otherwise measure instead of the kernels
    # The same loop fills the context's pool of timing events (created on first use; on some boxes
    # hipEventCreate is slow enough that creating them inside the timed region cost +0.2 ms per step).
    # sample the per-kernel events: bracketing every launch costs ~6 % of the frame (the driver's --steps 20 run of
    # round 1 paid that); every 8th frame in long runs, and never fewer than 8 sampled frames (or all of them) in short ones
    timing_period = max(1, min(8, steps // 8))
    if args.prewarm_seconds > 0:
        ctx.timing_enable(timing_period if collect_kernels else 0)
        t_wake = time.perf_counter()
        n_wake = 0
        # with several ranks the number of wake-up frames is fixed, not timed: ranks that exchange rows every frame
        # (a moving camera, --halo exchange) must all draw the same frames
        fixed = None if dist is None else max(steps, 16 * int(args.prewarm_seconds * 1e3 / (16 * 0.2)))
        while (n_wake < fixed) if fixed is not None else (
                time.perf_counter() - t_wake < args.prewarm_seconds or (n_wake < steps and time.perf_counter() - t_wake < 3.0)):
            for _ in range(16):
                draw()
            n_wake += 16
            for c in ctxs:
                c.sync()
        ctx.timing_collect()
        ctx.timing_enable(0)
    for _ in range(warmup):
        draw()
    fence()
    hist_bytes[0] = present_bytes[0] = 0
    for c in ctxs:
        c.reset_counters()
    # per-kernel HIP events on the launch stream, sampled: bracketing every launch costs ~6 % of the frame
    ctx.timing_enable(timing_period if collect_kernels else 0)
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        draw()
    HIST_BYTES[0] = hist_bytes[0]
    PRESENT_BYTES[0] = present_bytes[0]
    HOST_SECONDS[0] = time.perf_counter() - t0  # time the host needed to submit the K steps (diagnostic)
    for c in ctxs:
        c.sync()
    LOCAL_SECONDS[0] = time.perf_counter() - t0
    fence()
    elapsed = time.perf_counter() - t0
    ctx.timing_enable(0)
    kern = ctx.timing_collect() if collect_kernels else {}
    timed_frames = max(1, len([f for f in range(steps) if f % timing_period == 0])) if collect_kernels else steps
    rays = sum(c.raycount() for c in ctxs)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        r = torch.tensor([rays], dtype=torch.int64, device="cuda")
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        rays = int(r.item())
    plan = app.plan
    app.backend.close()
    return elapsed, rays, kern, plan, timed_frames


def kernel_report(kern, wl, plan, steps):
    """per-kernel average launch duration (HIP events on the launch stream) and algorithmic GB/s"""
    W = wl["width"]
    # sampled frames = launches of the once-per-frame kernel (K2 alone, or with K0 + K1 in its launch)
    steps = max(1, kern.get("k_pathtrace", (0.0, 0))[1] + kern.get("k_gbuffer_pathtrace", (0.0, 0))[1]) if kern else max(1, steps)
    rows = {
        "k_gbuffer": plan.gbuffer_rows(), "k_gbuffer_gradient": plan.gbuffer_rows(), "k_gradient": plan.gradient_rows(),
        "k_pathtrace": plan.raytrace_rows(), "k_gbuffer_pathtrace": plan.raytrace_rows(),
        "k_atrous_final": plan.filter_rows(wl["iterations"]), "k_present": plan.own,
    }
    out = {}
    for name, (ms, n) in kern.items():
        if not n:
            continue
        avg_us = ms / n * 1e3
        e = {"launches_per_frame": n / steps, "avg_us": round(avg_us, 3)}
        if name in ("k_atrous", "k_atrous_chain"):
            N = wl["iterations"]
            ks = [k for k in range(1, N + 1) if not (k == N and N & 1)]  # an odd final pass is k_atrous_final
            if name == "k_atrous_chain":
                ks = ks[CHAIN_LEVELS - 1::CHAIN_LEVELS]  # a chain stores the rows of its LAST iteration
            px = sum(b - a for a, b in map(plan.filter_rows, ks)) * W / max(1, len(ks))
        elif name in rows:
            a, b = rows[name]
            px = (b - a) * W
        else:
            px = None
        if px and name in BYTES_PER_PX:
            e["algorithmic_bytes"] = int(BYTES_PER_PX[name] * px)
            e["algorithmic_GBps"] = round(BYTES_PER_PX[name] * px / (avg_us * 1e-6) / 1e9, 1)
        req = REQUIRED_PER_PX_NRM if wl.get("instanced") else REQUIRED_PER_PX
        if px and name in req:
            e["required_bytes"] = int(req[name] * px)
        if px and name == "k_atrous_chain":
            e["replaced_pass_bytes"] = int(40 * CHAIN_LEVELS * px)
        out[name] = e
    return out


def traversal_report():
    """north_star: "L2-hit/occupancy on the traversal kernel": a committed PMC measurement (profiles/traversal.json,
    scripts/pmc_traversal.sh; counters cannot be collected inside a bench run), named by the commit it was taken at"""
    try:
        tv = json.load(open(os.path.join(ROOT, "profiles", "traversal.json")))
        kp = tv.get("k_pathtrace", {})
        return {"kernel": "k_pathtrace<BVH> (tile kernel + queue windows)",
                "l2_hit_rate": round(kp.get("l2_hit_rate", 0), 4),
                "occupancy_waves_per_simd": round(kp.get("waves_per_simd", 0), 2), "occupancy_max": 8,
                "valu_busy": round(kp.get("valu_busy", 0), 3),
                "lane_utilisation": round(kp.get("lane_utilisation", 0), 3),
                "committed_measurement": True,
                "source": "profiles/traversal.json @ " + str(tv.get("_source"))}
    except Exception:
        return None


def cpu_baseline(workload: str, budget_s=12.0):
    """the CPU oracle in a CHILD process (a fresh interpreter: no torch, no HIP runtime, none of their threads): the
    checker's worker threads never share a process with the GPU legs that ran before it"""
    import subprocess
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", workload, str(budget_s)],
                         stdout=subprocess.PIPE, stderr=sys.stderr, timeout=600, check=True)
    return json.loads(out.stdout.decode().strip().splitlines()[-1])


def _cpu_baseline_child(wl, budget_s=12.0):
    """the CPU oracle (oracle/rtpt_oracle.c, "port") on a bounded row band of the same workload"""
    from oracle import oracle as O
    from real_time_path_tracing_with_spatiotemporal_filtering_amd.app import DEFAULT_SCENE
    O.build()
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = min(cores, 16)  # the one-GPU box's CPU share
    O.set_threads(cores)
    xyz, idx = O.load_obj(DEFAULT_SCENE)
    W, H, N = wl["width"], wl["height"], wl["iterations"]
    import numpy as np
    if wl.get("instanced"):
        # BASELINE configs[4]: the oracle's closest hit is brute force over all 1,152,000 triangles (hit ids
        # independent of any BVH), ~13 G triangle tests for one 3840-pixel row of 8-segment paths — the bounded sample is ONE
        # row, every pass on it (the filter's taps above and below read untraced rows: a timing sample, not an image)
        from real_time_path_tracing_with_spatiotemporal_filtering_amd import scenes
        vx, ti, xf, cam, zfar = scenes.instanced_cornell(xyz, idx)
        tris = O.flatten(vx, ti, xf)
        app = O.OracleApp(W, H, tris, max_segments=wl["max_segments"], iterations=N, camera=cam, z_far=zfar,
                          light=(1.0, float(cam[1]), float(cam[2]) - 8.0))
    else:
        tris = O.flatten(xyz, idx)
        app = O.OracleApp(W, H, tris, max_segments=wl["max_segments"], iterations=N)
    app.update_scene()
    lut = O.lut(tris, np.array(app.ubo.model[:], np.float32))
    banded = not wl.get("instanced")

    def band_frame(rows, y0=None):
        y0 = H // 2 - rows // 2 if y0 is None else y0
        y1 = y0 + rows
        halo = N * (N + 1) // 2 if banded else 0
        g0, g1 = max(0, y0 - halo), min(H, y1 + halo)
        t0 = time.perf_counter()
        vis, wp, depth = O.gbuffer(app.cfg, tris, app.ubo, g0, g1)
        O.temporal_gradient(app.cfg, app.pc, vis, wp, lut, lut, y0, y1)
        img, rays, _ = O.raytrace(app.cfg, app.pc, tris, g0, g1, want_hit_id=False)
        app.pc.maxWaveletIteration = N
        cur = img
        for k in range(1, N + 1):
            app.pc.waveletIteration = k
            rem = sum(range(k + 1, N + 1)) if banded else 0
            cur = O.atrous(app.cfg, app.pc, app.ubo, cur, depth, vis, lut, lut, wp, None,
                           max(0, y0 - rem), min(H, y1 + rem))
        dt = time.perf_counter() - t0
        if banded:  # rays of the band rows only (the halo rows are overhead of banding, as on a strip rank)
            _, rays, _ = O.raytrace(app.cfg, app.pc, tris, y0, y1, want_hit_id=False)
        return dt, rays

    if not banded:
        # 2 rows per thread x 240 columns (two frame rows' worth of pixels at 16 threads, ~10 s), through the lattice's fourth
        # layer of boxes
        rows, cols = 2 * cores, W // 16
        y0, x0 = H // 2 - H // 16, W // 2 - cols // 2
        O.set_columns(x0, x0 + cols)
        dt, rays = band_frame(rows, y0)
        O.set_columns()
        return {
            "value": round(rays / dt / 1e6, 5), "unit": "Mray/s", "cores": cores, "kind": "port", "one_thread": None,
            "sample": f"{rows} x {cols} pixels at ({x0}, {y0}) of the {W}x{H} frame: {rays} closest-hit queries, each a brute force over "
                      f"{len(tris)} triangles, all passes on those pixels, {dt:.1f} s of oracle/rtpt_oracle.c with {cores} threads",
            "ms_per_frame_extrapolated": round(dt / (rows * cols) * W * H * 1e3, 1),
        }
    rows = 16
    dt, rays = band_frame(rows)
    while dt < budget_s / 2 and rows < H:
        rows = min(H, int(rows * min(8.0, max(2.0, budget_s / max(dt, 1e-3) * 0.8))))
        dt, rays = band_frame(rows)
    reps = 1
    while dt < budget_s * 0.8:  # the whole frame is cheaper than the budget: repeat it (successive frame numbers)
        app.pc.frameNumber += 1
        d2, r2 = band_frame(rows)
        dt, rays, reps = dt + d2, rays + r2, reps + 1
    # also the scalar figure — one thread, a short band (about 3 s)
    O.set_threads(1)
    rows1 = 16
    d1, r1 = band_frame(rows1)
    while d1 < 1.5 and rows1 < H:
        rows1 = min(H, rows1 * 2)
        d1, r1 = band_frame(rows1)
    O.set_threads(cores)
    return {
        "value": round(rays / dt / 1e6, 3), "unit": "Mray/s", "cores": cores, "kind": "port",
        "one_thread": {"value": round(r1 / d1 / 1e6, 3), "unit": "Mray/s", "sample": f"{rows1} centre rows, {d1:.1f} s"},
        "sample": f"{reps} x {rows} centre rows of the {W}x{H} frame (+{N * (N + 1) // 2} halo rows per side when banded), "
                  f"all passes, {dt:.1f} s of oracle/rtpt_oracle.c with {cores} threads",
        "ms_per_frame_extrapolated": round(dt / reps / rows * H * 1e3, 1),
    }


def _min_strip_rows(wl, args, world):
    """rows every strip keeps when the boundaries move: the halo of the plan (exchange: the longest reach; redundant: their sum)"""
    from real_time_path_tracing_with_spatiotemporal_filtering_amd.strips import StripPlan
    return max(1, StripPlan(wl["height"], world, 0, wl["iterations"], args.halo, args.flags & 0x9F0).halo)


def _strip_app(wl, args, rank, world, rows, torch_planes, halo="redundant"):
    """an application for rank `rank`'s strip of a `world`-rank job with the boundaries `rows`, for the balancing procedures'
    own measurements: no present, redundant halo rows (no rank waits for another while the camera rests, so the ranks may
    draw different numbers of frames)"""
    from real_time_path_tracing_with_spatiotemporal_filtering_amd.app import make_app
    extra = _scene_kwargs(wl)
    return make_app(wl["width"], wl["height"], max_segments=wl["max_segments"], iterations=wl["iterations"], rank=rank, world=world,
                    mode=halo, flags=args.flags, torch_planes=torch_planes, splits=rows if world > 1 else (), **extra)


def ray_profile(app, y0, y1, band=8, settle=4):
    """rays traced per row of rows [y0, y1) of this application's strip, one band of `band` rows per frame through the ray
    counter's row window (rtpt_set_count_rows) — the frames differ in their random numbers, a band's sum hardly does"""
    ctx = app.backend.ctx
    for _ in range(settle):
        app.drawScene(())
    out = []
    for a in range(y0, y1, band):
        b = min(a + band, y1)
        ctx.set_count_rows(a, b)
        ctx.reset_counters()
        app.drawScene(())
        n = ctx.raycount()
        out += [n / (b - a)] * (b - a)
    ctx.set_count_rows(*app.plan.own)
    ctx.reset_counters()
    return out


def _trace_share(kern):
    """share of the per-frame kernel time that scales with the rays (the tracing launch) in a timing_collect() table"""
    tot = sum(ms for ms, n in kern.values() if n)
    tr = sum(ms for k, (ms, n) in kern.items() if n and k in ("k_pathtrace", "k_gbuffer_pathtrace", "k_pathtrace_queue"))
    return min(0.95, max(0.05, tr / tot)) if tot > 0 else 0.5


def balance_live(wl, args, rank, world, torch, dist):
    """--balance: every rank measures the rays per row of its own strip, the ranks gather the profile, cut it into equal
    parts, and then ROUNDS times run a short batch of frames, gather their own times and let them correct the profile
    (strips.StripBalancer).  Every rank computes the same rows from the same gathered numbers.  Nothing of this enters the
    timed region."""
    from real_time_path_tracing_with_spatiotemporal_filtering_amd.strips import StripBalancer, StripPlan
    H = wl["height"]
    dev = "cpu" if dist.get_backend() == "gloo" else "cuda"

    def gather(values, n):
        mine = torch.zeros(n, dtype=torch.float64, device=dev)
        mine[:len(values)] = torch.tensor(values, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        return [t.cpu().tolist() for t in every]

    rows = args.strip_rows or tuple(StripPlan.bounds(H, world, r)[0] for r in range(world)) + (H,)
    app = _strip_app(wl, args, rank, world, rows, torch_planes=True)
    mine = ray_profile(app, *app.plan.own)
    app.backend.ctx.timing_enable(1)
    for _ in range(8):
        app.drawScene(())
    share = _trace_share(app.backend.ctx.timing_collect())
    app.backend.close()
    tallest = max(b - a for a, b in zip(rows, rows[1:]))
    parts = gather(mine + [share], tallest + 1)
    profile = []
    for r in range(world):
        profile += parts[r][:rows[r + 1] - rows[r]]
    share = sum(parts[r][rows[r + 1] - rows[r]] for r in range(world)) / world
    mean = sum(profile) / H
    bal = StripBalancer(H, world, _min_strip_rows(wl, args, world), profile, floor=mean * (1.0 - share) / share)
    rows = bal.splits()
    steps = max(8, min(args.steps, 40))
    # a strip's time is not additive in its rows (a launch of 4 140 tiles takes a third generation of workgroups that one of
    # 4 080 does not), so the rounds wander around the optimum: the job runs with the best boundaries it has SEEN
    best = None
    for i in range(max(1, args.balance)):
        args.strip_rows = rows
        # redundant halo rows and no present whatever the job will use: a rank's own time must not contain waiting for others
        run_gpu(wl, args, rank, world, steps, min(args.warmup, 5), torch, dist, collect_kernels=False, halo="redundant", present=None)
        times = [t[0] for t in gather([LOCAL_SECONDS[0]], 1)]
        if best is None or max(times) < best[0]:
            best = (max(times), rows)
        bal.update(rows, times)
        rows = bal.splits()
    return best[1]


def emulate_balance(wl, args, torch):
    """--emulate-balance N[:ROUNDS] on one GPU: what --balance does on N, with the strips run one after the other.  Round 0 is
    the equal division; round 1 the cut of the measured ray profile; the later rounds correct it with the strips' times.
    `plain` repeats the rounds without the profile (strips.balanced_splits: a strip's time spread evenly over its rows)."""
    from real_time_path_tracing_with_spatiotemporal_filtering_amd.strips import StripBalancer, StripPlan, balanced_splits
    spec = args.emulate_balance.split(":")
    n, rounds = int(spec[0]), int(spec[1]) if len(spec) > 1 else 3
    H = wl["height"]
    args.halo = "redundant"
    equal = tuple(StripPlan.bounds(H, n, r)[0] for r in range(n)) + (H,)
    min_rows = _min_strip_rows(wl, args, n)
    out = {"emulated_balance": n, "workload": args.workload, "steps": args.steps, "min_rows": min_rows, "rounds": [], "plain": []}

    def measure(rows, into):
        args.strip_rows = rows
        ms = []
        for r in range(n):
            e, _, _, _, _ = run_gpu(wl, args, r, n, args.steps, args.warmup, torch, None, collect_kernels=False,
                                    in_flight=args.frames_in_flight)
            ms.append(round(e / args.steps * 1e3, 4))
        into.append({"strip_rows": list(rows), "ms_per_strip": ms, "slowest": max(ms), "mean": round(sum(ms) / n, 4)})
        print(f"# rows {list(rows)} ms {ms} slowest {max(ms)}", file=sys.stderr, flush=True)
        return ms

    full = _strip_app(wl, args, 0, 1, (), torch_planes=False)
    profile = ray_profile(full, 0, H)
    full.backend.ctx.timing_enable(1)
    for _ in range(8):
        full.drawScene(())
    share = _trace_share(full.backend.ctx.timing_collect())
    full.backend.close()
    out["trace_share_of_kernel_time"] = round(share, 3)
    out["rays_per_row_by_16_rows"] = [round(sum(profile[y:y + 16]) / len(profile[y:y + 16]), 1) for y in range(0, H, 16)]
    bal = StripBalancer(H, n, min_rows, profile, floor=sum(profile) / H * (1.0 - share) / share)
    ms = measure(equal, out["rounds"])
    rows = bal.splits()
    for i in range(rounds):
        ms = measure(rows, out["rounds"])
        if i + 1 < rounds:
            bal.update(rows, ms)
            rows = bal.splits()
    best = min(out["rounds"], key=lambda r: r["slowest"])
    out["best"] = {"strip_rows": best["strip_rows"], "slowest": best["slowest"],
                   "speedup_of_the_slowest_strip_vs_equal_strips": round(out["rounds"][0]["slowest"] / best["slowest"], 3)}
    if args.emulate_balance_plain:
        rows, ms = equal, out["rounds"][0]["ms_per_strip"]
        for i in range(rounds):
            rows = balanced_splits(rows, ms, min_rows)
            ms = measure(rows, out["plain"])
    _RESULT_LINE.append(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="4k")
    # redundant halo rows are the default: at 4K / 8 ranks a strip is ~0.2 ms of work and five k*61 KB
    # exchanges per frame are latency-bound; --halo exchange runs the RCCL send/recv path
    ap.add_argument("--halo", choices=["exchange", "redundant"], default="redundant")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (nccl) and torch-owned planes even with one rank (rehearsal)")
    ap.add_argument("--emulate-strip", default=None, metavar="R/N",
                    help="diagnostic: this single process runs rank R's strip of an N-rank job (redundant halo, no "
                         "communication is needed while the camera rests) and prints its per-frame time")
    ap.add_argument("--splits", default=None, metavar="0,A,B,...,H",
                    help="unequal strips: world + 1 ascending rows (strips.StripPlan.splits); with --emulate-strip R/N the rows of "
                         "the N-rank job whose rank R this process runs")
    ap.add_argument("--balance", type=int, default=0, metavar="ROUNDS",
                    help="several ranks: before the measured run, measure the rays per row (every rank its own strip), cut that "
                         "profile into equal parts and correct it ROUNDS times with the ranks' own frame times "
                         "(strips.StripBalancer); the line reports the rows under strip_rows.  Off by default: equal strips")
    ap.add_argument("--emulate-balance", default=None, metavar="N[:ROUNDS]",
                    help="diagnostic on ONE GPU: run every strip of an N-rank job alone, one after the other (redundant halo), "
                         "balance the boundaries from the measured times and repeat ROUNDS times (default 3); prints the "
                         "slowest strip of every round")
    ap.add_argument("--emulate-balance-plain", action="store_true",
                    help="with --emulate-balance: also run the rounds of the profile-free procedure (strips.balanced_splits)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="multi-rank rehearsal on a single GPU: every rank uses device 0 and torch.distributed runs "
                         "on gloo (RCCL refuses two ranks on one device); exercises strips, halo exchange and the "
                         "timing protocol, not the interconnect")
    ap.add_argument("--frames-in-flight", type=int, choices=[1, 2], default=1,
                    help="2: the timed run itself uses app.PipelinedBackend (no per-kernel timing, roofline null); "
                         "the default run reports it under also.two_frames_in_flight")
    ap.add_argument("--prewarm-seconds", type=float, default=PREWARM_SECONDS,
                    help="untimed device wake-up before the W warm-up steps (0 makes the rendered frame numbers, and "
                         "with them the ray counts, a function of --steps/--warmup only)")
    ap.add_argument("--flags", type=lambda v: int(v, 0), default=0,
                    help="RTPT_FLAG_* bits for A/B runs (0x400 = one kernel per a-trous iteration, no chaining)")
    ap.add_argument("--camera-keys", default="",
                    help="keys held on successive frames, cycled (e.g. EQ: the camera moves up and down 0.1 per frame, "
                         "main.cpp:1119-1168), so that every frame reprojects and, on strips, exchanges history bands; "
                         "default: camera at rest, like every BASELINE config")
    ap.add_argument("--present", choices=["auto", "none", "rgba8", "f32"], default="auto",
                    help="the swapchain blit of main.cpp:1338-1361 inside every step: rgba8 = rtpt_present (B8G8R8A8_UNORM) "
                         "and, with several ranks, the converted strips gathered on rank 0; f32 = the float strips gathered "
                         "as they are; none = the frame stays where the final pass left it.  auto: none on one GPU (the "
                         "frame is already whole), rgba8 on several (a frame that is never assembled cannot be presented)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--secondary-timeout", type=float, default=300.0,
                    help="several ranks: seconds the legs behind the headline measurement may take before rank 0 prints the line "
                         "without them and every rank exits (a watchdog against a message that never arrives)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    args.present = {"auto": "rgba8" if max(world, args.gpus) > 1 else None, "none": None}.get(args.present, args.present)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if args.gpus > 1:
            sys.exit(f"--gpus {args.gpus} needs one process per GPU: launch with python -m torch.distributed.run "
                     f"--nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py --gpus {args.gpus} ...")
        world, rank, local_rank = 1, 0, 0

    import torch
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the hot path has no CPU fallback")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    wl = WORKLOADS[args.workload]
    args.strip_rows = tuple(int(v) for v in args.splits.split(",")) if args.splits else ()
    if args.emulate_balance:
        emulate_balance(wl, args, torch)
        return
    if dist is not None and args.balance > 0:   # (with --force-dist also on one rank: the procedure's collectives over RCCL)
        rows = balance_live(wl, args, rank, world, torch, dist)
        args.strip_rows = rows if world > 1 else ()
    if args.emulate_strip:
        r, n = map(int, args.emulate_strip.split("/"))
        args.halo = "redundant"
        elapsed, rays, kern, plan, timed_frames = run_gpu(wl, args, r, n, args.steps, args.warmup, torch, None,
                                                          in_flight=args.frames_in_flight)
        _RESULT_LINE.append(json.dumps({"emulated_strip": args.emulate_strip, "strip_rows": list(plan.splits) or None,
                          "rows_owned": plan.own, "rows_stored": plan.stored,
                          "ms_per_step": round(elapsed / args.steps * 1e3, 4), "rays_per_frame": rays / args.steps,
                          "kernels": kernel_report(kern, wl, plan, timed_frames)}))
        return
    elapsed, rays, kern, plan, timed_frames = run_gpu(wl, args, rank, world, args.steps, args.warmup, torch, dist,
                                                      in_flight=args.frames_in_flight)
    ms_per_step = elapsed / args.steps * 1e3
    host_ms = HOST_SECONDS[0] / args.steps * 1e3
    result = None
    if rank == 0:
        kr = kernel_report(kern, wl, plan, timed_frames)
        # the filter iterations k < N: chained pairs by default, single launches with RTPT_FLAG_NO_FILTER_FUSION
        rk = "k_atrous_chain" if "k_atrous_chain" in kr else "k_atrous"
        at = kr.get(rk, {})
        traffic, traffic_source, valu_tab = None, None, {}
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and world == 1:
            try:
                tj = json.load(open(tpath))
                traffic = tj.get(args.workload, {}).get(rk)
                traffic_source = tj.get("_source")
                valu_tab = tj.get("valu", {}).get(args.workload, {})
            except Exception:
                traffic = None
        # every kernel of the frame is bound by instruction issue, not by HBM: next to the HBM fractions, the share of
        # each launch that its VALU instructions (SQ_INSTS_VALU / _TRANS_F32 per launch, the same committed PMC run as
        # `traffic`) occupy the SIMDs for at the measured issue rates above
        for k, v in kr.items():
            c = valu_tab.get(k) or valu_tab.get({"k_gbuffer_gradient": "k_gbuffer"}.get(k, ""))  # the fused launch is k_gbuffer<fused>
            if c and v.get("avg_us"):
                issue_us = ((c["insts"] - c["trans"]) * VALU_NS + c["trans"] * TRANS_NS) * 1e-3 / N_SIMDS
                v["valu_issue"] = {"committed_measurement": True, "source": "profiles/traffic.json @ " + str(traffic_source),
                                   "insts_per_launch": int(c["insts"]), "transcendental": int(c["trans"]),
                                   "issue_us": round(issue_us, 1), "frac_of_launch": round(issue_us / v["avg_us"], 3)}
        achieved = at.get("algorithmic_GBps")
        result = {
            "metric": "Mray/s (closest-hit queries / whole-frame time: G-buffer + gradient + trace + a-trous)",
            "value": round(rays / elapsed / 1e6, 2),
            "unit": "Mray/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "host_submit_ms_per_step": round(host_ms, 4),
            "fps": round(1e3 / ms_per_step, 1),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (Cornell box OBJ, scripted static camera/light, RNG seeded by pixel+frame)",
            "config": {"workload": f"cornell-{args.workload}-1spp-{wl['max_segments']}seg-{wl['iterations']}atrous",
                       "width": wl["width"], "height": wl["height"], "max_segments": wl["max_segments"],
                       "atrous_iterations": wl["iterations"], "triangles": 1152000 if wl.get("instanced") else 32,
                       "parallelism": f"row-strips x{world}" + (f" ({args.halo} halo)" if world > 1 else ""),
                       "present": (args.present or "none") + (" gathered on rank 0" if world > 1 and args.present else ""),
                       "frames_in_flight": args.frames_in_flight},
            "rays_per_frame": round(rays / args.steps, 1),
            # unequal strips (--splits / --balance): the rows of the job that was measured; null = equal strips
            "strip_rows": list(plan.splits) or None,
            "history_exchange_bytes_per_frame_rank0": round(HIST_BYTES[0] / args.steps, 1),
            "present_gather_bytes_per_frame_rank0": round(PRESENT_BYTES[0] / args.steps, 1),
            # achieved / frac: ALGORITHMIC bytes of the operator one launch computes / the launch time measured
            # with HIP events on the launch stream.  frac_required prices the bytes this kernel as built must move (rgbd
            # cells), frac_traffic the fabric bytes the PMC counters saw (per launch, profiles/traffic.json: a committed
            # measurement of the commit named in traffic_source, not of this run).  For the chained kernel
            # frac_vs_separate_passes_at_peak is the old "bytes of the passes it replaces" figure: a speed-up, not a
            # roofline fraction.
            "roofline": {"kernel": rk + (f" ({CHAIN_LEVELS} a-trous iterations k < N per launch, intermediates in LDS)"
                                        if rk == "k_atrous_chain" else " (one a-trous iteration, k < N)"),
                         "bound": "hbm",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None,
                         "avg_launch_us": at.get("avg_us"), "algorithmic_bytes": at.get("algorithmic_bytes"),
                         "required_bytes": at.get("required_bytes"),
                         "frac_required": round(at["required_bytes"] / (at["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
                         if at.get("required_bytes") and at.get("avg_us") else None,
                         "traffic": traffic,
                         "frac_traffic": round(traffic / (at["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
                         if traffic and at.get("avg_us") else None,
                         "traffic_source": traffic_source,
                         "valu_issue": at.get("valu_issue"),
                         "frac_vs_separate_passes_at_peak": round(at["replaced_pass_bytes"] / (at["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
                         if at.get("replaced_pass_bytes") and at.get("avg_us") else None},
            "kernels": kr,
        }
        if wl.get("instanced"):
            result["traversal"] = traversal_report()
        pt = kr.get("k_pathtrace")  # the trace as a launch of its own (RTPT_NO_TRACE_FUSION=1): its Mray/s
        if pt:
            result["pathtrace_kernel_mray_s"] = round(rays / args.steps / (pt["avg_us"] * 1e-6) / 1e6, 1)

    # Several ranks: the legs behind the headline measurement exercise point-to-point messages (halo rows, history bands, the
    # float gather) that a one-GPU box can only rehearse over gloo.  If one of them never comes back on the real
    # interconnect, rank 0 still prints the line it has (with a note under `also`), and every rank leaves.
    watchdog = _arm_watchdog(result, rank, args.secondary_timeout) if dist is not None and world > 1 and not args.no_secondary else None
    try:
        if not args.no_secondary and args.frames_in_flight == 1:
            # the same workload with two frames in flight (app.PipelinedBackend): throughput only, the per-kernel
            # durations above come from the serial run because overlapping frames stretch every kernel
            e3, r3, _, _, _ = run_gpu(wl, args, rank, world, args.steps, args.warmup, torch, dist, collect_kernels=False,
                                      in_flight=2)
            if rank == 0:
                result.setdefault("also", {})["two_frames_in_flight"] = {
                    "value": round(r3 / e3 / 1e6, 2), "unit": "Mray/s", "ms_per_step": round(e3 / args.steps * 1e3, 4),
                    "fps": round(args.steps / e3, 1)}
        if world > 1 and not args.no_secondary:
            # the legs a scaling run must see (every rank takes part): the frame left distributed, the float gather, the RCCL
            # halo exchange (k rows per neighbour per iteration instead of redundant rows), and a camera that moves every
            # frame (history bands travel between strips).  Same steps/warm-up, throughput only.
            legs = [("without_output_gather", dict(present=None)),
                    ("with_f32_gather", dict(present="f32")),
                    ("halo_exchange" if args.halo != "exchange" else "halo_redundant",
                     dict(halo="exchange" if args.halo != "exchange" else "redundant")),
                    ("moving_camera", dict(camera_keys=args.camera_keys or "EQ"))]
            from real_time_path_tracing_with_spatiotemporal_filtering_amd.strips import StripPlan
            for name, kw in legs:
                try:   # every rank checks EVERY rank's plan, so a leg is skipped by all of them or by none
                    for r in range(world):
                        for k in range(1, wl["iterations"] + 1):
                            StripPlan(wl["height"], world, r, wl["iterations"], kw.get("halo", args.halo), args.flags & 0x9F0,
                                      args.strip_rows).exchange_rows(k)
                except ValueError as ex:   # strips shorter than the exchange halo
                    if rank == 0:
                        result.setdefault("also", {})[name] = {"skipped": str(ex)}
                    continue
                e5, r5, _, _, _ = run_gpu(wl, args, rank, world, args.steps, args.warmup, torch, dist, collect_kernels=False, **kw)
                if rank == 0:
                    result.setdefault("also", {})[name] = {
                        "value": round(r5 / e5 / 1e6, 2), "unit": "Mray/s", "ms_per_step": round(e5 / args.steps * 1e3, 4),
                        "history_exchange_bytes_per_frame_rank0": round(HIST_BYTES[0] / args.steps, 1),
                        "present_gather_bytes_per_frame_rank0": round(PRESENT_BYTES[0] / args.steps, 1)}
    except Exception as ex:   # noqa: BLE001
        # several ranks: the legs behind the headline are best effort — a rank whose peer has left (its watchdog fired) or whose
        # message failed must not turn the finished headline measurement into a failed run
        if watchdog is None:
            raise
        watchdog.cancel()
        print(f"rank {rank}: secondary legs abandoned: {ex!r}", file=sys.stderr, flush=True)
        if rank == 0:
            result.setdefault("also", {})["_error"] = f"secondary legs abandoned: {ex!r}"
            os.write(_STDOUT_FD[0] if _STDOUT_FD else 1, (json.dumps(result) + "\n").encode())
        os._exit(0)
    if world == 1 and rank == 0 and not args.no_secondary and args.present is None:
        e6, r6, k6, p6, tf6 = run_gpu(wl, args, 0, 1, args.steps, args.warmup, torch, None, present="rgba8")
        result.setdefault("also", {})["with_present_rgba8"] = {
            "value": round(r6 / e6 / 1e6, 2), "unit": "Mray/s", "ms_per_step": round(e6 / args.steps * 1e3, 4),
            "k_present_us": kernel_report(k6, wl, p6, tf6).get("k_present", {}).get("avg_us")}
    if world == 1 and rank == 0 and not args.no_secondary and args.workload == "4k":
        e2, r2, k2, p2, tf2 = run_gpu(WORKLOADS["1080p"], args, 0, 1, args.steps, args.warmup, torch, None)
        kr2 = kernel_report(k2, WORKLOADS["1080p"], p2, tf2)
        result.setdefault("also", {})["cornell-1080p-1spp-4seg-5atrous"] = {
            "value": round(r2 / e2 / 1e6, 2), "unit": "Mray/s", "ms_per_step": round(e2 / args.steps * 1e3, 4),
            "kernels": kr2,
            "atrous_kernel": "k_atrous_chain" if "k_atrous_chain" in kr2 else "k_atrous",
            "atrous_GBps": (kr2.get("k_atrous_chain") or kr2.get("k_atrous", {})).get("algorithmic_GBps"),
            "atrous_frac": round((kr2.get("k_atrous_chain") or kr2.get("k_atrous", {})).get("algorithmic_GBps", 0) / HBM_PEAK_GBS, 4)}
        e4, r4, _, _, _ = run_gpu(WORKLOADS["1080p"], args, 0, 1, args.steps, args.warmup, torch, None, collect_kernels=False,
                                  in_flight=2)
        result["also"]["cornell-1080p-1spp-4seg-5atrous"]["two_frames_in_flight"] = {
            "value": round(r4 / e4 / 1e6, 2), "ms_per_step": round(e4 / args.steps * 1e3, 4)}
    if world == 1 and rank == 0 and not args.no_secondary and args.workload == "4k":
        # BASELINE.json configs[4] beside the headline (its defining 8-GPU deployment is the driver's to launch: --workload
        # instanced --gpus 8): the 1,152,000-triangle lattice on this one GPU, per-kernel durations, the traversal's committed
        # PMC figures, and the CPU oracle on a bounded pixel sample of the same frame
        wi = WORKLOADS["instanced"]
        si, wu = min(args.steps, 40), min(args.warmup, 5)
        e7, r7, k7, p7, tf7 = run_gpu(wi, args, 0, 1, si, wu, torch, None)
        kr7 = kernel_report(k7, wi, p7, tf7)
        inst = {"value": round(r7 / e7 / 1e6, 2), "unit": "Mray/s", "ms_per_step": round(e7 / si * 1e3, 4), "steps": si, "warmup": wu,
                "triangles": 1152000, "max_segments": wi["max_segments"], "rays_per_frame": round(r7 / si, 1), "kernels": kr7,
                "traversal": traversal_report()}
        if kr7.get("k_pathtrace"):  # (absent when K0 + K1 share the launch)
            inst["pathtrace_kernel_mray_s"] = round(r7 / si / (kr7["k_pathtrace"]["avg_us"] * 1e-6) / 1e6, 1)
        if not args.no_cpu_baseline:
            inst["cpu_baseline"] = cpu_baseline("instanced")
        result.setdefault("also", {})["instanced-4k-1spp-8seg-5atrous"] = inst
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args.workload)
    if watchdog is not None:
        watchdog.cancel()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        _RESULT_LINE.append(json.dumps(result))


def _arm_watchdog(result, rank, seconds):
    def fire():
        try:
            if rank == 0 and result is not None:
                result.setdefault("also", {})["_watchdog"] = (f"the legs behind the headline measurement did not finish within {seconds:g} s; "
                                                             "this line was printed without the missing ones")
                os.write(_STDOUT_FD[0] if _STDOUT_FD else 1, (json.dumps(result) + "\n").encode())
        finally:
            os._exit(0)
    t = threading.Timer(seconds + (0.0 if rank == 0 else 2.0), fire)   # rank 0 first: its line is what matters
    t.daemon = True
    t.start()
    return t


def _main_with_clean_stdout():
    """stdout carries exactly one JSON line: RCCL prints a version banner to fd 1 when its first communicator
    comes up, so fd 1 points at stderr while the benchmark runs and is restored for the result line."""
    sys.stdout.flush()
    saved = os.dup(1)
    _STDOUT_FD.append(saved)
    os.dup2(2, 1)
    try:
        main()
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)
        if _RESULT_LINE:
            print(_RESULT_LINE[0], flush=True)


_RESULT_LINE = []
_STDOUT_FD = []   # the real stdout while fd 1 points at stderr (_main_with_clean_stdout)

if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "--cpu-baseline-child":
        print(json.dumps(_cpu_baseline_child(WORKLOADS[sys.argv[2]], float(sys.argv[3]) if len(sys.argv) > 3 else 12.0)))
    else:
        _main_with_clean_stdout()
