#!/usr/bin/env python3
"""Timeline of the workgroups of the last K0 / K2 launch of a frame (the TIMELINE build of the library:
scripts/build_variant.sh timeline -DRTPT_TILE_TIMELINE=1): how many workgroups are resident over time, how long a
workgroup lives by the time it starts, where the launch's tail is.

    RTPT_LIB_PATH=.../variants/librtpt_timeline.so python scripts/tile_timeline.py [--strip 3/8] [--workload 4k] [--out f.json]
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--strip", default=None, metavar="R/N")
    ap.add_argument("--workload", default="4k")
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--bucket-us", type=float, default=4.0)
    ap.add_argument("--out", default=None)
    ap.add_argument("--lpt", action="store_true",
                    help="experiment: after the first measurement, hand K2 its tiles longest-first (by the lifetimes just measured) "
                         "and measure again — what a longest-processing-time-first dispatch order is worth")
    args = ap.parse_args()
    import bench
    from real_time_path_tracing_with_spatiotemporal_filtering_amd import abi, scenes
    from real_time_path_tracing_with_spatiotemporal_filtering_amd.app import DEFAULT_SCENE, make_app
    lib = abi.load()
    try:
        fn = lib.rtpt_debug_timeline
    except AttributeError:
        sys.exit("this library was not built with -DRTPT_TILE_TIMELINE=1")
    fn.argtypes = [C.c_int, C.POINTER(C.c_ulonglong), C.c_uint32]
    wl = bench.WORKLOADS[args.workload]
    r, n = map(int, args.strip.split("/")) if args.strip else (0, 1)
    extra = {}
    if wl.get("instanced"):
        xyz, idx = abi.load_obj(DEFAULT_SCENE)
        vx, ti, xf, cam, zfar = scenes.instanced_cornell(xyz, idx)
        extra = dict(mesh=(vx, ti), instance_xforms=xf, cameraOrigin=cam, z_far=zfar, lightPos=(1.0, float(cam[1]), float(cam[2]) - 8.0))
    app = make_app(wl["width"], wl["height"], max_segments=wl["max_segments"], iterations=wl["iterations"], rank=r, world=n,
                   mode="redundant", torch_planes=False, **extra)
    for _ in range(args.frames):
        app.drawScene(())
    app.backend.ctx.sync()
    W = wl["width"]
    res = {"strip": args.strip, "workload": args.workload}
    for round_ in range(2 if args.lpt else 1):
      if round_ == 1:
        fo = lib.rtpt_debug_tile_order
        fo.argtypes = [C.POINTER(C.c_uint32), C.c_uint32]
        order = np.argsort(-k2_dur, kind="stable").astype(np.uint32)   # timeline entries are indexed by tile (by * gx + bx)
        assert fo(order.ctypes.data_as(C.POINTER(C.c_uint32)), len(order)) == 0
        import time
        for _ in range(args.frames):
            app.drawScene(())
        app.backend.ctx.sync()
        t0 = time.perf_counter()
        for _ in range(200):
            app.drawScene(())
        app.backend.ctx.sync()
        print(f"--- tiles longest-first: {(time.perf_counter() - t0) / 200 * 1e3:.4f} ms per frame")
      elif args.lpt:
        import time
        t0 = time.perf_counter()
        for _ in range(200):
            app.drawScene(())
        app.backend.ctx.sync()
        print(f"--- built-in order: {(time.perf_counter() - t0) / 200 * 1e3:.4f} ms per frame")
      for kernel, name, rows_of in ((0, "K0+K1", app.plan.gbuffer_rows()), (1, "K2", app.plan.raytrace_rows())):
          rows = rows_of[1] - rows_of[0]
          gx, gy = (W + 63) // 64, (rows + 3) // 4
          nb = gx * gy
          buf = (C.c_ulonglong * (3 * nb))()
          assert fn(kernel, buf, nb) == 0
          t = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 3)
          t0 = t[:, 0].astype(np.int64)
          t1 = t[:, 1].astype(np.int64)
          base = t0.min()
          s_us = (t0 - base) / 100.0   # 100 MHz wall clock
          e_us = (t1 - base) / 100.0
          dur = e_us - s_us
          if kernel == 1:
              k2_dur = dur
          span = e_us.max()
          nbk = int(span / args.bucket_us) + 1
          resident = np.zeros(nbk)
          for b in range(nbk):   # workgroups resident at the bucket's middle
              m = (b + 0.5) * args.bucket_us
              resident[b] = np.count_nonzero((s_us <= m) & (e_us > m))
          order = np.argsort(s_us)
          q = np.array_split(order, 8)
          print(f"{name}: {nb} workgroups ({gx} x {gy}), launch span {span:.1f} us, sum of lifetimes {dur.sum():.0f} us "
                f"(= {dur.sum() / span:.0f} resident on average), longest {dur.max():.1f} us, median {np.median(dur):.1f} us")
          print("   resident per %.0f us:" % args.bucket_us, " ".join(f"{int(v)}" for v in resident))
          print("   by start order (eighths): start us / mean lifetime us:",
                " ".join(f"{s_us[i].mean():.0f}/{dur[i].mean():.1f}" for i in q))
          # the last workgroups to end: where they sit in the grid and when they started
          last = np.argsort(e_us)[-8:]
          print("   last to end (block x, y, start, end):", [(int(b % gx), int(b // gx), round(float(s_us[b]), 1), round(float(e_us[b]), 1)) for b in last])
          xcc = (t[:, 2] >> np.uint64(32)).astype(np.int64) & 0xF
          print("   workgroups per XCC:", np.bincount(xcc, minlength=8).tolist())
          res[name] = {"workgroups": nb, "grid": [gx, gy], "span_us": float(span), "resident": resident.tolist(),
                       "bucket_us": args.bucket_us, "start_us": s_us.round(2).tolist(), "end_us": e_us.round(2).tolist()}
    if args.out:
        json.dump(res, open(args.out, "w"))


if __name__ == "__main__":
    main()
