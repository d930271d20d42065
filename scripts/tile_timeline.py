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
    ap.add_argument("--splits", default=None, metavar="0,A,...,H", help="with --strip R/N: unequal strips (strips.StripPlan.splits)")
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
                   mode="redundant", torch_planes=False, splits=tuple(int(v) for v in args.splits.split(",")) if args.splits else (), **extra)
    for _ in range(args.frames):
        app.drawScene(())
    app.backend.ctx.sync()
    # the filter's launches (atrous_chain.hip / atrous.hip, experiments/span_instrumentation.inc) fold every workgroup of every
    # launch since the last clear: re-arm them, draw ONE frame, and every launch of that frame sits on one clock
    spans, reader_names = [], {}
    for reader, slots, names in (("rtpt_debug_chain_span", 4, ["chain (1,2)", "chain (3,4)", "chain (1,2) final", "chain (3,4)+ final"]),
                                 ("rtpt_debug_comb_span", 2, ["single pass k < N", "final pass"])):
        f = getattr(lib, reader, None)
        if f is not None:
            f.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
            assert f(None, 1) == 0
            spans.append((f, slots, names))
            reader_names[id(f)] = reader
    if spans:
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
                       "bucket_us": args.bucket_us, "start_us": s_us.round(2).tolist(), "end_us": e_us.round(2).tolist(),
                       "abs_start": t0.tolist(), "abs_end": t1.tolist()}
    # K0 + K1 + K2 as ONE launch (the default): both kinds of workgroups on one clock — the G-buffer tiles are dispatched behind the
    # tracing tiles and run while those drain
    if "K0+K1" in res and "K2" in res and "abs_start" in res["K0+K1"]:
        a0, a1 = np.array(res["K0+K1"]["abs_start"]), np.array(res["K0+K1"]["abs_end"])
        b0, b1 = np.array(res["K2"]["abs_start"]), np.array(res["K2"]["abs_end"])
        base = min(a0.min(), b0.min())
        if abs(a0.min() - b0.min()) < 100000:   # within 1 ms of each other: the same launch
            span = (max(a1.max(), b1.max()) - base) / 100.0
            nbk = int(span / args.bucket_us) + 1
            rows = []
            for b in range(nbk):
                m = base + (b + 0.5) * args.bucket_us * 100.0
                rows.append((int(np.count_nonzero((b0 <= m) & (b1 > m))), int(np.count_nonzero((a0 <= m) & (a1 > m)))))
            print(f"one launch: span {span:.1f} us; resident per {args.bucket_us:.0f} us (tracing tiles + G-buffer tiles):",
                  " ".join(f"{t}+{g}" for t, g in rows))
            print(f"   first G-buffer tile starts at {(a0.min() - base) / 100.0:.1f} us, last tracing tile ends at {(b1.max() - base) / 100.0:.1f} us, "
                  f"last G-buffer tile at {(a1.max() - base) / 100.0:.1f} us")
            res["one_launch"] = {"span_us": float(span), "resident_trace_gbuffer": rows}
        # the whole frame on that clock: the launches of the filter behind the trace
        raws, raw_idx = {}, {}
        frame = [("K0+K1+K2", base, max(a1.max(), b1.max()), float((a1 - a0).sum() + (b1 - b0).sum()) / 100.0, len(a0) + len(b0), None)]
        for f, slots, names in spans:
            buf = (C.c_ulonglong * (slots * 4 + slots * 132))()
            assert f(buf, 0) == 0
            v = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)
            for sl in range(slots):
                first, last_, life, n = v[4 * sl: 4 * sl + 4]
                if n:
                    st = v[4 * slots + 132 * sl: 4 * slots + 132 * (sl + 1)]
                    frame.append((names[sl], first, last_, life / 100.0, int(n), st if st[2] > 0 else None))
                    fr = getattr(lib, reader_names[id(f)] + "_raw", None)
                    if fr is not None:
                        fr.argtypes = [C.c_int, C.POINTER(C.c_ulonglong), C.c_uint32]
                        rb = (C.c_ulonglong * (3 * 4096))()
                        if fr(sl, rb, 4096) == 0:
                            rr = np.frombuffer(rb, dtype=np.uint64).astype(np.int64).reshape(4096, 3)
                            raws[names[sl]] = rr[rr[:, 0] != 0]
                            raw_idx[names[sl]] = np.nonzero(rr[:, 0] != 0)[0]
        frame.sort(key=lambda e: e[1])
        if len(frame) > 1 and frame[-1][2] - base < 10000000:   # the same frame
            print("the frame on one clock (us from the first tracing tile; gap = idle time since the previous launch's last workgroup ended):")
            prev_end = None
            res["frame"] = []
            for name, first, last_, life, n, st in frame:
                gap = (first - prev_end) / 100.0 if prev_end is not None else 0.0
                print(f"   {name:20s} start {(first - base) / 100.0:7.1f}  end {(last_ - base) / 100.0:7.1f}  span {(last_ - first) / 100.0:6.1f}  gap {gap:5.1f}  "
                      f"{n:5d} workgroups, mean lifetime {life / n:6.1f}")
                e = {"launch": name, "start_us": (first - base) / 100.0, "end_us": (last_ - base) / 100.0, "gap_us": gap, "workgroups": n,
                     "mean_lifetime_us": life / n}
                raw = raws.get(name)
                if raw is not None and len(raw):
                    lt = (raw[:, 1] - raw[:, 0]) / 100.0
                    st0 = (raw[:, 0] - first) / 100.0
                    hw = raw[:, 2]
                    cu = ((hw >> 32) & 0xF) * 4096 + ((hw >> 13) & 0x7) * 512 + ((hw >> 12) & 0x1) * 256 + ((hw >> 8) & 0xF)
                    per_cu = {c: int(np.count_nonzero(cu == c)) for c in np.unique(cu)}
                    share = np.array([per_cu[c] for c in cu])
                    print(f"      lifetimes us: min {lt.min():.1f}  median {np.median(lt):.1f}  p90 {np.percentile(lt, 90):.1f}  max {lt.max():.1f};  "
                          f"starts: median {np.median(st0):.1f}  p90 {np.percentile(st0, 90):.1f}  last {st0.max():.1f};  {len(per_cu)} CUs used, "
                          "workgroups per CU -> CUs / mean lifetime: " +
                          "  ".join(f"{k}: {sum(1 for v in per_cu.values() if v == k)} / {lt[share == k].mean():.1f}" for k in sorted(set(per_cu.values()))))
                    idx = raw_idx[name]
                    print("      mean lifetime by XCD (workgroup index & 7): " + " ".join(f"{lt[(idx & 7) == x].mean():.1f}" for x in range(8) if np.any((idx & 7) == x)) +
                          ";  by position in the launch (eighths of the workgroup index): " +
                          " ".join(f"{lt[q].mean():.1f}" for q in np.array_split(np.argsort(idx), 8) if len(q)) +
                          ";  end of the first / median / last workgroup: " + f"{(raw[:, 1].min() - first) / 100.0:.1f} / {(np.median(raw[:, 1]) - first) / 100.0:.1f} / {(raw[:, 1].max() - first) / 100.0:.1f}")
                    e["lifetime_us"] = {"min": float(lt.min()), "median": float(np.median(lt)), "p90": float(np.percentile(lt, 90)), "max": float(lt.max())}
                    e["start_us"] = {"median": float(np.median(st0)), "p90": float(np.percentile(st0, 90)), "last": float(st0.max())}
                if st is not None:
                    T = int(min(st[2], 128))
                    steps = np.diff(np.concatenate([[st[1]], st[3:3 + T]])) / 100.0
                    print(f"      probe workgroup: starts {(st[0] - first) / 100.0:.1f} us into the launch, prologue {(st[1] - st[0]) / 100.0:.1f} us, {int(st[2])} steps, "
                          f"mean {steps.mean():.2f} us: " + " ".join(f"{x:.1f}" for x in steps))
                    e["probe"] = {"prologue_us": (st[1] - st[0]) / 100.0, "steps_us": steps.round(2).tolist()}
                res["frame"].append(e)
                prev_end = last_
            print(f"   frame: {(frame[-1][2] - base) / 100.0:.1f} us from the first workgroup's start to the last one's end")
        for k in ("K0+K1", "K2"):
            res[k].pop("abs_start", None)
            res[k].pop("abs_end", None)
    if args.out:
        json.dump(res, open(args.out, "w"))


if __name__ == "__main__":
    main()
