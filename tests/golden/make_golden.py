#!/usr/bin/env python3
"""Regenerates tests/golden/cornell_64x48.npz from the CPU oracle (oracle/rtpt_oracle.c).

The reference ships no golden images and cannot run here, so these vectors pin the
ORACLE against drift (compiler, flags, platform fma) and give the GPU tests a fixture that does not
need the oracle at run time.  Scenario = BASELINE.json configs[0] scaled down: Cornell box, 64x48,
1 spp, 2 segments, N = 5; frames 0-1 static, light.x -0.1 on frame 2 (key J, main.cpp:1157-1158),
camera x +0.1 on frame 3 (key D, main.cpp:1132-1135: a ~2 px reprojection shift).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

W, H, SEGMENTS, ITERATIONS = 64, 48, 2, 5
SCRIPT = [dict(), dict(), dict(move_light=(-0.1, 0.0, 0.0)), dict(move_camera=(0.1, 0.0, 0.0))]
KEYS = [(), (), ("J",), ("D",)]


def generate():
    from oracle import oracle as O
    scene = os.path.join(ROOT, "real_time_path_tracing_with_spatiotemporal_filtering_amd", "scenes",
                         "CornellBox-Original-Merged.obj")
    xyz, idx = O.load_obj(scene)
    app = O.OracleApp(W, H, O.flatten(xyz, idx), max_segments=SEGMENTS, iterations=ITERATIONS)
    out = {}
    for f, kw in enumerate(SCRIPT):
        fo = app.draw_scene(**kw)
        for name in ("vis", "worldpos", "depth", "gradient", "traced", "hit_id", "image", "prev_pixel"):
            out[f"f{f}_{name}"] = getattr(fo, name)
        out[f"f{f}_rays"] = np.array([fo.rays], np.uint64)
    return out


if __name__ == "__main__":
    data = generate()
    path = os.path.join(HERE, "cornell_64x48.npz")
    np.savez_compressed(path, **data)
    print(path, os.path.getsize(path), "bytes")
