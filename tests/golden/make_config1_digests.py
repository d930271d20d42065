#!/usr/bin/env python3
"""Regenerates tests/golden/config1_256x256_sha256.json from the CPU oracle: BASELINE.json configs[0] at its real
size — CornellBox-Original-Merged.obj, 256x256, 1 spp, 2 segments, N = 5, frames 0-1 static, light.x -0.1 on frame 2
(SURVEY.md 8d "Config 1") — as SHA-256 digests of every plane of every frame ("SHA-256 digests for
256x256 planes").  Like the 64x48 fixture these pin the ORACLE against drift and give the GPU tests something to
compare with when the oracle is not at hand; the reference itself produces no output that could be committed."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

W = H = 256
SEGMENTS, ITERATIONS = 2, 5
SCRIPT = [dict(), dict(), dict(move_light=(-0.1, 0.0, 0.0))]
KEYS = [(), (), ("J",)]
PLANES = ("vis", "worldpos", "depth", "gradient", "traced", "hit_id", "image", "prev_pixel")


def digest(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def generate(exact_filter_only=False):
    from oracle import oracle as O
    scene = os.path.join(ROOT, "real_time_path_tracing_with_spatiotemporal_filtering_amd", "scenes",
                         "CornellBox-Original-Merged.obj")
    xyz, idx = O.load_obj(scene)
    app = O.OracleApp(W, H, O.flatten(xyz, idx), max_segments=SEGMENTS, iterations=ITERATIONS)
    out = {"width": W, "height": H, "segments": SEGMENTS, "iterations": ITERATIONS, "frames": []}
    for kw in SCRIPT:
        fo = app.draw_scene(**kw)
        out["frames"].append({**{name: digest(getattr(fo, name)) for name in PLANES}, "rays": int(fo.rays)})
    return out


if __name__ == "__main__":
    path = os.path.join(HERE, "config1_256x256_sha256.json")
    with open(path, "w") as f:
        json.dump(generate(), f, indent=1)
    print(path)
