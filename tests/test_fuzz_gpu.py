"""Seeded sweep over the configuration space of the path (sizes, segment bounds, filter iterations, kernel-variant
flags, extension flags, key scripts): every case runs a few frames through the C ABI and is compared with the oracle —
traced colour, ids, ray count and reprojected pixels bit for bit, the final image bit for bit with the exact filter
and within FILTER_TOL otherwise."""
import os

import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu

KEYS = "WASDQEIJKLUO"


def _cases(n, seed):
    rng = np.random.default_rng(seed)
    for i in range(n):
        w = int(rng.choice([1, 7, 64, 65, 100, 191, 256, 333]))
        h = int(rng.choice([1, 3, 4, 5, 33, 64, 100, 121]))
        seg = int(rng.choice([1, 2, 3, 4, 5, 8, 9, 17, 32]))
        n_it = int(rng.choice([1, 2, 3, 5, 6, 9]))
        flags = 0
        for bit in (0x1, 0x2, 0x4, 0x8, 0x200):          # exact, force BVH, direct filter, no compaction, single launch
            if rng.random() < 0.4:
                flags |= bit
        if rng.random() < 0.4:                              # extension modes
            flags |= int(rng.choice([0x10, 0x20, 0x40, 0x80, 0x100, 0xF0, 0x1F0]))
        script = ["".join(rng.choice(list(KEYS), size=rng.integers(0, 3))) for _ in range(int(rng.integers(2, 5)))]
        yield i, w, h, seg, n_it, flags, script


# RTPT_FUZZ_SEEDS=3,4,...: a longer sweep than the two seeds of every run (used at the end of a round's kernel changes)
@pytest.mark.parametrize("seed", [int(v) for v in os.environ.get("RTPT_FUZZ_SEEDS", "1,2").split(",")])
def test_seeded_configuration_sweep(hip_lib, oracle, cornell, seed):
    from real_time_path_tracing_with_spatiotemporal_filtering_amd.app import make_app
    moves = {"S": (2, +0.1), "W": (2, -0.1), "A": (0, -0.1), "D": (0, +0.1), "E": (1, +0.1), "Q": (1, -0.1)}
    for i, w, h, seg, n_it, flags, script in _cases(14, seed):
        tag = (seed, i, w, h, seg, n_it, hex(flags), script)
        app = make_app(w, h, max_segments=seg, iterations=n_it, flags=flags,
                       debug_mask=hip_lib.DEBUG_HIT_ID | hip_lib.DEBUG_PREV_PIXEL)
        ref = oracle.OracleApp(w, h, cornell[2], max_segments=seg, iterations=n_it, ext_flags=flags & 0x1F0)
        ctx = app.backend.ctx
        total = 0
        for keys in script:
            app.updateScene(tuple(keys))
            app.drawVisbilityBuffer()
            app.computeTemporalGradient()
            app.drawSceneToImage()
            traced = ctx.readback(hip_lib.PLANE_IMAGE)
            hit = ctx.readback(hip_lib.PLANE_HIT_ID)
            app.applyTemporalFiltering()
            final = ctx.readback(hip_lib.PLANE_IMAGE)
            pp = ctx.readback(hip_lib.PLANE_PREV_PIXEL)
            app.copyImageToSwapChainsCurrentImage()
            app.frameCount += 1
            # the oracle's scripted moves: the same float32 additions the key handler performs, light wrap included
            ref.camera[:] = app.cameraOrigin
            ref.light[:] = app.lightPos
            ref.camera_moved = any(k in moves for k in keys)
            fo = ref.draw_scene()
            total += fo.rays
            assert bytes(app.pushConstants) == bytes(ref.pc), tag
            assert np.array_equal(hit.reshape(h, w), fo.hit_id), tag
            assert np.array_equal(bits(traced.reshape(h, w, 4)), bits(fo.traced)), tag
            if n_it & 1:
                assert np.array_equal(pp.reshape(h, w, 2), fo.prev_pixel), tag
            got = final.reshape(h, w, 4)
            if flags & 1:
                assert np.array_equal(bits(got), bits(fo.image)), tag
            else:
                err = np.linalg.norm((got[..., :3] - fo.image[..., :3]).astype(np.float64), axis=-1)
                # FILTER_TOL of the reference filter; the extension modes (25 taps, strides to 2^(N-1), weights
                # divided by a standard deviation that can be 1e-4) amplify the 1-ulp error of v_exp_f32: 1e-4 there
                tol = 1e-4 if flags & 0x1F0 else 1e-5
                lim = tol * (1.0 + np.linalg.norm(fo.image[..., :3].astype(np.float64), axis=-1))
                fin = np.isfinite(fo.image[..., :3]).all(-1)
                assert (err[fin] <= lim[fin]).all(), (tag, float((err[fin] / lim[fin]).max()))
        assert ctx.raycount() == total, tag
        app.backend.close()
