"""K2 on small scenes, RTPT_FLAG_BINNED_PATHS: paths handed from segment to segment through queues binned by which
objects' bounds the next ray enters (kernels.hpp ClusterInfo), against the default segment-window path — the triangles a
class leaves out cannot be hit, so image, first-hit ids and ray count must not change by a bit.  (By default the mode
runs for launches of >= 4 M pixels, where it is faster; the flags force it on / off at any size.)"""
import numpy as np
import pytest

from conftest import bits


def test_cornell_clusters(cornell):
    """host-only: the Cornell box splits into the room (5 walls sharing corner positions, always tested together with the
    light quad, which saves too little to earn a class bit) and the two boxes, 10 triangles each"""
    from real_time_path_tracing_with_spatiotemporal_filtering_amd import abi
    abi.load()
    _, _, tris = cornell
    always, masks, bounds = abi.clusters(tris)
    box = lambda lo, hi: sum(1 << t for t in range(lo, hi))
    assert sorted(masks) == sorted([box(10, 20), box(20, 30)])
    assert always == box(0, 10) | box(30, 32)
    for m, (lo, hi) in zip(masks, bounds):
        ids = [t for t in range(32) if m >> t & 1]
        v = tris[ids].reshape(-1, 3)
        assert (lo < v.min(0)).all() and (hi > v.max(0)).all() and (v.min(0) - lo).max() < 1e-3
    # one connected object: nothing to cull against
    always, masks, _ = abi.clusters(tris[:10])
    assert masks == [] and always == box(0, 10)


@pytest.mark.gpu
@pytest.mark.parametrize("size,seg", [((333, 190), 2), ((333, 190), 5), ((64, 4), 8), ((1, 1), 4), ((1000, 800), 8), ((3840, 2160), 4)])
def test_binned_paths_equal_segment_windows(hip_lib, size, seg):
    from real_time_path_tracing_with_spatiotemporal_filtering_amd.app import make_app
    w, h = size
    outs = []
    for flags in (hip_lib.FLAG_BINNED_PATHS, hip_lib.FLAG_NO_BINNED_PATHS):
        app = make_app(w, h, max_segments=seg, iterations=3, flags=flags | hip_lib.FLAG_EXACT_FILTER, debug_mask=hip_lib.DEBUG_HIT_ID)
        ctx = app.backend.ctx
        for f in range(3):
            app.updateScene(("J",) if f == 1 else (("D",) if f == 2 else ()))
            app.drawVisbilityBuffer()
            app.computeTemporalGradient()
            app.drawSceneToImage()
            traced, hit = ctx.readback(hip_lib.PLANE_IMAGE), ctx.readback(hip_lib.PLANE_HIT_ID)
            app.applyTemporalFiltering()
            app.copyImageToSwapChainsCurrentImage()
            app.frameCount += 1
        outs.append((traced, hit, ctx.readback(hip_lib.PLANE_PREVIOUS), ctx.raycount()))
        app.backend.close()
    assert outs[0][3] == outs[1][3] and outs[0][3] >= 3 * w * h
    assert np.array_equal(outs[0][1], outs[1][1])
    assert np.array_equal(bits(outs[0][0]), bits(outs[1][0])) and np.array_equal(bits(outs[0][2]), bits(outs[1][2]))


@pytest.mark.gpu
def test_binned_paths_follow_a_moving_model(hip_lib, oracle, cornell):
    """the cluster bounds are recomputed with the pose (rtpt_gbuffer's model matrix): a rotating scene against the oracle"""
    from test_parity_gpu import make_pair
    from test_scene_ext import rot_y_translate
    app, ref = make_pair(hip_lib, oracle, cornell, w=120, h=90, seg=6, n=1, flags=hip_lib.FLAG_BINNED_PATHS)
    total = 0
    for f in range(4):
        m = rot_y_translate(0.4 * f, (0.1 * f, 0, -0.05 * f))
        app.modelMatrix = m
        ref.model = m
        app.updateScene()
        app.drawVisbilityBuffer()
        app.computeTemporalGradient()
        app.drawSceneToImage()
        traced = app.backend.ctx.readback(hip_lib.PLANE_IMAGE)
        app.applyTemporalFiltering()
        app.copyImageToSwapChainsCurrentImage()
        app.frameCount += 1
        fo = ref.draw_scene()
        assert np.array_equal(bits(traced), bits(fo.traced)), f
        total += fo.rays
    assert app.backend.ctx.raycount() == total
    app.backend.close()
