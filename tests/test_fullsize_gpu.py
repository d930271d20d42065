"""Full-size GPU parity for the two BASELINE.json configurations that round 1 only covered through samples:

* configs[4] — 10x10x10 lattice of 6x6-tessellated Cornell boxes (1,152,000 triangles), 3840x2160, 8 segments, N = 5:
  the shipped kernel set for it (k_pathtrace<BVH, compact> + k_pathtrace_queue<BVH> windows + the per-pixel-normal
  variant of the LDS-staged filter) against the oracle on a row band (the oracle's closest hit is O(rays x triangles):
  one 3840-pixel row of 8-segment paths is ~15 k queries = ~10 s on the box's 16 host threads), plus whole-frame
  equalities that need no oracle (queued vs single-launch paths, LDS-staged vs direct filter), bit for bit.
* configs[3] — the 4K frame split into 8 row strips (270 rows + halo), both halo modes, a vertical camera move in the
  sequence (the reprojected history pixel leaves the strip), against the single-context frame, bit for bit.

PARITY UNPINNED against the reference (nothing of it can run here): the checker is the builder's
restatement in oracle/.
"""
import os

import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu

W4K, H4K = 3840, 2160


def _instanced(oracle, cornell):
    from real_time_path_tracing_with_spatiotemporal_filtering_amd import scenes
    xyz, idx, _ = cornell
    vx, ti, xf, cam, zfar = scenes.instanced_cornell(xyz, idx)
    return vx, ti, xf, cam, zfar


def _make_instanced_app(hip_lib, scene, flags, debug=True, seg=8, n=5):
    from real_time_path_tracing_with_spatiotemporal_filtering_amd.app import HipBackend, PathTracingApplication
    from real_time_path_tracing_with_spatiotemporal_filtering_amd.strips import StripPlan
    vx, ti, xf, cam, zfar = scene
    be = HipBackend(W4K, H4K, StripPlan(H4K, 1, 0, n), max_segments=seg, flags=flags,
                    debug_mask=(hip_lib.DEBUG_HIT_ID | hip_lib.DEBUG_PREV_PIXEL) if debug else 0)
    app = PathTracingApplication(be, W4K, H4K, n, cameraOrigin=cam, z_far=zfar,
                                 lightPos=(1.0, float(cam[1]), float(cam[2]) - 8.0))
    app.objVertices, app.objIndices = vx, ti
    app.buildAccelerationStructure(xf)
    return app


def _ostruct(oracle_type, abi_struct):
    return oracle_type.from_buffer_copy(bytes(abi_struct))


def test_million_triangle_4k_frame_row_band_against_oracle(hip_lib, oracle, cornell):
    """BASELINE configs[4] at its real size.  Frame 0 and frame 1 (camera moved left: reprojection + history blend):
    * row 1080: G-buffer ids / world position / depth, first-hit ids, traced colour and the ray count against the
      oracle's brute force over all 1,152,000 triangles — bit for bit;
    * rows 1075..1085 of the final image and the reprojected pixel against the oracle's a-trous chain run on the
      frame's own (GPU-produced, read back) traced / depth / id / world-position planes: the per-pixel-normal LDS-staged
      kernel at strides 1..5 and the fused reprojection + blend, FILTER_TOL (fast weights)."""
    from test_parity_gpu import l2_ok
    oracle.set_threads(min(16, os.cpu_count() or 1))
    scene = _instanced(oracle, cornell)
    vx, ti, xf, cam, zfar = scene
    tris = oracle.flatten(vx, ti, xf)
    assert len(tris) == 1_152_000
    app = _make_instanced_app(hip_lib, scene, 0)
    ctx = app.backend.ctx
    N = 5
    lut = oracle.lut(tris, np.eye(4, dtype=np.float32).ravel())
    ocfg = oracle.config_default(W4K, H4K)
    ocfg.max_segments = 8
    history = None
    try:
        for frame, keys in enumerate([(), ("A",)]):
            app.updateScene(keys)
            app.drawVisbilityBuffer()
            app.computeTemporalGradient()
            vis, wp, depth = (ctx.readback(p) for p in (hip_lib.PLANE_VIS_ID, hip_lib.PLANE_WORLDPOS, hip_lib.PLANE_DEPTH))
            if frame == 0:
                # the row with the most geometry near the middle of the frame (the exact middle looks into the gap
                # between two layers of boxes)
                cover = (vis[H4K // 2 - 200:H4K // 2 + 200] > 0).sum(axis=1)
                ROW = H4K // 2 - 200 + int(np.argmax(cover))
                B0, B1 = ROW - 5, ROW + 6
                assert cover.max() > W4K // 4
            ctx.set_count_rows(ROW, ROW + 1)
            ctx.reset_counters()
            app.drawSceneToImage()
            hit, traced, rays = ctx.readback(hip_lib.PLANE_HIT_ID), ctx.readback(hip_lib.PLANE_IMAGE), ctx.raycount()
            app.applyTemporalFiltering()
            final, pp = ctx.readback(hip_lib.PLANE_IMAGE), ctx.readback(hip_lib.PLANE_PREV_PIXEL)
            app.copyImageToSwapChainsCurrentImage()
            app.frameCount += 1
            opc, oubo = _ostruct(oracle.PushConstants, app.pushConstants), _ostruct(oracle.Ubo, app.ubo)
            # ---- the traced row, brute force
            ovis, owp, odepth = oracle.gbuffer(ocfg, tris, oubo, ROW, ROW + 1)
            assert np.array_equal(vis[ROW], ovis[ROW]) and vis[ROW].max() > 100_000 and (vis[ROW] > 0).sum() > W4K // 4
            assert np.array_equal(bits(wp[ROW]), bits(owp[ROW]))
            assert np.array_equal(bits(depth[ROW]), bits(odepth[ROW]))
            oimg, orays, ohit = oracle.raytrace(ocfg, opc, tris, ROW, ROW + 1)
            assert np.array_equal(hit[ROW], ohit[ROW])
            assert np.array_equal(bits(traced[ROW, :, :3]), bits(oimg[ROW, :, :3])), "traced colour of the row"
            assert rays == orays and rays > 2 * W4K
            # ---- the filter chain on the band, from this frame's own planes
            cur = traced.copy()
            cur[..., 3] = 0.0
            opc.maxWaveletIteration = N
            opp = None
            for k in range(1, N + 1):
                opc.waveletIteration = k
                rem = sum(range(k + 1, N + 1))
                res = oracle.atrous(ocfg, opc, oubo, cur, depth, vis, lut, lut, wp, history, B0 - rem, B1 + rem,
                                    want_prev_pixel=(k == N))
                cur, opp = res if k == N else (res, None)
            assert np.array_equal(pp[B0:B1], opp[B0:B1]), "reprojected pixels of the band"
            ok, rel = l2_ok(final[B0:B1], cur[B0:B1])
            assert ok, f"frame {frame}: filtered band outside FILTER_TOL: {rel}"
            if frame == 1:
                assert (pp[B0:B1, :, 0] != np.arange(W4K)[None, :]).mean() > 0.2, "the camera move did shift the history fetch"
            history = final
    finally:
        ctx.set_count_rows(0, H4K)
        oracle.set_threads(min(8, os.cpu_count() or 1))
        app.backend.close()


def test_million_triangle_4k_frame_kernel_variant_equalities(hip_lib, oracle, cornell):
    """BASELINE configs[4], whole frames, no oracle needed: the path queue (k_pathtrace_queue windows) against one
    launch per path, and the LDS-staged per-pixel-normal filter against the direct-load kernel, under the exact filter
    arithmetic — traced image, ray count and final image bit for bit over three frames with a light and a camera move."""
    scene = _instanced(oracle, cornell)
    X = hip_lib.FLAG_EXACT_FILTER
    outs = []
    for flags in (X, X | hip_lib.FLAG_SINGLE_LAUNCH_PATHS, X | hip_lib.FLAG_DIRECT_FILTER):
        app = _make_instanced_app(hip_lib, scene, flags, debug=False)
        ctx = app.backend.ctx
        for f, keys in enumerate([(), ("J",), ("D",)]):
            app.updateScene(keys)
            app.drawVisbilityBuffer()
            app.computeTemporalGradient()
            app.drawSceneToImage()
            if f == 2:
                traced = ctx.readback(hip_lib.PLANE_IMAGE)
            app.applyTemporalFiltering()
            app.copyImageToSwapChainsCurrentImage()
            app.frameCount += 1
        outs.append((traced, ctx.readback(hip_lib.PLANE_PREVIOUS), ctx.raycount()))
        app.backend.close()
    base = outs[0]
    assert base[2] > 3 * 2 * W4K * H4K
    assert np.isfinite(base[1][..., :3]).mean() > 0.999 and not base[1][..., 3].any()
    for name, o in zip(("single-launch paths", "direct filter"), outs[1:]):
        assert o[2] == base[2], name
        assert np.array_equal(bits(o[0]), bits(base[0])), name + ": traced image"
        assert np.array_equal(bits(o[1]), bits(base[1])), name + ": final image"


@pytest.mark.parametrize("mode", ["redundant", "exchange"])
def test_4k_eight_strips_equal_single_frame(hip_lib, mode):
    """BASELINE configs[3] at its real size: the 3840x2160 frame as 8 strip contexts (270 rows + 15 / 5 halo rows) on
    one GPU, four frames with a vertical camera move (history fetched from the neighbouring strip through the gathered
    previous frame) and a light move, against the single-context frame — every pixel, bit for bit."""
    from test_parity_gpu import _strips_vs_single
    _strips_vs_single(W4K, H4K, 4, 5, 8, mode, 0, [(), ("E",), ("J",), ()])


@pytest.mark.parametrize("mode", ["redundant", "exchange"])
def test_million_triangle_4k_eight_strips_equal_single_frame(hip_lib, oracle, cornell, mode):
    """BASELINE configs[4] as BASELINE defines it — the 1,152,000-triangle lattice at 3840x2160, 8 segments, on EIGHT row
    strips: BVH traversal over fan pairs + per-pixel-normal filter + strip contexts + a camera move (history fetched across
    strips) + a light move, against the single-context frame, every pixel bit for bit, both halo modes."""
    from test_parity_gpu import _strips_vs_single
    vx, ti, xf, cam, zfar = _instanced(oracle, cornell)
    # the strips balanced from their measured times (profiles/r04_instanced_balanced_strips.json) in one mode, equal ones in the other
    splits = (0, 338, 566, 793, 1072, 1353, 1585, 1818, 2160) if mode == "redundant" else ()
    _strips_vs_single(W4K, H4K, 8, 5, 8, mode, 0, [(), ("E",), ("J",)], mesh=(vx, ti), instance_xforms=xf, cameraOrigin=cam, z_far=zfar,
                      lightPos=(1.0, float(cam[1]), float(cam[2]) - 8.0), splits=splits)


@pytest.mark.parametrize("strips", [["--ranks", "8"], ["--ranks", "4", "--frames-in-flight", "2", "--splits", "0,600,1080,1500,2160"]])
def test_million_triangle_4k_cpp_host_on_eight_strips_equals_python_host(hip_lib, oracle, cornell, tmp_path, strips):
    """The same configuration driven by the C++ host (north_star: "the host stays C++"): `rtpt_app --lattice 10x10x10
    --tessellate 6 --ranks 8` builds the scene itself (host/scene_gen.cpp), runs eight in-process strip contexts with
    redundant halo rows and, in the frames where the camera moved, swaps the history bands bounded with the posed, instanced
    scene box (host/strips.cpp: reprojection_rows) — against the Python host's single context, bit for bit.  Second case: four
    strips of unequal height with two frames in flight (the bands then come from the ranks' other contexts)."""
    import json
    import subprocess
    from test_cpp_host import APP, PKG, read_pfm
    subprocess.check_call(["make", "-C", os.path.join(PKG, "host"), "-s"])
    keys = ["", "E", "A"]
    pfm = tmp_path / "out.pfm"
    out = subprocess.run([APP, "--width", str(W4K), "--height", str(H4K), "--segments", "8", "--iterations", "5", "--frames", str(len(keys)),
                          "--script", ",".join(keys), "--dump", str(pfm), "--lattice", "10x10x10", "--tessellate", "6"] + strips,
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    stats = json.loads(out.stdout.strip().splitlines()[-1])
    assert stats["bytes_sent"] > 0, "the camera moves did exchange history bands"
    app = _make_instanced_app(hip_lib, _instanced(oracle, cornell), 0, debug=False)
    try:
        for k in keys:
            app.drawScene(tuple(k))
        want = app.backend.ctx.readback(hip_lib.PLANE_IMAGE)
        assert stats["rays"] == app.backend.ctx.raycount()
    finally:
        app.backend.close()
    got = read_pfm(pfm)
    assert np.array_equal(bits(got), bits(np.ascontiguousarray(want[..., :3])))


def test_two_contexts_are_independent(hip_lib, oracle, cornell):
    """rtpt.h: "distinct contexts are independent".  The CU count and the raised dynamic-LDS limit of the staged filter
    kernels are per-context / per-device state (they were process-global statics keyed on the first device).  Two
    contexts — on two devices when the box has them, otherwise two on one device driven from two host threads — run
    the N = 9 chain (strides 5..9 need the > 64 KiB LDS instance) concurrently and must both match the oracle."""
    import threading
    import torch
    from test_parity_gpu import l2_ok
    from real_time_path_tracing_with_spatiotemporal_filtering_amd.app import HipBackend, PathTracingApplication
    from real_time_path_tracing_with_spatiotemporal_filtering_amd.strips import StripPlan
    ndev = torch.cuda.device_count()
    w, h, n = 333, 170, 9
    xyz, idx, tris = cornell
    apps = []
    for i in range(2):
        be = HipBackend(w, h, StripPlan(h, 1, 0, n), max_segments=4, device=(i % ndev))
        a = PathTracingApplication(be, w, h, n)
        a.objVertices, a.objIndices = xyz, idx
        a.buildAccelerationStructure()
        apps.append(a)
    errs = []

    def run(a):
        try:
            for f in range(3):
                a.drawScene(("J",) if f == 1 else ())
            a.backend.sync()
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=run, args=(a,)) for a in apps]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    ref = oracle.OracleApp(w, h, tris, max_segments=4, iterations=n)
    for f in range(3):
        fo = ref.draw_scene(move_light=(-0.1, 0, 0) if f == 1 else None)
    for a in apps:
        ok, rel = l2_ok(a.backend.readback_rows(hip_lib.PLANE_PREVIOUS, 0, h), fo.image)
        assert ok, rel
        a.backend.close()


@pytest.mark.parametrize("n_iter", [4, 5])
def test_device_side_alpha_is_zero_after_the_last_iteration(hip_lib, n_iter):
    """The reference stores vec4(rgb, 0) (temporalFiltering.comp.glsl:152,:263).  Internally alpha carries the G-buffer
    depth between passes; a device-side consumer (rtpt_bind_plane / rtpt_plane_ptr — the swapchain / torch interop
    route of INTEGRATION.md) must still see alpha 0 on IMAGE after the last iteration of a frame, for odd AND even N,
    and on PREVIOUS after the hand-over."""
    from real_time_path_tracing_with_spatiotemporal_filtering_amd import abi
    from real_time_path_tracing_with_spatiotemporal_filtering_amd.app import make_app
    w, h = 130, 70
    app = make_app(w, h, max_segments=3, iterations=n_iter, torch_planes=True)
    be = app.backend
    for f in range(2):
        with be.stream_scope():
            app.updateScene()
            app.drawVisbilityBuffer()
            app.computeTemporalGradient()
            app.drawSceneToImage()
            app.applyTemporalFiltering()
        be.sync()
        img = be.color_rows(abi.PLANE_IMAGE, 0, h).cpu().numpy()   # the torch tensor bound as IMAGE: device bytes as they are
        assert np.isfinite(img[..., :3]).all() and img[..., :3].max() > 0
        assert not img[..., 3].any(), f"N = {n_iter}: depth leaked into IMAGE.alpha on the device"
        with be.stream_scope():
            app.copyImageToSwapChainsCurrentImage()
        app.frameCount += 1
        be.sync()
        prev = be.color_rows(abi.PLANE_PREVIOUS, 0, h).cpu().numpy()
        assert np.array_equal(bits(prev), bits(img))
    be.close()


def test_1080p_frames_against_oracle_and_chain_equals_separate_passes(hip_lib, oracle, cornell):
    """BASELINE configs[1] at its size (1920x1080, 4 segments, N = 5) under the DEFAULT kernel policy — at this size the
    iterations k < N run as chained pairs on 32 row segments of 34 rows x 16 column strips with an unrounded last step
    (a geometry no other size exercises).  Two frames, the light moves in the second:
    * every observable against the oracle: ids, first hits, traced colour, gradient, reprojected pixel bit for bit, the
      final image within FILTER_TOL;
    * with RTPT_FLAG_EXACT_FILTER the chained run equals the one-kernel-per-iteration run (RTPT_FLAG_NO_FILTER_FUSION)
      AND the oracle's image bit for bit."""
    from test_parity_gpu import l2_ok, make_pair
    oracle.set_threads(min(16, os.cpu_count() or 1))
    W, H = 1920, 1080
    runs = {}
    try:
        for name, flags in (("default", 0), ("exact", hip_lib.FLAG_EXACT_FILTER),
                            ("exact_unfused", hip_lib.FLAG_EXACT_FILTER | hip_lib.FLAG_NO_FILTER_FUSION)):
            app, ref = make_pair(hip_lib, oracle, cornell, w=W, h=H, seg=4, n=5, flags=flags)
            ctx = app.backend.ctx
            ctx.timing_enable(1)
            frames, total_rays = [], 0
            for keys, light_move in (((), None), (("J",), (-0.1, 0, 0))):
                app.updateScene(keys)
                app.drawVisbilityBuffer()
                app.computeTemporalGradient()
                app.drawSceneToImage()
                obs = dict(vis=ctx.readback(hip_lib.PLANE_VIS_ID), hit=ctx.readback(hip_lib.PLANE_HIT_ID),
                           traced=ctx.readback(hip_lib.PLANE_IMAGE), grad=ctx.readback(hip_lib.PLANE_GRADIENT), rays=ctx.raycount())
                app.applyTemporalFiltering()
                obs["final"], obs["pp"] = ctx.readback(hip_lib.PLANE_IMAGE), ctx.readback(hip_lib.PLANE_PREV_PIXEL)
                app.copyImageToSwapChainsCurrentImage()
                app.frameCount += 1
                if name != "exact_unfused":   # the oracle's frames are the same for both arithmetics of the GPU run
                    fo = ref.draw_scene(move_light=light_move)
                    total_rays += fo.rays
                    assert np.array_equal(obs["vis"], fo.vis) and np.array_equal(obs["hit"], fo.hit_id)
                    assert np.array_equal(bits(obs["traced"]), bits(fo.traced))
                    assert np.array_equal(bits(obs["grad"]), bits(fo.gradient))
                    assert np.array_equal(obs["pp"], fo.prev_pixel)
                    assert obs["rays"] == total_rays
                    if name == "exact":
                        assert np.array_equal(bits(obs["final"]), bits(fo.image)), "exact filter: the oracle's bits"
                    else:
                        ok, rel = l2_ok(obs["final"], fo.image)
                        assert ok, rel
                frames.append(obs["final"])
            tm = ctx.timing_collect()
            chained = tm["k_atrous_chain"][1]
            assert chained == (0 if name == "exact_unfused" else 4), (name, tm)   # 2 pairs x 2 frames
            runs[name] = frames
            app.backend.close()
        for a, b in zip(runs["exact"], runs["exact_unfused"]):
            assert np.array_equal(bits(a), bits(b)), "chain == separate passes at 1080p"
    finally:
        oracle.set_threads(min(8, os.cpu_count() or 1))
