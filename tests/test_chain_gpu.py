"""K3 chained iterations (atrous_chain.hip) against the one-kernel-per-iteration path (RTPT_FLAG_NO_FILTER_FUSION): the
chain keeps the intermediate image in LDS instead of HBM and must not change a single bit — per-pixel arithmetic and
accumulation order are the separate passes' (temporalFiltering.comp.glsl:118-155).  Every other GPU test runs with the
chain on (it is the default), i.e. against the oracle; this file pins chain == separate passes at sizes the oracle is
too slow for, and the recording semantics of rtpt_temporal_filter."""
import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def chain_every_size(monkeypatch, request):
    """by default only launches of >= 1 M pixels are chained (shorter row segments do not pay for the pipeline fill);
    these tests want the chain at every size — the knob is read by rtpt_create"""
    if "default_policy" not in request.keywords:
        monkeypatch.setenv("RTPT_CHAIN_MIN_PIXELS", "0")


def _frames(hip_lib, w, h, n, flags, keys, seg=3, **kw):
    from real_time_path_tracing_with_spatiotemporal_filtering_amd.app import make_app
    app = make_app(w, h, max_segments=seg, iterations=n, flags=flags, debug_mask=hip_lib.DEBUG_PREV_PIXEL, **kw)
    outs = []
    for k in keys:
        app.drawScene(k)
        outs.append((app.backend.ctx.readback(hip_lib.PLANE_PREVIOUS), app.backend.ctx.readback(hip_lib.PLANE_PREV_PIXEL)))
    app.backend.close()
    return outs, None


@pytest.mark.parametrize("exact", [0, 1])
@pytest.mark.parametrize("size", [(1, 1), (63, 5), (65, 7), (130, 33), (121, 64), (333, 170), (1000, 800)])
def test_chain_equals_separate_passes(hip_lib, size, exact):
    """ragged widths (strip tails, a strip of one column), frames shorter than a ring, N = 2..9 (pairs (1,2), (3,4) run
    chained, larger strides as far as the LDS admits, odd and even N), fast and exact weights, camera + light moving"""
    w, h = size
    keys = [(), ("J",), ("D", "E"), ()]
    for n in (2, 3, 4, 5, 9):
        a, _ = _frames(hip_lib, w, h, n, exact, keys)
        b, _ = _frames(hip_lib, w, h, n, exact | hip_lib.FLAG_NO_FILTER_FUSION, keys)
        for f, ((ia, pa), (ib, pb)) in enumerate(zip(a, b)):
            assert np.array_equal(bits(ia), bits(ib)), (size, n, exact, f)
            assert np.array_equal(pa, pb)


@pytest.mark.parametrize("g", [2, 3, 6])
def test_sliding_window_variant_equals_separate_passes(hip_lib, monkeypatch, g):
    """RTPT_CHAIN_SW=1: the round-3 sliding-window kernel (csrc/experiments/: a wave owns a residue class of rows and keeps
    the 3x3 tap window in registers; measured slower, profiles/r03_chain_sw_ab.csv; only in -DRTPT_AB_VARIANTS=1 builds)
    computes the same bits, at the
    frame borders (top/bottom clamp events), on ragged strips and for every rows-per-step setting"""
    if not hasattr(hip_lib.load(), "rtpt_debug_ab_variants"):
        pytest.skip("the library was built without the A/B variants (scripts/build_variant.sh ab -DRTPT_AB_VARIANTS=1, RTPT_LIB_PATH)")
    monkeypatch.setenv("RTPT_CHAIN_SW", "1")
    monkeypatch.setenv("RTPT_CHAIN_SW_G1", str(g))
    monkeypatch.setenv("RTPT_CHAIN_SW_G3", str(g))
    keys = [(), ("J",), ("D", "E"), ()]
    for (w, h) in ((1, 1), (63, 5), (65, 7), (130, 33), (333, 170), (1000, 800)):
        for n in (2, 3, 5):
            for exact in (0, 1):
                a, _ = _frames(hip_lib, w, h, n, exact, keys)
                b, _ = _frames(hip_lib, w, h, n, exact | hip_lib.FLAG_NO_FILTER_FUSION, keys)
                for f, ((ia, pa), (ib, pb)) in enumerate(zip(a, b)):
                    assert np.array_equal(bits(ia), bits(ib)), (w, h, n, exact, f)
                    assert np.array_equal(pa, pb)


@pytest.mark.parametrize("g,generic", [(2, 0), (3, 0), (4, 0), (2, 1), (3, 1), (4, 1)])
def test_rows_per_step_of_a_pair(hip_lib, monkeypatch, g, generic):
    """a chained pair runs four rows per level and step where its row segments are short (strips, small frames), the pair
    (1,2) two on tall frames, three otherwise (chain_g(), atrous_chain.hip): RTPT_CHAIN_G1 pins the choice (4: every pair), and
    all give the bits of the separate passes at every frame shape — from the instantiations with the strides of the default
    pairs compiled in and (RTPT_CHAIN_GENERIC=1) from the one that reads them"""
    monkeypatch.setenv("RTPT_CHAIN_G1", str(g))
    monkeypatch.setenv("RTPT_CHAIN_GENERIC", str(generic))
    keys = [(), ("J",), ("D", "E"), ()]
    for (w, h) in ((1, 1), (63, 5), (65, 7), (130, 33), (333, 170), (1000, 800)):
        for n in (2, 3, 5):
            for exact in (0, 1):
                a, _ = _frames(hip_lib, w, h, n, exact, keys)
                b, _ = _frames(hip_lib, w, h, n, exact | hip_lib.FLAG_NO_FILTER_FUSION, keys)
                for f, ((ia, pa), (ib, pb)) in enumerate(zip(a, b)):
                    assert np.array_equal(bits(ia), bits(ib)), (w, h, n, exact, f)
                    assert np.array_equal(pa, pb)


@pytest.mark.parametrize("env", [{"RTPT_CHAIN_SKEW": "0"}, {"RTPT_CHAIN_SKEW": "80,60"}, {"RTPT_CHAIN_BW": "120"}, {"RTPT_CHAIN_BW": "96", "RTPT_CHAIN_SKEW": "45,45"},
                                 {"RTPT_CHAIN_G1": "14"}, {"RTPT_CHAIN_WG_PER_CU": "1"}, {"RTPT_CHAIN_WG_PER_CU": "3", "RTPT_CHAIN_SKEW": "50"}])
def test_chain_geometry_switches_do_not_change_the_frames(hip_lib, monkeypatch, env):
    """how a chained launch cuts the frame — row segments sized by the age of their workgroups (RTPT_CHAIN_SKEW; the default is
    a skew too), the strip width of the pair (1,2), rows per step of that pair alone, workgroups per CU — is speed only: every
    cut gives the bits of the separate passes, on frames tall enough for several segments per XCD and on ragged ones"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    keys = [(), ("J",), ("D", "E")]
    for (w, h) in ((130, 33), (333, 700), (1000, 800), (2000, 1100)):
        for n in (3, 5):
            a, _ = _frames(hip_lib, w, h, n, 0, keys)
            b, _ = _frames(hip_lib, w, h, n, hip_lib.FLAG_NO_FILTER_FUSION, keys)
            for f, ((ia, pa), (ib, pb)) in enumerate(zip(a, b)):
                assert np.array_equal(bits(ia), bits(ib)), (env, w, h, n, f)
                assert np.array_equal(pa, pb)


@pytest.mark.parametrize("window", [1, 2, 3])
def test_segment_window_of_the_tile_kernel(hip_lib, oracle, cornell, monkeypatch, window):
    """RTPT_PT_WINDOW: the tile kernel hands its survivors to the queue kernels after `window` segments instead of 4 (brute
    force) / 8 (BVH) — same traced and filtered frames, same ray count, on the Cornell box and on a BVH scene"""
    from real_time_path_tracing_with_spatiotemporal_filtering_amd import scenes
    xyz, idx, _ = cornell
    vx, ti, xf, cam, zfar = scenes.instanced_cornell(xyz, idx, lattice=(2, 2, 2), tess=2)
    lattice = dict(mesh=(vx, ti), instance_xforms=xf, cameraOrigin=cam, z_far=zfar, lightPos=(1.0, float(cam[1]), float(cam[2]) - 8.0))
    keys = [(), ("J",), ("E",)]
    def run(**kw):
        from real_time_path_tracing_with_spatiotemporal_filtering_amd.app import make_app
        app = make_app(200, 150, max_segments=8, iterations=3, **kw)
        out = []
        for k in keys:
            app.drawScene(k)
            out.append(app.backend.ctx.readback(hip_lib.PLANE_PREVIOUS))
        rays = app.backend.ctx.raycount()
        app.backend.close()
        return out, rays
    want = [run(), run(**lattice)]
    monkeypatch.setenv("RTPT_PT_WINDOW", str(window))
    got = [run(), run(**lattice)]
    for (fa, ra), (fb, rb) in zip(want, got):
        assert ra == rb
        for a, b in zip(fa, fb):
            assert np.array_equal(bits(a), bits(b))


@pytest.mark.parametrize("exact", [0, 1])
def test_chain_equals_separate_passes_4k(hip_lib, exact):
    """BASELINE configs[2] at its size: 3840x2160, 4 segments, N = 5, three frames"""
    keys = [(), ("J",), ("A",)]
    a, _ = _frames(hip_lib, 3840, 2160, 5, exact, keys, seg=4)
    b, _ = _frames(hip_lib, 3840, 2160, 5, exact | hip_lib.FLAG_NO_FILTER_FUSION, keys, seg=4)
    for (ia, pa), (ib, pb) in zip(a, b):
        assert np.array_equal(bits(ia), bits(ib))
        assert np.array_equal(pa, pb)
        assert np.isfinite(ia[..., :3]).all() and not ia[..., 3].any()


@pytest.mark.default_policy
def test_chain_is_what_runs_by_default_on_large_frames(hip_lib):
    """the timing hooks name the launches: at 4K N = 5 is two chained pairs + the final pass, not four k_atrous
    launches; a small frame, and RTPT_FLAG_NO_FILTER_FUSION at any size, run one kernel per iteration"""
    from real_time_path_tracing_with_spatiotemporal_filtering_amd.app import make_app
    for (w, h, flags, want) in ((3840, 2160, 0, (6, 0, 3)), (1920, 1080, 0, (6, 0, 3)), (320, 200, 0, (0, 12, 3)),
                                (3840, 2160, hip_lib.FLAG_NO_FILTER_FUSION, (0, 12, 3))):
        app = make_app(w, h, max_segments=1, iterations=5, flags=flags)
        ctx = app.backend.ctx
        ctx.timing_enable(1)
        for _ in range(3):
            app.drawScene()
        tm = ctx.timing_collect()
        assert (tm["k_atrous_chain"][1], tm["k_atrous"][1], tm["k_atrous_final"][1]) == want, (w, h, flags, tm)
        # K0, K1 and K2: one launch unless every pass is launched when called
        fused = not (flags & hip_lib.FLAG_NO_FILTER_FUSION)
        assert (tm["k_gbuffer_pathtrace"][1], tm["k_gbuffer_gradient"][1], tm["k_gbuffer"][1], tm["k_gradient"][1], tm["k_pathtrace"][1]) == \
            ((3, 0, 0, 0, 0) if fused else (0, 0, 3, 3, 3)), tm
        app.backend.close()


def test_gbuffer_gradient_fusion_and_observation(hip_lib):
    """rtpt_gbuffer records; rtpt_temporal_gradient right behind it runs both in one launch; a readback in between (or a
    gradient over rows the G-buffer call did not cover) launches K0 alone first.  Same planes either way."""
    from real_time_path_tracing_with_spatiotemporal_filtering_amd.app import make_app
    outs = []
    for mode in ("fused", "observed", "unfused"):
        app = make_app(200, 90, max_segments=2, iterations=1, flags=hip_lib.FLAG_NO_FILTER_FUSION if mode == "unfused" else 0)
        ctx = app.backend.ctx
        ctx.timing_enable(1)
        planes = []
        for f in range(3):
            app.updateScene(("J",) if f == 1 else (("D",) if f == 2 else ()))
            app.drawVisbilityBuffer()
            if mode == "observed":
                planes.append(ctx.readback(hip_lib.PLANE_VIS_ID))
            app.computeTemporalGradient()
            planes += [ctx.readback(p) for p in (hip_lib.PLANE_VIS_ID, hip_lib.PLANE_WORLDPOS, hip_lib.PLANE_DEPTH, hip_lib.PLANE_GRADIENT)]
            if mode == "observed":
                planes.pop(-5)
            app.drawSceneToImage()
            app.applyTemporalFiltering()
            app.copyImageToSwapChainsCurrentImage()
            app.frameCount += 1
        tm = ctx.timing_collect()
        # fused: K0 + K1 stay recorded until rtpt_raytrace and share its launch; here the planes are read back right behind
        # rtpt_temporal_gradient, so the two run as a launch of their own (k_gbuffer_gradient) and K2 as another
        assert tm["k_gbuffer_gradient"][1] == (3 if mode == "fused" else 0) and tm["k_gradient"][1] == (0 if mode == "fused" else 3)
        assert tm["k_gbuffer_pathtrace"][1] == 0 and tm["k_pathtrace"][1] == 3
        outs.append(planes)
        app.backend.close()
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert np.array_equal(bits(a), bits(b))
    assert (outs[0][-1][..., 0] > 0).any(), "the light / camera moves did produce a gradient"


@pytest.mark.parametrize("flags", [0, 2])        # brute force / forced BVH
@pytest.mark.parametrize("spp", [1, 4])
def test_gbuffer_gradient_raytrace_in_one_launch_equals_three(hip_lib, monkeypatch, flags, spp):
    """rtpt_gbuffer and rtpt_temporal_gradient stay recorded until rtpt_raytrace arrives (main.cpp:1105-1107 is that order) and
    then run in ITS launch, the G-buffer's tiles dispatched behind the tracing tiles (kernels.hip: k_gbuffer_pathtrace): the
    traced image (alpha = the G-buffer depth, written by the G-buffer workgroups while the tracing ones store 12 bytes), every
    G-buffer plane, the gradient and the ray count equal the three separate launches (RTPT_NO_TRACE_FUSION=1), bit for bit —
    whole frames and a row range whose G-buffer rows exceed the traced ones (the strips' exchange mode)."""
    from real_time_path_tracing_with_spatiotemporal_filtering_amd.app import make_app
    planes = (hip_lib.PLANE_IMAGE, hip_lib.PLANE_VIS_ID, hip_lib.PLANE_WORLDPOS, hip_lib.PLANE_DEPTH, hip_lib.PLANE_GRADIENT)
    outs = []
    for fused in (True, False):
        monkeypatch.setenv("RTPT_NO_TRACE_FUSION", "0" if fused else "1")
        res = []
        for (w, h, rank, world) in ((200, 90, 0, 1), (130, 97, 1, 3)):
            app = make_app(w, h, max_segments=3, iterations=1, flags=flags, rank=rank, world=world, mode="exchange", torch_planes=False,
                           samples_per_pixel=spp)
            ctx = app.backend.ctx
            ctx.timing_enable(1)
            for f in range(3):
                app.updateScene(("J",) if f == 1 else (("D",) if f == 2 else ()))
                app.drawVisbilityBuffer()
                app.computeTemporalGradient()
                app.drawSceneToImage()
                res += [ctx.readback(p) for p in planes]
                if world == 1:   # (a lone rank of three has nobody to swap halo rows with: K0-K2 are what is under test)
                    app.applyTemporalFiltering()
                    res.append(ctx.readback(hip_lib.PLANE_IMAGE))   # the filter reads the depth out of the traced image's alpha
                ctx.end_frame()
                app.frameCount += 1
            tm = ctx.timing_collect()
            assert (tm["k_gbuffer_pathtrace"][1], tm["k_pathtrace"][1], tm["k_gbuffer_gradient"][1]) == ((3, 0, 0) if fused else (0, 3, 3)), tm
            res.append(ctx.raycount())
            app.backend.close()
        outs.append(res)
    for a, b in zip(*outs):
        assert np.array_equal(bits(a), bits(b)) if isinstance(a, np.ndarray) else a == b


def test_observation_between_iterations_sees_the_separate_pass_state(hip_lib, oracle, cornell):
    """rtpt_temporal_filter records; a readback between iterations must show exactly what the separate dispatches leave
    (main.cpp:1264-1281: odd k writes filteredImageBuffer, even k writes image), and the frame must still finish right"""
    from real_time_path_tracing_with_spatiotemporal_filtering_amd.app import make_app
    w, h, n = 200, 96, 5
    X = hip_lib.FLAG_EXACT_FILTER
    app = make_app(w, h, max_segments=3, iterations=n, flags=X)
    ref = oracle.OracleApp(w, h, cornell[2], max_segments=3, iterations=n)
    ctx = app.backend.ctx
    for frame in range(2):
        app.updateScene(("J",) if frame else ())
        app.drawVisbilityBuffer()
        app.computeTemporalGradient()
        app.drawSceneToImage()
        traced = ctx.readback(hip_lib.PLANE_IMAGE)
        depth, vis, wp, lut, lutp = (ctx.readback(p) for p in (hip_lib.PLANE_DEPTH, hip_lib.PLANE_VIS_ID, hip_lib.PLANE_WORLDPOS,
                                                             hip_lib.PLANE_LUT, hip_lib.PLANE_LUT_PREV))
        pc = app.pushConstants
        pc.maxWaveletIteration = n
        opc = oracle.PushConstants.from_buffer_copy(bytes(pc))
        oubo = oracle.Ubo.from_buffer_copy(bytes(app.ubo))
        cur = traced
        for k in range(1, n + 1):
            pc.waveletIteration = k
            opc.waveletIteration = k
            app.backend.temporal_filter(pc, app.ubo, 0, h)
            if k < n:
                cur = oracle.atrous(ref.cfg, opc, oubo, cur, depth, vis, lut, lutp, wp, None)
            if k in (2, 3):  # look after an even and an odd iteration: 1+2 were recorded, then 3 alone
                got = ctx.readback(hip_lib.PLANE_IMAGE if k % 2 == 0 else hip_lib.PLANE_FILTERED)
                assert np.array_equal(bits(got[..., :3]), bits(cur[..., :3])), (frame, k)
        final = ctx.readback(hip_lib.PLANE_IMAGE)
        app.copyImageToSwapChainsCurrentImage()
        app.frameCount += 1
        fo = ref.draw_scene(move_light=(-0.1, 0, 0) if frame else None)
        assert np.array_equal(bits(final), bits(fo.image))
    app.backend.close()


@pytest.mark.parametrize("mode", ["redundant"])
def test_chain_on_strips(hip_lib, mode):
    """redundant-halo strips record and chain too (their row ranges shrink by the next stride per iteration, which is
    exactly what a chain needs); exchange-mode strips look at the planes between iterations and run separate passes"""
    from test_parity_gpu import _strips_vs_single
    _strips_vs_single(250, 301, 3, 5, 4, mode, 0, [(), ("E",), ("J",)])
    _strips_vs_single(250, 301, 3, 4, 3, mode, hip_lib.FLAG_EXACT_FILTER, [(), ("Q",), ()])


@pytest.mark.parametrize("ext", [0x20, 0x40, 0x60, 0x70])
@pytest.mark.parametrize("exact", [0, 1])
def test_lds_staged_extension_taps_equal_the_direct_kernel(hip_lib, ext, exact):
    """RTPT_FLAG_EXT_GAUSS5 / _POW2_STRIDE (not reference behaviour): the non-final passes run in the comb kernel's 5x5 /
    wide-stride instances; RTPT_FLAG_DIRECT_FILTER forces the generic direct-load kernel the oracle tests pin.  Same
    arithmetic, so the frames are equal bit for bit — ragged sizes, N = 5 (strides up to 16, halos up to 32 columns)"""
    keys = [(), ("J",), ("D", "E")]
    for (w, h) in ((130, 33), (333, 170), (1000, 800)):
        a, _ = _frames(hip_lib, w, h, 5, ext | exact, keys)
        b, _ = _frames(hip_lib, w, h, 5, ext | exact | hip_lib.FLAG_DIRECT_FILTER, keys)
        for f, ((ia, pa), (ib, pb)) in enumerate(zip(a, b)):
            assert np.array_equal(bits(ia), bits(ib)), (w, h, hex(ext), exact, f)
            assert np.array_equal(pa, pb)


@pytest.mark.parametrize("exact", [0, 1])
def test_three_level_and_final_chains(hip_lib, monkeypatch, exact):
    """the chain kernel's other instances — a chain that ends in the FINAL pass (reprojection + blend in the last level's
    epilogue) and, where the workgroup size admits it (two waves per row and level, so only with G = 2 rows per step), three
    iterations per launch — selected through the tuning knobs rtpt_create reads.  Same bits as one kernel per iteration,
    whatever the grouping."""
    monkeypatch.setenv("RTPT_CHAIN_MAX", "3")
    monkeypatch.setenv("RTPT_CHAIN_FINAL", "1")
    keys = [(), ("J",), ("D", "E"), ()]
    for (w, h) in ((121, 64), (333, 170), (1000, 800)):
        for n in (3, 4, 5):
            a, _ = _frames(hip_lib, w, h, n, exact, keys)
            b, _ = _frames(hip_lib, w, h, n, exact | hip_lib.FLAG_NO_FILTER_FUSION, keys)
            for f, ((ia, pa), (ib, pb)) in enumerate(zip(a, b)):
                assert np.array_equal(bits(ia), bits(ib)), (w, h, n, exact, f)
                assert np.array_equal(pa, pb)
