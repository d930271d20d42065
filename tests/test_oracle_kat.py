"""Known-answer tests pinning the oracle to the reference *sources* (SURVEY.md 4).  The reference
ships no tests, fixtures or golden images, so these vectors — derived from the GLSL/C++ text and
the constants folded into the .spv files — are the only reference-side anchors that exist."""
import ctypes as C

import numpy as np


def test_pcg_words_seed0(oracle):
    # raytrace.comp.glsl:71-78
    words, floats, _ = oracle.rng_steps(0, 4)
    assert words == [0x108EF29B, 0x00033628, 0xDCCC2102, 0xD3BB3506]
    np.testing.assert_array_equal(np.array(floats, np.float32),
                                  np.array([0.06468121, 4.9004331e-05, 0.86248976, 0.8270753], np.float32))


def test_rng_float_is_word_times_2_pow_minus_32(oracle):
    # :77 — 4294967295.0f rounds to 2^32 in fp32; range [0,1] inclusive
    for state in (0, 1, 12345, 0xFFFFFFFF, 0x9E3779B9):
        w, f, _ = oracle.rng_steps(state, 3)
        for wi, fi in zip(w, f):
            assert fi == np.float32(np.float32(wi) * np.float32(2.0 ** -32))
            assert 0.0 <= fi <= 1.0


def test_seed_hash(oracle):
    # raytrace.comp.glsl:297, uint32 wrap-around
    kat = {(1, 0, 0): 0xC2B2AE3D, (0, 1, 0): 0x27D4EB2F, (128, 128, 0): 0x43CCB600, (128, 128, 1): 0x559AD1B1,
           (999, 799, 0): 0x18CFF7BC, (3839, 2159, 7): 0x3E7283F3}
    for (x, y, f), want in kat.items():
        assert oracle.rng_seed(x, y, f, 0) == want


def test_camera_slope_constant(oracle):
    # common.h:16 FOV 0.20 -> tan folded to 0.20271003 in raytrace.comp.glsl.spv
    cfg = oracle.config_default(256, 256)
    assert np.float32(cfg.fov_slope) == np.float32(np.tan(0.20))
    assert np.float32(cfg.fov_slope) == np.float32(0.20271003)


def test_struct_layouts(oracle):
    # main.cpp:35-49 / :82-90 offsets verified against the .spv
    P = oracle.PushConstants
    assert [getattr(P, n).offset for n in ("sample_batch", "frameNumber", "cameraPos", "lightPos", "lightPosPrev",
                                            "currentCameraColor", "previousCameraColor", "waveletIteration",
                                            "maxWaveletIteration")] == [0, 4, 16, 32, 48, 64, 80, 92, 96]
    assert C.sizeof(P) == 112
    assert C.sizeof(oracle.Ubo) == 384
    assert [getattr(oracle.Ubo, n).offset for n in ("model", "view", "proj", "modelPrev", "viewPrev", "projPrev")] == \
        [0, 64, 128, 192, 256, 320]


def test_scene_facts(cornell):
    # scenes/CornellBox-Original-Merged.obj: 64 v, 16 quads -> 32 triangles, AABB
    xyz, idx, tris = cornell
    assert xyz.shape == (64, 3) and idx.shape == (32, 3)
    assert len(np.unique(xyz, axis=0)) == 28
    np.testing.assert_allclose(xyz.min(0), [-1.02, 0.0, -1.04], atol=1e-6)
    np.testing.assert_allclose(xyz.max(0), [1.0, 1.99, 0.99], atol=1e-6)
    # D5 fan triangulation in file order: triangle t = 2*quad + {0,1}
    assert idx[0].tolist() == [0, 1, 2] and idx[1].tolist() == [0, 2, 3]
    assert idx[6].tolist() == [12, 13, 14] and idx[7].tolist() == [12, 14, 15]


def _normal(tri):
    n = np.cross(tri[3:6] - tri[0:3], tri[6:9] - tri[0:3])
    return n / np.linalg.norm(n)


def test_albedo_keyed_by_normal(cornell):
    # raytrace.comp.glsl:155-163: quad 3 (verts 13-16, x = +1 wall, normal -x) green,
    # quad 4 (verts 17-20, x ~ -1 wall, normal +x) red, rest grey
    _, _, tris = cornell
    nx = np.array([_normal(t)[0] for t in tris])
    red = np.where(nx > 0.99)[0].tolist()
    green = np.where(-nx > 0.99)[0].tolist()
    assert red == [8, 9] and green == [6, 7]


def test_direct_light_pixel(oracle, cornell):
    # :226-231 with main.cpp:70-72: a primary ray meeting the sphere returns 0.5*30/5 = 3 exactly
    _, _, tris = cornell
    app = oracle.OracleApp(256, 256, tris, max_segments=2, iterations=1)
    fo = app.draw_scene()
    lit = (fo.traced[..., :3] == 3.0).all(-1)
    assert lit.sum() > 100
    # the light centre (1,1,-0.4) projects to the right of the image centre at mid height
    ys, xs = np.nonzero(lit)
    assert 120 < ys.mean() < 136 and xs.mean() > 128
    assert (fo.traced[..., 3] == 0).all()  # :343 alpha 0


def test_static_scene_gradient_is_zero(oracle, cornell):
    # temporalGradient.comp.glsl:163-167: unchanged light => lambda == 0 on id != 0, id == 0 => 0.
    # Frame 0 has lightPosPrev = (0,0,0) (main.cpp:71 zero-initialised global), frame 1 is static.
    _, _, tris = cornell
    app = oracle.OracleApp(96, 64, tris, max_segments=1, iterations=1)
    f0 = app.draw_scene()
    assert f0.gradient.max() > 0.0
    f1 = app.draw_scene()
    assert f1.gradient.max() < 1e-3
    assert (f1.gradient[f1.vis == 0] == 0).all()


def test_filter_fixed_point(oracle, cornell):
    # temporalFiltering.comp.glsl:150: a constant colour image is a fixed point of num/den
    _, _, tris = cornell
    app = oracle.OracleApp(64, 48, tris, max_segments=1, iterations=1)
    fo = app.draw_scene()
    const = np.zeros((48, 64, 4), np.float32)
    const[..., :3] = (0.25, 0.5, 0.75)
    pc = oracle.PushConstants()
    pc.waveletIteration, pc.maxWaveletIteration = 2, 5
    out = oracle.atrous(app.cfg, pc, app.ubo, const, fo.depth, fo.vis, fo.lut, fo.lut, fo.worldpos, None)
    np.testing.assert_allclose(out[..., :3], const[..., :3], rtol=5e-7)  # a few ulp of the 9-term sums
    assert (out[..., 3] == 0).all()


def test_frame0_final_is_filtered_and_reprojection_identity(oracle, cornell):
    # :251-259 frame 0 => final == filtered; :178-189/:238 static camera => prev pixel == pixel
    _, _, tris = cornell
    W, H = 96, 64
    app = oracle.OracleApp(W, H, tris, max_segments=2, iterations=3)
    f0 = app.draw_scene()
    pc = oracle.PushConstants()
    pc.frameNumber = 0
    pc.maxWaveletIteration = 3
    cur = f0.traced
    for k in (1, 2):
        pc.waveletIteration = k
        cur = oracle.atrous(app.cfg, pc, app.ubo, cur, f0.depth, f0.vis, f0.lut, f0.lut, f0.worldpos, None)
    pc.waveletIteration = 3
    pc.maxWaveletIteration = 99  # same taps, not final
    plain = oracle.atrous(app.cfg, pc, app.ubo, cur, f0.depth, f0.vis, f0.lut, f0.lut, f0.worldpos, None)
    np.testing.assert_array_equal(plain, f0.image)
    f1 = app.draw_scene()  # static camera (frame-0 viewPrev differs: it looks at (0,1,0), main.cpp:482)
    xs, ys = np.meshgrid(np.arange(W), np.arange(H))
    np.testing.assert_array_equal(f1.prev_pixel[..., 0], xs)
    np.testing.assert_array_equal(f1.prev_pixel[..., 1], ys)


def test_grid_counts():
    # main.cpp:1216-1217 with common.h:18-19 (reference launch shape; informational)
    for (w, h), want in {(256, 256): (16, 32), (1000, 800): (63, 100), (1920, 1080): (120, 135),
                         (3840, 2160): (240, 270)}.items():
        assert ((w + 15) // 16, (h + 7) // 8) == want


def test_even_final_iteration_does_not_blend(oracle, cornell):
    # main.cpp:55 "must be an odd number": with an even N the blend lands in a buffer nothing reads
    _, _, tris = cornell
    app = oracle.OracleApp(48, 32, tris, max_segments=1, iterations=2)
    app.draw_scene()
    f1 = app.draw_scene()
    pc = oracle.PushConstants()
    pc.frameNumber, pc.waveletIteration, pc.maxWaveletIteration = 1, 1, 9
    a = oracle.atrous(app.cfg, pc, app.ubo, f1.traced, f1.depth, f1.vis, f1.lut, f1.lut, f1.worldpos, None)
    pc.waveletIteration = 2
    b = oracle.atrous(app.cfg, pc, app.ubo, a, f1.depth, f1.vis, f1.lut, f1.lut, f1.worldpos, None)
    np.testing.assert_array_equal(b, f1.image)


# ------------------------------------------------------------------------------ extension modes (oracle side)
def _ext_inputs(oracle, w=24, h=20, seed=3):
    rng = np.random.default_rng(seed)
    cfg = oracle.config_default(w, h)
    img = rng.random((h, w, 4), dtype=np.float32)
    img[..., 3] = 0
    depth = np.full((h, w), 0.5, np.float32)
    vis = np.ones((h, w), np.uint32)
    lut = np.zeros((2, 12), np.float32)
    lut[1, 0:3] = (0, 0, 1)          # vertex a
    lut[1, 4:7] = (1, 0, 1)          # vertex b
    lut[1, 8:11] = (0, 1, 1)         # vertex c
    return cfg, img, depth, vis, lut


def test_ext_flags_default_off_and_gauss5_flat_field(oracle):
    """a constant image stays constant under any tap set; with equal guides the 5x5 table reduces to the
    normalised gaussianKernel2D (sum 273, temporalFiltering.comp.glsl:93-99)"""
    cfg, img, depth, vis, lut = _ext_inputs(oracle)
    assert cfg.ext_flags == 0
    pc, ubo = oracle.PushConstants(), oracle.Ubo()
    pc.waveletIteration, pc.maxWaveletIteration = 1, 3
    wp = np.zeros_like(img)
    flat = np.full_like(img, 0.25)
    flat[..., 3] = 0
    for ext in (oracle.EXT_GAUSS5, oracle.EXT_POW2_STRIDE, oracle.EXT_GAUSS5 | oracle.EXT_POW2_STRIDE):
        cfg.ext_flags = ext
        out = oracle.atrous(cfg, pc, ubo, flat, depth, vis, lut, lut, wp, flat)
        assert np.allclose(out[..., :3], 0.25, rtol=1e-6)
    # one bright pixel, huge sigma_l => pure spatial kernel: the response IS the table
    cfg.ext_flags = oracle.EXT_GAUSS5
    cfg.sigma_l = 1e30
    imp = np.zeros_like(img)
    imp[10, 12, :3] = 273.0
    out = oracle.atrous(cfg, pc, ubo, imp, depth, vis, lut, lut, wp, imp)
    table = np.array([[1, 4, 7, 4, 1], [4, 16, 26, 16, 4], [7, 26, 41, 26, 7], [4, 16, 26, 16, 4], [1, 4, 7, 4, 1]], np.float32)
    assert np.allclose(out[8:13, 10:15, 0], table, rtol=1e-5)
    # stride 2^(k-1): iteration 3 taps land 4 pixels apart
    cfg.ext_flags = oracle.EXT_POW2_STRIDE
    pc.waveletIteration = 3
    pc.maxWaveletIteration = 4
    out = oracle.atrous(cfg, pc, ubo, imp, depth, vis, lut, lut, wp, imp)
    ys, xs = np.nonzero(out[..., 0])
    assert sorted(set(ys)) == [6, 10, 14] and sorted(set(xs)) == [8, 12, 16]


def test_ext_adaptive_alpha_and_disocclusion(oracle):
    cfg, img, depth, vis, lut = _ext_inputs(oracle)
    w, h = cfg.width, cfg.height
    pc, ubo = oracle.PushConstants(), oracle.Ubo()
    pc.waveletIteration = pc.maxWaveletIteration = 1
    pc.frameNumber = 3
    # identity reprojection: background pixels (id 0) keep their own coordinate (:213-217)
    vis0 = np.zeros((h, w), np.uint32)
    wp = np.zeros_like(img)
    hist = np.full_like(img, 2.0)
    base = oracle.atrous(cfg, pc, ubo, img, depth, vis0, lut, lut, wp, hist)
    filt_only = oracle.atrous(cfg, oracle.PushConstants.from_buffer_copy(bytes(pc)), ubo, img, depth, vis0, lut, lut, wp, hist)
    assert base.tobytes() == filt_only.tobytes()
    pc0 = oracle.PushConstants.from_buffer_copy(bytes(pc))
    pc0.frameNumber = 0
    filtered = oracle.atrous(cfg, pc0, ubo, img, depth, vis0, lut, lut, wp, hist)   # frame 0: no blend (:258)
    cfg.ext_flags = oracle.EXT_ADAPTIVE_ALPHA
    g = np.zeros_like(img)
    out0 = oracle.atrous(cfg, pc, ubo, img, depth, vis0, lut, lut, wp, hist, gradient=g)
    assert out0.tobytes() == base.tobytes(), "gradient 0 keeps the constant alpha"
    g[..., 0] = 1.0
    out1 = oracle.atrous(cfg, pc, ubo, img, depth, vis0, lut, lut, wp, hist, gradient=g)
    assert out1.tobytes() == filtered.tobytes(), "gradient 1 drops the history"
    g[..., 0] = 0.5
    outh = oracle.atrous(cfg, pc, ubo, img, depth, vis0, lut, lut, wp, hist, gradient=g)
    a = np.float32(0.5) * np.float32(0.3) + np.float32(0.5)
    assert np.allclose(outh[..., :3], filtered[..., :3] * a + 2.0 * (1 - a), rtol=1e-6)
    # disocclusion: history only where the previous id plane agrees
    cfg.ext_flags = oracle.EXT_DISOCCLUSION
    pv = np.zeros((h, w), np.uint32)
    pv[:, w // 2:] = 7
    outd = oracle.atrous(cfg, pc, ubo, img, depth, vis0, lut, lut, wp, hist, prev_vis=pv)
    assert outd[:, :w // 2].tobytes() == base[:, :w // 2].tobytes()
    assert outd[:, w // 2:].tobytes() == filtered[:, w // 2:].tobytes()


def test_ext_variance_moments_definitions(oracle):
    """EXT_VARIANCE: closed-form checks of oracle_moments and of the variance-guided weight"""
    cfg, img, depth, vis, lut = _ext_inputs(oracle)
    cfg.ext_flags = oracle.EXT_VARIANCE
    w, h = cfg.width, cfg.height
    pc, ubo = oracle.PushConstants(), oracle.Ubo()
    vis0 = np.zeros((h, w), np.uint32)            # background: the reprojection is the identity (:216)
    wp = np.zeros_like(img)
    lum = (0.2126 * img[..., 0] + 0.7152 * img[..., 1] + 0.0722 * img[..., 2]).astype(np.float64)
    # frame 0: no history
    pc.frameNumber = 0
    mo, var = oracle.moments(cfg, pc, ubo, img, vis0, wp, lut, None, None)
    assert np.allclose(mo[..., 0], lum, rtol=1e-6) and np.allclose(mo[..., 1], lum * lum, rtol=1e-6)
    assert (mo[..., 2] == 1).all() and np.array_equal(mo[..., 3], var) and var.max() < 1e-6
    # frame 1 with the same ids: a = max(0.3, 1/2) = 0.5, n = 2, var = (m2 - m1^2) * 4/2
    pc.frameNumber = 1
    img2 = np.random.default_rng(5).random(img.shape, dtype=np.float32)
    lum2 = (0.2126 * img2[..., 0] + 0.7152 * img2[..., 1] + 0.0722 * img2[..., 2]).astype(np.float64)
    mo2, var2 = oracle.moments(cfg, pc, ubo, img2, vis0, wp, lut, vis0, mo)
    m1 = 0.5 * lum + 0.5 * lum2
    m2 = 0.5 * lum * lum + 0.5 * lum2 * lum2
    assert (mo2[..., 2] == 2).all()
    assert np.allclose(mo2[..., 0], m1, rtol=1e-5) and np.allclose(var2, np.maximum(0, m2 - m1 * m1) * 2.0, rtol=1e-3, atol=2e-6)
    # a changed id at the reprojected pixel resets the history
    pv = vis0.copy()
    pv[:, : w // 2] = 9
    mo3, var3 = oracle.moments(cfg, pc, ubo, img2, vis0, wp, lut, pv, mo)
    assert (mo3[:, : w // 2, 2] == 1).all() and (mo3[:, w // 2:, 2] == 2).all()
    # the filter: zero variance => the colour term only passes equal luminances; huge variance => it passes everything
    pc.waveletIteration, pc.maxWaveletIteration = 1, 3
    flat_var = np.zeros((h, w), np.float32)
    out0, v0 = oracle.atrous(cfg, pc, ubo, img2, depth, vis0, lut, lut, wp, img2, var_in=flat_var)
    same = np.isclose(out0[..., :3], img2[..., :3], rtol=1e-3, atol=1e-3).all(-1)
    assert same.mean() > 0.95, "sigma_l * 0 + 1e-4: neighbours are rejected unless their luminance is within ~1e-4"
    big = np.full((h, w), 1e12, np.float32)
    out1, v1 = oracle.atrous(cfg, pc, ubo, img2, depth, vis0, lut, lut, wp, img2, var_in=big)
    box = sum(np.roll(np.roll(img2[..., :3], i, 0), j, 1) for i in (-1, 0, 1) for j in (-1, 0, 1)) / 9
    assert np.allclose(out1[2:-2, 2:-2, :3], box[2:-2, 2:-2], rtol=1e-4)
    assert np.allclose(v1[2:-2, 2:-2], 1e12 / 9, rtol=1e-4), "equal weights: var' = sum(h^2 var) / (sum h)^2 = var / 9"
