"""A second, independent transcription of the reference's shaders — vectorised numpy in float64, written from the GLSL
text with libm's pow/exp/sqrt — against the C oracle (contract arithmetic in binary32).  The reference cannot be run
here, so this does not pin the oracle to the reference's *output*; it does pin it against transcription
mistakes: two differently written restatements of the same GLSL must agree to binary32 accuracy, and integer
observables must agree except where a value sits on a rounding boundary."""
import numpy as np
import pytest

W, H = 96, 72
rng0 = np.random.default_rng(11)


def _norm(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def _area(a, b, c):  # getAreaOfTriangle
    return 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=-1)


def _bary(p, a, b, c):  # getBarycentricCoordinates, temporalGradient.comp.glsl:56-68
    t = _area(a, b, c)
    return np.stack([_area(p, b, c) / t, _area(a, p, c) / t, _area(a, b, p) / t], -1)


def _frame(oracle, cornell, w=W, h=H, frames=2, move=(0.1, 0.05, 0)):
    """a small oracle run with a camera move on the last frame, returning everything the restatements need"""
    app = oracle.OracleApp(w, h, cornell[2], max_segments=3, iterations=3)
    for f in range(frames):
        last = f == frames - 1
        fo = app.draw_scene(move_camera=move if last else None, move_light=(-0.1, 0, 0) if last else None)
    return app, fo


# ---------------------------------------------------------------------------------------- closest hit
def test_closest_hit_against_textbook_moller_trumbore(oracle, cornell):
    tris = cornell[2].astype(np.float64).reshape(-1, 3, 3)
    n = 4000
    o = rng0.uniform([-0.9, 0.1, -0.9], [0.9, 1.9, 3.0], (n, 3))
    d = _norm(rng0.normal(size=(n, 3)))
    ids, ts = oracle.trace_rays(cornell[2], np.concatenate([o, d], 1).astype(np.float32))
    o32, d32 = o.astype(np.float32).astype(np.float64), d.astype(np.float32).astype(np.float64)
    v0, e1, e2 = tris[:, 0], tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0]
    # Moller & Trumbore 1997, per ray x triangle
    p = np.cross(d32[:, None, :], e2[None])
    det = (e1[None] * p).sum(-1)
    tv = o32[:, None, :] - v0[None]
    u = (tv * p).sum(-1) / det
    q = np.cross(tv, e1[None])
    v = (d32[:, None, :] * q).sum(-1) / det
    t = (e2[None] * q).sum(-1) / det
    hit = (np.abs(det) > 0) & (u >= 0) & (v >= 0) & (u + v <= 1) & (t > 0)
    tt = np.where(hit, t, np.inf)
    best = tt.argmin(1)
    tbest = tt.min(1)
    second = np.partition(tt, 1, axis=1)[:, 1]
    edge = np.minimum.reduce([u, v, 1 - u - v])[np.arange(n), best]
    hit_any = np.isfinite(tbest)
    gap = np.where(hit_any & np.isfinite(second), second - np.where(hit_any, tbest, 0.0), np.inf)
    clear = hit_any & (gap > 1e-4) & (edge > 1e-5)   # away from ties and edges
    assert hit_any.mean() > 0.3 and clear.sum() > 0.95 * hit_any.sum()
    assert np.array_equal(ids[clear], best[clear] + 1)
    assert np.allclose(ts[clear], tbest[clear], rtol=2e-5)
    missed = ~np.isfinite(tbest)
    assert (ids[missed] == 0).all()


# ---------------------------------------------------------------------------------------- K1
def test_temporal_gradient_against_float64_phong(oracle, cornell):
    app, fo = _frame(oracle, cornell)
    pc = app.pc
    cam, lp, lpp = (np.array(x[:], np.float64) for x in (pc.cameraPos, pc.lightPos, pc.lightPosPrev))
    lc, lcp = np.array(pc.currentCameraColor[:], np.float64), np.array(pc.previousCameraColor[:], np.float64)
    vis = fo.vis
    lut = fo.lut.reshape(-1, 3, 4)[..., :3].astype(np.float64)
    wp = fo.worldpos[..., :3].astype(np.float64)
    tri = lut[vis]                                   # [H, W, 3 vertices, 3]
    a, b, c = tri[..., 0, :], tri[..., 1, :], tri[..., 2, :]
    with np.errstate(all="ignore"):
        nrm = _norm(np.cross(b - a, c - a))
        bc = _bary(wp, a, b, c)
        wpp = bc[..., 0:1] * a + bc[..., 1:2] * b + bc[..., 2:3] * c   # LUTprev == LUT: the model never moves

        def phong(p, n, light, col):                 # temporalGradient.comp.glsl:70-101
            ld = _norm(light - p)
            diff = np.maximum((n * ld).sum(-1, keepdims=True), 0)
            vd = _norm(cam - p)
            inc = -ld
            rd = inc - 2 * (n * inc).sum(-1, keepdims=True) * n
            spec = np.maximum((vd * rd).sum(-1, keepdims=True), 0) ** 128
            return (0.1 * col + diff * col + 0.5 * spec * col) * 0.7

        cur, prv = phong(wp, nrm, lp, lc), phong(wpp, nrm, lpp, lcp)
        lam = np.minimum(1.0, np.linalg.norm(cur - prv, axis=-1) / np.maximum(np.linalg.norm(cur, axis=-1), np.linalg.norm(prv, axis=-1)))
    lam = np.where(vis == 0, 0.0, lam)
    got = fo.gradient[..., 0].astype(np.float64)
    ok = np.isfinite(lam)
    assert ok.mean() > 0.99 and (vis > 0).mean() > 0.2 and lam[ok].max() > 0.05, "the light moved: the gradient must be visible"
    # lambda is a ratio of differences of nearby colours: absolute agreement at binary32 level
    assert np.abs(got[ok] - lam[ok]).max() < 2e-4
    assert np.array_equal(fo.gradient[..., 0], fo.gradient[..., 1]) and not fo.gradient[..., 3].any()


# ---------------------------------------------------------------------------------------- K3
def _normals(vis, lut):
    tri = lut[vis]
    with np.errstate(all="ignore"):
        n = _norm(np.cross(tri[..., 1, :] - tri[..., 0, :], tri[..., 2, :] - tri[..., 0, :]))
    n[vis == 0] = (0.0, 0.0, 1.0)                    # temporalFiltering.comp.glsl:83
    return n


@pytest.mark.parametrize("k", [1, 2, 5])
def test_atrous_pass_against_float64_restatement(oracle, cornell, k):
    app, fo = _frame(oracle, cornell)
    cfg = app.cfg
    pc = oracle.PushConstants.from_buffer_copy(bytes(app.pc))
    pc.waveletIteration, pc.maxWaveletIteration = k, 9
    img = fo.traced
    out = oracle.atrous(cfg, pc, app.ubo, img, fo.depth, fo.vis, fo.lut, fo.lut, fo.worldpos, fo.traced)
    c = img[..., :3].astype(np.float64)
    dep = fo.depth.astype(np.float64)
    nrm = _normals(fo.vis, fo.lut.reshape(-1, 3, 4)[..., :3].astype(np.float64))
    h, w = dep.shape
    ys, xs = np.mgrid[0:h, 0:w]
    num = np.zeros_like(c)
    den = np.zeros((h, w))
    for i in (-1, 0, 1):                             # :132-147
        for j in (-1, 0, 1):
            qx = np.clip(xs + i * k, 0, w - 1)
            qy = np.clip(ys + j * k, 0, h - 1)
            cq, dq, nq = c[qy, qx], dep[qy, qx], nrm[qy, qx]
            wgt = (np.maximum(0.0, (nrm * nq).sum(-1)) ** 128.0) * np.exp(-np.abs(dep - dq) / 1.0) * np.exp(-np.linalg.norm(c - cq, axis=-1) / 4.0)
            num += (1 / 9.0) * wgt[..., None] * cq
            den += (1 / 9.0) * wgt
    want = num / den[..., None]
    err = np.linalg.norm(out[..., :3] - want, axis=-1) / (1 + np.linalg.norm(want, axis=-1))
    assert np.isfinite(want).all() and err.max() < 2e-5, err.max()


def test_reprojection_and_blend_against_float64_restatement(oracle, cornell):
    app, fo = _frame(oracle, cornell, frames=3)
    w, h = app.cfg.width, app.cfg.height
    vis = fo.vis
    lutp = fo.lut.reshape(-1, 3, 4)[..., :3].astype(np.float64)   # LUTprev == LUT
    wp = fo.worldpos[..., :3].astype(np.float64)
    tri = lutp[vis]
    a, b, c = tri[..., 0, :], tri[..., 1, :], tri[..., 2, :]
    with np.errstate(all="ignore"):
        bc = _bary(wp, a, b, c)
        wpp = bc[..., 0:1] * a + bc[..., 1:2] * b + bc[..., 2:3] * c
        V = np.array(app.ubo.viewPrev[:], np.float64).reshape(4, 4).T   # column-major -> matrix
        P = np.array(app.ubo.projPrev[:], np.float64).reshape(4, 4).T
        clip = np.concatenate([wpp, np.ones(wpp.shape[:-1] + (1,))], -1) @ (P @ V).T
        ndc = clip[..., :2] / clip[..., 3:4]
        scr = (ndc * 0.5 + 0.5) * np.array([w, h], np.float64)            # :186
    ys, xs = np.mgrid[0:h, 0:w]
    want = np.where((vis < 1)[..., None], np.stack([xs, ys], -1), np.trunc(np.nan_to_num(scr)).astype(np.int64))
    got = fo.prev_pixel.astype(np.int64)
    frac = np.abs(scr - np.round(scr)).min(-1)
    clear = (vis < 1) | (np.isfinite(scr).all(-1) & (frac > 1e-3))
    assert clear.mean() > 0.95 and np.array_equal(got[clear], want[clear])
    assert (got[vis > 0] != np.stack([xs, ys], -1)[vis > 0]).any(-1).mean() > 0.5, "the camera moved: pixels must reproject elsewhere"


# ---------------------------------------------------------------------------------------- K2
def test_first_segment_against_float64_restatement(oracle, cornell):
    """raytrace.comp.glsl with the loop bound 1: seed hash -> PCG -> Box-Muller jitter -> camera ray -> light test ->
    closest hit -> albedo / sky.  Integers (the RNG) are reproduced exactly, the float part in float64; the colour of a
    one-segment path is one of a few discrete values (or the sky gradient), so the images must agree pixel for pixel
    except where the jittered ray grazes a silhouette or the light's rim."""
    w, h, frame = 120, 90, 7
    cfg = oracle.config_default(w, h)
    cfg.max_segments = 1
    pc = oracle.PushConstants()
    pc.frameNumber, pc.sample_batch = frame, 0
    cam = np.array([-0.001, 1.0, 6.0])
    light = np.array([1.0, 1.0, -0.4])
    pc.cameraPos[:] = cam
    pc.lightPos[:] = light
    pc.currentCameraColor[:] = (0.5, 0.5, 0.5)
    img, rays, hit = oracle.raytrace(cfg, pc, cornell[2], 0, h)
    assert rays == w * h

    M = 0xFFFFFFFF
    ys, xs = np.mgrid[0:h, 0:w].astype(np.uint64)
    state = ((xs * 3266489917 + ys * 668265263) & M) ^ ((frame * 374761393) & M) ^ 0     # :297

    def step(s):                                                                           # :71-78
        s = (s * 747796405 + 1) & M
        word = ((((s >> ((s >> 28) + 4)) ^ s) & M) * 277803737) & M
        word = ((word >> 22) ^ word) & M
        return s, np.float32(word.astype(np.float64) / 4294967295.0).astype(np.float64)    # float(word) / 4294967295.0f

    state, u1 = step(state)
    state, u2 = step(state)
    u1 = np.maximum(1e-38, u1)
    r = np.sqrt(-2.0 * np.log(u1))
    theta = 2 * 3.14159265 * u2
    cx = xs + 0.5 + 0.375 * r * np.cos(theta)                                              # :314
    cy = ys + 0.5 + 0.375 * r * np.sin(theta)
    slope = np.tan(0.20)                                                                   # common.h FOV
    d = np.stack([slope * (2 * cx - w) / h, slope * -(2 * cy - h) / h, -np.ones_like(cx)], -1)
    d = d / np.linalg.norm(d, axis=-1, keepdims=True)
    # checkRayLightIntersection, :168-198 (tested before the triangle hit, :226)
    oc = cam - light
    b = 2 * (d @ oc)
    disc = b * b - 4 * (oc @ oc - 0.2 * 0.2)
    sq = np.sqrt(np.maximum(disc, 0))
    lit = (disc >= 0) & (((-b - sq) / 2 > 0) | ((-b + sq) / 2 > 0))
    # closest hit, textbook Moller-Trumbore
    tris = cornell[2].astype(np.float64).reshape(-1, 3, 3)
    v0, e1, e2 = tris[:, 0], tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0]
    D = d.reshape(-1, 1, 3)
    p = np.cross(D, e2[None])
    det = (e1[None] * p).sum(-1)
    tv = (cam - v0)[None]
    u = (tv * p).sum(-1) / det
    q = np.cross(tv, e1[None])
    v = (D * q).sum(-1) / det
    t = (e2[None] * q).sum(-1) / det
    ok = (np.abs(det) > 0) & (u >= 0) & (v >= 0) & (u + v <= 1) & (t > 0)
    tt = np.where(ok, t, np.inf)
    best = tt.argmin(1).reshape(h, w)
    miss = ~np.isfinite(tt.min(1)).reshape(h, w)
    nrm = np.cross(e1, e2)
    nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    alb = np.where((nrm[:, 0] > 0.99)[:, None], [1.0, 0, 0], np.where((-nrm[:, 0] > 0.99)[:, None], [0, 1.0, 0], [0.7, 0.7, 0.7]))  # :155-163
    sky = np.where((d[..., 1] > 0)[..., None], 1.0 * (1 - d[..., 1:2]) + np.array([0.25, 0.5, 1.0]) * d[..., 1:2], 0.03)     # :95-107
    want = np.where(lit[..., None], 0.5 * 30 / 5.0, np.where(miss[..., None], sky, alb[best]))
    got = img[..., :3].astype(np.float64)
    same = np.abs(got - want).max(-1) < 1e-5
    assert same.mean() > 0.995, same.mean()
    assert lit.sum() > 20 and miss.sum() > 100 and (~miss & ~lit).sum() > 1000, "light, sky and surfaces must all be in view"
    agree_id = (hit == np.where(miss, 0, best + 1))
    assert agree_id.mean() > 0.995
    assert not img[..., 3].any()                                                            # :343 alpha 0
