"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes/numpy front-end of ``oracle/liboracle.so`` (the plain-C restatement of the reference's
compute chain, ``rtpt_oracle.c``).  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; the product package never does.
Parity is *unpinned* by reference fixtures (the reference ships none): the
oracle is pinned by the known-answer vectors derived from the reference sources
(``tests/test_oracle_kat.py``).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RTPT_ORACLE_LIB: a sanitizer build of the same source (oracle/Makefile targets asan / tsan), CPU only
_LIB_PATH = os.environ.get("RTPT_ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")


class OracleConfig(C.Structure):
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32),
        ("max_segments", C.c_uint32), ("samples_per_pixel", C.c_uint32),
        ("sigma_n", C.c_int32), ("sigma_z", C.c_float), ("sigma_l", C.c_float),
        ("alpha", C.c_float), ("light_radius", C.c_float), ("light_intensity", C.c_float),
        ("first_hit_light_divisor", C.c_float), ("fov_slope", C.c_float),
        ("pixel_jitter", C.c_float), ("ray_offset", C.c_float), ("ray_tmax", C.c_float),
        ("ext_flags", C.c_uint32),
    ]


EXT_ADAPTIVE_ALPHA, EXT_GAUSS5, EXT_POW2_STRIDE, EXT_DISOCCLUSION, EXT_VARIANCE = 0x10, 0x20, 0x40, 0x80, 0x100
EXT_SVGF_VARIANCE = 0x800


class PushConstants(C.Structure):
    """main.cpp:35-49 — 112 bytes."""
    _fields_ = [
        ("sample_batch", C.c_uint32), ("frameNumber", C.c_uint32), ("_pad0", C.c_uint32 * 2),
        ("cameraPos", C.c_float * 3), ("_pad1", C.c_float),
        ("lightPos", C.c_float * 3), ("_pad2", C.c_float),
        ("lightPosPrev", C.c_float * 3), ("_pad3", C.c_float),
        ("currentCameraColor", C.c_float * 3), ("_pad4", C.c_float),
        ("previousCameraColor", C.c_float * 3),
        ("waveletIteration", C.c_int32), ("maxWaveletIteration", C.c_int32),
        ("_pad5", C.c_uint32 * 3),
    ]


class Ubo(C.Structure):
    """main.cpp:82-90 — six column-major mat4."""
    _fields_ = [(n, C.c_float * 16) for n in ("model", "view", "proj", "modelPrev", "viewPrev", "projPrev")]


assert C.sizeof(PushConstants) == 112 and C.sizeof(Ubo) == 384

_lib = None


def build(force: bool = False) -> str:
    """compile liboracle.so with gcc (building the checker is not using it)."""
    if os.environ.get("RTPT_ORACLE_LIB"):
        return _LIB_PATH
    srcs = [os.path.join(_HERE, f) for f in ("rtpt_oracle.c", "rtpt_oracle.h", "det_math.h", "Makefile")]
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.oracle_log.restype = C.c_float
        _lib.oracle_log.argtypes = [C.c_float]
        for n in ("oracle_sin2pi", "oracle_cos2pi", "oracle_exp", "oracle_sqrt", "oracle_rcp"):
            getattr(_lib, n).restype = C.c_float
            getattr(_lib, n).argtypes = [C.c_float]
        _lib.oracle_powi.restype = C.c_float
        _lib.oracle_powi.argtypes = [C.c_float, C.c_int]
        _lib.oracle_rng_seed.restype = C.c_uint32
        _lib.oracle_rng_seed.argtypes = [C.c_uint32] * 4
        _lib.oracle_rng_step.restype = C.c_uint32
        _lib.oracle_rng_step.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_float)]
        _lib.oracle_closest_hit.restype = C.c_uint32
        _lib.oracle_get_threads.restype = C.c_int
    return _lib


def _p(a, t=None):
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)


def config_default(width: int, height: int) -> OracleConfig:
    c = OracleConfig()
    lib().oracle_config_default(C.byref(c), C.c_uint32(width), C.c_uint32(height))
    return c


def set_threads(n: int) -> None:
    lib().oracle_set_threads(C.c_int(n))


def set_columns(x0: int = 0, x1: int = 0) -> None:
    """restrict the per-pixel passes to columns [x0, x1) — timing samples only; set_columns() restores every column"""
    lib().oracle_set_columns(C.c_int(x0), C.c_int(x1))


def math_array(op: int, x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    lib().oracle_math_array(C.c_int(op), _p(x), _p(out), C.c_uint64(x.size))
    return out


def rng_seed(px, py, frame, batch=0) -> int:
    return int(lib().oracle_rng_seed(px, py, frame, batch))


def rng_steps(state: int, n: int):
    """returns (words, floats, final_state)."""
    s = C.c_uint32(state)
    f = C.c_float()
    words, floats = [], []
    for _ in range(n):
        w = lib().oracle_rng_step(C.byref(s), C.byref(f))
        words.append(int(w))
        floats.append(np.float32(f.value))
    return words, floats, int(s.value)


def look_at(eye, center, up) -> np.ndarray:
    out = np.zeros(16, np.float32)
    e, c, u = (np.asarray(v, np.float32) for v in (eye, center, up))
    lib().oracle_look_at(_p(e), _p(c), _p(u), _p(out))
    return out


def perspective(fovy, aspect, zn, zf) -> np.ndarray:
    out = np.zeros(16, np.float32)
    lib().oracle_perspective(C.c_float(fovy), C.c_float(aspect), C.c_float(zn), C.c_float(zf), _p(out))
    return out


def load_obj(path: str):
    nv, nt = C.c_uint32(), C.c_uint32()
    if lib().oracle_load_obj(path.encode(), None, C.byref(nv), None, C.byref(nt)) != 0:
        raise FileNotFoundError(path)
    xyz = np.zeros((nv.value, 3), np.float32)
    idx = np.zeros((nt.value, 3), np.uint32)
    lib().oracle_load_obj(path.encode(), _p(xyz), C.byref(nv), _p(idx), C.byref(nt))
    return xyz, idx


def flatten(xyz: np.ndarray, idx: np.ndarray, xforms: np.ndarray | None = None) -> np.ndarray:
    xyz = np.ascontiguousarray(xyz, np.float32)
    idx = np.ascontiguousarray(idx, np.uint32)
    ni = 1 if xforms is None else len(xforms)
    if xforms is not None:
        xforms = np.ascontiguousarray(xforms, np.float32).reshape(ni, 12)
    tris = np.zeros((ni * len(idx), 9), np.float32)
    lib().oracle_flatten(_p(xyz), _p(idx), C.c_uint32(len(idx)), _p(xforms),
                         C.c_uint32(0 if xforms is None else ni), _p(tris))
    return tris


def trace_rays(tris: np.ndarray, rays: np.ndarray, tmax: float = 10000.0):
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
    ids = np.zeros(len(rays), np.uint32)
    ts = np.zeros(len(rays), np.float32)
    lib().oracle_trace_rays(_p(tris), C.c_uint32(len(tris)), _p(rays), C.c_uint64(len(rays)),
                            C.c_float(tmax), _p(ids), _p(ts))
    return ids, ts


def lut(tris: np.ndarray, model: np.ndarray) -> np.ndarray:
    out = np.zeros((len(tris) + 1, 12), np.float32)
    model = np.ascontiguousarray(model, np.float32)
    lib().oracle_lut(_p(tris), C.c_uint32(len(tris)), _p(model), _p(out))
    return out


def gbuffer(cfg: OracleConfig, tris, ubo: Ubo, y0=0, y1=None):
    W, H = cfg.width, cfg.height
    y1 = H if y1 is None else y1
    vis = np.zeros((H, W), np.uint32)
    wp = np.zeros((H, W, 4), np.float32)
    depth = np.zeros((H, W), np.float32)
    lib().oracle_gbuffer(C.byref(cfg), _p(tris), C.c_uint32(len(tris)), C.byref(ubo),
                         C.c_uint32(y0), C.c_uint32(y1), _p(vis), _p(wp), _p(depth))
    return vis, wp, depth


def temporal_gradient(cfg, pc: PushConstants, vis, worldpos, lut_, lut_prev, y0=0, y1=None):
    W, H = cfg.width, cfg.height
    y1 = H if y1 is None else y1
    grad = np.zeros((H, W, 4), np.float32)
    lib().oracle_temporal_gradient(C.byref(cfg), C.byref(pc), _p(vis), _p(worldpos), _p(lut_), _p(lut_prev),
                                   C.c_uint32(y0), C.c_uint32(y1), _p(grad))
    return grad


def material_records(tri_material, materials) -> np.ndarray:
    """n_base x 8 floats (Kd.rgb, 0, Ke.rgb, emissive flag) from per-triangle indices + (Kd, Ke) rows"""
    mats = np.asarray(materials, np.float32).reshape(-1, 6)[np.asarray(tri_material, np.int64)]
    rec = np.zeros((len(mats), 8), np.float32)
    rec[:, 0:3] = mats[:, 0:3]
    rec[:, 4:7] = mats[:, 3:6]
    rec[:, 7] = (mats[:, 3:6] != 0).any(axis=1)
    return rec


def raytrace(cfg, pc: PushConstants, tris, y0=0, y1=None, want_hit_id=True, tri_mat=None):
    """tri_mat: material_records(...) of the base mesh (extension), None = the reference's normal-keyed colours"""
    W, H = cfg.width, cfg.height
    y1 = H if y1 is None else y1
    img = np.zeros((H, W, 4), np.float32)
    rc = C.c_uint64(0)
    hid = np.zeros((H, W), np.uint32) if want_hit_id else None
    if tri_mat is not None:
        tri_mat = np.ascontiguousarray(tri_mat, np.float32)
    lib().oracle_raytrace_mat(C.byref(cfg), C.byref(pc), _p(tris), C.c_uint32(len(tris)), _p(tri_mat),
                              C.c_uint32(0 if tri_mat is None else len(tri_mat)),
                              C.c_uint32(y0), C.c_uint32(y1), _p(img), C.byref(rc), _p(hid))
    return img, int(rc.value), hid


def moments(cfg, pc: PushConstants, ubo: Ubo, traced, vis, worldpos, lut_prev, prev_vis, moments_prev, y0=0, y1=None):
    """extension EXT_VARIANCE: (moments[H,W,4] = m1, m2, n, var; variance[H,W])"""
    W, H = cfg.width, cfg.height
    y1 = H if y1 is None else y1
    if prev_vis is None:
        prev_vis = np.zeros((H, W), np.uint32)
    if moments_prev is None:
        moments_prev = np.zeros((H, W, 4), np.float32)
    mo = np.zeros((H, W, 4), np.float32)
    var = np.zeros((H, W), np.float32)
    lib().oracle_moments(C.byref(cfg), C.byref(pc), C.byref(ubo), _p(traced), _p(vis), _p(worldpos), _p(lut_prev),
                         _p(prev_vis), _p(moments_prev), C.c_uint32(y0), C.c_uint32(y1), _p(mo), _p(var))
    return mo, var


def var_prefilter(cfg, var: np.ndarray) -> np.ndarray:
    """EXT_SVGF_VARIANCE: the 3x3 Gaussian of the variance plane that scales an iteration's luminance weight"""
    var = np.ascontiguousarray(var, np.float32)
    out = np.zeros_like(var)
    lib().oracle_var_prefilter(C.byref(cfg), _p(var), C.c_uint32(0), C.c_uint32(cfg.height), _p(out))
    return out


def atrous(cfg, pc: PushConstants, ubo: Ubo, img_in, depth, vis, lut_, lut_prev, worldpos, history,
           y0=0, y1=None, want_prev_pixel=False, gradient=None, prev_vis=None, var_in=None):
    """with var_in (EXT_VARIANCE) the filtered variance is appended to the result"""
    W, H = cfg.width, cfg.height
    y1 = H if y1 is None else y1
    out = np.zeros((H, W, 4), np.float32)
    pp = np.zeros((H, W, 2), np.int32) if want_prev_pixel else None
    if history is None:
        history = np.zeros((H, W, 4), np.float32)
    if prev_vis is None:
        prev_vis = np.zeros((H, W), np.uint32)
    if gradient is None:
        gradient = np.zeros((H, W, 4), np.float32)
    var_out = np.zeros((H, W), np.float32) if var_in is not None else None
    lib().oracle_atrous_var(C.byref(cfg), C.byref(pc), C.byref(ubo), _p(img_in), _p(depth), _p(vis), _p(lut_),
                            _p(lut_prev), _p(worldpos), _p(history), _p(gradient), _p(prev_vis), _p(var_in),
                            C.c_uint32(y0), C.c_uint32(y1), _p(out), _p(pp), _p(var_out))
    res = (out, pp) if want_prev_pixel else (out,)
    if var_in is not None:
        res = res + (var_out,)
    return res if len(res) > 1 else res[0]


# ------------------------------------------------------------------------------------------
# frame driver: the reference's drawScene() order (main.cpp:1090-1113) on the oracle passes
# ------------------------------------------------------------------------------------------

@dataclass
class FrameOut:
    vis: np.ndarray
    worldpos: np.ndarray
    depth: np.ndarray
    gradient: np.ndarray
    traced: np.ndarray
    hit_id: np.ndarray
    image: np.ndarray       # final blended image (== history handed to the next frame)
    prev_pixel: np.ndarray
    rays: int
    lut: np.ndarray


class OracleApp:
    """CPU statement of PathTracingApplication's per-frame state machine (main.cpp:1090-1185,
    :1255-1306, :1361-1372, :1463-1475).  Camera/light moves are scripted (keyboard input is out
    of scope)."""

    def __init__(self, width, height, tris, max_segments=32, iterations=9,
                 camera=(-0.001, 1.0, 6.0), light=(1.0, 1.0, -0.4), light_color=(0.5, 0.5, 0.5), z_near=0.1, z_far=10.0,
                 ext_flags=0, tri_mat=None):
        self.cfg = config_default(width, height)
        self.cfg.max_segments = max_segments
        self.cfg.ext_flags = ext_flags
        self.iterations = iterations          # main.cpp:55
        self.tris = np.ascontiguousarray(tris, np.float32)   # as uploaded: the model matrix poses them per frame
        self.model = np.eye(4, dtype=np.float32).T.ravel().copy()  # column-major, main.cpp:1469 (identity there)
        self.tri_mat = tri_mat
        self.camera = np.array(camera, np.float32)   # main.cpp:65
        self.light = np.array(light, np.float32)     # main.cpp:70
        self.light_color = np.array(light_color, np.float32)  # main.cpp:72
        self.camera_moved = False
        self.z_near, self.z_far = z_near, z_far
        self.frame = 0
        self.pc = PushConstants()
        self.ubo = Ubo()
        # uploadBuffers main.cpp:481-489 — initial matrices look at (0,1,0)
        self.ubo.model[:] = np.eye(4, dtype=np.float32).ravel()
        self.ubo.view[:] = look_at(self.camera, (0.0, 1.0, 0.0), (0.0, 1.0, 0.0))
        proj = perspective(np.float32(0.20) * 2, np.float32(width) / np.float32(height), self.z_near, self.z_far)
        proj[5] *= -1
        self.ubo.proj[:] = proj
        self.ubo.modelPrev[:] = self.ubo.model[:]
        self.ubo.viewPrev[:] = self.ubo.view[:]
        self.ubo.projPrev[:] = self.ubo.proj[:]
        # initializeSceneConstants main.cpp:661-666 (lightPosPrev is a zero-initialised global)
        self.pc.currentCameraColor[:] = self.light_color
        self.pc.lightPos[:] = self.light
        self.pc.lightPosPrev[:] = (0.0, 0.0, 0.0)
        self.history = None
        self.lut_prev = None
        self.prev_vis = None

    def update_ubo(self):  # main.cpp:1463-1475
        u = self.ubo
        u.modelPrev[:] = u.model[:]
        u.viewPrev[:] = u.view[:]
        u.projPrev[:] = u.proj[:]
        u.model[:] = self.model
        c = self.camera
        u.view[:] = look_at(c, (c[0], c[1], np.float32(c[2] - np.float32(6.0))), (0.0, 1.0, 0.0))
        proj = perspective(np.float32(0.20) * 2, np.float32(self.cfg.width) / np.float32(self.cfg.height), self.z_near, self.z_far)
        proj[5] *= -1
        u.proj[:] = proj

    def update_scene(self, move_camera=None, move_light=None):  # main.cpp:1115-1185
        if move_camera is not None:
            self.camera = (self.camera + np.asarray(move_camera, np.float32)).astype(np.float32)
            self.camera_moved = True
        if move_light is not None:
            self.light = (self.light + np.asarray(move_light, np.float32)).astype(np.float32)
        pc = self.pc
        pc.frameNumber = self.frame
        pc.previousCameraColor[:] = pc.currentCameraColor[:]
        pc.currentCameraColor[:] = self.light_color
        pc.lightPosPrev[:] = pc.lightPos[:]
        pc.lightPos[:] = self.light
        self.update_ubo()
        if self.camera_moved or self.frame == 0:
            pc.cameraPos[:] = self.camera
            self.camera_moved = False

    def draw_scene(self, move_camera=None, move_light=None) -> FrameOut:  # main.cpp:1090-1113
        cfg = self.cfg
        self.update_scene(move_camera, move_light)
        model = np.array(self.ubo.model[:], np.float32)
        lut_ = lut(self.tris, model)
        if self.lut_prev is None:
            self.lut_prev = lut_.copy()  # D3
        posed = self.tris
        if not np.array_equal(model, np.eye(4, dtype=np.float32).ravel()):
            # world triangle = model * uploaded triangle (visibility.vert.glsl:24): exactly the LUT's vertices
            posed = np.ascontiguousarray(lut_[1:].reshape(-1, 3, 4)[:, :, :3].reshape(-1, 9))
        tris_up, self.tris = self.tris, posed
        try:
            return self._draw_posed(lut_)
        finally:
            self.tris = tris_up

    def _draw_posed(self, lut_) -> FrameOut:
        cfg = self.cfg
        vis, wp, depth = gbuffer(cfg, self.tris, self.ubo)
        grad = temporal_gradient(cfg, self.pc, vis, wp, lut_, self.lut_prev)
        self.pc.sample_batch = 0  # main.cpp:1237
        traced, rays, hid = raytrace(cfg, self.pc, self.tris, tri_mat=self.tri_mat)
        self.pc.maxWaveletIteration = self.iterations  # main.cpp:1258
        cur = traced
        pp = None
        var = None
        if cfg.ext_flags & EXT_VARIANCE:
            self.moments, var = moments(cfg, self.pc, self.ubo, traced, vis, wp, self.lut_prev, self.prev_vis,
                                        getattr(self, "moments", None))
            self.variance0 = var
        for k in range(1, self.iterations + 1):
            self.pc.waveletIteration = k
            res = atrous(cfg, self.pc, self.ubo, cur, depth, vis, lut_, self.lut_prev, wp, self.history,
                         want_prev_pixel=(k == self.iterations), gradient=grad, prev_vis=self.prev_vis, var_in=var)
            if var is not None:
                var = res[-1]
                res = res[:-1]
                res = res if len(res) > 1 else res[0]
            if k == self.iterations:
                cur, pp = res
            else:
                cur = res
        self.variance = var
        # history hand-over main.cpp:1361-1372
        self.history = cur
        self.lut_prev = lut_
        self.prev_vis = vis  # main.cpp:1367
        self.frame += 1
        return FrameOut(vis, wp, depth, grad, traced, hid, cur, pp, rays, lut_)


def present_bgra8(image: np.ndarray) -> np.ndarray:
    """main.cpp:1338-1361: the blit RGBA32F `image` -> B8G8R8A8_UNORM swapchain image, as the build defines the
    conversion (include/rtpt.h rtpt_present): clamp to [0,1], x*255 + 0.5 (binary32 multiply, then add) truncated,
    NaN -> 0.  Returns [H, W, 4] uint8 in memory order B,G,R,A."""
    c = np.asarray(image, np.float32)
    c = np.where(np.isnan(c), np.float32(0), c)
    c = np.minimum(np.maximum(c, np.float32(0)), np.float32(1))
    q = (c * np.float32(255.0) + np.float32(0.5)).astype(np.uint8)
    return np.ascontiguousarray(q[..., [2, 1, 0, 3]])
