"""ctypes binding of the C ABI declared in ``include/rtpt.h`` (``librtpt_hip.so``).

This is the only way Python reaches the hot path: there is no CPU fallback.  ``load()`` raises
``RtptLibraryMissing`` when the HIP library has not been built and every call raises
``RtptError`` on a negative status code, carrying ``rtpt_last_error()``.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)
# RTPT_LIB_PATH: a differently-built library for compile-time A/B runs on the GPU box (never set in tests or bench lines)
LIB_PATH = os.environ.get("RTPT_LIB_PATH") or os.path.join(PKG_DIR, "librtpt_hip.so")
HEADER_PATH = os.path.join(REPO_DIR, "include", "rtpt.h")

RTPT_OK, RTPT_E_INVALID, RTPT_E_NOMEM, RTPT_E_DEVICE, RTPT_E_NO_SCENE, RTPT_E_NO_GPU = 0, -1, -2, -3, -4, -5
FLAG_EXACT_FILTER, FLAG_FORCE_BVH, FLAG_DIRECT_FILTER, FLAG_NO_PATH_COMPACTION = 0x1, 0x2, 0x4, 0x8
# extension modes (not reference behaviour, see include/rtpt.h)
FLAG_EXT_ADAPTIVE_ALPHA, FLAG_EXT_GAUSS5, FLAG_EXT_POW2_STRIDE, FLAG_EXT_DISOCCLUSION = 0x10, 0x20, 0x40, 0x80
FLAG_EXT_VARIANCE = 0x100
FLAG_SINGLE_LAUNCH_PATHS = 0x200
FLAG_NO_FILTER_FUSION = 0x400
FLAG_EXT_SVGF_VARIANCE = 0x800
FLAG_EXT_MASK = 0x9F0
DEBUG_HIT_ID, DEBUG_PREV_PIXEL = 0x1, 0x2

# rtpt_plane
(PLANE_IMAGE, PLANE_FILTERED, PLANE_PREVIOUS, PLANE_WORLDPOS, PLANE_GRADIENT, PLANE_DEPTH, PLANE_VIS_ID,
 PLANE_PREV_VIS_ID, PLANE_LUT, PLANE_LUT_PREV, PLANE_PREV_PIXEL, PLANE_RAYCOUNT, PLANE_HIT_ID, PLANE_MOMENTS,
 PLANE_VARIANCE, PLANE_MOMENTS_PREV) = range(16)
# rtpt_kernel_id
(K_GBUFFER, K_LUT, K_GRADIENT, K_PATHTRACE, K_ATROUS, K_ATROUS_FINAL, K_ATROUS_CHAIN, K_ATROUS_CHAIN_FINAL,
 K_GBUFFER_GRADIENT, K_PRESENT, K_GBUFFER_PATHTRACE, K_COUNT) = range(12)
KERNEL_NAMES = ["k_gbuffer", "k_lut", "k_gradient", "k_pathtrace", "k_atrous", "k_atrous_final", "k_atrous_chain",
                "k_atrous_chain_final", "k_gbuffer_gradient", "k_present", "k_gbuffer_pathtrace"]


class RtptLibraryMissing(RuntimeError):
    pass


class RtptError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"rtpt error {code}: {msg}")
        self.code = code


class PushConstants(C.Structure):
    """rtpt_push_constants == PushConstants, main.cpp:35-49 (112 bytes)."""
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
    """rtpt_ubo == UniformBufferObject, main.cpp:82-90 (384 bytes, column-major)."""
    _fields_ = [(n, C.c_float * 16) for n in ("model", "view", "proj", "modelPrev", "viewPrev", "projPrev")]


class Material(C.Structure):
    """rtpt_material: .mtl Kd / Ke"""
    _fields_ = [("albedo", C.c_float * 3), ("emission", C.c_float * 3)]


class Config(C.Structure):
    """rtpt_config."""
    _fields_ = [
        ("struct_size", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32),
        ("row_begin", C.c_uint32), ("row_end", C.c_uint32),
        ("max_segments", C.c_uint32), ("samples_per_pixel", C.c_uint32),
        ("sigma_n", C.c_int32), ("sigma_z", C.c_float), ("sigma_l", C.c_float), ("alpha", C.c_float),
        ("light_radius", C.c_float), ("light_intensity", C.c_float), ("first_hit_light_divisor", C.c_float),
        ("fov_slope", C.c_float), ("pixel_jitter", C.c_float), ("ray_offset", C.c_float), ("ray_tmax", C.c_float),
        ("flags", C.c_uint32), ("device", C.c_int32),
    ]


assert C.sizeof(PushConstants) == 112 and C.sizeof(Ubo) == 384

# every symbol include/rtpt.h declares (tests check the header and this list agree)
SYMBOLS = [
    "rtpt_config_default", "rtpt_create", "rtpt_destroy", "rtpt_resize", "rtpt_last_error", "rtpt_set_stream", "rtpt_bind_plane",
    "rtpt_plane_ptr", "rtpt_plane_bytes", "rtpt_set_external_history", "rtpt_stream_wait", "rtpt_scene_upload", "rtpt_gbuffer", "rtpt_temporal_gradient",
    "rtpt_raytrace", "rtpt_temporal_filter", "rtpt_end_frame", "rtpt_sync", "rtpt_readback", "rtpt_set_plane",
    "rtpt_reset_counters", "rtpt_set_count_rows", "rtpt_enable_debug", "rtpt_timing_enable", "rtpt_timing_collect", "rtpt_kernel_name",
    "rtpt_selftest_math", "rtpt_selftest_exhaustive", "rtpt_selftest_div", "rtpt_selftest_trace", "rtpt_util_look_at", "rtpt_util_perspective", "rtpt_util_load_obj", "rtpt_util_bvh_check",
    "rtpt_scene_set_materials", "rtpt_util_load_obj_materials", "rtpt_util_bvh_refit_check", "rtpt_set_external_guides",
    "rtpt_present", "rtpt_debug_bvh_check", "rtpt_present_target",
]

_lib = None


def load() -> C.CDLL:
    """dlopen librtpt_hip.so; fails loudly when the HIP extension is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RtptLibraryMissing(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the hot path.")
    lib = C.CDLL(LIB_PATH)
    vp, u32, sz = C.c_void_p, C.c_uint32, C.c_size_t
    lib.rtpt_last_error.restype = C.c_char_p
    lib.rtpt_last_error.argtypes = [vp]
    lib.rtpt_kernel_name.restype = C.c_char_p
    lib.rtpt_kernel_name.argtypes = [C.c_int]
    sigs = {
        "rtpt_config_default": [C.POINTER(Config), u32, u32],
        "rtpt_create": [C.POINTER(Config), C.POINTER(vp)],
        "rtpt_destroy": [vp],
        "rtpt_resize": [vp, u32, u32, u32, u32],
        "rtpt_set_stream": [vp, vp],
        "rtpt_bind_plane": [vp, C.c_int, vp, sz],
        "rtpt_plane_ptr": [vp, C.c_int, C.POINTER(vp)],
        "rtpt_plane_bytes": [vp, C.c_int, C.POINTER(sz)],
        "rtpt_set_external_history": [vp, vp, u32, u32],
        "rtpt_stream_wait": [vp, vp],
        "rtpt_set_external_guides": [vp, vp, vp, u32, u32],
        "rtpt_scene_upload": [vp, vp, u32, vp, u32, vp, u32],
        "rtpt_gbuffer": [vp, C.POINTER(Ubo), u32, u32],
        "rtpt_temporal_gradient": [vp, C.POINTER(PushConstants), u32, u32],
        "rtpt_raytrace": [vp, C.POINTER(PushConstants), u32, u32],
        "rtpt_temporal_filter": [vp, C.POINTER(PushConstants), C.POINTER(Ubo), u32, u32],
        "rtpt_end_frame": [vp],
        "rtpt_present": [vp, vp, u32, u32],
        "rtpt_present_target": [vp, vp, u32, u32],
        "rtpt_sync": [vp],
        "rtpt_readback": [vp, C.c_int, vp, sz],
        "rtpt_set_plane": [vp, C.c_int, vp, sz],
        "rtpt_reset_counters": [vp],
        "rtpt_enable_debug": [vp, u32],
        "rtpt_set_count_rows": [vp, u32, u32],
        "rtpt_timing_enable": [vp, C.c_int],
        "rtpt_timing_collect": [vp, C.POINTER(C.c_double * K_COUNT), C.POINTER(u32 * K_COUNT)],
        "rtpt_selftest_math": [vp, C.c_int, vp, vp, sz],
        "rtpt_selftest_exhaustive": [vp, C.c_int, vp, vp],
        "rtpt_selftest_div": [vp, C.c_int, C.c_uint32, C.c_uint32, vp, vp],
        "rtpt_selftest_trace": [vp, vp, sz, vp, vp],
        "rtpt_util_load_obj": [C.c_char_p, vp, C.POINTER(u32), vp, C.POINTER(u32)],
        "rtpt_util_bvh_check": [vp, u32, C.POINTER(C.c_uint64 * 8)],
        "rtpt_scene_set_materials": [vp, vp, u32, vp, u32],
        "rtpt_util_bvh_refit_check": [vp, vp, u32, C.POINTER(C.c_uint64 * 8)],
        "rtpt_debug_bvh_check": [vp, C.POINTER(C.c_uint64 * 8)],
        "rtpt_util_load_obj_materials": [C.c_char_p, vp, C.POINTER(u32), vp, C.POINTER(u32)],
    }
    for name, args in sigs.items():
        fn = getattr(lib, name)
        fn.restype = C.c_int
        fn.argtypes = args
    lib.rtpt_util_look_at.restype = None
    lib.rtpt_util_look_at.argtypes = [vp, vp, vp, vp]
    lib.rtpt_util_perspective.restype = None
    lib.rtpt_util_perspective.argtypes = [C.c_float, C.c_float, C.c_float, C.c_float, vp]
    _lib = lib
    return lib


def _check(rc: int) -> None:
    if rc != 0:
        msg = load().rtpt_last_error(None)
        raise RtptError(rc, msg.decode() if msg else "")


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def config_default(width: int, height: int) -> Config:
    cfg = Config()
    _check(load().rtpt_config_default(C.byref(cfg), width, height))
    return cfg


def look_at(eye, center, up) -> np.ndarray:
    out = np.zeros(16, np.float32)
    e, c, u = (np.ascontiguousarray(v, np.float32) for v in (eye, center, up))
    load().rtpt_util_look_at(_ptr(e), _ptr(c), _ptr(u), _ptr(out))
    return out


def perspective(fovy, aspect, z_near, z_far) -> np.ndarray:
    out = np.zeros(16, np.float32)
    load().rtpt_util_perspective(fovy, aspect, z_near, z_far, _ptr(out))
    return out


def load_obj(path: str):
    """(xyz[n,3] f32, idx[t,3] u32) — `v`/`f` records, fan triangulation (main.cpp:416-428, D5)."""
    nv, nt = C.c_uint32(), C.c_uint32()
    _check(load().rtpt_util_load_obj(path.encode(), None, C.byref(nv), None, C.byref(nt)))
    xyz = np.zeros((nv.value, 3), np.float32)
    idx = np.zeros((nt.value, 3), np.uint32)
    _check(load().rtpt_util_load_obj(path.encode(), _ptr(xyz), C.byref(nv), _ptr(idx), C.byref(nt)))
    return xyz, idx


def load_obj_materials(path: str):
    """(tri_material[t] u32, materials[m, 6] f32 = Kd, Ke) of an OBJ's `mtllib`/`usemtl`; (None, None) when the OBJ names
    no readable library (the reference's Cornell box: its .mtl is missing upstream)"""
    nt, nm = C.c_uint32(), C.c_uint32()
    _check(load().rtpt_util_load_obj_materials(path.encode(), None, C.byref(nt), None, C.byref(nm)))
    if nm.value == 0:
        return None, None
    tri = np.zeros(nt.value, np.uint32)
    mats = np.zeros((nm.value, 6), np.float32)
    _check(load().rtpt_util_load_obj_materials(path.encode(), _ptr(tri), C.byref(nt), _ptr(mats), C.byref(nm)))
    return tri, mats


_PLANE_DTYPE = {
    PLANE_IMAGE: (np.float32, 4), PLANE_FILTERED: (np.float32, 4), PLANE_PREVIOUS: (np.float32, 4),
    PLANE_WORLDPOS: (np.float32, 4), PLANE_GRADIENT: (np.float32, 4), PLANE_DEPTH: (np.float32, 1),
    PLANE_VIS_ID: (np.uint32, 1), PLANE_PREV_VIS_ID: (np.uint32, 1), PLANE_PREV_PIXEL: (np.int32, 2),
    PLANE_HIT_ID: (np.uint32, 1), PLANE_MOMENTS: (np.float32, 4), PLANE_VARIANCE: (np.float32, 1),
    PLANE_MOMENTS_PREV: (np.float32, 4),
}


class Context:
    """RAII wrapper of rtpt_ctx."""

    def __init__(self, cfg: Config):
        self._lib = load()
        self.cfg = cfg
        h = C.c_void_p()
        _check(self._lib.rtpt_create(C.byref(cfg), C.byref(h)))
        self._h = h
        self.n_tris = 0

    # -- lifetime
    def close(self):
        if getattr(self, "_h", None):
            self._lib.rtpt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def rows(self) -> int:
        return self.cfg.row_end - self.cfg.row_begin

    def resize(self, width: int, height: int, row_begin: int = 0, row_end: int = 0):
        _check(self._lib.rtpt_resize(self._h, width, height, row_begin, row_end))
        self.cfg.width, self.cfg.height = width, height
        self.cfg.row_begin, self.cfg.row_end = (row_begin, row_end) if (row_begin or row_end) else (0, height)

    # -- configuration
    def set_stream(self, stream_handle: int | None):
        _check(self._lib.rtpt_set_stream(self._h, C.c_void_p(stream_handle or 0)))

    def bind_plane(self, which: int, device_ptr: int | None, nbytes: int = 0):
        _check(self._lib.rtpt_bind_plane(self._h, which, C.c_void_p(device_ptr or 0), nbytes))

    def plane_ptr(self, which: int) -> int:
        p = C.c_void_p()
        _check(self._lib.rtpt_plane_ptr(self._h, which, C.byref(p)))
        return p.value or 0

    def plane_bytes(self, which: int) -> int:
        n = C.c_size_t()
        _check(self._lib.rtpt_plane_bytes(self._h, which, C.byref(n)))
        return n.value

    def set_external_history(self, device_ptr: int | None, row_begin: int = 0, row_end: int = 0):
        _check(self._lib.rtpt_set_external_history(self._h, C.c_void_p(device_ptr or 0), row_begin, row_end))

    def set_external_guides(self, prev_vis_ptr: int | None, moments_ptr: int | None, row_begin: int = 0, row_end: int = 0):
        """previous frame's id (u32) and moment (float4) rows [row_begin,row_end) gathered across strips; None, None:
        the context's own planes"""
        _check(self._lib.rtpt_set_external_guides(self._h, C.c_void_p(prev_vis_ptr or 0), C.c_void_p(moments_ptr or 0),
                                                  row_begin, row_end))

    def plane_dtype(self, which: int):
        return _PLANE_DTYPE[which]

    def stream_wait(self, other: "Context"):
        """work submitted to this context from now on starts after everything submitted to `other` so far"""
        _check(self._lib.rtpt_stream_wait(self._h, other._h))

    def enable_debug(self, mask: int):
        _check(self._lib.rtpt_enable_debug(self._h, mask))

    # -- scene
    def scene_upload(self, xyz: np.ndarray, idx: np.ndarray, instance_xforms: np.ndarray | None = None):
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        idx = np.ascontiguousarray(idx, np.uint32).reshape(-1, 3)
        ni = 0
        xf = None
        if instance_xforms is not None:
            xf = np.ascontiguousarray(instance_xforms, np.float32).reshape(-1, 12)
            ni = len(xf)
        _check(self._lib.rtpt_scene_upload(self._h, _ptr(xyz), len(xyz), _ptr(idx), len(idx), _ptr(xf), ni))
        self.n_tris = len(idx) * max(ni, 1)

    def set_materials(self, tri_material: np.ndarray | None, materials: np.ndarray | None):
        """per-triangle material indices + (Kd, Ke) rows; None returns to the reference's normal-keyed colours"""
        if tri_material is None or materials is None:
            _check(self._lib.rtpt_scene_set_materials(self._h, None, 0, None, 0))
            return
        tri = np.ascontiguousarray(tri_material, np.uint32)
        mats = np.ascontiguousarray(materials, np.float32).reshape(-1, 6)
        _check(self._lib.rtpt_scene_set_materials(self._h, _ptr(tri), len(tri), _ptr(mats), len(mats)))

    # -- passes
    def gbuffer(self, ubo: Ubo, y0=0, y1=0):
        _check(self._lib.rtpt_gbuffer(self._h, C.byref(ubo), y0, y1))

    def temporal_gradient(self, pc: PushConstants, y0=0, y1=0):
        _check(self._lib.rtpt_temporal_gradient(self._h, C.byref(pc), y0, y1))

    def raytrace(self, pc: PushConstants, y0=0, y1=0):
        _check(self._lib.rtpt_raytrace(self._h, C.byref(pc), y0, y1))

    def temporal_filter(self, pc: PushConstants, ubo: Ubo | None, y0=0, y1=0):
        _check(self._lib.rtpt_temporal_filter(self._h, C.byref(pc), C.byref(ubo) if ubo is not None else None, y0, y1))

    def end_frame(self):
        _check(self._lib.rtpt_end_frame(self._h))

    def present_target(self, dst_device_ptr: int | None, y0=0, y1=0):
        """name the swapchain rows of the frame being built: the final filter pass writes them too (fused blit)"""
        _check(self._lib.rtpt_present_target(self._h, C.c_void_p(dst_device_ptr) if dst_device_ptr else None, y0, y1))

    def present(self, dst_device_ptr: int, y0=0, y1=0):
        """main.cpp:1338-1361: rows [y0,y1) of the finished frame -> B8G8R8A8_UNORM at the device address"""
        _check(self._lib.rtpt_present(self._h, C.c_void_p(dst_device_ptr), y0, y1))

    def sync(self):
        _check(self._lib.rtpt_sync(self._h))

    def debug_bvh_check(self):
        """the acceleration structure as it stands on the device (after an upload or a device-side refit), checked on the host"""
        st = (C.c_uint64 * 8)()
        _check(self._lib.rtpt_debug_bvh_check(self._h, C.byref(st)))
        return dict(zip(("nodes", "leaves", "depth", "largest_leaf", "bad_refs_to_triangles", "boxes_not_containing", "boxes_beyond_scene",
                         "dangling"), (int(v) for v in st)))

    # -- data movement
    def readback(self, which: int) -> np.ndarray:
        if which in (PLANE_LUT, PLANE_LUT_PREV):
            out = np.zeros((self.n_tris + 1, 12), np.float32)
        elif which == PLANE_RAYCOUNT:
            out = np.zeros(1, np.uint64)
        else:
            dt, ch = _PLANE_DTYPE[which]
            shape = (self.rows, self.cfg.width) + ((ch,) if ch > 1 else ())
            out = np.zeros(shape, dt)
        _check(self._lib.rtpt_readback(self._h, which, _ptr(out), out.nbytes))
        return out

    def set_plane(self, which: int, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        _check(self._lib.rtpt_set_plane(self._h, which, _ptr(arr), arr.nbytes))

    def reset_counters(self):
        _check(self._lib.rtpt_reset_counters(self._h))

    def set_count_rows(self, y0: int, y1: int):
        _check(self._lib.rtpt_set_count_rows(self._h, y0, y1))

    def raycount(self) -> int:
        return int(self.readback(PLANE_RAYCOUNT)[0])

    # -- timing
    def timing_enable(self, period):
        """0/False off; n: bracket the kernels of every n-th frame with HIP events (True == 1)."""
        _check(self._lib.rtpt_timing_enable(self._h, int(period)))

    def timing_collect(self):
        ms = (C.c_double * K_COUNT)()
        n = (C.c_uint32 * K_COUNT)()
        _check(self._lib.rtpt_timing_collect(self._h, C.byref(ms), C.byref(n)))
        return {KERNEL_NAMES[i]: (ms[i], n[i]) for i in range(K_COUNT)}

    # -- self tests
    def selftest_math(self, op: int, x: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(x, np.float32)
        out = np.empty_like(x)
        _check(self._lib.rtpt_selftest_math(self._h, op, _ptr(x), _ptr(out), x.size))
        return out

    def selftest_exhaustive(self, op: int):
        """(mismatches, first offending patterns) of exact sqrt (op 3) / reciprocal (op 4) over all 2^32 binary32 patterns"""
        n = C.c_uint64(0)
        first = np.zeros(4, np.uint32)
        _check(self._lib.rtpt_selftest_exhaustive(self._h, op, C.byref(n), _ptr(first)))
        return int(n.value), first

    def selftest_div(self, mode: int, first_pass: int, n_passes: int):
        """(mismatches, bits of one offending (a, b)) of exact::div_ against the compiler's division; see rtpt.h"""
        n = C.c_uint64(0)
        first = np.zeros(2, np.uint32)
        _check(self._lib.rtpt_selftest_div(self._h, mode, first_pass, n_passes, C.byref(n), _ptr(first)))
        return int(n.value), first

    def selftest_trace(self, rays: np.ndarray):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        ids = np.zeros(len(rays), np.uint32)
        ts = np.zeros(len(rays), np.float32)
        _check(self._lib.rtpt_selftest_trace(self._h, _ptr(rays), len(rays), _ptr(ids), _ptr(ts)))
        return ids, ts


def bvh_check(tris: np.ndarray, built_for: np.ndarray | None = None) -> dict:
    """host-only self check of the BVH builder + device node packing (needs no GPU): see rtpt_util_bvh_check;
    with `built_for` the tree is built over those triangles and REFIT to `tris` (rtpt_util_bvh_refit_check)"""
    tris = np.ascontiguousarray(tris, np.float32).reshape(-1, 9)
    st = (C.c_uint64 * 8)()
    if built_for is not None:
        built_for = np.ascontiguousarray(built_for, np.float32).reshape(-1, 9)
        assert built_for.shape == tris.shape
        _check(load().rtpt_util_bvh_refit_check(_ptr(built_for), _ptr(tris), len(tris), C.byref(st)))
    else:
        _check(load().rtpt_util_bvh_check(_ptr(tris), len(tris), C.byref(st)))
    keys = ("nodes", "leaves", "max_depth", "largest_leaf", "bad_triangle_refs", "loose_boxes", "loose_device_boxes", "bad_child_refs")
    return dict(zip(keys, (int(v) for v in st)))
