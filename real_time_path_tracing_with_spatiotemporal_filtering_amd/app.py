"""Headless Python mirror of the reference's ``PathTracingApplication`` (main.cpp:179-1529): the
same method names, the same per-frame call order and push-constant/UBO update rules, driving the
HIP hot path through the C ABI (``abi.Context``).  Window, swapchain and keyboard are out of
scope; input is a scripted set of pressed keys per frame (the keys of main.cpp:1119-1168).

The frame loop is written against a small *backend* protocol so that the multi-rank strip logic
(``strips.py``) can be exercised on CPU with gloo in the tests; the product backend is
``HipBackend`` below and nothing else ships.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import abi
from .strips import StripPlan, exchange_halo, exchange_history, gather_frame, reprojection_rows

FOV = 0.20                 # common.h:16
SPEED = np.float32(0.1)    # main.cpp:68
DEFAULT_SCENE = os.path.join(abi.PKG_DIR, "scenes", "CornellBox-Original-Merged.obj")


def _null_scope():
    import contextlib
    return contextlib.nullcontext()


class HipBackend:
    """abi.Context + (optionally) torch-owned colour planes so halo rows can be sent with RCCL."""

    def __init__(self, width, height, plan: StripPlan | None = None, max_segments=32, flags=0, device=-1,
                 torch_planes=False, debug_mask=0, samples_per_pixel=1):
        self.plan = plan or StripPlan(height, 1, 0, 1)
        cfg = abi.config_default(width, height)
        cfg.row_begin, cfg.row_end = self.plan.stored
        cfg.max_segments = max_segments
        cfg.samples_per_pixel = samples_per_pixel   # raytrace.comp.glsl:306 (the reference runs 1)
        cfg.flags = flags
        cfg.device = device
        self.ctx = abi.Context(cfg)
        self.width, self.height = width, height
        self._tensors = None
        self.torch_stream = None
        self.ctx.set_count_rows(*self.plan.own)
        if debug_mask:
            self.ctx.enable_debug(debug_mask)
        if torch_planes:
            import torch
            dev = torch.device("cuda", torch.cuda.current_device())
            rows = cfg.row_end - cfg.row_begin
            self._tensors = [torch.zeros((rows, width, 4), dtype=torch.float32, device=dev) for _ in range(3)]
            for role, t in zip((abi.PLANE_IMAGE, abi.PLANE_FILTERED, abi.PLANE_PREVIOUS), self._tensors):
                self.ctx.bind_plane(role, t.data_ptr(), t.numel() * 4)
            # a dedicated non-default stream: the kernels and the RCCL point-to-point ops issued while
            # it is current are ordered with each other (the default stream's handle is 0, which the
            # ABI reads as "use the context's own stream")
            self.torch_stream = torch.cuda.Stream(device=dev)
            self.ctx.set_stream(self.torch_stream.cuda_stream)

    def stream_scope(self):
        """context manager making the stream the kernels run on torch's current stream"""
        if self.torch_stream is None:
            import contextlib
            return contextlib.nullcontext()
        import torch
        return torch.cuda.stream(self.torch_stream)

    # passes ------------------------------------------------------------------------------
    def scene_upload(self, xyz, idx, xforms=None):
        self.ctx.scene_upload(xyz, idx, xforms)

    def gbuffer(self, ubo, y0, y1):
        self.ctx.gbuffer(ubo, y0, y1)

    def temporal_gradient(self, pc, y0, y1):
        self.ctx.temporal_gradient(pc, y0, y1)

    def raytrace(self, pc, y0, y1):
        self.ctx.raytrace(pc, y0, y1)

    def temporal_filter(self, pc, ubo, y0, y1):
        self.ctx.temporal_filter(pc, ubo, y0, y1)

    def end_frame(self):
        self.ctx.end_frame()

    def sync(self):
        self.ctx.sync()

    # halo exchange support ---------------------------------------------------------------
    def color_rows(self, plane: int, y0: int, y1: int):
        """torch view of frame rows [y0,y1) of the buffer currently playing colour role `plane`."""
        if self._tensors is None:
            raise RuntimeError("HipBackend(torch_planes=True) is required for halo exchange")
        ptr = self.ctx.plane_ptr(plane)
        for t in self._tensors:
            if t.data_ptr() == ptr:
                base = self.ctx.cfg.row_begin
                return t[y0 - base:y1 - base]
        raise RuntimeError("colour plane is not one of the bound torch tensors")

    # previous-frame guide planes (ids, moments) of the extension modes, gathered like the history ---------------
    def guide_rows(self, plane: int, y0: int, y1: int):
        """torch view (no copy) of frame rows [y0,y1) of a context-owned per-pixel plane, through the device pointer"""
        import torch
        dt, ch = self.ctx.plane_dtype(plane)
        base = self.ctx.cfg.row_begin
        ptr = self.ctx.plane_ptr(plane) + (y0 - base) * self.width * ch * 4
        shape = (y1 - y0, self.width) + ((ch,) if ch > 1 else ())

        class _Dev:  # the CUDA array interface is how torch adopts foreign device memory (ROCm included)
            __cuda_array_interface__ = {"shape": shape, "typestr": "<f4" if dt == np.float32 else "<i4",
                                        "data": (ptr, False), "version": 2}   # ids travel as int32 bit patterns
        return torch.as_tensor(_Dev(), device=torch.device("cuda", torch.cuda.current_device()))

    def guides_full(self):
        """(ids [H, W] int32, moments [H, W, 4] f32) buffers the other strips' rows are gathered into"""
        if getattr(self, "_guides_full", None) is None:
            import torch
            dev = torch.device("cuda", torch.cuda.current_device())
            self._guides_full = (torch.zeros((self.height, self.width), dtype=torch.int32, device=dev),
                                 torch.zeros((self.height, self.width, 4), dtype=torch.float32, device=dev))
        return self._guides_full

    def use_external_guides(self, on: bool, rows=None, moments=True):
        if on:
            ids, mom = self.guides_full()
            a, b = rows
            self.ctx.set_external_guides(ids.data_ptr() + a * self.width * 4,
                                         mom.data_ptr() + a * self.width * 16 if moments else None, a, b)
        else:
            self.ctx.set_external_guides(None, None)

    # history all-gather support (multi-rank, moving camera) ----------------------------------
    def history_full(self):
        """torch tensor [H, W, 4] that receives every rank's final strip (allocated on first use)"""
        if getattr(self, "_hist_full", None) is None:
            import torch
            dev = torch.device("cuda", torch.cuda.current_device())
            self._hist_full = torch.zeros((self.height, self.width, 4), dtype=torch.float32, device=dev)
        return self._hist_full

    def use_external_history(self, on: bool, rows=None):
        """rows = (a, b): the frame rows of history_full() that hold the previous frame (default: all)"""
        if on:
            t = self.history_full()
            a, b = rows if rows is not None else (0, self.height)
            self.ctx.set_external_history(t.data_ptr() + a * self.width * 16, a, b)
        else:
            self.ctx.set_external_history(None)

    def readback_rows(self, plane: int, y0: int, y1: int) -> np.ndarray:
        base = self.ctx.cfg.row_begin
        return self.ctx.readback(plane)[y0 - base:y1 - base]

    # presenting the frame (main.cpp:1338-1361) ---------------------------------------------------
    on_device = True

    def alloc(self, shape, dtype):
        import torch
        return torch.zeros(shape, dtype=getattr(torch, dtype), device=torch.device("cuda", torch.cuda.current_device()))

    def present_rows(self, image8, y0: int, y1: int):
        """rtpt_present: rows [y0,y1) of the finished frame -> rows [y0,y1) of the [H, W, 4] uint8 swapchain image"""
        self.ctx.present(image8.data_ptr() + y0 * self.width * 4, y0, y1)

    def present_target(self, image8, y0: int, y1: int):
        """rtpt_present_target: the frame's final filter pass writes these rows of the swapchain image itself"""
        self.ctx.present_target(image8.data_ptr() + y0 * self.width * 4, y0, y1)

    def final_rows(self, y0: int, y1: int):
        """torch view of rows of the finished frame (after end_frame: the PREVIOUS plane)"""
        return self.color_rows(abi.PLANE_PREVIOUS, y0, y1)

    # two frames in flight (PipelinedBackend) ------------------------------------------------
    def wait_for(self, other: "HipBackend"):
        self.ctx.stream_wait(other.ctx)

    def set_history_from(self, other: "HipBackend", y0: int, y1: int):
        """history of the next final pass = frame rows [y0,y1) of `other`'s PREVIOUS plane"""
        base = other.ctx.cfg.row_begin
        ptr = other.ctx.plane_ptr(abi.PLANE_PREVIOUS) + (y0 - base) * self.width * 16
        self.ctx.set_external_history(ptr, y0, y1)

    def set_guides_from(self, other: "HipBackend"):
        """the previous frame's id plane (and, with RTPT_FLAG_EXT_VARIANCE, its moment plane) = `other`'s, which ended that
        frame: what the moment accumulation and the disocclusion test of this context's frame read at reprojected pixels"""
        flags = self.ctx.cfg.flags
        y0, y1 = other.ctx.cfg.row_begin, other.ctx.cfg.row_end
        self.ctx.set_external_guides(other.ctx.plane_ptr(abi.PLANE_PREV_VIS_ID),
                                     other.ctx.plane_ptr(abi.PLANE_MOMENTS_PREV) if flags & abi.FLAG_EXT_VARIANCE else None, y0, y1)

    @property
    def guided(self) -> bool:
        return bool(self.ctx.cfg.flags & (abi.FLAG_EXT_VARIANCE | abi.FLAG_EXT_DISOCCLUSION))

    def close(self):
        self.ctx.close()


class PipelinedBackend:
    """Two frames in flight: even frames run in one backend, odd frames in another (two contexts, two streams on
    the same GPU).  The only dependency between consecutive frames of this path is the final pass's history
    fetch (temporalFiltering.comp.glsl:253), so the next frame's G-buffer, gradient, trace and non-final filter
    passes overlap the tail of the previous frame; the finished frame is handed across as external history
    behind a stream-to-stream wait.  Same results as one backend (the RNG is seeded by pixel + frame number),
    ~16 % more frames per second at 4K on one MI355X and ~31 % on a 270-row strip (frame latency unchanged).
    RTPT_FLAG_EXT_DISOCCLUSION / _VARIANCE read the previous frame's id and moment planes as well: those are handed across the
    same way before the first filter iteration (each context's own "previous" planes are two frames old) — the filter of a
    frame then starts behind the previous frame's, the passes before it still overlap.

    `backends` are two objects with the backend protocol (HipBackend; the CPU tests pass oracle backends)."""

    def __init__(self, backends):
        self.be = list(backends)
        assert len(self.be) == 2
        self.plan = self.be[0].plan
        self.width, self.height = self.be[0].width, self.be[0].height
        self.frame = 0           # frames ended
        self._ext_by_app = False  # the application registered an all-gathered history (multi-rank, moving camera)
        self._guides_by_app = False  # ... or the gathered bands of the previous frame's id / moment planes (extension flags)

    # the backend of the frame being built / of the last finished frame
    @property
    def cur(self):
        return self.be[self.frame & 1]

    @property
    def prev(self):
        return self.be[(self.frame & 1) ^ 1]

    @property
    def ctx(self):
        return self.cur.ctx

    def stream_scope(self):
        scope = getattr(self.cur, "stream_scope", None)
        return scope() if scope else _null_scope()

    def scene_upload(self, xyz, idx, xforms=None):
        for b in self.be:
            b.scene_upload(xyz, idx, xforms)

    def gbuffer(self, ubo, y0, y1):
        self.cur.gbuffer(ubo, y0, y1)

    def temporal_gradient(self, pc, y0, y1):
        self.cur.temporal_gradient(pc, y0, y1)

    def raytrace(self, pc, y0, y1):
        self.cur.raytrace(pc, y0, y1)

    def _wait_prev(self):
        wait = getattr(self.cur, "wait_for", None)
        if wait:
            wait(self.prev)

    def temporal_filter(self, pc, ubo, y0, y1):
        k, n = pc.waveletIteration, pc.maxWaveletIteration
        if k == 1 and self.frame > 0 and getattr(self.cur, "guided", False) and not self._guides_by_app:
            # the moment accumulation (this iteration) and the disocclusion test (the final one) read the previous frame's id /
            # moment planes, which live in the other backend
            self._wait_prev()
            self.cur.set_guides_from(self.prev)
        if k == n and (k & 1) and self.frame > 0 and not self._ext_by_app:
            # the previous frame's final strip lives in the other backend's PREVIOUS plane, rows = its final rows
            self._wait_prev()
            o0, o1 = self.plan.own
            self.cur.set_history_from(self.prev, o0, o1)
        self.cur.temporal_filter(pc, ubo, y0, y1)

    def end_frame(self):
        self.cur.end_frame()
        self.frame += 1
        self._ext_by_app = False
        self._guides_by_app = False

    def sync(self):
        for b in self.be:
            b.sync()

    # halo exchange: the planes of the frame being built
    def color_rows(self, plane, y0, y1):
        if plane == abi.PLANE_PREVIOUS:  # the previous frame's output (history all-gather)
            self._wait_prev()
            return self.prev.color_rows(plane, y0, y1)
        return self.cur.color_rows(plane, y0, y1)

    def history_full(self):
        return self.cur.history_full()

    # strips + extension flags, camera moved: the previous frame's id / moment planes rest in the other backend, the gathered
    # bands go to the backend of the frame being built (app._prepare_guides)
    def guide_rows(self, plane, y0, y1):
        if plane in (abi.PLANE_PREV_VIS_ID, abi.PLANE_MOMENTS_PREV):   # what the previous frame left
            self._wait_prev()
            return self.prev.guide_rows(plane, y0, y1)
        return self.cur.guide_rows(plane, y0, y1)   # planes of the frame being built (the variance that travels with the halo rows)

    def guides_full(self):
        return self.cur.guides_full()

    def use_external_guides(self, on, rows=None, moments=True):
        self._guides_by_app = bool(on)
        if on:
            self.cur.use_external_guides(True, rows, moments)
        # off: temporal_filter hands the other backend's strip-local planes across (frame > 0)

    def use_external_history(self, on, rows=None):
        self._ext_by_app = bool(on)
        if on:
            self.cur.use_external_history(True, rows)
        # off: temporal_filter installs the other backend's plane (frame > 0) — nothing to undo here

    def readback_rows(self, plane, y0, y1):
        return self.cur.readback_rows(plane, y0, y1)

    def final_image_rows(self, y0, y1):
        """rows of the last finished frame"""
        return self.prev.readback_rows(abi.PLANE_PREVIOUS, y0, y1)

    # presenting: called after end_frame, i.e. the finished frame lives in `prev`; no stream-to-stream wait — the
    # caller is still on the finished frame's stream (drawScene's scope)
    @property
    def on_device(self):
        return getattr(self.be[0], "on_device", False)

    def alloc(self, shape, dtype):
        return self.be[0].alloc(shape, dtype)

    def present_rows(self, image8, y0, y1):
        self.prev.present_rows(image8, y0, y1)

    def present_target(self, image8, y0, y1):   # called while the frame is being built: its backend is `cur`
        fn = getattr(self.cur, "present_target", None)   # optional in the backend protocol (the fused blit is a shortcut)
        if fn:
            fn(image8, y0, y1)

    def final_rows(self, y0, y1):
        return self.prev.final_rows(y0, y1)

    def raycount(self):
        return sum(b.ctx.raycount() for b in self.be)

    def close(self):
        for b in self.be:
            b.close()


class PathTracingApplication:
    """main.cpp:179-1529, headless.  Method names follow the reference."""

    def __init__(self, backend, width=1000, height=800, maxWaveletIteration=9, plan: StripPlan | None = None,
                 cameraOrigin=(-0.001, 1.0, 6.0), lightPos=(1.0, 1.0, -0.4), lightColor=(0.5, 0.5, 0.5),
                 group=None, z_near=0.1, z_far=10.0, present=None, present_root=0):
        self.backend = backend
        # present: None — the finished frame stays where the final pass left it (strips stay on their ranks);
        # "rgba8" — every frame is converted to the swapchain format (rtpt_present, B8G8R8A8_UNORM) and, with several
        # ranks, gathered on rank `present_root`; "f32" — the float strips are gathered as they are (main.cpp:1338-1361)
        if present not in (None, "rgba8", "f32"):
            raise ValueError("present must be None, 'rgba8' or 'f32'")
        self.present, self.present_root = present, present_root
        self._present_images = None
        self._present_stream = None
        self._present_done = {}
        self.present_bytes_sent = 0
        self.render_width, self.render_height = width, height          # main.cpp:52-53
        self.maxWaveletIteration = maxWaveletIteration                 # main.cpp:55
        self.plan = plan or StripPlan(height, 1, 0, maxWaveletIteration)
        self.group = group
        self.z_near, self.z_far = z_near, z_far                        # main.cpp:483 (0.1, 10)
        self.cameraOrigin = np.array(cameraOrigin, np.float32)         # main.cpp:65
        self.lightPos = np.array(lightPos, np.float32)                 # main.cpp:70
        self.lightColor = np.array(lightColor, np.float32)             # main.cpp:72
        self.cameraMoved = False
        # ubo.model: the reference recomputes it every frame as the identity (main.cpp:1469); an animated scene sets
        # modelMatrix (16 floats, column-major, affine) before drawScene — SURVEY 8(f) rank 4
        self.modelMatrix = np.eye(4, dtype=np.float32).ravel()
        self.frameCount = 0
        self.history_bytes_sent = 0   # bytes this rank sent for the last frame's history exchange
        self.history_rows = None
        self.pushConstants = abi.PushConstants()
        self.ubo = abi.Ubo()
        self._upload_initial_ubo()
        self.initializeSceneConstants()

    # ---- initVulkan() pieces -----------------------------------------------------------
    def loadMesh(self, path: str = DEFAULT_SCENE):
        """main.cpp:409-462 — objVertices / objIndices (the RT arrays)."""
        self.objVertices, self.objIndices = abi.load_obj(path)
        return self.objVertices, self.objIndices

    def buildAccelerationStructure(self, instance_xforms=None):
        """main.cpp:687-742 — one BLAS, identity instance unless transforms are given."""
        self.backend.scene_upload(self.objVertices, self.objIndices, instance_xforms)
        # world-space bounds of the scene (strips bound the reprojection reach with them, strips.reprojection_rows)
        v = np.asarray(self.objVertices, np.float64).reshape(-1, 3)
        v = v[np.unique(np.asarray(self.objIndices).ravel())]
        if instance_xforms is None:
            self.sceneBounds = (v.min(0), v.max(0))
        else:
            m = np.asarray(instance_xforms, np.float64).reshape(-1, 3, 4)
            c = np.array([[x, y, z] for x in (v[:, 0].min(), v[:, 0].max()) for y in (v[:, 1].min(), v[:, 1].max())
                          for z in (v[:, 2].min(), v[:, 2].max())])
            w = np.einsum("nij,kj->nki", m[:, :, :3], c) + m[:, None, :, 3]
            self.sceneBounds = (w.reshape(-1, 3).min(0), w.reshape(-1, 3).max(0))

    def _perspective(self):
        proj = abi.perspective(np.float32(FOV) * 2, np.float32(self.render_width) / np.float32(self.render_height),
                               self.z_near, self.z_far)
        proj[5] *= -1  # main.cpp:484
        return proj

    def _upload_initial_ubo(self):
        """uploadBuffers main.cpp:481-489: the first view looks at (0,1,0)."""
        u = self.ubo
        u.model[:] = np.eye(4, dtype=np.float32).ravel()
        u.view[:] = abi.look_at(self.cameraOrigin, (0.0, 1.0, 0.0), (0.0, 1.0, 0.0))
        u.proj[:] = self._perspective()
        u.modelPrev[:] = u.model[:]
        u.viewPrev[:] = u.view[:]
        u.projPrev[:] = u.proj[:]

    def initializeSceneConstants(self):
        """main.cpp:661-666 (lightPosPrev is a zero-initialised global at that point)."""
        pc = self.pushConstants
        pc.currentCameraColor[:] = self.lightColor
        pc.lightPos[:] = self.lightPos
        pc.lightPosPrev[:] = (0.0, 0.0, 0.0)

    # ---- per frame -----------------------------------------------------------------------
    def updateUBO(self):
        """main.cpp:1463-1475."""
        u = self.ubo
        u.modelPrev[:] = u.model[:]
        u.viewPrev[:] = u.view[:]
        u.projPrev[:] = u.proj[:]
        u.model[:] = np.asarray(self.modelMatrix, np.float32).ravel()   # :1469
        c = self.cameraOrigin
        u.view[:] = abi.look_at(c, (c[0], c[1], np.float32(c[2] - np.float32(6.0))), (0.0, 1.0, 0.0))
        u.proj[:] = self._perspective()

    def updateScene(self, keys=()):
        """main.cpp:1115-1185 with `keys` = the set of keys held this frame."""
        keys = set(keys)
        cam, lp = self.cameraOrigin, self.lightPos
        for key, axis, sign in (("S", 2, +1), ("W", 2, -1), ("A", 0, -1), ("D", 0, +1), ("E", 1, +1), ("Q", 1, -1)):
            if key in keys:
                cam[axis] = np.float32(cam[axis] + sign * SPEED)
                self.cameraMoved = True
        if "I" in keys:
            lp[2] = np.float32(lp[2] - SPEED)
        if "K" in keys:
            lp[2] = np.float32(lp[2] + SPEED)
        if "L" in keys:
            lp[0] = np.float32(lp[0] + SPEED)
            if lp[0] > 2:
                lp[0] = -20
        if "J" in keys:
            lp[0] = np.float32(lp[0] - SPEED)
            if lp[0] < -20:
                lp[0] = 2
        if "O" in keys:
            lp[1] = np.float32(lp[1] + SPEED)
        if "U" in keys:
            lp[1] = np.float32(lp[1] - SPEED)
        pc = self.pushConstants
        pc.frameNumber = self.frameCount                       # :1171
        pc.previousCameraColor[:] = pc.currentCameraColor[:]   # :1173
        pc.currentCameraColor[:] = self.lightColor             # :1175
        pc.lightPosPrev[:] = pc.lightPos[:]                    # :1177
        pc.lightPos[:] = lp                                    # :1178
        self.updateUBO()                                       # :1180
        if self.cameraMoved or self.frameCount == 0:           # :1181-1184
            pc.cameraPos[:] = cam
            self.cameraMoved = False

    def drawVisbilityBuffer(self):
        """main.cpp:1187-1199 (K0)."""
        self.backend.gbuffer(self.ubo, *self.plan.gbuffer_rows())

    def computeTemporalGradient(self):
        """main.cpp:1201-1220 (K1)."""
        self.backend.temporal_gradient(self.pushConstants, *self.plan.gradient_rows())

    def drawSceneToImage(self):
        """main.cpp:1222-1253 (K2), NUM_SAMPLE_BATCHES = 1."""
        self.pushConstants.sample_batch = 0  # :1237
        self.backend.raytrace(self.pushConstants, *self.plan.raytrace_rows())

    def applyTemporalFiltering(self):
        """main.cpp:1255-1306: k = 1..N, ping-pong by parity; one ABI call per iteration."""
        pc = self.pushConstants
        pc.maxWaveletIteration = self.maxWaveletIteration  # :1258
        if self.present == "rgba8" and hasattr(self.backend, "present_target"):
            # name this frame's swapchain rows now: the final pass then stores them in swapchain format itself (one
            # launch and one read of the frame less); _present() below calls rtpt_present all the same
            self.backend.present_target(self._present_image(self.frameCount & 1), *self.plan.own)
        if self.plan.world > 1 and (self.plan.ext_flags & (abi.FLAG_EXT_VARIANCE | abi.FLAG_EXT_DISOCCLUSION)):
            self._prepare_guides()
        for k in range(1, self.maxWaveletIteration + 1):   # :1259
            pc.waveletIteration = k
            if self.plan.world > 1 and self.plan.mode == "exchange":
                in_plane = abi.PLANE_IMAGE if (k & 1) else abi.PLANE_FILTERED
                exchange_halo(self.plan, k, lambda a, b: self.backend.color_rows(in_plane, a, b), self.group)
                if k > 1 and (self.plan.ext_flags & abi.FLAG_EXT_VARIANCE):
                    # the variance iteration k-1 filtered travels with the colour it guides (north_star: "colour/moment/
                    # depth/normal strips between iterations").  Iteration 1 needs no message: its variance comes from the
                    # moment accumulation, which every rank also runs on the halo rows it just received the colour of
                    exchange_halo(self.plan, k, lambda a, b: self.backend.guide_rows(abi.PLANE_VARIANCE, a, b), self.group)
            if self.plan.world > 1 and k == self.maxWaveletIteration and (k & 1):
                self._prepare_history()
            self.backend.temporal_filter(pc, self.ubo, *self.plan.filter_rows(k))

    def copyImageToSwapChainsCurrentImage(self):
        """main.cpp:1308-1406: the history hand-over (:1364-1372) and, when `present` is set, the blit to the
        swapchain image (:1338-1361) — with several ranks: assembled on the presenting rank."""
        self.backend.end_frame()
        if self.present:
            self._present()

    def presented_image(self, frame: int | None = None):
        """the swapchain image ("rgba8": [H, W, 4] uint8, bytes B,G,R,A; "f32": [H, W, 4] float) holding frame
        `frame` (default: the last one drawn) — complete on the presenting rank once its stream has been synchronised"""
        f = self.frameCount - 1 if frame is None else frame
        return self._present_images[f & 1]

    def present_sync(self):
        """host-blocks until every posted gather has landed"""
        for ev in self._present_done.values():
            ev.synchronize()

    def _acquire(self, idx: int):
        """vkAcquireNextImageKHR (main.cpp:1310-1316): image idx may be written again once the gather that read it —
        two frames ago — is done.  The same wait protects the float strip a 'f32' gather is still sending: its buffer
        becomes IMAGE again, and is overwritten by the trace, two frames after it was finished."""
        ev = self._present_done.pop(idx, None)
        if ev is not None:
            import torch
            torch.cuda.current_stream().wait_event(ev)

    def _present_image(self, idx: int):
        be, plan = self.backend, self.plan
        H, W = plan.height, self.render_width
        if self._present_images is None:
            if self.present == "rgba8":
                self._present_images = [be.alloc((H, W, 4), "uint8") for _ in range(2)]
            elif plan.world > 1 and plan.rank == self.present_root:
                self._present_images = [be.alloc((H, W, 4), "float32") for _ in range(2)]
            else:
                self._present_images = [None, None]
        return self._present_images[idx]

    def _present(self):
        be, plan = self.backend, self.plan
        o0, o1 = plan.own
        f = self.frameCount            # drawScene increments after this call
        idx = f & 1
        rgba8 = self.present == "rgba8"
        img = self._present_image(idx)
        if rgba8:
            be.present_rows(img, o0, o1)
            mine = img[o0:o1]
        else:
            mine = be.final_rows(o0, o1) if plan.world > 1 else None
            if img is not None:
                img[o0:o1].copy_(mine)
        self.present_bytes_sent = 0
        if plan.world == 1:
            return
        if getattr(be, "on_device", False):
            # the gather runs on its own stream behind the frame's kernels, so the next frame's passes do not wait for
            # the wire; _acquire() closes the loop two frames later
            import torch
            main = torch.cuda.current_stream()
            if self._present_stream is None:
                self._present_stream = torch.cuda.Stream()
            ps = self._present_stream
            ps.wait_stream(main)
            with torch.cuda.stream(ps):
                self.present_bytes_sent = gather_frame(plan, mine, img, self.present_root, self.group)
                ev = torch.cuda.Event()
                ev.record(ps)
            self._present_done[idx] = ev
        else:
            self.present_bytes_sent = gather_frame(plan, mine, img, self.present_root, self.group)

    def _camera_static(self):
        """every pixel reprojects onto itself: view, proj AND model unchanged since the previous frame (an animated
        model matrix moves pixels across strips exactly like a camera move does)"""
        u = self.ubo
        return (list(u.view) == list(u.viewPrev) and list(u.proj) == list(u.projPrev)
                and list(u.model) == list(u.modelPrev))

    def _world_bounds(self):
        """sceneBounds (of the uploaded, un-posed mesh) under the current model matrix: the box of its 8 posed corners"""
        lo, hi = self.sceneBounds
        m = np.asarray(self.ubo.model[:], np.float64).reshape(4, 4).T
        c = np.array([[x, y, z, 1.0] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])]) @ m.T
        return c[:, :3].min(0), c[:, :3].max(0)

    def _prepare_history(self):
        """With several ranks the final pass may fetch history at a reprojected pixel of another strip
        (temporalFiltering.comp.glsl:253).  While the camera rests the reprojection is the identity and the
        strip-local PREVIOUS plane is enough.  In a frame where view/proj differ from viewPrev/projPrev every rank
        bounds the previous-frame rows its own pixels can reach (strips.reprojection_rows: camera matrices + scene
        bounds, the same on every rank) and the ranks swap exactly those bands of their finished strips — k*W*16 B
        messages to the neighbours the band overlaps instead of the whole previous frame to everybody."""
        be = self.backend
        if self.frameCount == 0 or self._camera_static():
            be.use_external_history(False)
            self.history_bytes_sent = 0
            return
        H, R = self.plan.height, self.plan.world
        bounds = self._world_bounds()
        needs = [reprojection_rows(self.ubo, self.render_width, H, self.plan.rows_of(r), bounds, self.z_near)
                 for r in range(R)]
        full = be.history_full()
        self.history_bytes_sent = exchange_history(self.plan, needs, lambda a, b: be.color_rows(abi.PLANE_PREVIOUS, a, b),
                                                   full, self.group) + getattr(self, "_guide_bytes", 0)
        self.history_rows = needs[self.plan.rank]
        be.use_external_history(True, needs[self.plan.rank])

    def _prepare_guides(self):
        """RTPT_FLAG_EXT_VARIANCE / _DISOCCLUSION on strips: the moment accumulation (before iteration 1) and the
        disocclusion test (final pass) read the PREVIOUS frame's id and moment planes at reprojected pixels.  Every
        rank holds them for its stored rows; under camera motion the rows a rank's stored pixels can reach beyond
        that are gathered from their owners, with the same bound and plan as the history image."""
        be = self.backend
        self._guide_bytes = 0
        if self.frameCount == 0 or self._camera_static():
            be.use_external_guides(False)
            return
        from .strips import exchange_history
        H, R = self.plan.height, self.plan.world
        plans = [StripPlan(H, R, r, self.plan.iterations, self.plan.mode, self.plan.ext_flags, self.plan.splits) for r in range(R)]
        bounds = self._world_bounds()
        needs = [reprojection_rows(self.ubo, self.render_width, H, p.stored, bounds, self.z_near) for p in plans]
        ids, mom = be.guides_full()
        variance = bool(self.plan.ext_flags & abi.FLAG_EXT_VARIANCE)
        sent = exchange_history(self.plan, needs, lambda a, b: be.guide_rows(abi.PLANE_PREV_VIS_ID, a, b), ids, self.group)
        if variance:
            sent += exchange_history(self.plan, needs, lambda a, b: be.guide_rows(abi.PLANE_MOMENTS_PREV, a, b), mom, self.group)
        self._guide_bytes = sent
        be.use_external_guides(True, needs[self.plan.rank], moments=variance)

    def drawScene(self, keys=()):
        """main.cpp:1090-1113."""
        scope = getattr(self.backend, "stream_scope", None)
        with (scope() if scope else _null_scope()):
            if self.present and self._present_done:
                self._acquire(self.frameCount & 1)
            self.updateScene(keys)
            self.drawVisbilityBuffer()
            self.computeTemporalGradient()
            self.drawSceneToImage()
            self.applyTemporalFiltering()
            self.copyImageToSwapChainsCurrentImage()
        self.frameCount += 1

    def run(self, frames: int, script=None):
        """mainLoop (main.cpp:301) for a fixed number of frames; script[f] = keys held on frame f."""
        for f in range(frames):
            self.drawScene(script[f] if script and f < len(script) else ())


def make_app(width, height, max_segments=4, iterations=5, rank=0, world=1, mode="exchange", flags=0,
             torch_planes=None, debug_mask=0, scene=DEFAULT_SCENE, instance_xforms=None, group=None, mesh=None,
             frames_in_flight=1, samples_per_pixel=1, splits=(), **app_kw):
    """createBuffers + loadMesh + buildAccelerationStructure for one rank.  `mesh` = (xyz, idx) replaces
    the OBJ (synthetic scenes of scenes.py); `splits` = unequal strips (strips.StripPlan.splits, the same on every rank)."""
    plan = StripPlan(height, world, rank, iterations, mode, flags & abi.FLAG_EXT_MASK, tuple(splits) if world > 1 else ())
    if torch_planes is None:
        torch_planes = world > 1  # halo exchange and the history all-gather move rows of torch-owned planes
    def one():
        return HipBackend(width, height, plan, max_segments=max_segments, flags=flags, torch_planes=torch_planes,
                          debug_mask=debug_mask, samples_per_pixel=samples_per_pixel)
    if frames_in_flight == 2:
        be = PipelinedBackend([one(), one()])
    elif frames_in_flight == 1:
        be = one()
    else:
        raise ValueError("frames_in_flight must be 1 or 2")
    app = PathTracingApplication(be, width, height, iterations, plan, group=group, **app_kw)
    if mesh is not None:
        app.objVertices, app.objIndices = mesh
    else:
        app.loadMesh(scene)
    app.buildAccelerationStructure(instance_xforms)
    return app
