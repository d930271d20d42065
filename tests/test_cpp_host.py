"""The headless C++ host (host/app.cpp: the reference's PathTracingApplication over the C ABI)."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, bits

PKG = os.path.join(ROOT, "real_time_path_tracing_with_spatiotemporal_filtering_amd")
APP = os.path.join(PKG, "rtpt_app")


@pytest.fixture(scope="module")
def app_binary():
    subprocess.check_call(["make", "-C", os.path.join(PKG, "csrc"), "-s"])
    subprocess.check_call(["make", "-C", os.path.join(PKG, "host"), "-s"])
    return APP


def test_cli_help_and_bad_option(app_binary):
    out = subprocess.run([app_binary, "--help"], capture_output=True, text=True)
    assert out.returncode == 0 and "--iterations" in out.stdout
    assert subprocess.run([app_binary, "--bogus"], capture_output=True).returncode == 2


def test_no_gpu_is_an_error(app_binary):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    out = subprocess.run([app_binary, "--frames", "1", "--width", "32", "--height", "32"], capture_output=True, text=True)
    assert out.returncode == 1 and "no CPU fallback" in out.stderr


def read_pfm(path):  # the C++ host's --dump and output.py agree on the format
    from real_time_path_tracing_with_spatiotemporal_filtering_amd.output import read_pfm as rd
    return rd(str(path))


@pytest.mark.gpu
@pytest.mark.parametrize("in_flight", [1, 2])
def test_cpp_host_equals_python_host(app_binary, hip_lib, tmp_path, in_flight):
    from real_time_path_tracing_with_spatiotemporal_filtering_amd.app import make_app
    W, H, SEG, N = 96, 64, 3, 5
    keys = ["", "", "J", "D", "SI"]
    pfm = tmp_path / "out.pfm"
    out = subprocess.run([app_binary, "--width", str(W), "--height", str(H), "--segments", str(SEG), "--iterations", str(N),
                          "--frames", str(len(keys)), "--script", ",".join(keys), "--dump", str(pfm),
                          "--frames-in-flight", str(in_flight)],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    stats = json.loads(out.stdout.strip().splitlines()[-1])
    app = make_app(W, H, max_segments=SEG, iterations=N)
    for k in keys:
        app.drawScene(tuple(k))
    want = app.backend.ctx.readback(hip_lib.PLANE_IMAGE)
    got = read_pfm(pfm)
    assert np.array_equal(bits(got), bits(np.ascontiguousarray(want[..., :3])))
    assert stats["rays"] == app.backend.ctx.raycount() and stats["frames"] == len(keys)


@pytest.mark.gpu
@pytest.mark.parametrize("ranks,halo,splits", [(2, "redundant", ""), (2, "exchange", ""), (3, "exchange", ""), (4, "redundant", ""),
                                               (3, "exchange", "0,20,88,121"), (4, "redundant", "0,40,49,100,121")])
def test_cpp_host_strips_equal_python_single_context(app_binary, hip_lib, tmp_path, ranks, halo, splits):
    """--ranks R without --rank: R strip contexts in one process on one GPU, the transport's messages are device copies
    (with --rank each strip is a process on its own GPU and the same messages are ncclSend/ncclRecv).  Halo rows per
    iteration (exchange) or redundant rows, and the previous frame's bands under vertical camera moves (E, Q): the
    assembled frame must equal the single-context Python host's, bit for bit, and so must the ray count.  `splits`: strips of
    different heights (--splits)."""
    from real_time_path_tracing_with_spatiotemporal_filtering_amd.app import make_app
    W, H, SEG, N = 160, 121, 3, 5
    keys = ["", "E", "J", "QA", "", "E"]
    pfm = tmp_path / "strips.pfm"
    out = subprocess.run([app_binary, "--width", str(W), "--height", str(H), "--segments", str(SEG), "--iterations", str(N),
                          "--frames", str(len(keys)), "--script", ",".join(keys), "--dump", str(pfm),
                          "--ranks", str(ranks), "--halo", halo] + (["--splits", splits] if splits else []),
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    stats = json.loads(out.stdout.strip().splitlines()[-1])
    app = make_app(W, H, max_segments=SEG, iterations=N)
    for k in keys:
        app.drawScene(tuple(k))
    want = app.backend.ctx.readback(hip_lib.PLANE_IMAGE)
    got = read_pfm(pfm)
    assert np.array_equal(bits(got), bits(np.ascontiguousarray(want[..., :3])))
    assert stats["rays"] == app.backend.ctx.raycount() and stats["ranks"] == ranks
    # what travelled: history bands in the frames whose camera moved (+ the k-row halos per iteration in exchange mode)
    assert stats["bytes_sent"] > 0
    if halo == "redundant":
        assert stats["bytes_sent"] < 3 * (ranks - 1) * 2 * H * W * 16, "bands, not whole frames to every rank"


@pytest.mark.gpu
@pytest.mark.parametrize("ranks,halo,present,splits,flags", [(3, "exchange", "", "", 0), (4, "redundant", "rgba8", "0,40,49,100,121", 0),
                                                             (2, "redundant", "f32", "", 0), (3, "exchange", "rgba8", "0,20,88,121", 0),
                                                             (3, "redundant", "", "", 0x180), (3, "exchange", "", "0,20,88,121", 0x900),
                                                             (2, "redundant", "rgba8", "", 0x1F0)])
def test_cpp_host_strips_with_two_frames_in_flight(app_binary, hip_lib, oracle, tmp_path, ranks, halo, present, splits, flags):
    """--ranks R --frames-in-flight 2: every rank builds even frames in one context and odd frames in another (one stream per
    parity); the finished strip is handed to the other context for the blend (strip-local while the camera rests, through the
    bands of the ranks' OTHER contexts in the frames where it moved: E, Q), halo rows and the presenting rank's gather run on
    the frame's own stream.  With the guided extension modes (flags) the previous frame's id / moment planes cross the same
    two ways.  Frames, ray count and presented image equal the single-context Python host's."""
    from real_time_path_tracing_with_spatiotemporal_filtering_amd.app import make_app
    W, H, SEG, N = 160, 121, 3, 5
    keys = ["", "E", "J", "QA", "", "", "E"]
    pfm, raw = tmp_path / "s.pfm", tmp_path / "p.raw"
    cmd = [app_binary, "--width", str(W), "--height", str(H), "--segments", str(SEG), "--iterations", str(N), "--frames", str(len(keys)),
           "--script", ",".join(keys), "--dump", str(pfm), "--ranks", str(ranks), "--halo", halo, "--frames-in-flight", "2"]
    cmd += (["--splits", splits] if splits else []) + (["--present", present, "--dump-present", str(raw)] if present else [])
    cmd += ["--flags", hex(flags)] if flags else []
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    stats = json.loads(out.stdout.strip().splitlines()[-1])
    app = make_app(W, H, max_segments=SEG, iterations=N, flags=flags)
    for k in keys:
        app.drawScene(tuple(k))
    want = app.backend.ctx.readback(hip_lib.PLANE_IMAGE)
    assert np.array_equal(bits(read_pfm(pfm)), bits(np.ascontiguousarray(want[..., :3])))
    assert stats["rays"] == app.backend.ctx.raycount()
    if present == "rgba8":
        assert np.fromfile(raw, np.uint8).tobytes() == oracle.present_bgra8(want).tobytes()
    elif present == "f32":
        assert np.fromfile(raw, np.uint8).tobytes() == want.tobytes()


@pytest.mark.gpu
def test_cpp_host_rccl_transport_single_rank(app_binary, tmp_path):
    """--ranks 1 --rank 0 is refused (one rank needs no transport), --ranks 2 --rank 0 needs a peer: what CAN run on a
    one-GPU box is the rendezvous + communicator bring-up of a world of one — done through a private world: ranks = 1 is
    the plain path, so this only checks the CLI contract; the RCCL point-to-point path itself needs two GPUs."""
    out = subprocess.run([app_binary, "--width", "64", "--height", "48", "--frames", "1", "--segments", "2", "--iterations", "3",
                          "--ranks", "2", "--rank", "5", "--rccl-id-file", str(tmp_path / "id")], capture_output=True, text=True)
    assert out.returncode == 1 and "rank out of range" in out.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("ranks,halo,flags", [(3, "exchange", 0xF0), (3, "redundant", 0x1F0), (2, "redundant", 0x180), (3, "exchange", 0x190),
                                              (3, "redundant", 0x900), (2, "exchange", 0x900)])
def test_cpp_host_strips_serve_the_extension_modes(app_binary, hip_lib, tmp_path, ranks, halo, flags):
    """SURVEY 8(f) rank 1 in the product host's strip mode: 5x5 taps / 2^(k-1) stride widen the halo (StripPlan::reach),
    adaptive alpha reads the gradient, and the disocclusion test / moment accumulation read the previous frame's id and
    moment planes at reprojected pixels — bands of them travel between the strips when the camera moves
    (rtpt_set_external_guides).  R in-process strips equal the Python single context bit for bit, vertical camera moves
    in the script."""
    from real_time_path_tracing_with_spatiotemporal_filtering_amd.app import make_app
    W, H, SEG, N = 160, 144, 3, 3
    keys = ["", "E", "J", "QA", "", "E"]
    pfm = tmp_path / "ext.pfm"
    out = subprocess.run([app_binary, "--width", str(W), "--height", str(H), "--segments", str(SEG), "--iterations", str(N),
                          "--frames", str(len(keys)), "--script", ",".join(keys), "--dump", str(pfm), "--flags", hex(flags),
                          "--ranks", str(ranks), "--halo", halo], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    app = make_app(W, H, max_segments=SEG, iterations=N, flags=flags)
    for k in keys:
        app.drawScene(tuple(k))
    want = app.backend.ctx.readback(hip_lib.PLANE_IMAGE)
    got = read_pfm(pfm)
    assert np.array_equal(bits(got), bits(np.ascontiguousarray(want[..., :3])))


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [0x180, 0x900])
def test_cpp_host_two_frames_in_flight_serve_the_guided_extension_modes(app_binary, hip_lib, tmp_path, flags):
    """--frames-in-flight 2 with the disocclusion test / variance guidance: the previous frame's id and moment planes are
    handed from the context that ended that frame to the one building the next (rtpt_stream_wait + rtpt_set_external_guides
    before the first filter iteration) — the serial Python host's frame, bit for bit (exact filter)."""
    from real_time_path_tracing_with_spatiotemporal_filtering_amd.app import make_app
    W, H, SEG, N = 160, 144, 3, 5
    keys = ["", "E", "J", "QA", "", "E", "D", ""]
    pfm = tmp_path / "fif.pfm"
    out = subprocess.run([app_binary, "--width", str(W), "--height", str(H), "--segments", str(SEG), "--iterations", str(N),
                          "--frames", str(len(keys)), "--script", ",".join(keys), "--dump", str(pfm), "--flags", hex(flags),
                          "--exact-filter", "--frames-in-flight", "2"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    app = make_app(W, H, max_segments=SEG, iterations=N, flags=flags | hip_lib.FLAG_EXACT_FILTER)
    for k in keys:
        app.drawScene(tuple(k))
    want = app.backend.ctx.readback(hip_lib.PLANE_IMAGE)
    assert np.array_equal(bits(read_pfm(pfm)), bits(np.ascontiguousarray(want[..., :3])))


@pytest.mark.gpu
def test_cpp_host_rccl_rendezvous_ignores_a_stale_id_and_never_hangs(app_binary, tmp_path):
    """host/strips.cpp RcclTransport: (a) a rank > 0 that finds an id file of ANOTHER launch (different nonce) does not
    trust it and gives up with an error when its own launch's id never appears; (b) rank 0 removes the stale file,
    publishes its own, and when its peer never joins the communicator the watchdog ends the process with status 3
    instead of hanging in ncclCommInitRank"""
    import struct
    import time
    idf = tmp_path / "id"
    stale = struct.pack("<QQ", 0x3144494C43435452, 6) + bytes(128)
    idf.write_bytes(stale)
    common = [app_binary, "--width", "64", "--height", "48", "--frames", "1", "--segments", "2", "--iterations", "3", "--ranks", "2",
              "--rccl-id-file", str(idf), "--rccl-nonce", "7"]
    t0 = time.time()
    out = subprocess.run(common + ["--rank", "1", "--rccl-timeout", "2"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 1 and "this launch's nonce" in out.stderr, out.stderr
    assert idf.read_bytes() == stale
    out = subprocess.run(common + ["--rank", "0", "--rccl-timeout", "6"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 3 and "did not come up in time" in out.stderr, (out.returncode, out.stderr)
    fresh = idf.read_bytes()
    assert fresh[:16] == struct.pack("<QQ", 0x3144494C43435452, 7) and fresh != stale, "rank 0 replaced the stale file"
    assert time.time() - t0 < 100


@pytest.mark.gpu
@pytest.mark.parametrize("ranks,halo,present,splits", [(1, "redundant", "rgba8", ""), (3, "redundant", "rgba8", ""), (4, "exchange", "f32", ""),
                                                       (3, "redundant", "rgba8", "0,61,80,121")])
def test_cpp_host_presents_the_assembled_frame(app_binary, hip_lib, oracle, tmp_path, ranks, halo, present, splits):
    """--present: main.cpp:1338-1361 in the C++ host.  One context converts its frame with rtpt_present; with --ranks the
    presenting rank's swapchain image is assembled from every strip (converted rows, or float rows) on the present
    stream.  The last image equals the Python single-context frame (through the oracle's restatement of the blit)."""
    from real_time_path_tracing_with_spatiotemporal_filtering_amd.app import make_app
    W, H, SEG, N = 160, 121, 3, 5
    keys = ["", "E", "J", "QA", ""]
    raw = tmp_path / "present.raw"
    cmd = [app_binary, "--width", str(W), "--height", str(H), "--segments", str(SEG), "--iterations", str(N), "--frames", str(len(keys)),
           "--script", ",".join(keys), "--present", present, "--dump-present", str(raw)]
    if ranks > 1:
        cmd += ["--ranks", str(ranks), "--halo", halo] + (["--splits", splits] if splits else [])
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    app = make_app(W, H, max_segments=SEG, iterations=N)
    for k in keys:
        app.drawScene(tuple(k))
    want = app.backend.ctx.readback(hip_lib.PLANE_IMAGE)
    got = np.fromfile(raw, np.uint8)
    if present == "rgba8":
        assert got.tobytes() == oracle.present_bgra8(want).tobytes()
    else:
        assert got.tobytes() == want.tobytes()


@pytest.mark.gpu
def test_cpp_host_present_with_two_frames_in_flight_and_odd_sizes(app_binary, hip_lib, oracle, tmp_path):
    """the fused blit (rtpt_present_target armed on the context that builds the frame, rtpt_present on the one that finished
    it) with even / odd frames in two contexts, on a frame whose width is not a multiple of the 64-pixel segments"""
    from real_time_path_tracing_with_spatiotemporal_filtering_amd.app import make_app
    W, H, SEG, N = 203, 77, 3, 5
    keys = ["", "J", "E", "", "Q"]
    raw = tmp_path / "p.raw"
    out = subprocess.run([app_binary, "--width", str(W), "--height", str(H), "--segments", str(SEG), "--iterations", str(N),
                          "--frames", str(len(keys)), "--script", ",".join(keys), "--present", "rgba8", "--dump-present", str(raw),
                          "--frames-in-flight", "2"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    app = make_app(W, H, max_segments=SEG, iterations=N)
    for k in keys:
        app.drawScene(tuple(k))
    want = app.backend.ctx.readback(hip_lib.PLANE_IMAGE)
    assert np.fromfile(raw, np.uint8).tobytes() == oracle.present_bgra8(want).tobytes()


@pytest.mark.parametrize("halo,flags,splits", [("redundant", 0, ()), ("exchange", 0, ()), ("redundant", 0x900, ()), ("exchange", 0x960, ()),
                                               ("redundant", 0, (0, 31, 120, 150, 217)), ("exchange", 0x900, (0, 80, 100, 190, 217))])
def test_cpp_strip_plan_and_history_bands_equal_the_python_mirror(app_binary, halo, flags, splits):
    """host-only (no GPU): `rtpt_app --plan-only` prints the C++ host's strip plan and, per scripted frame, the
    previous-frame rows every rank's final pass can reach (host/strips.cpp); the Python mirror (strips.py), which the gloo
    tests exercise end to end, must produce the same rows — same ownership, same halos, same reprojection bound."""
    from test_host_logic import Recorder
    from real_time_path_tracing_with_spatiotemporal_filtering_amd import abi
    from real_time_path_tracing_with_spatiotemporal_filtering_amd.app import DEFAULT_SCENE, PathTracingApplication
    from real_time_path_tracing_with_spatiotemporal_filtering_amd.strips import StripPlan, reprojection_rows
    abi.load()
    W, H, N, R = 333, 217, 5, 4
    keys = ["", "E", "E", "D", "QS", "", "W"]
    out = subprocess.run([app_binary, "--plan-only", "--width", str(W), "--height", str(H), "--iterations", str(N), "--ranks", str(R),
                          "--halo", halo, "--frames", str(len(keys)), "--script", ",".join(keys), "--flags", hex(flags)] +
                         (["--splits", ",".join(map(str, splits))] if splits else []),
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    plan = json.loads(out.stdout)
    assert plan["splits"] == [StripPlan.bounds(H, R, r, splits)[0] for r in range(R)] + [H]
    for r, p in enumerate(plan["ranks"]):
        sp = StripPlan(H, R, r, N, halo, flags, splits)
        assert tuple(p["own"]) == sp.own and tuple(p["stored"]) == sp.stored and tuple(p["raytrace"]) == sp.raytrace_rows()
        assert [tuple(x) for x in p["filter"]] == [sp.filter_rows(k) for k in range(1, N + 1)]
    app = PathTracingApplication(Recorder(), W, H, N)
    app.loadMesh(DEFAULT_SCENE)
    app.buildAccelerationStructure()
    moved_any = False
    for f, k in enumerate(keys):
        app.updateScene(tuple(k))
        needs = [list(reprojection_rows(app.ubo, W, H, StripPlan.bounds(H, R, r, splits), app.sceneBounds, app.z_near)) for r in range(R)]
        assert plan["frames"][f]["needs"] == needs, (f, k)
        assert plan["frames"][f]["moved"] == (f > 0 and not app._camera_static())
        moved_any |= plan["frames"][f]["moved"]
        app.frameCount += 1
    assert moved_any


def test_cpp_balanced_splits_equal_the_python_mirror(app_binary):
    """host-only: the boundaries the C++ host derives from the ranks' frame times (host/strips.cpp balanced_splits, printed by
    --plan-only --balance) are the Python mirror's, for measured and for random times, from equal and from unequal strips"""
    from real_time_path_tracing_with_spatiotemporal_filtering_amd.strips import StripPlan, balanced_splits
    rng = np.random.default_rng(11)
    cases = [(2160, 8, 5, "redundant", (), [0.4063, 0.6799, 0.686, 0.5498, 0.5539, 0.6709, 0.6652, 0.3998])]
    for _ in range(12):
        H, R = int(rng.integers(200, 2200)), int(rng.integers(2, 9))
        halo = ("redundant", "exchange")[int(rng.integers(0, 2))]
        splits = ()
        if rng.integers(0, 2):
            cuts = sorted(int(v) for v in rng.choice(np.arange(1, H // 20), R - 1, replace=False) * 20)
            splits = (0, *cuts, H)
        cases.append((H, R, 5, halo, splits, [float(v) for v in rng.uniform(0.05, 3.0, R)]))
    for H, R, N, halo, splits, cost in cases:
        out = subprocess.run([app_binary, "--plan-only", "--width", "64", "--height", str(H), "--iterations", str(N), "--ranks", str(R),
                              "--halo", halo, "--frames", "1", "--balance", ",".join(repr(c) for c in cost)] +
                             (["--splits", ",".join(map(str, splits))] if splits else []), capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        cur = tuple(StripPlan.bounds(H, R, r, splits)[0] for r in range(R)) + (H,)
        want = balanced_splits(cur, cost, max(1, StripPlan(H, R, 0, N, halo).halo))
        assert tuple(json.loads(out.stdout)["balanced"]) == want, (H, R, halo, splits, cost)


def _lattice_args(lattice, tess):
    return ["--lattice", "x".join(map(str, lattice)), "--tessellate", str(tess)]


def test_cpp_lattice_scene_and_its_strip_plan_equal_the_python_mirror(app_binary, tmp_path):
    """host-only (no GPU): BASELINE configs[4]'s scene as the C++ host builds it (host/scene_gen.cpp: tessellated quads on a
    lattice of instances, main.cpp:728-741 being the instance list it extends) is, bit for bit, what scenes.py hands to
    rtpt_scene_upload — vertices, indices, transforms, camera, light, far plane — and the history bands of its strips, bounded
    with the POSED, INSTANCED scene box, are the Python mirror's."""
    from test_host_logic import Recorder
    from real_time_path_tracing_with_spatiotemporal_filtering_amd import abi, scenes
    from real_time_path_tracing_with_spatiotemporal_filtering_amd.app import DEFAULT_SCENE, PathTracingApplication
    from real_time_path_tracing_with_spatiotemporal_filtering_amd.strips import StripPlan, reprojection_rows
    abi.load()
    W, H, N, R = 320, 180, 5, 4
    lattice, tess = (3, 4, 2), 3
    keys = ["", "E", "E", "D", "QS", ""]
    dump = tmp_path / "scene.bin"
    out = subprocess.run([app_binary, "--plan-only", "--width", str(W), "--height", str(H), "--iterations", str(N), "--ranks", str(R),
                          "--frames", str(len(keys)), "--script", ",".join(keys), "--dump-scene", str(dump)] + _lattice_args(lattice, tess),
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    xyz, idx = abi.load_obj(DEFAULT_SCENE)
    vx, ti, xf, cam, zfar = scenes.instanced_cornell(xyz, idx, lattice=lattice, tess=tess)
    raw = np.fromfile(dump, np.uint8)
    nv, nt, ni = raw[:12].view(np.uint32)
    assert (nv, nt, ni) == (len(vx), len(ti), len(xf)) and nt == 32 * tess * tess
    o = 12
    got_vx = raw[o:o + 12 * nv].view(np.float32).reshape(-1, 3); o += 12 * nv
    got_ti = raw[o:o + 12 * nt].view(np.uint32).reshape(-1, 3); o += 12 * nt
    got_xf = raw[o:o + 48 * ni].view(np.float32).reshape(-1, 12); o += 48 * ni
    tail = raw[o:o + 28].view(np.float32)
    assert np.array_equal(bits(got_vx), bits(vx)) and np.array_equal(got_ti, ti) and np.array_equal(bits(got_xf), bits(xf))
    light = np.array((1.0, float(cam[1]), float(cam[2]) - 8.0), np.float32)   # bench.py's light for this scene
    assert np.array_equal(bits(tail[:3]), bits(np.array(cam, np.float32))) and np.array_equal(bits(tail[3:6]), bits(light))
    assert tail[6] == np.float32(zfar)
    plan = json.loads(out.stdout)
    app = PathTracingApplication(Recorder(), W, H, N, cameraOrigin=cam, z_far=zfar, lightPos=light)
    app.objVertices, app.objIndices = vx, ti
    app.buildAccelerationStructure(xf)
    moved_any = False
    for f, k in enumerate(keys):
        app.updateScene(tuple(k))
        needs = [list(reprojection_rows(app.ubo, W, H, StripPlan.bounds(H, R, r), app.sceneBounds, app.z_near)) for r in range(R)]
        assert plan["frames"][f]["needs"] == needs, (f, k)
        moved_any |= plan["frames"][f]["moved"]
        app.frameCount += 1
    assert moved_any
    # the full-size scene: counts only (1,152,000 triangles x 10^3 instances is what rtpt_scene_upload flattens)
    out = subprocess.run([app_binary, "--plan-only", "--frames", "1", "--dump-scene", str(dump)] + _lattice_args((10, 10, 10), 6),
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    nv, nt, ni = np.fromfile(dump, np.uint32, 3)
    assert nt * ni == 1_152_000


@pytest.mark.gpu
@pytest.mark.parametrize("ranks,halo", [(1, "redundant"), (3, "redundant"), (3, "exchange")])
def test_cpp_host_runs_the_lattice_scene_like_the_python_host(app_binary, hip_lib, tmp_path, ranks, halo):
    """configs[4]'s scene class through the C++ host (BVH traversal over fan pairs, per-pixel-normal filter, instance
    transforms through rtpt_scene_upload), one context and in-process strips, camera moving: the Python host's frame bit for
    bit.  (The 1.15 M-triangle size runs in test_fullsize_gpu.py.)"""
    from real_time_path_tracing_with_spatiotemporal_filtering_amd import abi, scenes
    from real_time_path_tracing_with_spatiotemporal_filtering_amd.app import DEFAULT_SCENE, make_app
    W, H, SEG, N = 192, 120, 4, 5
    lattice, tess = (3, 3, 2), 2
    keys = ["", "", "E", "D", "J"]
    pfm = tmp_path / "out.pfm"
    cmd = [app_binary, "--width", str(W), "--height", str(H), "--segments", str(SEG), "--iterations", str(N), "--frames", str(len(keys)),
           "--script", ",".join(keys), "--dump", str(pfm)] + _lattice_args(lattice, tess)
    if ranks > 1:
        cmd += ["--ranks", str(ranks), "--halo", halo]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    stats = json.loads(out.stdout.strip().splitlines()[-1])
    xyz, idx = abi.load_obj(DEFAULT_SCENE)
    vx, ti, xf, cam, zfar = scenes.instanced_cornell(xyz, idx, lattice=lattice, tess=tess)
    app = make_app(W, H, max_segments=SEG, iterations=N, mesh=(vx, ti), instance_xforms=xf, cameraOrigin=cam, z_far=zfar,
                   lightPos=(1.0, float(cam[1]), float(cam[2]) - 8.0))
    for k in keys:
        app.drawScene(tuple(k))
    want = app.backend.ctx.readback(hip_lib.PLANE_IMAGE)
    got = read_pfm(pfm)
    assert np.array_equal(bits(got), bits(np.ascontiguousarray(want[..., :3])))
    assert stats["rays"] == app.backend.ctx.raycount()


def _instance_file(tmp_path):
    """three instances with rotation, shear-free scale and translation (3x4 row-major), as text"""
    c, s_ = np.float32(np.cos(0.3)), np.float32(np.sin(0.3))
    xf = np.array([[1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0],
                   [c, 0, s_, 2.4, 0, 1, 0, 0.0, -s_, 0, c, -1.0],
                   [0.5, 0, 0, -2.2, 0, 0.5, 0, 0.0, 0, 0, 0.5, 0.5]], np.float32)
    path = tmp_path / "instances.txt"
    path.write_text("\n".join(" ".join(repr(float(v)) for v in row) for row in xf))
    return path, xf


def test_cpp_instance_file_is_what_the_python_mirror_uploads(app_binary, tmp_path):
    """host-only: `rtpt_app --instances file` (a general instance list in place of main.cpp:728-741's identity) hands
    rtpt_scene_upload the transforms of the file, bit for bit, and bounds its strips' history bands with the instanced box"""
    from test_host_logic import Recorder
    from real_time_path_tracing_with_spatiotemporal_filtering_amd import abi
    from real_time_path_tracing_with_spatiotemporal_filtering_amd.app import DEFAULT_SCENE, PathTracingApplication
    from real_time_path_tracing_with_spatiotemporal_filtering_amd.strips import StripPlan, reprojection_rows
    abi.load()
    path, xf = _instance_file(tmp_path)
    W, H, N, R = 320, 180, 5, 3
    keys = ["", "E", "D", "Q"]
    dump = tmp_path / "scene.bin"
    out = subprocess.run([app_binary, "--plan-only", "--width", str(W), "--height", str(H), "--iterations", str(N), "--ranks", str(R),
                          "--frames", str(len(keys)), "--script", ",".join(keys), "--dump-scene", str(dump), "--instances", str(path)],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    raw = np.fromfile(dump, np.uint8)
    nv, nt, ni = raw[:12].view(np.uint32)
    assert (nt, ni) == (32, 3)
    o = 12 + 12 * nv + 12 * nt
    assert np.array_equal(bits(raw[o:o + 48 * ni].view(np.float32).reshape(-1, 12)), bits(xf))
    plan = json.loads(out.stdout)
    app = PathTracingApplication(Recorder(), W, H, N)
    app.loadMesh(DEFAULT_SCENE)
    app.buildAccelerationStructure(xf)
    for f, k in enumerate(keys):
        app.updateScene(tuple(k))
        needs = [list(reprojection_rows(app.ubo, W, H, StripPlan.bounds(H, R, r), app.sceneBounds, app.z_near)) for r in range(R)]
        assert plan["frames"][f]["needs"] == needs, (f, k)
        app.frameCount += 1
    bad = subprocess.run([app_binary, "--plan-only", "--instances", str(path), "--lattice", "2x2x2"], capture_output=True, text=True)
    assert bad.returncode == 1 and "exclude" in bad.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [1, 3])
def test_cpp_host_with_an_instance_file_equals_the_python_host(app_binary, hip_lib, tmp_path, ranks):
    """rotated, scaled and translated instances of the Cornell box (96 triangles: BVH traversal over fan pairs) through the
    C++ host, one context and three in-process strips with a moving camera, against the Python host, bit for bit"""
    from real_time_path_tracing_with_spatiotemporal_filtering_amd.app import make_app
    path, xf = _instance_file(tmp_path)
    W, H, SEG, N = 192, 120, 4, 5
    keys = ["", "", "E", "D", "J"]
    pfm = tmp_path / "out.pfm"
    cmd = [app_binary, "--width", str(W), "--height", str(H), "--segments", str(SEG), "--iterations", str(N), "--frames", str(len(keys)),
           "--script", ",".join(keys), "--dump", str(pfm), "--instances", str(path)]
    if ranks > 1:
        cmd += ["--ranks", str(ranks)]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    app = make_app(W, H, max_segments=SEG, iterations=N, instance_xforms=xf)
    for k in keys:
        app.drawScene(tuple(k))
    want = app.backend.ctx.readback(hip_lib.PLANE_IMAGE)
    assert np.array_equal(bits(read_pfm(pfm)), bits(np.ascontiguousarray(want[..., :3])))
    assert json.loads(out.stdout.strip().splitlines()[-1])["rays"] == app.backend.ctx.raycount()
