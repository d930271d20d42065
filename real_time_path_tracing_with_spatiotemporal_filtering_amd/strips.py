"""Row-strip sharding of one frame across ranks — new work, the reference is
single-device (``useDeviceGroups`` in its vendored context is dead code, context.hpp:153).

Rank r of R owns frame rows [r*H/R, (r+1)*H/R).  K0/K1/K2 are per-pixel independent and the RNG
seed depends only on absolute (x, y, frame, batch) (raytrace.comp.glsl:297), so any partition
reproduces the single-GPU image bit for bit.  K3 iteration k is a stencil that reads rows y-k,
y, y+k (temporalFiltering.comp.glsl:135) with a *global* border clamp (:136), so a strip needs k
rows of the current colour plane from each neighbour before iteration k:

* ``exchange``  — the real exchange step: k rows per neighbour per iteration, point-to-point
  send/recv (RCCL over xGMI on GPUs, gloo in the CPU tests).  Guide planes (depth, id, world
  position) never travel: each rank rasterises its strip +- N rows itself.
* ``redundant`` — no data-path message at all: every rank traces and filters the extra rows it
  will need (sum_{j>k} j rows per side at iteration k; 15 rows for N = 5).

Both produce the identical frame; the choice is a latency/compute trade (messages are k*W*16 B,
i.e. latency-bound).

The strips need not be equally tall: ``splits`` (world + 1 ascending row numbers, 0 ... H) replaces the equal division
where the rows of a frame do not cost the same (the 1.15 M-triangle lattice: paths in the middle of the frame bounce twice
as often as at its top and bottom; equal strips leave the job at 4.9x of one GPU on eight).  ``balanced_splits`` moves the
boundaries from the ranks' measured frame times; any division reproduces the single-GPU image bit for bit.
"""
from __future__ import annotations

from dataclasses import dataclass


@dataclass(frozen=True)
class StripPlan:
    height: int
    world: int
    rank: int
    iterations: int          # maxWaveletIteration (main.cpp:55)
    mode: str = "exchange"   # "exchange" | "redundant"
    ext_flags: int = 0       # RTPT_FLAG_EXT_* (extension modes change how far the taps of iteration k reach)
    splits: tuple = ()       # () = equal strips; else world + 1 ascending rows, splits[r] .. splits[r + 1] is rank r's strip

    def reach(self, k: int) -> int:
        """rows above/below a pixel that iteration k reads: k for the reference's 3x3 linear-stride taps
        (temporalFiltering.comp.glsl:135); radius 2 with EXT_GAUSS5, stride 2^(k-1) with EXT_POW2_STRIDE."""
        stride = (1 << (k - 1)) if (self.ext_flags & 0x40) else k
        return stride * (2 if (self.ext_flags & 0x20) else 1)

    @property
    def svgf_pad(self) -> int:
        """RTPT_FLAG_EXT_SVGF_VARIANCE (with _EXT_VARIANCE): a pixel with a short moment history takes its variance from the
        7x7 neighbourhood of the traced frame, so the traced rows must reach 3 rows beyond every row whose variance
        iteration 1 reads"""
        return 3 if (self.ext_flags & 0x900) == 0x900 else 0

    def __post_init__(self):
        if self.mode not in ("exchange", "redundant"):
            raise ValueError(f"unknown halo mode {self.mode!r}")
        if not (0 <= self.rank < self.world):
            raise ValueError("rank out of range")
        if self.world > self.height:
            raise ValueError("more ranks than rows")
        if self.splits:
            sp = tuple(int(v) for v in self.splits)
            object.__setattr__(self, "splits", sp)
            if len(sp) != self.world + 1 or sp[0] != 0 or sp[-1] != self.height or any(b <= a for a, b in zip(sp, sp[1:])):
                raise ValueError(f"splits must be {self.world + 1} ascending rows from 0 to {self.height}, every strip at least one row: {sp}")

    # ---- ownership
    @staticmethod
    def bounds(height: int, world: int, rank: int, splits=()):
        if splits:
            return int(splits[rank]), int(splits[rank + 1])
        return (rank * height) // world, ((rank + 1) * height) // world

    def rows_of(self, rank: int):
        """rows [a, b) rank `rank` owns."""
        return self.bounds(self.height, self.world, rank, self.splits)

    @property
    def own(self):
        return self.rows_of(self.rank)

    @property
    def halo(self) -> int:
        """rows stored beyond the owned strip on each side."""
        if self.world == 1:
            return 0
        n = self.iterations
        if self.mode == "exchange":
            return max(max(self.reach(k) for k in range(1, n + 1)), self.reach(1) + self.svgf_pad)
        return sum(self.reach(k) for k in range(1, n + 1)) + self.svgf_pad

    @property
    def stored(self):
        o0, o1 = self.own
        return max(0, o0 - self.halo), min(self.height, o1 + self.halo)

    def _grow(self, rows: int):
        o0, o1 = self.own
        return max(0, o0 - rows), min(self.height, o1 + rows)

    # ---- per-pass row ranges
    def gbuffer_rows(self):
        # guides are needed wherever a filter tap of a computed row can land
        return self.stored

    def gradient_rows(self):
        return self.own

    def raytrace_rows(self):
        if self.world == 1 or self.mode == "exchange":
            return self.own
        return self._grow(sum(self.reach(k) for k in range(1, self.iterations + 1)) + self.svgf_pad)

    def filter_rows(self, k: int):
        """rows iteration k must produce on this rank."""
        if self.world == 1 or self.mode == "exchange":
            return self.own
        remaining = sum(self.reach(j) for j in range(k + 1, self.iterations + 1))
        return self._grow(remaining)

    # ---- exchange mode: what travels before iteration k
    def neighbours(self):
        up = self.rank - 1 if self.rank > 0 else None
        down = self.rank + 1 if self.rank + 1 < self.world else None
        return up, down

    def exchange_rows(self, k: int):
        """[(peer, send_rows, recv_rows)] for iteration k; rows are frame-coordinate [a, b) ranges.
        A strip shorter than k rows would need rows from beyond its neighbour; such plans are
        rejected (use fewer ranks or the redundant mode)."""
        if self.world == 1 or self.mode != "exchange":
            return []
        o0, o1 = self.own
        out = []
        up, down = self.neighbours()
        r = self.reach(k) + (self.svgf_pad if k == 1 else 0)   # iteration 1's variance taps look 3 traced rows further
        if up is not None:
            u0, u1 = self.rows_of(up)
            if u1 - u0 < r or o1 - o0 < r:
                raise ValueError(f"strip shorter than the {r}-row halo of iteration {k}")
            out.append((up, (o0, o0 + r), (o0 - r, o0)))
        if down is not None:
            d0, d1 = self.rows_of(down)
            if d1 - d0 < r or o1 - o0 < r:
                raise ValueError(f"strip shorter than the {r}-row halo of iteration {k}")
            out.append((down, (o1 - r, o1), (o1, o1 + r)))
        return out


def balanced_splits(splits, cost, min_rows: int = 1):
    """New strip boundaries from the ranks' measured frame times: ``cost[r]`` is what rank r spent on rows
    [splits[r], splits[r + 1]) (any unit), taken as spread evenly over those rows; the new boundaries cut the resulting
    piecewise-linear cumulative cost into equal parts.  A strip's time also holds work that does not scale with its rows
    (the halo rows of the redundant mode, launches), so one application does not land exactly on equal times: apply it
    again to the times measured with the new boundaries — equal times are its fixed point; two or three rounds settle
    within a row or two.  Every strip keeps at least ``min_rows`` rows (the exchange mode needs as many as the longest
    reach).  Pure arithmetic on floats and ints, mirrored by host/strips.cpp balanced_splits (same operations in the
    same order, so the two hosts agree on the rows)."""
    sp = [int(v) for v in splits]
    world = len(sp) - 1
    if world < 1 or len(cost) != world or any(b <= a for a, b in zip(sp, sp[1:])):
        raise ValueError("splits must be world + 1 ascending rows and cost one value per rank")
    if any(not (c > 0.0) for c in cost):
        raise ValueError("every rank's cost must be positive")
    height = sp[-1] - sp[0]
    if world * min_rows > height:
        raise ValueError("min_rows does not fit the frame")
    total = 0.0
    for c in cost:
        total += float(c)
    new = [sp[0]]
    r, acc = 0, 0.0   # acc = cost of the rows above splits[r]
    for j in range(1, world):
        target = total * j / world
        while r < world - 1 and acc + float(cost[r]) < target:
            acc += float(cost[r])
            r += 1
        y = sp[r] + (target - acc) * (sp[r + 1] - sp[r]) / float(cost[r])
        new.append(int(y + 0.5))
    new.append(sp[-1])
    for j in range(1, world):
        new[j] = max(new[j], new[j - 1] + min_rows)
    for j in range(world - 1, 0, -1):
        new[j] = min(new[j], new[j + 1] - min_rows)
    return tuple(new)


class StripBalancer:
    """Strip boundaries from a per-row cost profile that the ranks' measured frame times keep correcting.

    ``balanced_splits`` spreads a strip's time evenly over its rows.  Where the cost of a row changes faster than a strip is
    tall — the lattice of configs[4] puts a layer of boxes every ~216 rows of the 4K frame, a strip of an 8-rank job is 270 —
    that model moves a boundary by 18 rows and finds 0.1 ms behind it (profiles/r04_instanced_balanced_strips.json): the cut
    oscillates.  Here the shape inside a strip comes from a profile (``profile[y]``: rays traced in row y, measured with the
    ray counter's row window, rtpt_set_count_rows; ``floor`` = what a row costs besides its rays, in the same unit), and
    ``update`` rescales the rows of every strip so that they add up to what the strip was measured to take (iterative
    proportional fitting: the profile keeps its shape inside a strip, the measurements set its level strip by strip).
    Plain float arithmetic in a fixed order: every rank computes the same rows from the same gathered numbers."""

    def __init__(self, height: int, world: int, min_rows: int = 1, profile=None, floor: float = 0.0):
        if world < 1 or world * min_rows > height:
            raise ValueError("min_rows does not fit the frame")
        self.height, self.world, self.min_rows = int(height), int(world), int(min_rows)
        if profile is None:
            self.density = [1.0] * self.height
        else:
            if len(profile) != self.height:
                raise ValueError("one profile value per row")
            self.density = [float(v) + float(floor) for v in profile]
            if any(not (v > 0.0) for v in self.density):
                raise ValueError("every row must cost something (raise `floor`)")

    def update(self, splits, cost):
        """the ranks' measured times for the strips `splits`: rows [splits[r], splits[r + 1]) took cost[r]"""
        sp = [int(v) for v in splits]
        if len(sp) != self.world + 1 or len(cost) != self.world or sp[0] != 0 or sp[-1] != self.height or \
                any(b <= a for a, b in zip(sp, sp[1:])) or any(not (c > 0.0) for c in cost):
            raise ValueError("splits must be world + 1 ascending rows from 0 to height and cost one positive value per rank")
        d = self.density
        for r in range(self.world):
            s = 0.0
            for y in range(sp[r], sp[r + 1]):
                s += d[y]
            f = float(cost[r]) / s
            for y in range(sp[r], sp[r + 1]):
                d[y] *= f

    def splits(self):
        """world + 1 rows cutting the profile's cumulative cost into equal parts (each cut at the nearer row)"""
        d, world = self.density, self.world
        total = 0.0
        for v in d:
            total += v
        new, y, acc = [0], 0, 0.0
        for j in range(1, world):
            target = total * j / world
            while y < self.height - 1 and acc + d[y] <= target:
                acc += d[y]
                y += 1
            new.append(y + 1 if target - acc > 0.5 * d[y] else y)
        new.append(self.height)
        for j in range(1, world):
            new[j] = max(new[j], new[j - 1] + self.min_rows)
        for j in range(world - 1, 0, -1):
            new[j] = min(new[j], new[j + 1] - self.min_rows)
        return tuple(new)


def _post(sends, recvs, group=None):
    """point-to-point exchange of [(tensor, peer)] lists.  RCCL ("nccl") takes device tensors and orders the transfers
    on the current stream.  gloo — the CPU tests, and the rehearsal of several ranks on ONE GPU, where RCCL refuses to
    run — only moves host memory: handing it a device tensor makes its TCP transport read the allocation through the
    PCIe aperture with no stream ordering at all (a race that returns stale rows), so device tensors are staged
    through host copies made and consumed on the current stream."""
    import torch
    import torch.distributed as dist
    if not sends and not recvs:
        return
    staged = dist.get_backend(group) == "gloo" and any(t.is_cuda for t, _ in list(sends) + list(recvs))
    ops, landing = [], []
    for t, peer in sends:
        ops.append(dist.P2POp(dist.isend, t.cpu() if staged and t.is_cuda else t, peer, group))  # .cpu() waits for the stream
    for t, peer in recvs:
        if staged and t.is_cuda:
            h = torch.empty(t.shape, dtype=t.dtype, device="cpu")
            landing.append((t, h))
            ops.append(dist.P2POp(dist.irecv, h, peer, group))
        else:
            ops.append(dist.P2POp(dist.irecv, t, peer, group))
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    for t, h in landing:
        t.copy_(h)


def exchange_halo(plan: StripPlan, k: int, rows_view, group=None):
    """Exchange the k-row colour halos for iteration k.

    ``rows_view(y0, y1)`` returns a contiguous torch tensor viewing frame rows [y0, y1) of the
    iteration's *input* colour plane on this rank (device memory on GPUs).  Uses
    torch.distributed point-to-point ops: backend "nccl" is RCCL on ROCm (xGMI), "gloo" on CPU.
    The ops are enqueued on the current stream, i.e. in order with the filter kernels.
    """
    todo = plan.exchange_rows(k)
    if not todo:
        return
    _post([(rows_view(*send_rows), peer) for peer, send_rows, _ in todo],
          [(rows_view(*recv_rows), peer) for peer, _, recv_rows in todo], group)


# ------------------------------------------------------------------------------------------
# history under camera motion: which rows of the previous frame can a strip's final pass fetch?
# ------------------------------------------------------------------------------------------
def _mat(m16):
    import numpy as np
    return np.asarray(m16, np.float64).reshape(4, 4).T  # column-major float[16] -> row-major 4x4


def reprojection_rows(ubo, width: int, height: int, rows, bounds, z_near: float = 0.1, pad: int = 2):
    """Frame rows [a, b) of the PREVIOUS frame that the final pass of frame rows `rows` = (y0, y1) can fetch history
    from (temporalFiltering.comp.glsl:213-239,:253), given the scene's world-space bounds ((min xyz), (max xyz)).

    The G-buffer's world position of pixel (x, y) lies on that pixel's centre ray at a view depth z between the
    nearest and the farthest corner of the bounds; its previous-frame pixel row is
        ppy = ((P_prev V_prev p).y / (...).w * 0.5 + 0.5) * H        (worldToPixel, :178-189)
    (with model == modelPrev the previous world position IS the current one; `bounds` are those of the POSED scene)
    which is a ratio of functions affine in x, in y and in z separately — monotone along each of the three axes — so
    over the box [0, W-1] x [y0, y1-1] x [z_lo, z_hi] it takes its extremes at the 8 corners.  `pad` rows cover the
    binary32 rounding of the device arithmetic (the kernel's value is within a small fraction of a pixel of this
    float64 one).  A corner behind the previous camera makes the projection unbounded: the whole frame is returned.
    Background pixels (id 0) fetch their own pixel (:215-217), so `rows` itself is always part of the result."""
    import numpy as np
    y0, y1 = rows
    if list(ubo.model) != list(ubo.modelPrev):
        # A model matrix that changed since the previous frame: the shader takes the CURRENT world position's area-ratio
        # barycentrics against the PREVIOUS frame's triangle (both vertex sets come from LUTprev,
        # temporalFiltering.comp.glsl:223-233) — a point off that triangle's plane, whose "barycentrics" do not sum to
        # one — so the previous position is not M_prev M^-1 p and no ray/depth argument bounds it: the whole frame.
        return 0, height
    V, P = _mat(ubo.view), _mat(ubo.proj)
    PVp = _mat(ubo.projPrev) @ _mat(ubo.viewPrev)
    R, t = V[:3, :3], V[:3, 3]
    org = -R.T @ t
    lo, hi = (np.asarray(b, np.float64) for b in bounds)
    corners = np.array([[x, y, z] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])])
    depth = -(corners @ R.T + t)[:, 2]
    z_lo = max(float(z_near), float(depth.min()))
    z_hi = max(z_lo, float(depth.max()))
    a, b = min(y0, height - 1), max(y0, min(y1, height) - 1)
    ppy = []
    for x in (0, width - 1):
        for y in (a, b):
            nx = (2.0 * (x + 0.5) - width) / width
            ny = (2.0 * (y + 0.5) - height) / height
            d = R.T @ np.array([nx / P[0, 0], ny / P[1, 1], -1.0])
            for z in (z_lo, z_hi):
                clip = PVp @ np.append(org + z * d, 1.0)
                if not clip[3] > 1e-6:
                    return 0, height
                ppy.append((clip[1] / clip[3] * 0.5 + 0.5) * height)
    n0 = int(np.floor(min(ppy))) - pad
    n1 = int(np.floor(max(ppy))) + 1 + pad
    # background pixels fetch their own pixel: the strip's own rows always belong to the range (they are local anyway)
    n0, n1 = min(n0, y0), max(n1, y1)
    return max(0, min(n0, height)), max(0, min(n1, height))


def history_exchange_plan(height: int, world: int, needs, splits=()):
    """needs[q] = (a, b): previous-frame rows rank q's final pass can fetch.  Returns, per rank r,
    [(peer, 'send' | 'recv', (y0, y1))] — r sends the part of ITS OWN rows that peer needs, receives the part of the
    peer's rows it needs itself.  Every rank computes the same table from the same camera matrices: no negotiation."""
    table = [[] for _ in range(world)]
    own = [StripPlan.bounds(height, world, r, splits) for r in range(world)]
    for r in range(world):
        for q in range(world):
            if q == r:
                continue
            s0, s1 = max(own[r][0], needs[q][0]), min(own[r][1], needs[q][1])
            if s1 > s0:
                table[r].append((q, "send", (s0, s1)))
            g0, g1 = max(own[q][0], needs[r][0]), min(own[q][1], needs[r][1])
            if g1 > g0:
                table[r].append((q, "recv", (g0, g1)))
    return table


def exchange_history(plan: StripPlan, needs, prev_rows_view, full, group=None) -> int:
    """Move the previous frame's rows between ranks so that `full` (a [H, W, 4] tensor on every rank) holds rows
    needs[plan.rank] of it; `prev_rows_view(y0, y1)` views rows of this rank's finished strip.  Returns bytes sent."""
    r = plan.rank
    o0, o1 = plan.own
    a, b = max(o0, needs[r][0]), min(o1, needs[r][1])
    if b > a:
        full[a:b].copy_(prev_rows_view(a, b))
    sends, recvs, sent = [], [], 0
    for peer, what, (y0, y1) in history_exchange_plan(plan.height, plan.world, needs, plan.splits)[r]:
        if what == "send":
            t = prev_rows_view(y0, y1)
            sent += t.numel() * t.element_size()
            sends.append((t, peer))
        else:
            recvs.append((full[y0:y1], peer))
    _post(sends, recvs, group)
    return sent


# ------------------------------------------------------------------------------------------
# presenting the frame: the strips meet on one rank
# ------------------------------------------------------------------------------------------
def gather_frame(plan: StripPlan, mine, full, root: int = 0, group=None) -> int:
    """The reference presents every frame (main.cpp:1338-1361: `image` blitted to the swapchain image); with the frame
    sharded by rows, the presenting rank `root` first has to hold all of it.  `mine` views the rows this rank owns of
    the finished frame ([rows, W, C]: the float image, or its swapchain-format conversion, rtpt_present), `full` is the
    [H, W, C] image on the root.  Strips may differ by a row, so this is a group of point-to-point messages rather than
    a gather collective: the root posts its R-1 receives together (RCCL then drives every xGMI link into the root at
    once: 7 x 16.6 MB at 4K f32, a quarter of that in swapchain format), every other rank posts one send.  Enqueued on
    the current stream.  Returns the bytes this rank put on or took off the wire."""
    if plan.world == 1:
        return 0
    o0, o1 = plan.own
    if plan.rank == root:
        recvs = []
        for r in range(plan.world):
            if r != root:
                a, b = plan.rows_of(r)
                recvs.append((full[a:b], r))
        _post([], recvs, group)
        return sum(t.numel() * t.element_size() for t, _ in recvs)
    _post([(mine, root)], [], group)
    return mine.numel() * mine.element_size()
