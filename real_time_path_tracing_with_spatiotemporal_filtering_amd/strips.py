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

    def reach(self, k: int) -> int:
        """rows above/below a pixel that iteration k reads: k for the reference's 3x3 linear-stride taps
        (temporalFiltering.comp.glsl:135); radius 2 with EXT_GAUSS5, stride 2^(k-1) with EXT_POW2_STRIDE."""
        stride = (1 << (k - 1)) if (self.ext_flags & 0x40) else k
        return stride * (2 if (self.ext_flags & 0x20) else 1)

    def __post_init__(self):
        if self.mode not in ("exchange", "redundant"):
            raise ValueError(f"unknown halo mode {self.mode!r}")
        if not (0 <= self.rank < self.world):
            raise ValueError("rank out of range")
        if self.world > self.height:
            raise ValueError("more ranks than rows")

    # ---- ownership
    @staticmethod
    def bounds(height: int, world: int, rank: int):
        return (rank * height) // world, ((rank + 1) * height) // world

    @property
    def own(self):
        return self.bounds(self.height, self.world, self.rank)

    @property
    def halo(self) -> int:
        """rows stored beyond the owned strip on each side."""
        if self.world == 1:
            return 0
        n = self.iterations
        if self.mode == "exchange":
            return max(self.reach(k) for k in range(1, n + 1))
        return sum(self.reach(k) for k in range(1, n + 1))

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
        return self._grow(sum(self.reach(k) for k in range(1, self.iterations + 1)))

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
        r = self.reach(k)
        if up is not None:
            u0, u1 = self.bounds(self.height, self.world, up)
            if u1 - u0 < r or o1 - o0 < r:
                raise ValueError(f"strip shorter than the {r}-row halo of iteration {k}")
            out.append((up, (o0, o0 + r), (o0 - r, o0)))
        if down is not None:
            d0, d1 = self.bounds(self.height, self.world, down)
            if d1 - d0 < r or o1 - o0 < r:
                raise ValueError(f"strip shorter than the {r}-row halo of iteration {k}")
            out.append((down, (o1 - r, o1), (o1, o1 + r)))
        return out


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
    import torch.distributed as dist

    ops = []
    for peer, send_rows, recv_rows in todo:
        ops.append(dist.P2POp(dist.isend, rows_view(*send_rows), peer, group))
        ops.append(dist.P2POp(dist.irecv, rows_view(*recv_rows), peer, group))
    for req in dist.batch_isend_irecv(ops):
        req.wait()
