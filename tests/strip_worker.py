"""worker of tests/test_bench_gpu.py::test_gloo_ranks_on_one_gpu_*: one rank of a multi-rank PathTracingApplication on
GPU 0 (gloo carries the messages: RCCL refuses two ranks on one device), dumping the rows it owns of every frame.
python -m torch.distributed.run --nproc-per-node R tests/strip_worker.py <out_dir> <mode> <flags> <keys,keys,...> W H [present [in_flight]]
(present = rgba8 | f32: rank 0 also dumps the frame it assembled, app._present)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

torch.cuda.set_device(0)
from real_time_path_tracing_with_spatiotemporal_filtering_amd import abi  # noqa: E402
from real_time_path_tracing_with_spatiotemporal_filtering_amd.app import make_app  # noqa: E402

out_dir, mode, flags, keys, W, H = sys.argv[1], sys.argv[2], int(sys.argv[3], 0), sys.argv[4].split(","), int(sys.argv[5]), int(sys.argv[6])
present = (sys.argv[7] if len(sys.argv) > 7 else None) or None
in_flight = int(sys.argv[8]) if len(sys.argv) > 8 else 1   # 2: app.PipelinedBackend (multi-rank runs only; the single-rank reference stays serial)
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
if world > 1:
    dist.init_process_group("gloo", rank=rank, world_size=world)
app = make_app(W, H, max_segments=3, iterations=3, rank=rank, world=world, mode=mode, flags=flags, torch_planes=world > 1,
               present=present, frames_in_flight=in_flight if world > 1 else 1)
done = (lambda: app.backend.prev) if in_flight == 2 and world > 1 else (lambda: app.backend)   # the backend that ended the last frame
frames, shown, sent = [], {}, 0
for i, k in enumerate(keys):
    app.drawScene(tuple(k))
    o0, o1 = app.plan.own
    frames.append(done().readback_rows(abi.PLANE_PREVIOUS, o0, o1).copy())
    sent += app.history_bytes_sent
    if present and rank == 0 and (present == "rgba8" or world > 1):
        app.present_sync()
        torch.cuda.synchronize()
        shown[f"shown_{i}"] = app.presented_image().cpu().numpy()
np.savez(os.path.join(out_dir, f"w{world}_r{rank}.npz"), *frames, rays=np.array([sum(b.ctx.raycount() for b in getattr(app.backend, 'be', [app.backend]))]), sent=np.array([sent]),
         **shown)
app.backend.close()
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
