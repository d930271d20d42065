# upper bound for n frames in flight: n independent contexts
import sys, time
sys.path.insert(0, '.')
import torch; torch.cuda.is_available()
from real_time_path_tracing_with_spatiotemporal_filtering_amd.app import make_app
W, H = 3840, 2160
for (rank, world) in [(0, 1), (3, 8)]:
    kw = dict(max_segments=4, iterations=5, rank=rank, world=world, mode="redundant", torch_planes=False)
    apps = [make_app(W, H, **kw) for _ in range(4)]
    for a in apps:
        for _ in range(20): a.drawScene()
        a.backend.ctx.sync()
    for n in (1, 2, 3, 4):
        frames = 1200
        t = time.perf_counter()
        for i in range(frames): apps[i % n].drawScene()
        for a in apps: a.backend.ctx.sync()
        dt = time.perf_counter() - t
        print(f"strip {rank}/{world}: {n} contexts {dt/frames*1e3:.4f} ms/frame")
    for a in apps: a.backend.close()
