# serial vs two independent contexts (upper bound) vs PipelinedBackend, same process
import sys, time
sys.path.insert(0, '.')
import torch; torch.cuda.is_available()
from real_time_path_tracing_with_spatiotemporal_filtering_amd.app import make_app
W, H = 3840, 2160
def run(app, n, sync):
    for _ in range(10): app.drawScene()
    sync()
    t = time.perf_counter()
    for _ in range(n): app.drawScene()
    t_host = time.perf_counter() - t
    sync()
    return (time.perf_counter() - t) / n * 1e3, t_host / n * 1e3
for (rank, world) in [(0, 1), (3, 8), (1, 4)]:
    kw = dict(max_segments=4, iterations=5, rank=rank, world=world, mode="redundant", torch_planes=False)
    a = make_app(W, H, **kw)
    s, hs = run(a, 400, a.backend.ctx.sync)
    a.backend.close()
    p = make_app(W, H, frames_in_flight=2, **kw)
    q, hq = run(p, 400, p.backend.sync)
    p.backend.close()
    print(f"strip {rank}/{world}: serial {s:.4f} ms (host {hs:.4f}), pipelined {q:.4f} ms (host {hq:.4f}), x{s/q:.3f}")
