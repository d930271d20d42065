import sys, time
sys.path.insert(0, '.')
from real_time_path_tracing_with_spatiotemporal_filtering_amd.app import make_app
app = make_app(64,64,max_segments=4,iterations=5)
ctx = app.backend.ctx
for _ in range(20): app.drawScene()
ctx.sync()
t=time.perf_counter()
for _ in range(500): app.drawScene()
t1=time.perf_counter(); ctx.sync(); t2=time.perf_counter()
print('host us/frame %.1f  (with final sync %.1f)'%((t1-t)/500*1e6,(t2-t)/500*1e6))
t=time.perf_counter()
for _ in range(500): app.updateScene()
print('updateScene us %.1f'%((time.perf_counter()-t)/500*1e6))
