# long paths at 4K: single-launch tile kernel vs queued segment windows
import sys
sys.path.insert(0, '.')
import torch; torch.cuda.is_available()
from real_time_path_tracing_with_spatiotemporal_filtering_amd import abi
from real_time_path_tracing_with_spatiotemporal_filtering_amd.app import make_app
for seg in (8, 16, 32):
    for fl in (abi.FLAG_SINGLE_LAUNCH_PATHS, 0):
        app = make_app(3840, 2160, max_segments=seg, iterations=1, flags=fl)
        app.updateScene(); app.drawVisbilityBuffer()
        ctx = app.backend.ctx
        for _ in range(3): app.drawSceneToImage()
        ctx.sync(); ctx.reset_counters(); ctx.timing_enable(True)
        for _ in range(10): app.drawSceneToImage()
        tm = ctx.timing_collect(); ctx.timing_enable(False)
        us = tm['k_pathtrace'][0] / tm['k_pathtrace'][1] * 1e3
        print('segments', seg, 'queued' if fl == 0 else 'single', '%.1f us' % us, '%.1f Gray/s' % (ctx.raycount() / 10 / us / 1e3))
        app.backend.close()
