import sys
sys.path.insert(0, '.')
import torch; torch.cuda.is_available()
from real_time_path_tracing_with_spatiotemporal_filtering_amd.app import make_app
apps = {}
for seg in (4, 32):
    for fl in (0, 2):
        app = make_app(3840, 2160, max_segments=seg, iterations=1, flags=fl)
        app.updateScene(); app.drawVisbilityBuffer()
        apps[(seg, fl)] = app
for rnd in range(2):
    for (seg, fl), app in apps.items():
        ctx = app.backend.ctx
        for _ in range(2): app.drawSceneToImage(); app.drawVisbilityBuffer()
        ctx.sync(); ctx.timing_enable(True)
        for _ in range(8): app.drawSceneToImage(); app.drawVisbilityBuffer()
        tm = ctx.timing_collect(); ctx.timing_enable(False)
        print('round', rnd, 'seg', seg, 'flags', fl, 'pathtrace %.1f us gbuffer %.1f us' % (tm['k_pathtrace'][0] / tm['k_pathtrace'][1] * 1e3, tm['k_gbuffer'][0] / tm['k_gbuffer'][1] * 1e3))
