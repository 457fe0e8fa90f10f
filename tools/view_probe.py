#!/usr/bin/env python3
"""Diagnostic: DVR kernel time per frame of BASELINE config 3 from several camera positions (the cell
order inside a brick favours some view directions; this shows how much)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import build_scene

r, msg, info = build_scene(1920, 1080, 512, 0, 1, 0)
if len(sys.argv) > 1:
    r.set_layout(int(sys.argv[1]))      # 1 cellquad (default), 2 brickf32 + LDS windows
r.settings.dvr_skip_empty = False
views = {"benchmark.json": None, "+x": (1.05, 0.05, 0.02), "-x": (-1.05, 0.03, 0.04), "+y": (0.03, 1.05, 0.05),
         "-z": (0.04, 0.02, -1.05), "+z": (0.02, 0.05, 1.05), "diag": (0.6, 0.6, -0.6), "diag2": (-0.55, 0.65, 0.6)}
tot = 0.0
for name, pos in views.items():
    if pos is not None:
        r.camera.pos = np.asarray(pos, dtype=np.float64)
        r.camera.view = np.zeros(3)
    r.restart_rendering()
    r.bind_uniforms()
    r.render(frames=4, rebind=False); r.finish(); r.reset_counters()
    r.render(frames=64, rebind=False, in_flight=16); r.finish()
    c = r.counters()
    ms = c.kernel_ms / c.frames
    tot += ms
    print(json.dumps(dict(view=name, ms_per_frame=round(ms, 4), Msamples=round(c.samples / c.frames / 1e6, 1),
                          gsps=round(c.samples / c.kernel_ms / 1e6, 1))), flush=True)
print(json.dumps(dict(mean_ms=round(tot / len(views), 4))))
