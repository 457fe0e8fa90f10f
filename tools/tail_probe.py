#!/usr/bin/env python3
"""Diagnostic: frame time and sample count of the DVR kernel as the per-ray step budget grows.
Separates throughput from the latency-bound tail of the longest rays."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import build_scene

layout = int(sys.argv[1]) if len(sys.argv) > 1 else 1
skip = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
r, msg, info = build_scene(1920, 1080, 512, 0, 1, 0)
r.set_layout(layout)
r.settings.dvr_skip_empty = skip
for ms in (25, 50, 100, 200, 400, 800, 1600, 1 << 20):
    r.settings.dvr_max_steps = ms
    r.bind_uniforms()
    r.render(frames=3, rebind=False); r.finish(); r.reset_counters()
    r.render(frames=10, rebind=False); r.finish()
    c = r.counters()
    ms_frame = c.kernel_ms / c.frames
    print(json.dumps(dict(layout=layout, skip=skip, max_steps=ms, ms=round(ms_frame, 4), Msamples=round(c.samples / c.frames / 1e6, 2),
                          gsps=round(c.samples / c.kernel_ms / 1e6, 1), util=round(c.samples / max(c.lane_slots, 1), 3))))
