#!/usr/bin/env python3
"""Diagnostic: what one GPU of an N-GPU run would do -- frame time of shard 0 of N on this GPU.
Predicts image-tile strong scaling (without the gather) from a single device."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_scene

skip = bool(int(sys.argv[1])) if len(sys.argv) > 1 else False
in_flight = int(sys.argv[2]) if len(sys.argv) > 2 else 1
base = None
for n in (1, 2, 4, 8):
    worst = 0.0; tot = 0
    for rank in ((0,) if n == 1 else (0, n // 2, n - 1)):
        r, msg, info = build_scene(1920, 1080, 512, rank, n, 0)
        r.settings.dvr_skip_empty = skip
        r.bind_uniforms()
        r.render(frames=4, rebind=False); r.finish(); r.reset_counters()
        r.render(frames=32, rebind=False, in_flight=in_flight); r.finish()
        c = r.counters()
        ms = c.kernel_ms / c.frames
        worst = max(worst, ms); tot = c.samples / c.frames
        r.close()
    if base is None: base = worst
    print(json.dumps(dict(skip=skip, in_flight=in_flight, shards=n, worst_rank_ms=round(worst, 4), predicted_speedup=round(base / worst, 2),
                          Msamples_rank=round(tot / 1e6, 1))))
