#!/usr/bin/env python3
"""Diagnostic: samples per rank (8 shards) with the default tile dealing and with the balanced order from
vx_probe_tile_costs, plus per-rank kernel time (32 frames per launch)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import build_scene
from volxel_amd import tiles

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
r, msg, info = build_scene(1920, 1080, 512, 0, 1, 0)
r.settings.dvr_skip_empty = False
costs = r.probe_tile_costs()
perm = tiles.balanced_order(costs, N)
for label, order in (("default", None), ("balanced", perm)):
    samples, ms = [], []
    for rank in range(N):
        r.shard_rank, r.shard_count = rank, N
        r.bind_uniforms()
        r.set_tile_order(order)
        r.bind_uniforms()
        r.render(frames=4, rebind=False); r.finish(); r.reset_counters()
        r.render(frames=64, rebind=False, in_flight=32); r.finish()
        c = r.counters()
        samples.append(c.samples / c.frames / 1e6); ms.append(c.kernel_ms / c.frames)
    s, m = np.array(samples), np.array(ms)
    print(json.dumps(dict(order=label, Msamples_max_over_mean=round(float(s.max() / s.mean()), 4),
                          ms_max=round(float(m.max()), 4), ms_mean=round(float(m.mean()), 4),
                          ms_max_over_mean=round(float(m.max() / m.mean()), 4), Msamples=[round(float(x), 2) for x in s])), flush=True)
print(json.dumps(dict(probe_cost_sum=int(costs.sum()), tiles=int(costs.size), zero_cost_tiles=int((costs == 0).sum()))))
