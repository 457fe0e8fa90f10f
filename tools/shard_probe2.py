#!/usr/bin/env python3
"""Diagnostic: per-frame kernel time of one rank of an N-GPU run (every rank measured in turn on this
GPU) for several frames-per-launch settings.  Predicts image-tile strong scaling without the gather."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_scene

fpls = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "32,64").split(",")]
shards = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,8").split(",")]
base = {}
r, msg, info = build_scene(1920, 1080, 512, 0, 1, 0)
r.settings.dvr_skip_empty = False
for n in shards:
    ranks = range(n)
    for fpl in fpls:
        per_rank = []
        for rank in ranks:
            r.shard_rank, r.shard_count = rank, n
            r.restart_rendering()
            r.bind_uniforms()
            if os.environ.get("SP_BALANCE", "1") != "0" and n > 1:
                r.balance_tiles()
            else:
                r.set_tile_order(None)
            r.bind_uniforms()
            r.render(frames=4, rebind=False); r.finish(); r.reset_counters()
            r.render(frames=2 * fpl, rebind=False, in_flight=fpl); r.finish()   # warm the pipes
            r.reset_counters()
            import time
            t0 = time.perf_counter()
            r.render(frames=4 * fpl, rebind=False, in_flight=fpl); r.finish()
            wall = (time.perf_counter() - t0) * 1e3 / (4 * fpl)
            c = r.counters()
            per_rank.append((c.kernel_ms / c.frames, wall))
        worst_k = max(k for k, _ in per_rank); worst_w = max(w for _, w in per_rank)
        base.setdefault(fpl, worst_w if n == 1 else None)
        print(json.dumps(dict(shards=n, frames_per_launch=fpl, worst_kernel_ms=round(worst_k, 4),
                              worst_wall_ms=round(worst_w, 4), mean_kernel_ms=round(sum(k for k, _ in per_rank) / len(per_rank), 4),
                              speedup_vs_1=round(base[fpl] / worst_w, 2) if base.get(fpl) else None)), flush=True)
