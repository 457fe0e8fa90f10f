#!/usr/bin/env python3
"""Diagnostic: true cost of every 64x64 tile (each rendered as its own shard) against vx_probe_tile_costs,
and the shard balance the probe-based dealing order achieves in terms of the true costs."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import build_scene
from volxel_amd import tiles

r, msg, info = build_scene(1920, 1080, 512, 0, 1, 0)
for skip in (False, True):
    r.settings.dvr_skip_empty = skip
    r.shard_rank, r.shard_count = 0, 1
    r.bind_uniforms(); r.set_tile_order(None)
    probe = r.probe_tile_costs().astype(np.float64)
    nt = probe.size
    true = np.zeros(nt)
    for t in range(nt):
        r.shard_rank, r.shard_count = t, nt
        r.bind_uniforms(); r.restart_rendering(); r.reset_counters()
        r.render(frames=1, rebind=False); r.finish()
        c = r.counters()
        true[t] = c.samples + c.skip_steps
    out = dict(skip=skip, corr=round(float(np.corrcoef(probe, true)[0, 1]), 4), probe_sum=probe.sum(), true_sum=true.sum())
    for N in (2, 4, 8):
        plain = np.array([true[k::N].sum() for k in range(N)])
        perm = tiles.balanced_order(probe, N)
        bal = np.array([true[perm[k::N]].sum() for k in range(N)])
        ideal = tiles.balanced_order(true, N)
        idl = np.array([true[ideal[k::N]].sum() for k in range(N)])
        out[f"N{N}"] = dict(plain=round(float(plain.max() / plain.mean()), 4), probe=round(float(bal.max() / bal.mean()), 4),
                            with_true_costs=round(float(idl.max() / idl.mean()), 4))
    print(json.dumps(out), flush=True)
    np.save(f"gpurun_out/tile_costs_skip{int(skip)}.npy", np.stack([probe, true]))
