#!/usr/bin/env python3
"""the render modes on BASELINE config 3 (1x MI355X): kernel ms per accumulation frame, lane utilisation where the kernel
counts it.  usage: python tools/mode_probe.py [--modes a,b] [--fpl 16,32] [bounces ...]
VX_PATHS_KERNEL=generic | packed selects the one-pixel-per-lane / segment re-packing kernels instead of the event-batched one"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

args = sys.argv[1:]
modes = ("default", "no_dda", "raymarch")
fpls = (1, 32)
if "--modes" in args:
    i = args.index("--modes"); modes = tuple(args[i + 1].split(",")); del args[i:i + 2]
if "--fpl" in args:
    i = args.index("--fpl"); fpls = tuple(int(x) for x in args[i + 1].split(",")); del args[i:i + 2]
r, msg, info = bench.build_scene(1920, 1080, 512, 0, 1, 0)
bl = [int(x) for x in args] or [1]
import time
t0 = time.perf_counter()
r.settings.render_mode = "dvr"; r.bind_uniforms()
while time.perf_counter() - t0 < 0.15:          # bring the device to its sustained clock
    r.render(frames=32, rebind=False, in_flight=32); r.finish()
for bounces in bl:
    for mode in modes:
        for P in fpls:
            r.settings.render_mode, r.settings.bounces = mode, bounces
            r.restart_rendering(); r.bind_uniforms()
            r.render(frames=3, rebind=False); r.finish(); r.reset_counters()
            for _ in range(2):
                r.render(frames=P, rebind=False, in_flight=P)
            r.finish()
            c = r.counters()
            util = (c.active_lane_slots / c.lane_slots) if (c.lane_slots and c.active_lane_slots) else ((c.samples / c.lane_slots) if c.lane_slots else float("nan"))
            print(f"{mode:9s} bounces {bounces} fpl {P:2d}: {c.kernel_ms / c.frames:.4f} ms/frame, samples/frame {c.samples // c.frames}, "
                  f"skip steps/frame {c.skip_steps // c.frames}, lane utilisation {util:.3f}", flush=True)
