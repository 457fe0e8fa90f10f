#!/usr/bin/env python3
"""the reference render modes on BASELINE config 3 (1x MI355X): kernel ms per accumulation frame.
usage: python tools/mode_probe.py [bounces ...]      VX_PATHS_KERNEL=packed selects the kernel that re-packs path segments through LDS (vx_paths.hpp)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

r, msg, info = bench.build_scene(1920, 1080, 512, 0, 1, 0)
bl = [int(x) for x in sys.argv[1:]] or [1]
for bounces in bl:
    for mode in ("default", "no_dda", "raymarch"):
        for P in (1, 16):
            r.settings.render_mode, r.settings.bounces = mode, bounces
            r.restart_rendering(); r.bind_uniforms()
            r.render(frames=3, rebind=False); r.finish(); r.reset_counters()
            for _ in range(2):
                r.render(frames=P, rebind=False, in_flight=P)
            r.finish()
            c = r.counters()
            print(f"{mode:9s} bounces {bounces} fpl {P:2d}: {c.kernel_ms / c.frames:.4f} ms/frame, samples/frame {c.samples // c.frames}, "
                  f"skip steps/frame {c.skip_steps // c.frames}", flush=True)
