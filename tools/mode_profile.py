#!/usr/bin/env python3
"""One render mode on config 3 for a rocprofv3 counter pass: python3 tools/mode_profile.py <mode> [bounces]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_scene
mode = sys.argv[1] if len(sys.argv) > 1 else "default"
r, msg, info = build_scene(1920, 1080, 512, 0, 1, 0)
r.settings.render_mode = mode
r.settings.bounces = int(sys.argv[2]) if len(sys.argv) > 2 else 1
r.settings.max_samples = 1 << 30
r.bind_uniforms()
r.render(frames=2, rebind=False); r.finish()
import time
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.15:          # bring the device to its sustained clock
    r.render(frames=32, rebind=False, in_flight=32); r.finish()
r.restart_rendering(); r.bind_uniforms()
r.render(frames=2, rebind=False); r.finish(); r.reset_counters()
r.render(frames=64, rebind=False, in_flight=32); r.finish()
c = r.counters()
print(mode, "ms/frame", c.kernel_ms / c.frames, "Msamples", c.samples / c.frames / 1e6, "pixels/frame", c.pixels / c.frames, "launches", c.launches, "frames", c.frames)
