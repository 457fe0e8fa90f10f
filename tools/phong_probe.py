#!/usr/bin/env python3
"""dvr_phong on BASELINE config 4: work counters and kernel time (1x MI355X)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

r, msg, info = bench.build_scene(1920, 1080, 512, 0, 1, 0)
if len(sys.argv) > 1:
    r.set_layout(int(sys.argv[1]))
for mode in ("dvr", "dvr_phong"):
    for jitter in (True,):
        for P in (1, 32):
            r.settings.render_mode = mode
            r.settings.dvr_jitter = jitter
            r.restart_rendering(); r.bind_uniforms()
            r.render(frames=4, rebind=False); r.finish(); r.reset_counters()
            for _ in range(4):
                r.render(frames=P, rebind=False, in_flight=P)
            r.finish()
            c = r.counters()
            print(f"{mode:10s} jitter {int(jitter)} fpl {P:2d}: {c.kernel_ms / c.frames:.4f} ms/frame, samples/frame {c.samples // c.frames}, "
                  f"grad samples/frame {c.grad_samples // c.frames} ({c.grad_samples / max(c.samples, 1):.3f}), "
                  f"lane slots/frame {c.lane_slots // c.frames}, gathers/frame {c.gathers // c.frames}, lds reads/frame {c.lds_reads // c.frames}", flush=True)
