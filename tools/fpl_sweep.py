#!/usr/bin/env python3
"""frames per launch vs kernel time per frame on BASELINE config 3, jitter on / off (1x MI355X).
usage: python tools/fpl_sweep.py [--volume 512] [--width 1920 --height 1080]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--volume", type=int, default=512)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--layout", type=int, default=None)
ap.add_argument("--frames", type=int, default=128)
ap.add_argument("--fpl", type=str, default="1,2,4,8,16,32,64")
ap.add_argument("--jitter", type=str, default="1,0")
ap.add_argument("--sync", type=float, default=-1.0, help="wait for every launch, then sleep this many ms before the next (isolated launches)")
a = ap.parse_args()
r, msg, info = bench.build_scene(a.width, a.height, a.volume, 0, 1, 0)
if a.layout is not None:
    r.set_layout(a.layout)
print(info)
for jitter in [bool(int(x)) for x in a.jitter.split(",")]:
    r.settings.dvr_jitter = jitter
    for P in [int(x) for x in a.fpl.split(",")]:
        r.restart_rendering(); r.bind_uniforms()
        r.render(frames=4, rebind=False); r.finish(); r.reset_counters()
        n = max(P, min(a.frames, 32 * P))
        done = 0
        while done < n:
            r.render(frames=P, rebind=False, in_flight=P); done += P
            if a.sync >= 0.0:
                r.finish()
                if a.sync > 0.0:
                    import time
                    time.sleep(a.sync * 1e-3)
        r.finish()
        c = r.counters()
        print(f"jitter {int(jitter)} frames/launch {P:2d}: {c.kernel_ms / c.frames:.4f} ms/frame kernel, "
              f"{c.merge_ms / c.frames:.4f} ms/frame blend, {c.samples / c.kernel_ms / 1e6:.1f} Gsamples/s in-kernel, "
              f"{c.samples // c.frames} samples/frame, skip/direct {c.skip_steps // c.frames}, launches {c.launches} (max {c.max_launch_frames} frames)", flush=True)
