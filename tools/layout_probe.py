#!/usr/bin/env python3
"""the path-traced render modes on each device layout, at BASELINE config 3 (512^3, 1920x1080) and on the
config-5 volume (1024^3, 3840x2160): kernel ms per accumulation frame and the device memory the layout takes.
usage: python tools/layout_probe.py [--volume 512,1024] [--fpl 32]   (1x MI355X)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

args = sys.argv[1:]
vols = (512, 1024)
fpl = 32
if "--volume" in args:
    i = args.index("--volume"); vols = tuple(int(x) for x in args[i + 1].split(",")); del args[i:i + 2]
if "--fpl" in args:
    i = args.index("--fpl"); fpl = int(args[i + 1]); del args[i:i + 2]
import torch  # noqa: E402  (device memory readings only)
LAYOUTS = {0: "reference", 1: "cellquad", 2: "brickf32", 3: "auto"}
for n in vols:
    W, H = (1920, 1080) if n <= 512 else (3840, 2160)
    free0, total = torch.cuda.mem_get_info()
    r, msg, info = bench.build_scene(W, H, n, 0, 1, 0)
    free1, _ = torch.cuda.mem_get_info()
    print(f"# volume {n}^3, {W}x{H}: reference textures + brickf32 (built at upload) + framebuffers take {(free0 - free1) / 2**30:.2f} GiB", flush=True)
    t0 = time.perf_counter()
    r.settings.render_mode = "dvr"; r.bind_uniforms()
    while time.perf_counter() - t0 < 0.15:
        r.render(frames=fpl, rebind=False, in_flight=fpl); r.finish()
    for layout in (0, 2, 1):          # cellquad last: its build is the big allocation
        for mode in ("default", "no_dda", "raymarch"):
            r.set_layout(layout)
            r.settings.render_mode, r.settings.bounces = mode, 1
            r.restart_rendering(); r.bind_uniforms()
            r.render(frames=3, rebind=False); r.finish(); r.reset_counters()
            for _ in range(2):
                r.render(frames=fpl, rebind=False, in_flight=fpl)
            r.finish()
            c = r.counters()
            free2, _ = torch.cuda.mem_get_info()
            print(f"{n}^3 {LAYOUTS[layout]:9s} {mode:9s}: {c.kernel_ms / c.frames:.4f} ms/frame, samples/frame {c.samples // c.frames}, "
                  f"DDA steps/frame {c.skip_steps // c.frames}, device memory in use {(total - free2) / 2**30:.2f} GiB", flush=True)
    r.close()
    del r
