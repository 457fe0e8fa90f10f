#!/usr/bin/env python3
"""Secondary measurements for DESIGN.md: every render mode on BASELINE config 3, config 2 (256^3 CT
phantom), config 4 (gradient + Phong) and -- with `big` -- one shard of config 5 (1024^3, 4K)."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from volxel_amd import BENCHMARK_SETTINGS, Volxel3DRenderer, read_u16_stack_to_grid, synth


def scene(vox, sp, w, h, shard=(0, 1)):
    msg = read_u16_stack_to_grid(vox, sp)
    r = Volxel3DRenderer(w, h, shard_rank=shard[0], shard_count=shard[1])
    r.setup_from_grid(msg)
    r.restore_settings(BENCHMARK_SETTINGS)
    r.settings.volume_clip_min = (0.25, 0.0, 0.0)
    r.settings.volume_clip_max = (1.0, 1.0, 0.75)
    r.settings.dvr_skip_empty = False
    r.settings.max_samples = 1 << 30
    return r


def measure(r, label, frames=128, in_flight=64, **extra):
    frames = int(os.environ.get("MB_FRAMES", frames))
    in_flight = int(os.environ.get("MB_INFLIGHT", in_flight))
    r.bind_uniforms()
    r.render(frames=4, rebind=False); r.finish(); r.reset_counters()
    r.render(frames=frames, rebind=False, in_flight=in_flight); r.finish()
    c = r.counters()
    print(json.dumps(dict(case=label, ms_per_frame=round(c.kernel_ms / c.frames, 4), Msamples=round(c.samples / c.frames / 1e6, 2),
                          gsps=round(c.samples / c.kernel_ms / 1e6, 1), skip_steps_M=round(c.skip_steps / c.frames / 1e6, 2),
                          **extra)), flush=True)


vox, sp = synth.value_noise(512, seed=42)
r = scene(vox, sp, 1920, 1080)
del vox
for mode, bounces in (("dvr", 1), ("dvr_phong", 1), ("raymarch", 1), ("no_dda", 1), ("default", 1), ("default", 3)):
    r.settings.render_mode = mode
    r.settings.bounces = bounces
    measure(r, f"config3 512^3 1080p {mode} bounces={bounces}")
r.close()
vox, sp = synth.ct_phantom(256)
r = scene(vox, sp, 1920, 1080)
r.settings.volume_clip_min = (0, 0, 0); r.settings.volume_clip_max = (1, 1, 1)
r.settings.render_mode = "dvr"
measure(r, "config2 256^3 CT phantom 1080p dvr")
r.close()
if len(sys.argv) > 1 and sys.argv[1] == "big":
    t0 = time.time()
    vox, sp = synth.value_noise(1024, seed=42)
    t1 = time.time()
    r = scene(vox, sp, 3840, 2160, shard=(0, 8))
    del vox
    r.settings.render_mode = "dvr"
    measure(r, "config5 1024^3 4K, shard 0 of 8, dvr", gen_s=round(t1 - t0, 1), setup_s=round(time.time() - t1, 1))
    r.close()
