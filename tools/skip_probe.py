#!/usr/bin/env python3
"""Diagnostic: exact empty-space skipping on / off (frame time, evaluated samples) on the CT phantom
(config 2: contiguous air around the body) and on the config-3 noise volume (scattered empty cells)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volxel_amd import BENCHMARK_SETTINGS, Volxel3DRenderer, read_u16_stack_to_grid, synth

for name, (vox, sp), clip in (("config2 CT phantom 256^3", synth.ct_phantom(256), False),
                              ("config3 noise 512^3", synth.value_noise(512, seed=42), True)):
    r = Volxel3DRenderer(1920, 1080)
    r.setup_from_grid(read_u16_stack_to_grid(vox, sp))
    r.restore_settings(BENCHMARK_SETTINGS)
    r.settings.render_mode = "dvr"
    r.settings.max_samples = 1 << 30
    if clip:
        r.settings.volume_clip_min = (0.25, 0.0, 0.0); r.settings.volume_clip_max = (1.0, 1.0, 0.75)
    for skip in (False, True):
        r.settings.dvr_skip_empty = skip
        r.restart_rendering(); r.bind_uniforms()
        r.render(frames=4, rebind=False); r.finish(); r.reset_counters()
        r.render(frames=64, rebind=False, in_flight=32); r.finish()
        c = r.counters()
        print(json.dumps(dict(case=name, skip=skip, ms_per_frame=round(c.kernel_ms / c.frames, 4),
                              Msamples=round(c.samples / c.frames / 1e6, 1), Mskipped=round(c.skip_steps / c.frames / 1e6, 1),
                              lane_util=round(c.samples / max(c.lane_slots, 1), 3),
                              Mwave_steps=round(c.lane_slots / 64 / c.frames / 1e6, 3),
                              Mwindows=round(c.gathers / 3 / c.frames / 1e6, 3))), flush=True)
    r.close()
