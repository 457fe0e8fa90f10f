"""Shared scene builders for the tests (CPU and GPU)."""
import numpy as np

from volxel_amd import ViewerSettings, compute_params, synth
from volxel_amd.scene import Camera, Grid, Volume, from_flat
from volxel_amd.transfer import default_transfer_function, generate_transfer_function
from volxel_amd.settings import BENCHMARK_SETTINGS


def make_scene(grid, width, height, mode="dvr", cam_pos=(0.0, 0.0, -1.0), look_at=(0, 0, 0),
               clip_min=(0, 0, 0), clip_max=(1, 1, 1), env=False, ortho=None, **kw):
    """env=True: uniforms as the viewer binds them with an environment map resident (use_env = 1);
    ortho=h: [build] orthographic camera of half height h (BASELINE config 1) instead of the reference's
    perspective camera"""
    s = ViewerSettings(render_mode=mode, bounces=kw.pop("bounces", 1),
                       volume_clip_min=clip_min, volume_clip_max=clip_max, **kw)
    cam = Camera(1)
    cam.pos = np.asarray(cam_pos, dtype=np.float64)
    cam.view = np.asarray(look_at, dtype=np.float64)
    cam.ortho_half_height = ortho
    vol = Volume(Grid(tuple(grid.min_maj), np.asarray(grid.index_extent, float), from_flat(grid.transform)))
    ds = vol.normalise()
    p = compute_params(s, cam, vol, ds, width, height, has_environment=env)
    return s, cam, vol, ds, p


def benchmark_tf():
    colors = BENCHMARK_SETTINGS["transfer"]["transfer"]["colors"]
    return generate_transfer_function(colors)


BENCH_CAM = dict(cam_pos=BENCHMARK_SETTINGS["other"]["cameraPos"],
                 look_at=BENCHMARK_SETTINGS["other"]["cameraLookAt"])


def small_noise(n=64, seed=7):
    """small 3-octave noise volume with empty space, for parity cases"""
    v, sp = synth.value_noise(n, seed=seed, zero_quantile=0.5)
    return v, sp


def default_environment(oracle):
    """the viewer's default map (environment.ts:102-130) as the oracle's Environment"""
    from volxel_amd import Environment
    e = Environment.default()
    return oracle.Environment(e.floats, e.width, e.height)
