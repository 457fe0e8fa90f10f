"""Host-side mirror of the reference's render core (volxel-3d-viewer/src/viewer.ts) above
the C ABI of libvolxel_hip.so.

`Volxel3DRenderer` keeps the names of the reference class `Volxel3DDicomRenderer`
(viewer.ts:111) for the calls on the hot path:

    setup_from_grid(grid)          viewer.ts:1080-1145   (setupFromGrid)
    change_transfer_func(data, n)  viewer.ts:1147-1153   (changeTransferFunc)
    restart_rendering()            viewer.ts:1155-1181   (frameIndex = 0)
    bind_uniforms()                viewer.ts:1295-1357   (+ scene.ts:53-56)
    render()                       viewer.ts:1183-1293   (one accumulation sample)
    restore_settings(json)         viewer.ts:704-713
    render_mode (property)         viewer.ts:1442-1452

Everything that touches pixels goes through the HIP library; there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import _abi
from .scene import Camera, Grid, Volume, flat, from_flat
from .environment import Environment
from .settings import ViewerSettings, verify_settings
from .transfer import default_transfer_function, generate_transfer_function

LOW_RESOLUTION_DURATION = 5  # viewer.ts:132


class VolxelError(RuntimeError):
    """What viewer.ts:797-816 handleError receives."""


def sample_weight(frame_index: int, low_res_duration: int = LOW_RESOLUTION_DURATION) -> float:
    """viewer.ts:1356."""
    if frame_index < low_res_duration:
        return 0.0
    return (frame_index - low_res_duration) / (frame_index - low_res_duration + 1)


def compute_params(settings: ViewerSettings, camera: Camera, volume: Volume, density_scale: float,
                   width: int, height: int, env_strength: float = 1.0, shard_rank: int = 0,
                   shard_count: int = 1, has_environment: bool = False) -> "_abi.VxParams":
    """bindUniforms (viewer.ts:1295-1357) + Camera.bindAsUniforms (scene.ts:53-56):
    doubles on the host, rounded to float32 on upload."""
    p = _abi.VxParams()

    def put(name, arr):
        a = np.asarray(arr, dtype=np.float64).astype(np.float32).reshape(-1)
        getattr(p, name)[:] = a.tolist()

    view = camera.view_matrix()
    proj = camera.proj_matrix(width / height)
    view32 = from_flat(flat(view).astype(np.float32))  # the shader only sees the f32 upload
    proj32 = from_flat(flat(proj).astype(np.float32))
    put("camera_view", flat(view))
    put("camera_proj", flat(proj))
    put("camera_view_inv", flat(np.linalg.inv(view32)))   # inverse(camera_view), utils.glsl:24
    put("camera_proj_inv", flat(np.linalg.inv(proj32)))   # inverse(camera_proj), utils.glsl:29
    p.camera_ortho = 1 if camera.ortho_half_height is not None else 0   # [build] BASELINE config 1

    mn, maj = volume.min_maj()
    lo, hi = volume.aabb_clipped(settings.volume_clip_min, settings.volume_clip_max)
    put("volume_aabb_min", lo)
    put("volume_aabb_max", hi)
    mult = settings.density_multiplier
    p.volume_min = mn * density_scale * mult                 # viewer.ts:1321
    p.volume_maj = maj * density_scale * mult                # viewer.ts:1322
    p.volume_inv_maj = 1.0 / (maj * density_scale * mult)    # viewer.ts:1323
    put("volume_albedo", [0.9, 0.9, 0.9])                    # viewer.ts:1325
    p.volume_phase_g = 0.0                                   # viewer.ts:1326
    p.volume_density_scale = density_scale * mult            # viewer.ts:1327
    combined = volume.combined_transform()
    put("density_transform", flat(combined))                 # viewer.ts:1330
    put("density_transform_inv", flat(np.linalg.inv(combined)))  # viewer.ts:1331
    put("sample_range", settings.sample_range)               # viewer.ts:1343

    put("light_dir", settings.light_dir)                     # viewer.ts:1303
    p.env_strength = env_strength                            # environment.ts:83
    p.show_environment = 1 if settings.show_environment else 0   # viewer.ts:1338
    # viewer.ts:1340; an environment map must be resident (the viewer always has the default one)
    p.use_env = 1 if (settings.use_env and has_environment) else 0
    p.bounces = int(settings.bounces)                        # viewer.ts:1339
    p.res[0], p.res[1] = int(width), int(height)             # viewer.ts:1353 (quirk Q2 dropped)
    p.debug_hits = 1 if settings.debug_hits else 0           # viewer.ts:1354
    p.render_mode = _abi.RENDER_MODES[settings.render_mode]

    p.dvr_step_voxels = settings.dvr_step_voxels
    p.dvr_ert_tau = -math.log(settings.dvr_ert_epsilon)
    p.dvr_jitter = 1 if settings.dvr_jitter else 0
    p.dvr_max_steps = int(settings.dvr_max_steps)
    p.dvr_skip_empty = 1 if settings.dvr_skip_empty else 0
    # K = albedo * mis * f_p * Le / pdf for the directional light (fragment.frag:94-97,
    # environment.glsl:30-33, utils.glsl:104,121-124), in float32 like the shader
    f32 = np.float32
    inv_4pi = f32(1.0) / (f32(4.0) * f32(math.pi))
    f_p = inv_4pi * (f32(1.0) - f32(0.0)) / (f32(1.0) * f32(1.0))
    mis = (f32(1.0) / (f32(1.0) + f_p * f_p)) if settings.show_environment else f32(1.0)
    le = f32(env_strength) * f32(4.01)
    gain = ((f32(0.9) * mis) * f_p) * le
    put("dvr_gain", [gain, gain, gain])
    p.phong_ka, p.phong_kd, p.phong_ks, p.phong_shininess = [float(x) for x in settings.phong]
    p.shard_rank, p.shard_count = int(shard_rank), int(shard_count)
    return p


_worker_factory = None
COMPONENTS = {}


def register_volxel_components(worker_factory=None):
    """registerVolxelComponents (viewer.ts:1455-1462): remembers the worker factory and registers the component
    classes under the reference's element names.  The four widget elements (slider, histogram viewer, cube direction,
    colour ramp) are browser UI and have no counterpart here; `volxel-3d-viewer` maps to the renderer class.  The
    factory is kept for hosts that run the preprocessor elsewhere: the renderer itself calls the native preprocessor
    in-process."""
    global _worker_factory
    _worker_factory = worker_factory
    COMPONENTS["volxel-3d-viewer"] = Volxel3DRenderer
    return COMPONENTS


class Volxel3DRenderer:
    """Headless counterpart of the `<volxel-3d-viewer>` element's render core."""

    def __init__(self, width: int = 1920, height: int = 1080, device: int = 0,
                 shard_rank: int = 0, shard_count: int = 1, layout: int | None = None,
                 low_res_preview: bool = False):
        """width, height: the canvas.  low_res_preview=True reproduces the viewer's interactive
        ramp (viewer.ts:1167-1188): after every restart the first `low_resolution_duration` frames
        are rendered at 0.33 x the canvas and shown NEAREST-magnified; a headless caller that wants
        full-size frames from frame 0 (tests, bench.py) leaves it off."""
        self._lib = _abi.load_library()
        self._ctx = C.c_void_p()
        rc = self._lib.vx_create(int(device), C.byref(self._ctx))
        if rc != 0:
            msg = self._lib.vx_last_error(None)
            self._ctx = None
            raise VolxelError(msg.decode() if msg else f"vx_create failed ({rc})")
        self.settings = ViewerSettings()
        self.camera = Camera(1)                       # viewer.ts:418
        self.volume: Volume | None = None
        self.density_scale = 1.0
        self.environment: Environment | None = None
        self._env_strength = 1.0
        self.frame_index = 0
        self.canvas_width, self.canvas_height = int(width), int(height)
        self.width, self.height = int(width), int(height)      # current render size
        self.shard_rank, self.shard_count = shard_rank, shard_count
        self.low_resolution_duration = LOW_RESOLUTION_DURATION
        self.low_res_preview = bool(low_res_preview)
        self.tile_order = None                                 # vx_set_tile_order permutation, if any
        self.resolution_factor = 1.0                           # viewer.ts:131, the ramp state
        if layout is not None:
            self._check(self._lib.vx_set_layout(self._ctx, int(layout)))
        self._check(self._lib.vx_resize(self._ctx, self.width, self.height))
        self.set_environment(Environment.default())   # viewer.ts:372-374
        data, length = default_transfer_function()    # viewer.ts:377-385
        self.change_transfer_func(data, length)

    # -- error contract (viewer.ts:797-816) -------------------------------------------
    def _check(self, rc):
        if rc != 0:
            raise VolxelError(self._lib.vx_last_error(self._ctx).decode())

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.vx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- viewer.ts:1442-1452 ------------------------------------------------------------
    @property
    def render_mode(self) -> str:
        return self.settings.render_mode

    @render_mode.setter
    def render_mode(self, to: str):
        if to not in _abi.RENDER_MODES:
            raise VolxelError(f"Unrecognized render mode provided: {to}")
        self.settings.render_mode = to
        self.restart_rendering()

    # -- viewer.ts:963-975 restartFromFiles (+ worker.ts:77-104: files -> bytes -> read_dicoms_to_grid)
    def restart_from_files(self, files, n_threads: int = 0):
        """files: DICOM slices as paths or bytes objects, in stacking order (the reference does not
        sort them either, lib.rs:150-176).  ZIP / URL variants are container I/O outside the path."""
        from .preprocessor import read_dicoms_to_grid
        blobs = []
        for f in files:
            if isinstance(f, (bytes, bytearray, memoryview)):
                blobs.append(bytes(f))
            else:
                with open(f, "rb") as fh:
                    blobs.append(fh.read())
        self.setup_from_grid(read_dicoms_to_grid(blobs, n_threads))

    # -- viewer.ts:977-1017: the other load methods.  The reference posts the bytes to its worker, which decodes
    #    them (worker.ts:60-76,105-126); here the host decodes what the Python standard library can (containers.py)
    def restart_from_zip(self, zip_file, n_threads: int = 0):
        """restartFromZip (viewer.ts:977-989): a ZIP of DICOM slices as bytes or a path; folder rule of zip.rs:54-70"""
        from .containers import read_zip_slices
        from .preprocessor import read_dicoms_to_grid
        if not isinstance(zip_file, (bytes, bytearray, memoryview)):
            with open(zip_file, "rb") as fh:
                zip_file = fh.read()
        self.setup_from_grid(read_dicoms_to_grid(read_zip_slices(zip_file), n_threads))

    def restart_from_zip_url(self, url: str, n_threads: int = 0):
        """restartFromZipUrl (viewer.ts:991-1003, worker.ts:115-118)"""
        from .containers import fetch_bytes
        self.restart_from_zip(fetch_bytes(url), n_threads)

    def restart_from_urls(self, urls, n_threads: int = 0):
        """restartFromURLs (viewer.ts:1005-1017, worker.ts:120-123): one DICOM slice per URL, in the order given"""
        from .containers import fetch_bytes
        self.restart_from_files([fetch_bytes(u) for u in urls], n_threads)

    def load_env(self, data):
        """loadEnv (viewer.ts:1019-1033, worker.ts:77-90, hdr.rs:23-36): an encoded environment map.  Radiance RGBE is
        decoded; OpenEXR raises (decode it elsewhere and call setup_env with the floats)."""
        from .containers import decode_environment
        try:
            floats, w, h = decode_environment(data)
        except ValueError as e:
            raise VolxelError(str(e)) from None
        self.setup_env({"floats": floats, "width": w, "height": h})

    def load_env_from_url(self, url: str):
        """loadEnvFromUrl (viewer.ts:1035-1040)"""
        from .containers import fetch_bytes
        self.load_env(fetch_bytes(url))

    # -- viewer.ts:443-449,543-551,789-795: the light follows the camera ("backlight") ---------
    def maybe_sync_light(self):
        """maybeSyncLight (viewer.ts:789-795): lightDir = -(view - pos); assigning it to the direction widget
        (cubeDirection.ts:177-206) normalises it and hands the unit vector back through the 'direction' event
        (viewer.ts:536-541), so the uniform of the next frame is the unit vector from the look-at point to the camera.
        As in the reference, restoring a settings file does NOT re-aim the light (viewer.ts:696-713 never calls it):
        the file's lightDir is used until the camera is rotated or the toggle changes."""
        if self.settings.sync_light_dir:
            d = np.asarray(self.camera.pos, dtype=np.float64) - np.asarray(self.camera.view, dtype=np.float64)
            n = float(np.linalg.norm(d))
            if n > 0.0:                                    # the widget ignores a zero vector (cubeDirection.ts:187-190)
                self.settings.light_dir = (float(d[0] / n), float(d[1] / n), float(d[2] / n))

    def rotate_camera(self, by):
        """the canvas drag of viewer.ts:443-449: orbit, re-aim the synced light, restart"""
        self.camera.rotate_around_view(by)
        self.maybe_sync_light()
        self.restart_rendering()

    @property
    def sync_light_dir(self) -> bool:
        return bool(self.settings.sync_light_dir)

    @sync_light_dir.setter
    def sync_light_dir(self, on: bool):               # the backlight toggle, viewer.ts:543-551
        self.settings.sync_light_dir = bool(on)
        self.maybe_sync_light()
        self.restart_rendering()

    # -- viewer.ts:1073-1078 setupEnv(WasmWorkerMessageEnvReturn) -----------------------------
    def setup_env(self, env_message):
        """env_message: mapping / object with width, height, floats (row 0 = top)"""
        g = (lambda k: env_message[k]) if isinstance(env_message, dict) else (lambda k: getattr(env_message, k))
        # a new Environment starts at strength 1 (environment.ts:15)
        self.set_environment(Environment(g("floats"), int(g("width")), int(g("height")), 1.0))

    # -- viewer.ts:1080-1145 ------------------------------------------------------------
    def setup_from_grid(self, grid):
        """grid: preprocessor.BrickGridMessage (= WasmWorkerMessageDicomReturn)."""
        self.density_scale = 1.0
        self.settings.volume_clip_max = (1.0, 1.0, 1.0)
        self.settings.volume_clip_min = (0.0, 0.0, 0.0)
        g = Grid(min_maj=tuple(grid.min_maj), index_extent=np.asarray(grid.index_extent, float),
                 transform=from_flat(grid.transform))
        self.volume = Volume(g)
        self.density_scale *= self.volume.normalise()
        u3 = lambda t: (C.c_uint32 * 3)(*[int(x) for x in t])
        ind = np.ascontiguousarray(grid.indirection, dtype=np.uint32)
        rng = np.ascontiguousarray(grid.range, dtype=np.uint16)
        atl = np.ascontiguousarray(grid.atlas, dtype=np.uint8)
        n = len(grid.range_mipmaps)
        mips = [np.ascontiguousarray(m, dtype=np.uint16) for m, _ in grid.range_mipmaps]
        # the C ABI takes raw pointers and cannot check lengths: the arrays must hold what the size fields say
        prod = lambda t: int(t[0]) * int(t[1]) * int(t[2])
        if ind.size < prod(grid.indirection_size):
            raise VolxelError("setup_from_grid: indirection shorter than indirection_size")
        if rng.size < 2 * prod(grid.range_size):
            raise VolxelError("setup_from_grid: range shorter than 2 * range_size")
        if atl.size < prod(grid.atlas_size):
            raise VolxelError("setup_from_grid: atlas shorter than atlas_size")
        for m, (_, st) in zip(mips, grid.range_mipmaps):
            if m.size < 2 * prod(st):
                raise VolxelError("setup_from_grid: range mipmap shorter than 2 * stride")
        mip_ptrs = (C.c_void_p * max(n, 1))(*[m.ctypes.data for m in mips])
        mip_sizes = (C.c_uint32 * (3 * max(n, 1)))(*[int(x) for _, s in grid.range_mipmaps for x in s])
        self._check(self._lib.vx_upload_volume(
            self._ctx, ind.ctypes.data, u3(grid.indirection_size), rng.ctypes.data,
            u3(grid.range_size), atl.ctypes.data if atl.size else None, u3(grid.atlas_size), n,
            mip_ptrs, C.cast(mip_sizes, C.c_void_p), u3(grid.index_extent)))
        self.restart_rendering()

    # -- viewer.ts:1147-1153 ------------------------------------------------------------
    def change_transfer_func(self, data, length: int):
        a = np.ascontiguousarray(data, dtype=np.float32).reshape(-1)
        if a.size != 4 * length:
            raise VolxelError("transfer function must hold length*4 floats")
        self._check(self._lib.vx_upload_transfer(self._ctx, a.ctypes.data, int(length)))
        self._tf = (a.copy(), int(length))
        self.frame_index = 0

    def set_color_stops(self, colors, steps: int = 128):
        data, length = generate_transfer_function(colors, steps)
        self.change_transfer_func(data, length)

    # -- viewer.ts:1073-1078 setupEnv --------------------------------------------------------
    def set_environment(self, env: "Environment | None"):
        """env = None removes the map: use_env then falls back to the directional light."""
        if env is None:
            self._check(self._lib.vx_upload_environment(self._ctx, None, 0, 0))
        else:
            self._check(self._lib.vx_upload_environment(self._ctx, env.floats.ctypes.data, env.width, env.height))
        self.environment = env
        self.restart_rendering()

    @property
    def env_strength(self) -> float:                  # environment.ts:15 `strength`
        return self.environment.strength if self.environment is not None else self._env_strength

    @env_strength.setter
    def env_strength(self, v: float):
        self._env_strength = float(v)
        if self.environment is not None:
            self.environment.strength = float(v)

    # -- viewer.ts:1155-1181 ------------------------------------------------------------
    def restart_rendering(self):
        if self.low_res_preview:                               # viewer.ts:1176-1178
            self.resolution_factor = 0.33
            self._resize_framebuffers_to_canvas()
        self.frame_index = 0

    def _resize_framebuffers_to_canvas(self):                  # viewer.ts:925-949
        f = self.resolution_factor * self.settings.resolution_factor if self.low_res_preview else 1.0
        w = max(1, math.floor(self.canvas_width * f))
        h = max(1, math.floor(self.canvas_height * f))
        if (w, h) != (self.width, self.height):
            self.width, self.height = w, h
            self._check(self._lib.vx_resize(self._ctx, w, h))
            self.tile_order = None   # a dealing order belongs to one tile grid; the library dropped it too

    def resize(self, width: int, height: int):
        if int(width) <= 0 or int(height) <= 0:
            self._check(self._lib.vx_resize(self._ctx, max(0, int(width)), max(0, int(height))))  # raises
        self.canvas_width, self.canvas_height = int(width), int(height)
        self._resize_framebuffers_to_canvas()
        self.restart_rendering()

    # -- viewer.ts:704-713 restoreSettings ---------------------------------------------
    def restore_settings(self, s: dict):
        verify_settings(s)
        t = s["transfer"]
        self.settings.density_multiplier = t["densityMultiplier"]
        self.settings.sample_range = tuple(t["histogramRange"])          # viewer.ts:650
        if t["transfer"]["type"] == "color_stops":
            self.set_color_stops(t["transfer"]["colors"])
        else:
            rows = np.asarray(t["transfer"]["colors"], dtype=np.float32)
            self.change_transfer_func(rows.reshape(-1), rows.shape[0])
        d = s["display"]
        self.settings.bounces = d["bounces"]
        self.settings.max_samples = d["samples"]
        self.settings.gamma, self.settings.exposure = d["gamma"], d["exposure"]
        self.settings.debug_hits = d["debugHits"]
        self.settings.render_mode = d["renderMode"]
        self.settings.resolution_factor = d["resolutionFactor"]
        l = s["lighting"]
        self.settings.show_environment = l["showEnv"]
        self.settings.use_env = l["useEnv"]
        self.env_strength = l["envStrength"]
        self.settings.sync_light_dir = l["syncLightDir"]
        self.settings.light_dir = tuple(l["lightDir"])
        o = s["other"]
        self.settings.volume_clip_max = tuple(o["clipMax"])
        self.settings.volume_clip_min = tuple(o["clipMin"])
        self.camera.pos = np.asarray(o["cameraPos"], dtype=np.float64)
        self.camera.view = np.asarray(o["cameraLookAt"], dtype=np.float64)
        self.restart_rendering()

    # -- viewer.ts:1295-1357 ------------------------------------------------------------
    def bind_uniforms(self):
        if self.volume is None:
            raise VolxelError("Trying to bind uniforms without a volume.")
        p = compute_params(self.settings, self.camera, self.volume, self.density_scale, self.width,
                           self.height, self.env_strength, self.shard_rank, self.shard_count,
                           has_environment=self.environment is not None)
        self._check(self._lib.vx_set_params(self._ctx, C.byref(p)))
        self._params = p
        return p

    # -- viewer.ts:1183-1293 ------------------------------------------------------------
    def render(self, frames: int = 1, rebind: bool = True, in_flight: int = 32):
        """Render `frames` accumulation samples (the body of render() while
        frameIndex <= maxSamples).  Asynchronous; call finish() or a read_* to wait.
        Up to `in_flight` frames go into one launch (vx_render_frames): same bits as one launch per frame
        (in_flight=1), 1.6x the speed at 32."""
        if self.low_res_preview and (self.frame_index < self.low_resolution_duration or self.resolution_factor != 1.0):
            # the ramp changes the framebuffer size at frame low_resolution_duration: step up to it
            while frames > 0 and self.frame_index <= self.settings.max_samples:
                if self.frame_index >= self.low_resolution_duration:
                    if self.resolution_factor != 1.0:          # viewer.ts:1185-1188
                        self.resolution_factor = 1.0
                        self._resize_framebuffers_to_canvas()
                    break
                self.bind_uniforms()
                self._check(self._lib.vx_render_frame(self._ctx, self.frame_index, 0.0))
                self.frame_index += 1
                frames -= 1
            if frames == 0:
                return
            rebind = True
        if rebind:
            self.bind_uniforms()
        if in_flight > 1 and frames > 1:
            n = max(0, min(frames, self.settings.max_samples + 1 - self.frame_index))
            if n:
                w = (C.c_float * n)(*[sample_weight(self.frame_index + i, self.low_resolution_duration)
                                      for i in range(n)])
                self._check(self._lib.vx_render_frames(self._ctx, self.frame_index, n, w, int(in_flight)))
                self.frame_index += n
            return
        for _ in range(frames):
            if self.frame_index > self.settings.max_samples:     # viewer.ts:1194
                break
            w = sample_weight(self.frame_index, self.low_resolution_duration)
            self._check(self._lib.vx_render_frame(self._ctx, self.frame_index, w))
            self.frame_index += 1

    def finish(self):  # gl.finish(), viewer.ts:1289
        self._check(self._lib.vx_finish(self._ctx))

    def read_accum(self) -> np.ndarray:
        out = np.empty((self.height, self.width, 4), dtype=np.float32)
        self._check(self._lib.vx_read_accum(self._ctx, out.ctypes.data))
        return out

    def read_display(self) -> np.ndarray:  # blit pass to the canvas, viewer.ts:1253-1265
        out = np.empty((self.canvas_height, self.canvas_width, 4), dtype=np.uint8)
        self._check(self._lib.vx_read_display_scaled(self._ctx, out.ctypes.data, self.canvas_width,
                                                     self.canvas_height, float(self.settings.exposure),
                                                     float(self.settings.gamma)))
        return out

    # -- viewer.ts:856-890,1213-1252: the data-benchmark-url runner ------------------------
    def start_benchmark(self, collection: dict, volumes: dict | None = None, progress=None) -> list:
        """Run a VolxelBenchmark collection (public/benchmark.json) and return the list of
        VolxelBenchmarkResult records the viewer would hand to saveBenchmark.

        Per entry, as startBenchmark/singleBenchmark do: optional volume switch, restoreSettings,
        renderMode, restart; then frames 0..maxSamples are rendered one by one with gl.finish()
        (vx_finish) inside the timed region, on the viewer's framebuffer sizing (resolutionFactor and
        the 0.33 preview for the first frames).  `volumes` maps an entry's "zip" string to a brick
        grid message (ZIP/DICOM container I/O is outside the path); entries without one keep the
        current volume.  Times are milliseconds like performance.now()."""
        import datetime
        import os
        import platform
        import time
        from .settings import verify_benchmark
        verify_benchmark(collection)
        name, cus, mem = self.device_info()
        device = {"platform": platform.platform(), "userAgent": "volxel_amd " + self._lib.vx_version().decode(),
                  "deviceMemory": mem / 2 ** 30, "hardwareConcurrency": os.cpu_count(),
                  "screen": {"width": self.canvas_width, "height": self.canvas_height, "pixelRatio": 1},
                  "gpu": {"vendor": "AMD", "renderer": name.strip(), "version": "HIP",
                          "shadingLanguageVersion": "gfx950", "supportedExtensions": [], "computeUnits": cus}}
        results = []
        saved_preview = self.low_res_preview
        self.low_res_preview = True
        try:
            for entry in collection["benchmarks"]:
                if entry.get("zip") is not None:
                    if not volumes or entry["zip"] not in volumes:
                        raise VolxelError(f"benchmark volume not provided: {entry['zip']}")
                    self.setup_from_grid(volumes[entry["zip"]])
                if entry.get("env") is not None:
                    raise VolxelError("benchmark entry asks for an environment map URL; pass the decoded "
                                      "map with set_environment() before the run")
                st = entry["settings"]
                self.restore_settings(collection["sharedSettings"][st] if isinstance(st, int) else st)
                if entry.get("renderMode"):
                    self.render_mode = entry["renderMode"]
                self.restart_rendering()
                total = 0.0
                while self.frame_index <= self.settings.max_samples:      # viewer.ts:1194
                    t0 = time.perf_counter()
                    self.render(1)
                    self.finish()                                         # viewer.ts:1213-1218
                    total += (time.perf_counter() - t0) * 1e3
                    if progress and self.frame_index % 100 == 0:
                        progress(entry.get("name"), self.frame_index, self.settings.max_samples)
                rf = self.settings.resolution_factor
                results.append({
                    "name": entry.get("name"), "settings": self.settings.to_viewer_dict(),
                    "timePerSample": total / self.frame_index, "totalTime": total,
                    "viewport": [0, 0, rf * self.canvas_width, rf * self.canvas_height],
                    "device": device,
                    "timestamp": datetime.datetime.now(datetime.timezone.utc).isoformat().replace("+00:00", "Z"),
                })
        finally:
            self.low_res_preview = saved_preview
            if not saved_preview:
                self.resolution_factor = 1.0
                self._resize_framebuffers_to_canvas()
        return results

    # -- measurement hooks (viewer.ts:1213-1252 benchmark harness) -----------------------
    def counters(self):
        c = _abi.VxCounters()
        self._check(self._lib.vx_get_counters(self._ctx, C.byref(c)))
        return c

    def reset_counters(self):
        self._check(self._lib.vx_reset_counters(self._ctx))

    def set_stream(self, hip_stream: int | None):
        self._check(self._lib.vx_set_stream(self._ctx, C.c_void_p(hip_stream or 0)))

    def set_layout(self, layout: int):
        self._check(self._lib.vx_set_layout(self._ctx, int(layout)))

    def upload_stats(self):
        """(seconds, host bytes, pinned) of the last volume upload: copies from pinned host memory in chunks,
        device-side layout build overlapped behind them (viewer.ts:1106-1142 is its texImage3D counterpart)"""
        s, b, pin = C.c_double(), C.c_uint64(), C.c_int()
        self._check(self._lib.vx_upload_stats(self._ctx, C.byref(s), C.byref(b), C.byref(pin)))
        return s.value, b.value, bool(pin.value)

    def probe_gather_rate(self, lines: int, distinct: int | None = None):
        """clocks per 16-byte-per-lane gather instruction per CU (nominal clock) when the 64 lanes form `lines` groups of
        consecutive lanes, each inside one L1-resident line, using `distinct` (default: lines) different lines; and the
        nominal clock in kHz"""
        clk, khz = C.c_double(), C.c_uint32()
        d = int(lines) if distinct is None else int(distinct)
        self._check(self._lib.vx_probe_gather_rate(self._ctx, int(lines), d, C.byref(clk), C.byref(khz)))
        return clk.value, khz.value

    def probe_valu_rate(self):
        """clocks (nominal) per wave64 VALU instruction per SIMD this device sustains, and the nominal clock in kHz"""
        clk, khz = C.c_double(), C.c_uint32()
        self._check(self._lib.vx_probe_valu_rate(self._ctx, C.byref(clk), C.byref(khz)))
        return clk.value, khz.value

    def probe_gather_spread(self, frame_index: int = 0):
        """(q0 gather instructions, wave-wide distinct lines, quad line look-ups) of one DVR frame"""
        out = (C.c_uint64 * 3)()
        self.bind_uniforms()
        self._check(self._lib.vx_probe_gather_spread(self._ctx, int(frame_index), out))
        return int(out[0]), int(out[1]), int(out[2])

    def device_info(self):
        name = C.create_string_buffer(256)
        cus, mem = C.c_uint32(), C.c_uint64()
        self._check(self._lib.vx_device_info(self._ctx, name, 256, C.byref(cus), C.byref(mem)))
        return name.value.decode(), cus.value, mem.value

    # -- multi-GPU load balance: dealing order of the 64x64 tiles (no counterpart in the reference) ----
    def probe_tile_costs(self) -> np.ndarray:
        """cost estimate of every tile of the image for the current volume / TF / uniforms (identical on
        every rank)"""
        from .tiles import tile_counts
        self.bind_uniforms()
        n = tile_counts(self.width, self.height, self.shard_count)[2]
        costs = np.zeros(n, dtype=np.uint32)
        self._check(self._lib.vx_probe_tile_costs(self._ctx, costs.ctypes.data, n))
        return costs

    def set_tile_order(self, perm):
        """perm: position -> tile (a permutation, the same on all ranks) or None for the default dealing;
        restarts the accumulation"""
        if perm is None:
            self._check(self._lib.vx_set_tile_order(self._ctx, None, 0))
        else:
            p = np.ascontiguousarray(perm, dtype=np.uint32)
            self._check(self._lib.vx_set_tile_order(self._ctx, p.ctypes.data, p.size))
        self.tile_order = None if perm is None else np.array(perm, dtype=np.uint32)
        self.restart_rendering()

    def balance_tiles(self):
        """probe the tile costs and deal the tiles so that every shard gets an equal share of the work"""
        from .tiles import balanced_order
        perm = balanced_order(self.probe_tile_costs(), self.shard_count)
        self.set_tile_order(perm)
        return perm

    # -- zero-copy slab access for the RCCL gather (volxel_amd/dist.py) -----------------
    def slab_info(self):
        n, t = C.c_uint64(), C.c_uint32()
        self._check(self._lib.vx_slab_info(self._ctx, C.byref(n), C.byref(t)))
        return n.value, t.value

    def slab_device_ptr(self) -> int:
        p = C.c_void_p()
        self._check(self._lib.vx_slab_device_ptr(self._ctx, C.byref(p)))
        return p.value

    def detile(self, gathered_dev_ptr: int, image_dev_ptr: int):
        self._check(self._lib.vx_detile(self._ctx, C.c_void_p(gathered_dev_ptr),
                                        C.c_void_p(image_dev_ptr)))
