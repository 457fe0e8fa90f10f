"""Viewer settings and the settings-JSON schema V3, mirroring volxel-3d-viewer/src/settings.ts
and the ViewerSettings defaults of viewer.ts:147-163."""
from __future__ import annotations

import json
import math
from dataclasses import dataclass, field

import numpy as np

from ._abi import RENDER_MODES


@dataclass
class ViewerSettings:  # settings.ts:45-61, defaults viewer.ts:147-163
    density_multiplier: float = 1.0
    max_samples: int = 2000
    debug_hits: bool = False
    volume_clip_min: tuple = (0.0, 0.0, 0.0)
    volume_clip_max: tuple = (1.0, 1.0, 1.0)
    show_environment: bool = True
    use_env: bool = True
    light_dir: tuple = (-1 / math.sqrt(3), -1 / math.sqrt(3), -1 / math.sqrt(3))
    sync_light_dir: bool = False
    bounces: int = 3
    gamma: float = 2.2
    exposure: float = 5.5
    sample_range: tuple = (0.0, 1.0)
    render_mode: str = "default"
    resolution_factor: float = 1.0
    # [build] parameters of the deterministic DVR modes (no reference counterpart)
    dvr_step_voxels: float = 0.5
    dvr_ert_epsilon: float = 1e-4
    dvr_jitter: bool = False
    dvr_max_steps: int = 1 << 20
    dvr_skip_empty: bool = True   # exact empty-space skipping (samples with alpha == 0 for sure)
    phong: tuple = (0.3, 0.7, 0.4, 32.0)  # ka, kd, ks, shininess


    def to_viewer_dict(self) -> dict:
        """The viewer's own `settings` object (viewer.ts:147-163) as JSON.stringify would emit it
        into a VolxelBenchmarkResult (viewer.ts:1233-1236); [build] fields under their JS-host names."""
        return {
            "densityMultiplier": self.density_multiplier, "maxSamples": self.max_samples,
            "debugHits": self.debug_hits, "volumeClipMin": list(self.volume_clip_min),
            "volumeClipMax": list(self.volume_clip_max), "showEnvironment": self.show_environment,
            "useEnv": self.use_env, "lightDir": list(self.light_dir), "syncLightDir": self.sync_light_dir,
            "bounces": self.bounces, "gamma": self.gamma, "exposure": self.exposure,
            "sampleRange": list(self.sample_range), "renderMode": self.render_mode,
            "resolutionFactor": self.resolution_factor,
            "dvrStepVoxels": self.dvr_step_voxels, "dvrErtEpsilon": self.dvr_ert_epsilon,
            "dvrJitter": self.dvr_jitter, "dvrMaxSteps": self.dvr_max_steps,
            "dvrSkipEmpty": self.dvr_skip_empty, "phong": list(self.phong),
        }


def _is_num(x):
    return isinstance(x, (int, float)) and not isinstance(x, bool)


def verify_vector(v):  # settings.ts:107-111
    if not isinstance(v, list) or len(v) != 3 or any(not _is_num(e) for e in v):
        raise ValueError("Malformed Vector in Settings detected.")


def verify_transfer_settings(s):  # settings.ts:75-92
    t = s.get("transfer", {})
    ok = (_is_num(s.get("densityMultiplier")) and isinstance(s.get("histogramRange"), list)
          and len(s["histogramRange"]) == 2 and all(_is_num(x) for x in s["histogramRange"])
          and t.get("type") in ("color_stops", "full"))
    if ok and t["type"] == "full":
        ok = all(_is_num(x) for row in t["colors"] for x in row)
    if ok and t["type"] == "color_stops":
        ok = all(_is_num(c.get("stop")) and all(_is_num(x) for x in c.get("color", [None]))
                 for c in t["colors"])
    if not ok:
        raise ValueError("Malformed Transfer Settings detected.")
    return s


def verify_display_settings(s):  # settings.ts:94-105
    if not (_is_num(s.get("samples")) and _is_num(s.get("bounces")) and _is_num(s.get("gamma"))
            and _is_num(s.get("exposure")) and isinstance(s.get("debugHits"), bool)
            and isinstance(s.get("renderMode"), str)
            and s["renderMode"] in ("default", "no_dda", "raymarch")
            and _is_num(s.get("resolutionFactor"))):
        raise ValueError("Malformed Display Settings detected.")


def verify_lighting_settings(s):  # settings.ts:113-118
    if not (_is_num(s.get("envStrength")) and isinstance(s.get("showEnv"), bool)
            and isinstance(s.get("useEnv"), bool) and isinstance(s.get("syncLightDir"), bool)):
        raise ValueError("Malformed Lighting Settings detected.")
    verify_vector(s.get("lightDir"))


def verify_settings(s):  # settings.ts:120-132
    if s.get("version") != "v3":
        raise ValueError(f"Unsupported Settings Format Version: {s.get('version')}")
    verify_transfer_settings(s["transfer"])
    verify_display_settings(s["display"])
    verify_lighting_settings(s["lighting"])
    for k in ("cameraLookAt", "cameraPos", "clipMax", "clipMin"):
        verify_vector(s["other"][k])
    return s


def load_settings(text_or_path):  # settings.ts:153-165
    text = text_or_path
    if isinstance(text_or_path, str) and not text_or_path.lstrip().startswith("{"):
        with open(text_or_path) as f:
            text = f.read()
    return verify_settings(json.loads(text))


def verify_benchmark(b):  # the VolxelBenchmark type, viewer.ts:72-82 (data-benchmark-url payload)
    if (not isinstance(b, dict) or not isinstance(b.get("sharedSettings"), list)
            or not isinstance(b.get("benchmarks"), list)):
        raise ValueError("Malformed benchmark collection.")
    for s in b["sharedSettings"]:
        verify_settings(s)
    for e in b["benchmarks"]:
        st = e.get("settings") if isinstance(e, dict) else None
        if isinstance(st, bool) or not isinstance(st, (int, dict)):
            raise ValueError("Malformed benchmark entry: settings must be an index or a settings export.")
        if isinstance(st, int):
            if not 0 <= st < len(b["sharedSettings"]):
                raise ValueError("Benchmark entry refers to a missing shared settings index.")
        else:
            verify_settings(st)
        if e.get("renderMode") is not None and e["renderMode"] not in RENDER_MODES:
            raise ValueError(f"Unrecognized render mode provided: {e['renderMode']}")
    return b


# public/benchmark.json:86-99: the three reference scenarios, all on shared settings 0
BENCHMARK_COLLECTION_MODES = ("default", "no_dda", "raymarch")


# public/benchmark.json:5-85 -- the only reference-supplied render inputs.
BENCHMARK_SETTINGS = {
    "version": "v3",
    "transfer": {
        "densityMultiplier": 0.99,
        "transfer": {"type": "color_stops", "colors": [
            {"color": [0.5686274509803921, 0.2549019607843137, 0.6745098039215687, 0.54], "stop": 0},
            {"color": [0.9725490196078431, 0.8941176470588236, 0.3607843137254902, 1],
             "stop": 0.17822873724342708},
            {"color": [0, 1, 1, 0.17], "stop": 0.3985239852398524}]},
        "histogramRange": [0.05645751953125, 1]},
    "lighting": {"useEnv": True, "showEnv": True, "envStrength": 1, "syncLightDir": False,
                 "lightDir": [-0.5773502691896258, -0.5773502691896257, -0.5773502691896257]},
    "display": {"bounces": 1, "samples": 500, "gamma": 2.2, "exposure": 5.5, "debugHits": False,
                "renderMode": "default", "resolutionFactor": 0.8},
    "other": {"clipMax": [1, 1, 1], "clipMin": [0, 0, 0],
              "cameraLookAt": [0.0025690138263833135, 0.027589039598078468, -0.04982115887377399],
              "cameraPos": [0.6609656965103848, 0.10997611027017339, -0.8109681692227295]},
}
