"""Environment map: host counterpart of volxel-3d-viewer/src/representation/environment.ts.

The base map is `width x height` RGBA float with row 0 = TOP (the `floats` of
WasmWorkerMessageEnvReturn, common.ts; HDR/EXR decoding itself -- hdr.rs -- is container I/O outside
the path).  The importance map and its mips are built on the device by vx_upload_environment.
"""
from __future__ import annotations

import numpy as np

IMPORTANCE_DIMENSION = 512   # environment.ts:9
IMPORTANCE_SAMPLES = 64      # environment.ts:10


class Environment:
    def __init__(self, floats, width: int, height: int, strength: float = 1.0):
        a = np.ascontiguousarray(floats, dtype=np.float32)
        if width <= 0 or height <= 0 or a.size != width * height * 4:
            raise ValueError("Environment: floats must hold width*height RGBA texels")
        self.floats = a.reshape(height, width, 4)
        self.width, self.height = int(width), int(height)
        self.strength = float(strength)      # environment.ts:15

    @classmethod
    def default(cls) -> "Environment":
        """environment.ts:102-130: 8x6 checkerboard, bright upper third (row 0 = TOP).  A new object
        per call: `strength` is per viewer here (the reference shares one instance per page)."""
        w, h = 8, 6
        d = np.zeros((h, w, 4), dtype=np.float32)
        for y in range(h):
            top = y < h // 3
            for x in range(w):
                light = ((x + y) & 1) == 0
                d[y, x, :3] = (3.0 if light else 0.9) if top else (0.1 if light else 0.0)
                d[y, x, 3] = 1.0
        return cls(d, w, h)
