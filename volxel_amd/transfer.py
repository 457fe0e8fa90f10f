"""Transfer-function generation, mirroring volxel-3d-viewer/src/utils/data.ts:1-60."""
from __future__ import annotations

import numpy as np


def generate_transfer_function(colors, generated_steps: int = 128):
    """data.ts:21-60.  colors: list of {"color": [r,g,b,density], "stop": s}.
    Returns (float32 array [steps*4], steps)."""
    if len(colors) < 1:
        raise ValueError("At least one color stop required")
    stops = sorted(colors, key=lambda c: c["stop"])
    if any(c["stop"] < 0.0 or c["stop"] > 1.0 for c in stops):
        raise ValueError("ColorStop outside stop range")
    current = -1
    out = []
    for i in range(generated_steps):
        pos = i / generated_steps
        if current < 0:
            if stops[0]["stop"] >= pos:
                current = 0
                out.append(list(stops[0]["color"]))
            else:
                out.append([0.0, 0.0, 0.0, 0.0])
        else:
            nxt = stops[current + 1] if current + 1 < len(stops) else None
            if nxt is None:
                out.append(list(stops[current]["color"]))
            else:
                cur = stops[current]
                denom = nxt["stop"] - cur["stop"]
                # JS: x/0 = +-Infinity or NaN; NaN >= 1.0 is false
                if denom == 0:
                    num = pos - cur["stop"]
                    progress = float("nan") if num == 0 else (float("inf") if num > 0 else float("-inf"))
                else:
                    progress = (pos - cur["stop"]) / denom
                if progress >= 1.0:
                    out.append(list(nxt["color"]))
                    current += 1
                    continue
                out.append([(1 - progress) * v + progress * nxt["color"][j]
                            for j, v in enumerate(cur["color"])])
    return np.asarray(out, dtype=np.float64).astype(np.float32).reshape(-1), generated_steps


def default_transfer_function():
    """viewer.ts:378-384: white, alpha ramp i/128."""
    return generate_transfer_function([{"color": [1, 1, 1, 0], "stop": 0},
                                       {"color": [1, 1, 1, 1], "stop": 1}])


def parse_transfer_function(text: str):
    """data.ts:1-14: one 'r g b density' line per entry."""
    rows = []
    for line in text.split("\n"):
        parts = line.split(" ")
        if len(parts) != 4:
            continue
        vals = []
        for p in parts:
            try:
                vals.append(float(p))
            except ValueError:
                vals.append(float("nan"))  # Number.parseFloat of garbage
        rows.append(vals)
    data = np.asarray(rows, dtype=np.float64).astype(np.float32).reshape(-1)
    return data, len(rows), rows
