"""Synthetic input volumes for the BASELINE.json configs.

The reference ships no volume data (public/Dicom is git-ignored), so every config is a
closed-form or seeded generator producing the u16 voxel stack that read_dicoms_internal
(dicom_preprocessor/src/lib.rs:142-191) would hand to BrickGrid::construct.
All generators return (voxels[z,y,x] uint16, spacing (x,y,z)).
"""
from __future__ import annotations

import numpy as np


def sphere(n: int = 64):
    """config 1: v = round(4095*max(0, 1 - |p - c|/r)), c = (n-1)/2, r = 28*n/64."""
    c = (n - 1) / 2.0
    r = 28.0 * n / 64.0
    ax = np.arange(n, dtype=np.float64) - c
    d = np.sqrt(ax[:, None, None] ** 2 + ax[None, :, None] ** 2 + ax[None, None, :] ** 2)
    v = np.round(4095.0 * np.maximum(0.0, 1.0 - d / r))
    return v.astype(np.uint16), (1.0, 1.0, 1.0)


def ct_phantom(n: int = 256, seed: int = 1234):
    """config 2: nested ellipsoids air 0 / fat 900 / tissue 1100 / bone 2400 (12 bit)
    + N(0,20) noise."""
    ax = (np.arange(n, dtype=np.float32) + 0.5) / n * 2.0 - 1.0
    z, y, x = ax[:, None, None], ax[None, :, None], ax[None, None, :]
    v = np.zeros((n, n, n), dtype=np.float32)

    def ell(cx, cy, cz, rx, ry, rz):
        return ((x - cx) / rx) ** 2 + ((y - cy) / ry) ** 2 + ((z - cz) / rz) ** 2 <= 1.0

    v[ell(0, 0, 0, 0.85, 0.65, 0.9)] = 900.0      # fat
    v[ell(0, 0, 0, 0.75, 0.55, 0.85)] = 1100.0    # tissue
    v[ell(0, 0.35, 0, 0.12, 0.12, 0.8)] = 2400.0  # spine
    for sx in (-1, 1):
        v[ell(sx * 0.45, -0.05, 0.0, 0.08, 0.30, 0.08)] = 2400.0   # rib-ish
        v[ell(sx * 0.35, 0.0, 0.1, 0.22, 0.28, 0.5)] = 300.0       # lung
    rng = np.random.default_rng(seed)
    v += rng.normal(0.0, 20.0, size=v.shape).astype(np.float32) * (v > 0)
    return np.clip(np.round(v), 0, 4095).astype(np.uint16), (0.7, 0.7, 1.0)


def _upsample_axis(a, n, axis):
    """linear interpolation of lattice values along one axis to n samples."""
    m = a.shape[axis]
    pos = (np.arange(n, dtype=np.float32) + 0.5) / n * (m - 1)
    i0 = np.minimum(np.floor(pos).astype(np.int64), m - 2)
    f = (pos - i0).astype(np.float32)
    w = np.zeros((n, m), dtype=np.float32)
    w[np.arange(n), i0] = 1.0 - f
    w[np.arange(n), i0 + 1] = f
    return np.moveaxis(np.tensordot(w, a, axes=([1], [axis])), 0, axis)


def value_noise(n: int = 512, seed: int = 42, zero_quantile: float = 0.62):
    """configs 3-5: three octaves of trilinearly interpolated lattice noise, thresholded so
    that roughly half of the 8^3 bricks are constant zero; 12 bit."""
    rng = np.random.default_rng(seed)
    acc = np.zeros((n, n, n), dtype=np.float32)
    amp = 1.0
    for period in (n // 8, n // 16, n // 32):
        m = n // period + 1
        lat = rng.random((m, m, m), dtype=np.float32)
        up = lat
        for axis in range(3):
            up = _upsample_axis(up, n, axis)
        acc += amp * up.astype(np.float32)
        amp *= 0.5
    # threshold on a subsample (quantile of the full array is slow and not needed exactly)
    sub = acc[::4, ::4, ::4]
    thr = float(np.quantile(sub, zero_quantile))
    hi = float(sub.max())
    acc -= thr
    np.maximum(acc, 0.0, out=acc)
    acc *= 4095.0 / max(hi - thr, 1e-6)
    np.minimum(acc, 4095.0, out=acc)
    return np.round(acc).astype(np.uint16), (1.0, 1.0, 1.0)


def make(config: str, n: int | None = None):
    if config == "sphere":
        return sphere(n or 64)
    if config == "ct":
        return ct_phantom(n or 256)
    if config == "noise":
        return value_noise(n or 512)
    raise ValueError(config)
