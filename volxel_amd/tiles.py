"""Image-space sharding bookkeeping; mirrors vx::TileMap /
vx::wave_pixel / vx::detile of csrc/vx_kernels.hpp in NumPy.  Pure index math: used by the
host to lay out / interpret slabs, by the gloo tests, and to de-tile gathered slabs that live
in host memory.  No pixel is computed here."""
from __future__ import annotations

import numpy as np

TILE = 64


def tile_counts(width: int, height: int, shard_count: int):
    tx = (width + TILE - 1) // TILE
    ty = (height + TILE - 1) // TILE
    n = tx * ty
    return tx, ty, n, (n + shard_count - 1) // shard_count


def _morton6():
    wx = np.zeros(64, dtype=np.int64)
    wy = np.zeros(64, dtype=np.int64)
    for m in range(64):
        wx[m] = (m & 1) | ((m >> 1) & 2) | ((m >> 2) & 4)
        wy[m] = ((m >> 1) & 1) | ((m >> 2) & 2) | ((m >> 3) & 4)
    return wx, wy


def balanced_order(costs, shard_count: int) -> np.ndarray:
    """Dealing order for vx_set_tile_order from per-tile cost estimates (vx_probe_tile_costs): tiles by
    descending cost (ties by tile id), dealt to the shards in boustrophedon order -- 0..N-1, N-1..0, ... --
    so that every shard gets one tile of every cost class and the running sums stay level.  Deterministic:
    every rank derives the same permutation from the same costs."""
    c = np.asarray(costs, dtype=np.int64)
    order = np.argsort(-c, kind="stable")
    n = int(shard_count)
    out = order.copy()
    for g in range(1, (len(order) + n - 1) // n, 2):
        out[g * n:(g + 1) * n] = order[g * n:(g + 1) * n][::-1]
    return out.astype(np.uint32)


def slab_pixel_coords(width: int, height: int, shard_rank: int, shard_count: int, perm=None):
    """(px, py) int arrays of length tiles_per_shard*4096 giving the pixel of every slab slot,
    -1 where the slot is padding (tile beyond the image or pixel beyond the border).
    perm: the dealing order installed with vx_set_tile_order (position -> tile), None = identity."""
    tx, ty, n, tps = tile_counts(width, height, shard_count)
    wx, wy = _morton6()
    lane = np.arange(64)
    lx, ly = lane & 7, lane >> 3
    # one tile: slot = wt*64 + lane
    ox = (wx[:, None] * 8 + lx[None, :]).reshape(-1)
    oy = (wy[:, None] * 8 + ly[None, :]).reshape(-1)
    px = np.full(tps * 4096, -1, dtype=np.int64)
    py = np.full(tps * 4096, -1, dtype=np.int64)
    for lt in range(tps):
        t = lt * shard_count + shard_rank
        if t >= n:
            continue
        if perm is not None:
            t = int(perm[t])
        x = (t % tx) * TILE + ox
        y = (t // tx) * TILE + oy
        ok = (x < width) & (y < height)
        px[lt * 4096:(lt + 1) * 4096] = np.where(ok, x, -1)
        py[lt * 4096:(lt + 1) * 4096] = np.where(ok, y, -1)
    return px, py


def detile_numpy(gathered: np.ndarray, width: int, height: int, shard_count: int, perm=None) -> np.ndarray:
    """gathered: [shard_count, tiles_per_shard*4096, 4] -> image [height, width, 4]."""
    gathered = np.asarray(gathered).reshape(shard_count, -1, 4)
    img = np.zeros((height, width, 4), dtype=gathered.dtype)
    for r in range(shard_count):
        px, py = slab_pixel_coords(width, height, r, shard_count, perm)
        ok = px >= 0
        img[py[ok], px[ok]] = gathered[r][ok]
    return img
