"""Host-side scene math, mirroring volxel-3d-viewer/src/representation/{scene,volume,grid}.ts.

The reference does this in JavaScript doubles through math.gl 4.1.0 (a gl-matrix wrapper,
not vendored in the reference: parity unpinned) and uploads float32 uniforms.  Here: numpy
float64, column-major 4x4 stored as arrays m[col, row] flattened like gl-matrix
(flat[4*col + row]).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np


def mat4_identity():
    return np.eye(4, dtype=np.float64)


# Matrices are kept as "math" matrices M[row, col]; flat() converts to gl-matrix order.
def flat(m) -> np.ndarray:
    return np.asarray(m, dtype=np.float64).T.reshape(16).copy()


def from_flat(a) -> np.ndarray:
    return np.asarray(a, dtype=np.float64).reshape(4, 4).T.copy()


def look_at(eye, center, up) -> np.ndarray:
    """gl-matrix mat4.lookAt as used by scene.ts:58-64."""
    eye = np.asarray(eye, dtype=np.float64)
    center = np.asarray(center, dtype=np.float64)
    up = np.asarray(up, dtype=np.float64)
    if np.all(np.abs(eye - center) < 1e-6):
        return mat4_identity()
    z = eye - center
    z = z / math.sqrt(float(z @ z))
    x = np.cross(up, z)
    lx = math.sqrt(float(x @ x))
    x = x / lx if lx else np.zeros(3)
    y = np.cross(z, x)
    ly = math.sqrt(float(y @ y))
    y = y / ly if ly else np.zeros(3)
    m = mat4_identity()
    m[0, :3], m[1, :3], m[2, :3] = x, y, z
    m[0, 3], m[1, 3], m[2, 3] = -(x @ eye), -(y @ eye), -(z @ eye)
    return m


def perspective(fovy, aspect, near, far) -> np.ndarray:
    """gl-matrix mat4.perspective (OpenGL clip z in [-1,1]), scene.ts:65-72."""
    f = 1.0 / math.tan(fovy / 2.0)
    nf = 1.0 / (near - far)
    m = np.zeros((4, 4), dtype=np.float64)
    m[0, 0] = f / aspect
    m[1, 1] = f
    m[2, 2] = (far + near) * nf
    m[3, 2] = -1.0
    m[2, 3] = 2.0 * far * near * nf
    return m


def ortho(left, right, bottom, top, near, far) -> np.ndarray:
    """gl-matrix mat4.ortho (OpenGL clip z in [-1,1]).  [build]: the reference camera is perspective
    only (scene.ts:65-72); BASELINE config 1 asks for parallel rays."""
    lr, bt, nf = 1.0 / (left - right), 1.0 / (bottom - top), 1.0 / (near - far)
    m = np.zeros((4, 4), dtype=np.float64)
    m[0, 0], m[1, 1], m[2, 2], m[3, 3] = -2.0 * lr, -2.0 * bt, 2.0 * nf, 1.0
    m[0, 3], m[1, 3], m[2, 3] = (left + right) * lr, (top + bottom) * bt, (far + near) * nf
    return m


def scale_m(s) -> np.ndarray:
    m = mat4_identity()
    s = np.broadcast_to(np.asarray(s, dtype=np.float64), (3,))
    m[0, 0], m[1, 1], m[2, 2] = s
    return m


def translate_m(v) -> np.ndarray:
    m = mat4_identity()
    m[:3, 3] = np.asarray(v, dtype=np.float64)
    return m


def _quat_axis(axis, rad):
    """math.gl Quaternion.fromAxisRotation (gl-matrix quat.setAxisAngle): (x, y, z, w)"""
    s = math.sin(rad * 0.5)
    return np.array([axis[0] * s, axis[1] * s, axis[2] * s, math.cos(rad * 0.5)])


def _quat_mul(a, b):
    """gl-matrix quat.multiply(out, a, b) = a * b"""
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([ax * bw + aw * bx + ay * bz - az * by, ay * bw + aw * by + az * bx - ax * bz,
                     az * bw + aw * bz + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz])


def _quat_rotate(v, q):
    """gl-matrix vec3.transformQuat"""
    qv, w = np.asarray(q[:3]), q[3]
    uv = np.cross(qv, v)
    uuv = np.cross(qv, uv)
    return np.asarray(v, dtype=np.float64) + 2.0 * w * uv + 2.0 * uuv


class Camera:
    """scene.ts:3-72: orbit camera.  The pointer handling that feeds the orbit methods is browser UI (out of scope);
    the methods themselves are part of the component's behaviour (the light follows the camera through them,
    viewer.ts:443-449,789-795)."""

    up = np.array([0.0, 1.0, 0.0])

    def __init__(self, distance: float = 1.0):
        self.view = np.zeros(3)                       # look-at point (scene.ts:11)
        self.pos = np.array([0.0, 0.0, -distance])    # scene.ts:12
        self.yaw = 0.0                                # scene.ts:6-7
        self.pitch = 0.0
        # [build] None: the reference's perspective camera; a number: orthographic camera whose image
        # spans +-ortho_half_height world units vertically (BASELINE config 1: 0.6)
        self.ortho_half_height: float | None = None

    def rotate_around_view(self, by):                 # scene.ts:15-33
        self.yaw += -float(by[0])
        self.pitch += float(by[1])
        max_pitch = math.pi / 2 - 0.01
        self.pitch = min(max(self.pitch, -max_pitch), max_pitch)
        q_yaw = _quat_axis(Camera.up, self.yaw)
        right = _quat_rotate(np.array([1.0, 0.0, 0.0]), q_yaw)
        right = right / np.linalg.norm(right)
        q_pitch = _quat_axis(right, self.pitch)
        orientation = _quat_mul(q_pitch, q_yaw)
        dist = float(np.linalg.norm(np.asarray(self.pos, dtype=np.float64) - self.view))
        self.pos = _quat_rotate(np.array([0.0, 0.0, -1.0]), orientation) * dist + self.view

    def zoom(self, by: float) -> bool:                # scene.ts:35-40
        d = np.asarray(self.pos, dtype=np.float64) - self.view
        n = float(np.linalg.norm(d))
        if n * by <= 0.1 or n * by >= 10:
            return False
        self.pos = d * by + self.view
        return True

    def translate_on_plane(self, by):                 # scene.ts:42-47
        d = np.asarray(self.pos, dtype=np.float64) - self.view
        right = np.cross(d, Camera.up)
        right = right / np.linalg.norm(right)
        local_up = np.cross(d, right)
        local_up = local_up / np.linalg.norm(local_up)
        self.translate(right * (float(by[0]) * 5) + local_up * (-float(by[1]) * 5))

    def translate(self, by):                          # scene.ts:49-52
        self.pos = np.asarray(self.pos, dtype=np.float64) + by
        self.view = np.asarray(self.view, dtype=np.float64) + by

    def view_matrix(self) -> np.ndarray:
        return look_at(self.pos, self.view, Camera.up)

    def proj_matrix(self, aspect: float, fov: float = math.pi / 3) -> np.ndarray:
        if self.ortho_half_height is not None:
            h = float(self.ortho_half_height)
            return ortho(-h * aspect, h * aspect, -h, h, 0.1, 1000.0)
        return perspective(fov, aspect, 0.1, 1000.0)


@dataclass
class Grid:
    """representation/grid.ts:4-13 -- the host-side facts of an uploaded brick grid."""
    min_maj: tuple
    index_extent: np.ndarray
    transform: np.ndarray  # math matrix (row, col)


class Volume:
    """representation/volume.ts:5-48."""

    def __init__(self, grid: Grid):
        self.grid = grid
        self.transform = mat4_identity()

    def combined_transform(self) -> np.ndarray:     # volume.ts:14-16
        return self.transform @ self.grid.transform

    def to_world(self, index4) -> np.ndarray:       # volume.ts:17-20
        return self.combined_transform() @ np.asarray(index4, dtype=np.float64)

    def aabb(self):                                 # volume.ts:25-31
        lo = self.to_world([0, 0, 0, 1])
        e = self.grid.index_extent
        hi = self.to_world([e[0], e[1], e[2], 1])
        return lo[:3].copy(), hi[:3].copy()

    def aabb_clipped(self, cmin, cmax):             # volume.ts:32-37
        lo, hi = self.aabb()
        cmin = np.asarray(cmin, dtype=np.float64)
        cmax = np.asarray(cmax, dtype=np.float64)
        return lo + (hi - lo) * cmin, lo + (hi - lo) * cmax

    def min_maj(self):
        return self.grid.min_maj

    def normalise(self) -> float:
        """viewer.ts:1086-1099: centre at the origin, longest side 1; returns densityScale."""
        lo, hi = self.aabb()
        extent = hi - lo
        size = float(max(extent[0], max(extent[1], extent[2])))
        density_scale = 1.0
        if size != 1:
            self.transform = scale_m(1.0 / size) @ translate_m(-lo - extent * 0.5)
            density_scale *= size
        return density_scale
