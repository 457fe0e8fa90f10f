"""ctypes wrapper of the CPU oracle (oracle/vx_oracle.c).  TEST INFRASTRUCTURE ONLY.

Imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by
anything under volxel_amd/.  PARITY UNPINNED (the reference ships no golden vectors): see
vx_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import re
import subprocess
from concurrent.futures import ThreadPoolExecutor

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)

_CT = {"float": C.c_float, "double": C.c_double, "int32_t": C.c_int32, "uint32_t": C.c_uint32,
       "uint64_t": C.c_uint64}


def _struct(header_path, name):
    text = re.sub(r"/\*.*?\*/", "", open(header_path).read(), flags=re.S)
    m = re.search(r"typedef\s+struct\s+%s\s*\{(.*?)\}\s*%s\s*;" % (name, name), text, flags=re.S)
    fields = []
    for decl in m.group(1).split(";"):
        decl = decl.strip()
        if not decl:
            continue
        mm = re.match(r"(\w+)\s+(.*)$", decl, flags=re.S)
        for item in mm.group(2).split(","):
            item = item.strip()
            am = re.match(r"(\w+)\s*\[(\d+)\]$", item)
            fields.append((am.group(1), _CT[mm.group(1)] * int(am.group(2))) if am
                          else (item, _CT[mm.group(1)]))
    return type(name, (C.Structure,), {"_fields_": fields})


VxParams = _struct(os.path.join(_ROOT, "include", "volxel_hip.h"), "VxParams")


class VxoVolume(C.Structure):
    _fields_ = [("indirection", C.c_void_p), ("ind_size", C.c_uint32 * 3),
                ("range", C.c_void_p), ("range_size", C.c_uint32 * 3),
                ("atlas", C.c_void_p), ("atlas_size", C.c_uint32 * 3),
                ("mips", C.c_void_p * 3), ("mip_size", (C.c_uint32 * 3) * 3),
                ("index_extent", C.c_uint32 * 3)]


class VxoEnvironment(C.Structure):
    _fields_ = [("texture", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32),
                ("importance", C.c_void_p)]


IMP_DIM, IMP_LEVELS, IMP_FLOATS = 512, 10, 349525


class Environment:
    """environment.ts: base map (row 0 = top, RGBA float) -> flipped texture + importance pyramid"""

    def __init__(self, floats, width, height):
        L = lib()
        src = np.ascontiguousarray(floats, dtype=np.float32).reshape(height, width, 4)
        self.width, self.height = int(width), int(height)
        self.texture = np.empty_like(src)
        L.vxo_env_flip_rows(src.ctypes.data, width, height, self.texture.ctypes.data)
        self.importance = np.empty(IMP_FLOATS, dtype=np.float32)
        L.vxo_env_build_importance(self.texture.ctypes.data, width, height, self.importance.ctypes.data)
        self.c = VxoEnvironment(self.texture.ctypes.data, width, height, self.importance.ctypes.data)

    def level(self, k):
        L = lib()
        n = IMP_DIM >> k
        o = L.vxo_imp_offset(k)
        return self.importance[o:o + n * n].reshape(n, n)

    def sample(self, u0, u1, strength=1.0):
        wi, lp = (C.c_float * 3)(), (C.c_float * 4)()
        lib().vxo_env_sample(C.byref(self.c), strength, u0, u1, wi, lp)
        return np.array(wi[:], dtype=np.float32), np.array(lp[:], dtype=np.float32)

    def lookup(self, d, strength=1.0):
        rgb = (C.c_float * 3)()
        lib().vxo_env_lookup(C.byref(self.c), strength, (C.c_float * 3)(*d), rgb)
        return np.array(rgb[:], dtype=np.float32)

    def pdf(self, d, strength=1.0):
        return lib().vxo_env_pdf(C.byref(self.c), strength, (C.c_float * 3)(*d))

    def texture_at(self, u, v):
        rgb = (C.c_float * 3)()
        lib().vxo_env_texture(C.byref(self.c), u, v, rgb)
        return np.array(rgb[:], dtype=np.float32)


class VxoCounters(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("rays", C.c_uint64), ("pixels", C.c_uint64),
                ("skip_steps", C.c_uint64), ("grad_samples", C.c_uint64), ("tf_samples", C.c_uint64)]


def _has_fma():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return " fma " in line + " "
    except OSError:
        pass
    return False


def build(force=False):
    so = os.path.join(_HERE, "_build", "libvx_oracle.so")
    if force or not os.path.exists(so) or \
            os.path.getmtime(so) < os.path.getmtime(os.path.join(_HERE, "vx_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


_lib = None
_variants = {}
VARIANTS = ("contract", "nocontract", "ulp2")


class variant:
    """`with oracle.variant("nocontract"): ...` -- route every oracle call inside the block through one of the
    tolerance-envelope builds of the same source (oracle/Makefile): "nocontract" rounds every a*b+c twice, "ulp2" biases
    the transcendental built-ins by 2 ulps; "contract" is the arithmetic contract itself (the default library)."""

    def __init__(self, name):
        assert name in VARIANTS, name
        self.name = name

    def __enter__(self):
        global _lib
        self.prev = _lib
        _lib = _load({"contract": None, "nocontract": "libvx_oracle_nocontract.so", "ulp2": "libvx_oracle_ulp2.so"}[self.name])
        return _lib

    def __exit__(self, *exc):
        global _lib
        _lib = self.prev


def lib():
    global _lib
    if _lib is None:
        _lib = _load(None)
    return _lib


def _load(name):
    if name is None:
        name = "libvx_oracle.so" if _has_fma() else "libvx_oracle_nofma.so"
    if name in _variants:
        return _variants[name]
    path = os.path.join(_HERE, "_build", name)
    if not os.path.exists(path):
        build(force=True)
    L = C.CDLL(path)
    u32, f32, vp, i32 = C.c_uint32, C.c_float, C.c_void_p, C.c_int32
    P = C.POINTER
    L.vxo_tea.argtypes = [u32, u32, u32]; L.vxo_tea.restype = u32
    L.vxo_wang.argtypes = [u32]; L.vxo_wang.restype = u32
    L.vxo_seed_xoshiro.argtypes = [u32, P(u32)]
    L.vxo_xoshiro_next.argtypes = [P(u32)]; L.vxo_xoshiro_next.restype = u32
    L.vxo_rng.argtypes = [P(u32)]; L.vxo_rng.restype = f32
    L.vxo_pixel_seed.argtypes = [u32, u32, u32, u32]; L.vxo_pixel_seed.restype = u32
    L.vxo_f32_to_f16.argtypes = [f32]; L.vxo_f32_to_f16.restype = C.c_uint16
    L.vxo_f16_to_f32.argtypes = [C.c_uint16]; L.vxo_f16_to_f32.restype = f32
    L.vxo_brick_count.argtypes = [P(u32), P(u32)]
    L.vxo_brick_construct.argtypes = [vp, P(u32), C.c_uint16, vp, vp, vp, vp, vp, vp, P(u32)]
    L.vxo_brick_construct.restype = C.c_int64
    L.vxo_dicom_lookup.argtypes = [vp, P(u32), C.c_uint16, u32, u32, u32]
    L.vxo_dicom_lookup.restype = f32
    L.vxo_brick_lookup.argtypes = [P(VxoVolume), u32, u32, u32]; L.vxo_brick_lookup.restype = f32
    L.vxo_histogram_gradient.argtypes = [vp, u32, vp, P(u32), P(u32)]
    L.vxo_lookup_density_brick.argtypes = [P(VxoVolume), i32, i32, i32]
    L.vxo_lookup_density_brick.restype = f32
    L.vxo_lookup_density_trilinear.argtypes = [P(VxoVolume), f32, f32, f32, f32]
    L.vxo_lookup_density_trilinear.restype = f32
    L.vxo_lookup_majorant.argtypes = [P(VxoVolume), f32, f32, f32, f32, i32]
    L.vxo_lookup_majorant.restype = f32
    L.vxo_lookup_transfer.argtypes = [vp, u32, P(f32), f32, P(f32)]
    L.vxo_render.argtypes = [P(VxParams), u32, f32, P(VxoVolume), vp, u32, vp, vp, i32, i32, i32,
                             i32, P(VxoCounters)]
    L.vxo_render.restype = C.c_int
    L.vxo_render_env.argtypes = [P(VxParams), u32, f32, P(VxoVolume), vp, u32, P(VxoEnvironment), vp, vp, i32,
                                 i32, i32, i32, P(VxoCounters)]
    L.vxo_render_env.restype = C.c_int
    L.vxo_imp_offset.argtypes = [u32]; L.vxo_imp_offset.restype = u32
    L.vxo_env_flip_rows.argtypes = [vp, u32, u32, vp]
    L.vxo_env_texture.argtypes = [P(VxoEnvironment), f32, f32, P(f32)]
    L.vxo_env_build_importance.argtypes = [vp, u32, u32, vp]
    L.vxo_env_sample.argtypes = [P(VxoEnvironment), f32, f32, f32, P(f32), P(f32)]
    L.vxo_env_lookup.argtypes = [P(VxoEnvironment), f32, P(f32), P(f32)]
    L.vxo_env_pdf.argtypes = [P(VxoEnvironment), f32, P(f32)]; L.vxo_env_pdf.restype = f32
    L.vxo_primary_ray.argtypes = [P(VxParams), u32, i32, i32, P(f32), P(f32)]
    L.vxo_blit.argtypes = [vp, u32, f32, f32, vp, vp]
    L.vxo_skip_level.argtypes = [P(VxoVolume)]; L.vxo_skip_level.restype = C.c_int
    L.vxo_skip_dims.argtypes = [P(VxoVolume), C.c_int, P(u32)]
    L.vxo_build_skip_mask.argtypes = [P(VxParams), P(VxoVolume), vp, u32, C.c_int, vp]
    L.vxo_render_ray_samples.argtypes = [P(VxParams), u32, P(VxoVolume), vp, u32, P(VxoEnvironment), vp, vp, i32, i32, i32,
                                         i32, P(VxoCounters)]
    L.vxo_render_ray_samples.restype = C.c_int
    L.vxo_set_dvr_march.argtypes = [C.c_int]
    L.vxo_get_dvr_march.restype = C.c_int
    _variants[name] = L
    return L


# ---------------------------------------------------------------------------------------
def rng_stream(seed: int, n: int):
    """first n raw xoshiro outputs and rng() floats for a seed (random.glsl:69-106)."""
    L = lib()
    s = (C.c_uint32 * 4)()
    L.vxo_seed_xoshiro(seed, s)
    raw = np.empty(n, dtype=np.uint32)
    for i in range(n):
        raw[i] = L.vxo_xoshiro_next(s)
    return raw, (raw >> 8).astype(np.float32) / np.float32(16777216.0)


class BrickGrid:
    """Result of the oracle's BrickGrid::construct; field names follow common.ts:37-55."""

    def __init__(self, voxels: np.ndarray, spacing=(1.0, 1.0, 1.0), max_value: int = 0):
        L = lib()
        v = np.ascontiguousarray(voxels, dtype=np.uint16)
        self.voxels = v
        self.dims = (v.shape[2], v.shape[1], v.shape[0])
        self.max_value = int(max_value) if max_value else int(v.max())
        dims = (C.c_uint32 * 3)(*self.dims)
        bc = (C.c_uint32 * 3)()
        L.vxo_brick_count(dims, bc)
        self.brick_count = tuple(int(x) for x in bc)
        nb = self.brick_count[0] * self.brick_count[1] * self.brick_count[2]
        self.indirection = np.zeros(nb, dtype=np.uint32)
        packed = np.zeros(nb, dtype=np.uint32)
        atlas = np.zeros(nb * 512, dtype=np.uint8)
        mp = [np.zeros(max(nb >> (3 * (k + 1)), 1), dtype=np.uint32) for k in range(3)]
        asz = (C.c_uint32 * 3)()
        n = L.vxo_brick_construct(v.ctypes.data, dims, self.max_value, self.indirection.ctypes.data,
                                  packed.ctypes.data, atlas.ctypes.data, mp[0].ctypes.data,
                                  mp[1].ctypes.data, mp[2].ctypes.data, asz)
        if n < 0:
            raise RuntimeError("Exceeded max brick count")
        self.brick_counter = int(n)
        self.indirection_size = self.brick_count
        self.range_size = self.brick_count
        self.atlas_size = tuple(int(x) for x in asz)
        self.atlas = atlas[: self.atlas_size[0] * self.atlas_size[1] * self.atlas_size[2]].copy()
        self.range_packed = packed
        self.range = packed.view(np.uint16).copy()       # LE: [max, min] per brick
        self.range_mipmaps = []
        for k in range(3):
            st = tuple(b >> (k + 1) for b in self.brick_count)
            self.range_mipmaps.append((mp[k][: st[0] * st[1] * st[2]].view(np.uint16).copy(), st))
        self.index_extent = tuple(b * 8 for b in self.brick_count)
        self.min_maj = (0.0, 1.0)                         # dicom.rs:19-21
        self.transform = np.diag([spacing[0], spacing[1], spacing[2], 1.0]).astype(np.float32).T.reshape(16)
        # histogram (lib.rs:87-102): 2^bits_stored bins; synthetic stacks are 12 bit
        bins = 4096 if self.max_value < 4096 else 65536
        self.histogram = np.bincount(v.reshape(-1), minlength=bins).astype(np.uint32)
        sm = np.zeros(bins, dtype=np.int32)
        gmin, gmax = C.c_uint32(), C.c_uint32()
        L.vxo_histogram_gradient(self.histogram.ctypes.data, bins, sm.ctypes.data, C.byref(gmin),
                                 C.byref(gmax))
        self.histogram_gradient = sm
        self.histogram_gradient_range = (gmin.value, gmax.value)

    def volume_struct(self) -> VxoVolume:
        return make_volume(self)


def make_volume(g) -> VxoVolume:
    """VxoVolume over any object with the WasmWorkerMessageDicomReturn fields."""
    vol = VxoVolume()
    keep = []

    def arr(a, dt):
        a = np.ascontiguousarray(a, dtype=dt)
        keep.append(a)
        return a.ctypes.data if a.size else None

    vol.indirection = arr(g.indirection, np.uint32)
    vol.range = arr(g.range, np.uint16)
    vol.atlas = arr(g.atlas, np.uint8)
    vol.ind_size[:] = list(g.indirection_size)
    vol.range_size[:] = list(g.range_size)
    vol.atlas_size[:] = list(g.atlas_size)
    for k, (m, st) in enumerate(g.range_mipmaps):
        vol.mips[k] = arr(m, np.uint16)
        vol.mip_size[k][:] = list(st)
    vol.index_extent[:] = list(g.index_extent)
    vol._keep = keep
    return vol


def copy_params(src) -> VxParams:
    """bytewise copy of a VxParams coming from the product's ctypes struct."""
    dst = VxParams()
    assert C.sizeof(dst) == C.sizeof(src)
    C.memmove(C.byref(dst), C.byref(src), C.sizeof(dst))
    return dst


class dvr_march:
    """`with oracle.dvr_march(walk_t=True): ...` -- render DVR with the march contract of rounds 1-2 (vx_oracle.h
    vxo_set_dvr_march); the shipped contract is restored on exit"""

    def __init__(self, walk_t):
        self.walk_t = int(bool(walk_t))

    def __enter__(self):
        self.L = lib()
        self.prev = self.L.vxo_get_dvr_march()
        self.L.vxo_set_dvr_march(self.walk_t)

    def __exit__(self, *exc):
        self.L.vxo_set_dvr_march(self.prev)


def render(params, grid, tf, tf_len, frame_index=0, sample_weight=0.0, prev=None, rect=None,
           threads=None, env=None, ray_samples=False):
    """fragment.frag main over the image (or rect = (x0,x1,y0,y1)); returns (rgba, counters) -- with ray_samples=True
    (rgba, counters, uint32 (H, W) samples evaluated per pixel)."""
    L = lib()
    p = copy_params(params)
    W, H = p.res[0], p.res[1]
    vol = grid if isinstance(grid, VxoVolume) else make_volume(grid)
    tf = np.ascontiguousarray(tf, dtype=np.float32)
    out = np.zeros((H, W, 4), dtype=np.float32)
    if prev is not None:
        prev = np.ascontiguousarray(prev, dtype=np.float32)
    x0, x1, y0, y1 = rect if rect else (0, W, 0, H)
    threads = threads or min(os.cpu_count() or 1, 16)
    bands = np.linspace(y0, y1, min(threads * 4, max(y1 - y0, 1)) + 1).astype(int)
    totals = VxoCounters()

    per_ray = np.zeros((H, W), dtype=np.uint32) if ray_samples else None

    def work(i):
        c = VxoCounters()
        if ray_samples:
            rc = L.vxo_render_ray_samples(C.byref(p), frame_index, C.byref(vol), tf.ctypes.data, tf_len,
                                          C.byref(env.c) if env is not None else None, out.ctypes.data,
                                          per_ray.ctypes.data, x0, x1, int(bands[i]), int(bands[i + 1]), C.byref(c))
            assert rc == 0, rc
            return c
        rc = L.vxo_render_env(C.byref(p), frame_index, sample_weight, C.byref(vol), tf.ctypes.data,
                              tf_len, C.byref(env.c) if env is not None else None,
                              prev.ctypes.data if prev is not None else None, out.ctypes.data,
                              x0, x1, int(bands[i]), int(bands[i + 1]), C.byref(c))
        assert rc == 0, rc
        return c

    if threads == 1:
        cs = [work(i) for i in range(len(bands) - 1)]
    else:
        with ThreadPoolExecutor(threads) as ex:
            cs = list(ex.map(work, range(len(bands) - 1)))
    for c in cs:
        for f, _ in VxoCounters._fields_:
            setattr(totals, f, getattr(totals, f) + getattr(c, f))
    if ray_samples:
        return out, totals, per_ray
    return out, totals


def primary_ray(params, px, py, frame_index=0):
    L = lib()
    p = copy_params(params)
    o, d = (C.c_float * 3)(), (C.c_float * 3)()
    L.vxo_primary_ray(C.byref(p), frame_index, px, py, o, d)
    return np.array(o[:], dtype=np.float32), np.array(d[:], dtype=np.float32)


def blit(accum, exposure=5.5, gamma=2.2):
    L = lib()
    a = np.ascontiguousarray(accum, dtype=np.float32)
    n = a.size // 4
    out8 = np.zeros((n, 4), dtype=np.uint8)
    outf = np.zeros((n, 4), dtype=np.float32)
    L.vxo_blit(a.ctypes.data, n, exposure, gamma, out8.ctypes.data, outf.ctypes.data)
    return out8.reshape(a.shape), outf.reshape(a.shape)


def skip_mask(params, grid, tf, tf_len):
    """(level, dims, bits) of the exact empty-space mask (vxo_build_skip_mask)"""
    L = lib()
    p = copy_params(params)
    vol = grid if isinstance(grid, VxoVolume) else make_volume(grid)
    tf = np.ascontiguousarray(tf, dtype=np.float32)
    level = L.vxo_skip_level(C.byref(vol))
    dims = (C.c_uint32 * 3)()
    L.vxo_skip_dims(C.byref(vol), level, dims)
    n = dims[0] * dims[1] * dims[2]
    bits = np.zeros((n + 31) // 32, dtype=np.uint32)
    L.vxo_build_skip_mask(C.byref(p), C.byref(vol), tf.ctypes.data, tf_len, level, bits.ctypes.data)
    return level, tuple(dims), bits
