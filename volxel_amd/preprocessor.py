"""Host wrapper of the native brick-layout producer (include/volxel_brick.h).

Mirrors the worker side of the reference: volxel-3d-viewer/src/worker.ts:19-58 calls the
wasm preprocessor, copies every buffer out, frees the grid and posts a
WasmWorkerMessageDicomReturn (common.ts:37-55).  `BrickGridMessage` is that message.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _abi


@dataclass
class BrickGridMessage:  # common.ts:37-55
    indirection_size: tuple
    range_size: tuple
    atlas_size: tuple
    transform: np.ndarray            # 16 float32, column major (brick.rs:305-307)
    histogram: np.ndarray
    histogram_gradient_range: tuple
    histogram_gradient: np.ndarray
    min_maj: tuple
    index_extent: tuple
    range_mipmaps: list              # [(uint16 array, (x,y,z))]
    indirection: np.ndarray          # uint32
    range: np.ndarray                # uint16 stream [max,min] per brick
    atlas: np.ndarray                # uint8
    brick_counter: int = 0           # not in the reference message; used for reporting


def _arr3(fn, g):
    a = (C.c_uint32 * 3)()
    fn(g, a)
    return tuple(int(x) for x in a)


def _view(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).copy()


def read_dicoms_to_grid(files, n_threads: int = 0) -> BrickGridMessage:
    """read_dicoms_to_grid (lib.rs:193-202): `files` = list of bytes objects, one DICOM slice (or
    multi-frame file) each, in stacking order -- the LOAD_FROM_BYTES message of worker.ts:101-104."""
    lib = _abi.load_library()
    bufs = [np.frombuffer(bytes(f), dtype=np.uint8) for f in files]
    n = len(bufs)
    ptrs = (C.c_void_p * max(n, 1))(*[b.ctypes.data for b in bufs])
    sizes = (C.c_uint64 * max(n, 1))(*[b.size for b in bufs])
    g = C.c_void_p()
    rc = lib.vxb_read_dicoms_to_grid(ptrs, sizes, n, int(n_threads), C.byref(g))
    if rc != 0:
        raise RuntimeError(lib.vxb_last_error().decode())
    return _message_from_grid(lib, g)


def read_u16_stack_to_grid(voxels: np.ndarray, spacing=(1.0, 1.0, 1.0), max_value: int = 0,
                           n_threads: int = 0) -> BrickGridMessage:
    """read_dicoms_to_grid (lib.rs:193-202) for an already decoded u16 stack [z,y,x]."""
    lib = _abi.load_library()
    v = np.ascontiguousarray(voxels, dtype=np.uint16)
    if v.ndim != 3:
        raise ValueError("voxels must be [z, y, x]")
    dims = (C.c_uint32 * 3)(v.shape[2], v.shape[1], v.shape[0])
    sp = (C.c_float * 3)(*[float(s) for s in spacing])
    g = C.c_void_p()
    rc = lib.vxb_build_from_u16(v.ctypes.data, dims, sp, int(max_value), int(n_threads), C.byref(g))
    if rc != 0:
        raise RuntimeError(lib.vxb_last_error().decode())
    return _message_from_grid(lib, g)


def _message_from_grid(lib, g) -> BrickGridMessage:
    try:
        ind_size = _arr3(lib.vxb_indirection_size, g)
        range_size = _arr3(lib.vxb_range_size, g)
        atlas_size = _arr3(lib.vxb_atlas_size, g)
        nb = ind_size[0] * ind_size[1] * ind_size[2]
        t = (C.c_float * 16)()
        lib.vxb_transform(g, t)
        mips = []
        for i in range(lib.vxb_range_mipmaps(g)):
            st = (C.c_uint32 * 3)()
            lib.vxb_range_mipmap_stride(g, i, st)
            st = tuple(int(x) for x in st)
            mips.append((_view(lib.vxb_range_mipmap(g, i), st[0] * st[1] * st[2] * 2, np.uint16), st))
        hl = lib.vxb_histogram_len(g)
        msg = BrickGridMessage(
            indirection_size=ind_size, range_size=range_size, atlas_size=atlas_size,
            transform=np.array(list(t), dtype=np.float32),
            histogram=_view(lib.vxb_histogram(g), hl, np.uint32),
            histogram_gradient_range=(lib.vxb_histogram_gradient_min(g),
                                      lib.vxb_histogram_gradient_max(g)),
            histogram_gradient=_view(lib.vxb_histogram_gradient(g), hl, np.int32),
            min_maj=(lib.vxb_minorant(g), lib.vxb_majorant(g)),
            index_extent=_arr3(lib.vxb_index_extent, g),
            range_mipmaps=mips,
            indirection=_view(lib.vxb_indirection_data(g), nb, np.uint32),
            range=_view(lib.vxb_range_data(g), nb * 2, np.uint16),
            atlas=_view(lib.vxb_atlas_data(g), atlas_size[0] * atlas_size[1] * atlas_size[2], np.uint8),
            brick_counter=lib.vxb_brick_counter(g))
    finally:
        lib.vxb_free(g)  # worker.ts:54 grid.free()
    return msg
