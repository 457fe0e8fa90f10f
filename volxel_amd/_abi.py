"""ctypes binding of the C ABI declared in include/volxel_hip.h and include/volxel_brick.h.

The struct layouts are parsed from the headers so that the header stays the single source
of truth for the boundary.  Loading fails loudly when libvolxel_hip.so is missing: there is
no CPU fallback in the product path.
"""
from __future__ import annotations

import ctypes as C
import os
import re

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INCLUDE_DIR = os.path.join(_ROOT, "include")
# VOLXEL_HIP_LIB: another build of the same library (variant probes of tools/); default: the in-tree build
LIB_PATH = os.environ.get("VOLXEL_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libvolxel_hip.so")

_CTYPES = {
    "float": C.c_float, "double": C.c_double, "int32_t": C.c_int32, "uint32_t": C.c_uint32,
    "uint64_t": C.c_uint64, "int64_t": C.c_int64,
}


def struct_from_header(header: str, name: str):
    """Build a ctypes.Structure from `typedef struct <name> { ... } <name>;` in a C header."""
    text = open(os.path.join(INCLUDE_DIR, header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    m = re.search(r"typedef\s+struct\s+%s\s*\{(.*?)\}\s*%s\s*;" % (name, name), text, flags=re.S)
    if not m:
        raise RuntimeError(f"struct {name} not found in {header}")
    fields = []
    for decl in m.group(1).split(";"):
        decl = decl.strip()
        if not decl:
            continue
        mm = re.match(r"(\w+)\s+(.*)$", decl, flags=re.S)
        ctype = _CTYPES[mm.group(1)]
        for item in mm.group(2).split(","):
            item = item.strip()
            am = re.match(r"(\w+)\s*\[(\d+)\]$", item)
            if am:
                fields.append((am.group(1), ctype * int(am.group(2))))
            else:
                fields.append((item, ctype))
    return type(name, (C.Structure,), {"_fields_": fields})


VxParams = struct_from_header("volxel_hip.h", "VxParams")
VxCounters = struct_from_header("volxel_hip.h", "VxCounters")

MODE_DEFAULT, MODE_NO_DDA, MODE_RAYMARCH, MODE_DVR, MODE_DVR_PHONG = range(5)
LAYOUT_REFERENCE, LAYOUT_CELLQUAD, LAYOUT_BRICKF32, LAYOUT_AUTO = 0, 1, 2, 3
RENDER_MODES = {"default": MODE_DEFAULT, "no_dda": MODE_NO_DDA, "raymarch": MODE_RAYMARCH,
                "dvr": MODE_DVR, "dvr_phong": MODE_DVR_PHONG}
SHARD_TILE = 64


def declared_symbols(header: str):
    """Names of all functions a header declares (used by the symbol-export test)."""
    text = open(os.path.join(INCLUDE_DIR, header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vxb?_\w+)\s*\(", text)))


_lib = None


class VolxelLibraryMissing(RuntimeError):
    pass


def load_library():
    """Load libvolxel_hip.so (built by __graft_entry__.build() / volxel_amd/csrc/Makefile)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VolxelLibraryMissing(
            f"{LIB_PATH} is missing: build it with `make -C volxel_amd/csrc` "
            "(python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
    # One HIP runtime per process: PyTorch wheels bundle their own libamdhip64.so.7 /
    # libhsa-runtime64.so.1.  If torch is going to be used in this process (RCCL gather,
    # zero-copy slabs) its runtime has to be loaded FIRST so that our NEEDED libamdhip64.so.7
    # binds to the same instance; loading the system runtime first and torch's second gives
    # two HSA runtimes and torch reports "No HIP GPUs are available".
    try:
        import torch  # noqa: F401
    except Exception:  # torch absent (e.g. Node host): the system ROCm runtime is used
        pass
    lib = C.CDLL(LIB_PATH)
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
    P = C.POINTER
    sig = {
        "vx_create": ([i32, P(vp)], i32),
        "vx_destroy": ([vp], None),
        "vx_last_error": ([vp], C.c_char_p),
        "vx_set_stream": ([vp, vp], i32),
        "vx_upload_volume": ([vp, vp, P(u32), vp, P(u32), vp, P(u32), i32, P(vp), vp, P(u32)], i32),
        "vx_set_layout": ([vp, i32], i32),
        "vx_upload_transfer": ([vp, vp, u32], i32),
        "vx_upload_brick_grid": ([vp, vp], i32),
        "vx_upload_environment": ([vp, vp, u32, u32], i32),
        "vx_debug_read_importance": ([vp, vp], i32),
        "vx_set_params": ([vp, P(VxParams)], i32),
        "vx_resize": ([vp, u32, u32], i32),
        "vx_render_frame": ([vp, u32, C.c_float], i32),
        "vx_render_frames": ([vp, u32, u32, P(C.c_float), i32], i32),
        "vx_finish": ([vp], i32),
        "vx_read_accum": ([vp, vp], i32),
        "vx_read_display": ([vp, vp, C.c_float, C.c_float], i32),
        "vx_read_display_scaled": ([vp, vp, u32, u32, C.c_float, C.c_float], i32),
        "vx_slab_info": ([vp, P(u64), P(u32)], i32),
        "vx_probe_tile_costs": ([vp, vp, u32], i32),
        "vx_set_tile_order": ([vp, vp, u32], i32),
        "vx_render_size": ([vp, P(u32), P(u32)], i32),
        "vx_slab_device_ptr": ([vp, P(vp)], i32),
        "vx_detile": ([vp, vp, vp], i32),
        "vx_get_counters": ([vp, P(VxCounters)], i32),
        "vx_reset_counters": ([vp], i32),
        "vx_device_info": ([vp, C.c_char_p, u32, P(u32), P(u64)], i32),
        "vx_version": ([], C.c_char_p),
        "vx_debug_unorm_table": ([vp, vp], i32),
        "vx_debug_rng": ([vp, i32, vp, vp, u32, vp], i32),
        "vx_probe_gather_rate": ([vp, u32, u32, P(C.c_double), P(u32)], i32),
        "vx_probe_gather_spread": ([vp, u32, P(u64)], i32),
        "vx_probe_valu_rate": ([vp, P(C.c_double), P(u32)], i32),
        "vx_upload_stats": ([vp, P(C.c_double), P(u64), P(i32)], i32),
        "vx_debug_build_skip_mask": ([vp, P(u32), vp, u32, P(VxParams), vp, P(u32), P(u32)], i32),
        # preprocessor
        "vxb_build_from_u16": ([vp, P(u32), P(C.c_float), C.c_uint16, i32, P(vp)], i32),
        "vxb_read_dicoms_to_grid": ([P(vp), P(u64), u32, i32, P(vp)], i32),
        "vxb_free": ([vp], None),
        "vxb_last_error": ([], C.c_char_p),
        "vxb_indirection_size": ([vp, P(u32)], None),
        "vxb_range_size": ([vp, P(u32)], None),
        "vxb_atlas_size": ([vp, P(u32)], None),
        "vxb_indirection_data": ([vp], vp),
        "vxb_range_data": ([vp], vp),
        "vxb_atlas_data": ([vp], vp),
        "vxb_range_mipmaps": ([vp], u32),
        "vxb_range_mipmap": ([vp, u32], vp),
        "vxb_range_mipmap_stride": ([vp, u32, P(u32)], None),
        "vxb_transform": ([vp, P(C.c_float)], None),
        "vxb_minorant": ([vp], C.c_float),
        "vxb_majorant": ([vp], C.c_float),
        "vxb_index_extent": ([vp, P(u32)], None),
        "vxb_histogram_len": ([vp], u32),
        "vxb_histogram": ([vp], vp),
        "vxb_histogram_gradient": ([vp], vp),
        "vxb_histogram_gradient_min": ([vp], u32),
        "vxb_histogram_gradient_max": ([vp], u32),
        "vxb_brick_counter": ([vp], u32),
        "vxb_lookup": ([vp, u32, u32, u32], C.c_float),
    }
    for name, (args, res) in sig.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.argtypes = args
        fn.restype = res
    _lib = lib
    return lib
