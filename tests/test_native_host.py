"""CPU-side tests of the product: the C-ABI library loads and exports every declared symbol,
the native C++ brick builder is byte-identical to the oracle's restatement of
BrickGrid::construct, and the host math mirrors the reference's TS."""
import ctypes as C
import math
import os
import subprocess
import tempfile

import numpy as np
import pytest

from volxel_amd import _abi, synth
from volxel_amd.preprocessor import read_u16_stack_to_grid


def test_library_exports_every_declared_symbol(native_lib):
    for header in ("volxel_hip.h", "volxel_brick.h"):
        for name in _abi.declared_symbols(header):
            assert hasattr(native_lib, name), f"{name} declared in {header} but not exported"
    assert b"gfx950" in native_lib.vx_version()


def test_library_contains_gfx950_code_object():
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o",
                          f"--input={_abi.LIB_PATH}"], capture_output=True, text=True)
    data = open(_abi.LIB_PATH, "rb").read()
    assert b"gfx950" in data and b"render_dvr_cq" in data


def test_vx_create_fails_loudly_without_gpu(native_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ctx = C.c_void_p()
    rc = native_lib.vx_create(0, C.byref(ctx))
    assert rc != 0 and not ctx.value
    assert b"no HIP device" in native_lib.vx_last_error(None) or b"failed" in native_lib.vx_last_error(None)
    from volxel_amd import Volxel3DRenderer, VolxelError
    with pytest.raises(VolxelError):
        Volxel3DRenderer(64, 64)


@pytest.mark.parametrize("case", ["sphere32", "noise_ragged", "noise64", "tiny"])
def test_native_brick_builder_matches_oracle(native_lib, oracle, case):
    if case == "sphere32":
        vox, sp = synth.sphere(32)
    elif case == "noise_ragged":
        vox, sp = synth.value_noise(64, seed=3)
        vox = np.ascontiguousarray(vox[:50, :41, :37])   # ragged extents, padded bricks
        sp = (0.7, 0.7, 1.0)
    elif case == "noise64":
        vox, sp = synth.value_noise(64, seed=9, zero_quantile=0.6)
    else:
        vox = np.array([[[7]]], dtype=np.uint16)           # 1 voxel: everything border
        sp = (1, 1, 1)
    g = oracle.BrickGrid(vox, sp)
    for threads in (1, 5):
        m = read_u16_stack_to_grid(vox, sp, n_threads=threads)
        assert m.indirection_size == g.indirection_size and m.range_size == g.range_size
        assert m.atlas_size == g.atlas_size and m.index_extent == g.index_extent
        assert m.brick_counter == g.brick_counter
        assert np.array_equal(m.indirection, g.indirection)
        assert np.array_equal(m.range, g.range)
        assert np.array_equal(m.atlas, g.atlas)
        for (a, sa), (b, sb) in zip(m.range_mipmaps, g.range_mipmaps):
            assert sa == sb and np.array_equal(a, b)
        assert np.array_equal(m.transform, g.transform)
        assert m.min_maj == (0.0, 1.0)
        assert np.array_equal(m.histogram, g.histogram)
        assert np.array_equal(m.histogram_gradient, g.histogram_gradient)
        assert tuple(m.histogram_gradient_range) == tuple(g.histogram_gradient_range)


def test_native_brick_builder_error_paths(native_lib):
    dims = (C.c_uint32 * 3)(8200, 8, 8)     # 1025 bricks -> "Exceeded max brick count" (brick.rs:79-81)
    sp = (C.c_float * 3)(1, 1, 1)
    vox = np.zeros(8200 * 64, dtype=np.uint16)
    vox[0] = 1
    g = C.c_void_p()
    rc = native_lib.vxb_build_from_u16(vox.ctypes.data, dims, sp, 0, 1, C.byref(g))
    assert rc == 2 and b"Exceeded max brick count" in native_lib.vxb_last_error()
    dims = (C.c_uint32 * 3)(8, 8, 8)
    zeros = np.zeros(512, dtype=np.uint16)
    rc = native_lib.vxb_build_from_u16(zeros.ctypes.data, dims, sp, 0, 1, C.byref(g))
    assert rc == 1 and b"all zero" in native_lib.vxb_last_error()
    rc = native_lib.vxb_build_from_u16(None, dims, sp, 0, 1, C.byref(g))
    assert rc == 1


def test_native_lookup_twin(native_lib, oracle):
    vox, sp = synth.value_noise(32, seed=2)
    dims = (C.c_uint32 * 3)(32, 32, 32)
    spc = (C.c_float * 3)(*sp)
    g = C.c_void_p()
    assert native_lib.vxb_build_from_u16(vox.ctypes.data, dims, spc, 0, 2, C.byref(g)) == 0
    og = oracle.BrickGrid(vox, sp)
    vol = og.volume_struct()
    rng = np.random.default_rng(0)
    for _ in range(500):
        x, y, z = (int(v) for v in rng.integers(0, 32, 3))
        assert native_lib.vxb_lookup(g, x, y, z) == oracle.lib().vxo_brick_lookup(vol, x, y, z)
    native_lib.vxb_free(g)


# ---- host math (scene.ts / volume.ts / data.ts) ----------------------------------------
def _ref_invert(m):
    """gl-matrix style cofactor inverse, pure Python (independent of numpy.linalg)."""
    a = [float(x) for x in m]
    (a00, a01, a02, a03, a10, a11, a12, a13, a20, a21, a22, a23, a30, a31, a32, a33) = a
    b00 = a00 * a11 - a01 * a10; b01 = a00 * a12 - a02 * a10; b02 = a00 * a13 - a03 * a10
    b03 = a01 * a12 - a02 * a11; b04 = a01 * a13 - a03 * a11; b05 = a02 * a13 - a03 * a12
    b06 = a20 * a31 - a21 * a30; b07 = a20 * a32 - a22 * a30; b08 = a20 * a33 - a23 * a30
    b09 = a21 * a32 - a22 * a31; b10 = a21 * a33 - a23 * a31; b11 = a22 * a33 - a23 * a32
    det = b00 * b11 - b01 * b10 + b02 * b09 + b03 * b08 - b04 * b07 + b05 * b06
    d = 1.0 / det
    return [(a11 * b11 - a12 * b10 + a13 * b09) * d, (a02 * b10 - a01 * b11 - a03 * b09) * d,
            (a31 * b05 - a32 * b04 + a33 * b03) * d, (a22 * b04 - a21 * b05 - a23 * b03) * d,
            (a12 * b08 - a10 * b11 - a13 * b07) * d, (a00 * b11 - a02 * b08 + a03 * b07) * d,
            (a32 * b02 - a30 * b05 - a33 * b01) * d, (a20 * b05 - a22 * b02 + a23 * b01) * d,
            (a10 * b10 - a11 * b08 + a13 * b06) * d, (a01 * b08 - a00 * b10 - a03 * b06) * d,
            (a30 * b04 - a31 * b02 + a33 * b00) * d, (a21 * b02 - a20 * b04 - a23 * b00) * d,
            (a11 * b07 - a10 * b09 - a12 * b06) * d, (a00 * b09 - a01 * b07 + a02 * b06) * d,
            (a31 * b01 - a30 * b03 - a32 * b00) * d, (a20 * b03 - a21 * b01 + a22 * b00) * d]


def test_camera_matrices_gl_matrix_conventions():
    from volxel_amd.scene import Camera, flat
    cam = Camera(1)
    v = flat(cam.view_matrix())
    # default eye (0,0,-1) looking at the origin, up +y (scene.ts:8-13): z axis = (0,0,-1)
    assert np.allclose(v, [-1, 0, 0, 0, 0, 1, 0, 0, 0, 0, -1, 0, 0, 0, -1, 1])
    pr = flat(cam.proj_matrix(16 / 9))
    f = 1 / math.tan(math.pi / 6)
    assert pr[0] == pytest.approx(f / (16 / 9)) and pr[5] == pytest.approx(f)
    assert pr[10] == pytest.approx((1000 + 0.1) / (0.1 - 1000)) and pr[11] == -1
    assert pr[14] == pytest.approx(2 * 1000 * 0.1 / (0.1 - 1000)) and pr[15] == 0
    cam.pos = np.array([0.66, 0.11, -0.81]); cam.view = np.array([0.0, 0.03, -0.05])
    m = flat(cam.view_matrix())
    inv = _ref_invert(m)
    assert np.allclose(np.array(inv).reshape(4, 4).T @ np.array(m).reshape(4, 4).T, np.eye(4), atol=1e-12)
    # the eye maps to the view-space origin
    eye = np.array(m).reshape(4, 4).T @ np.array([0.66, 0.11, -0.81, 1.0])
    assert np.allclose(eye[:3], 0, atol=1e-12)


def test_uniforms_follow_bind_uniforms(oracle):
    from tests.common import make_scene
    vox, sp = synth.sphere(32)
    g = oracle.BrickGrid(vox, (0.7, 0.7, 1.0))
    s, cam, vol, ds, p = make_scene(g, 64, 48, "dvr", clip_min=(0.25, 0, 0), clip_max=(1, 1, 0.75),
                                    density_multiplier=0.99)
    # viewer.ts:1089-1099: extent = spacing*index_extent, size = longest side
    assert g.index_extent == (64, 64, 64)   # padded to 8 bricks per axis (brick.rs:77, quirk Q7)
    ext = np.array([0.7, 0.7, 1.0]) * 64
    size = ext.max()
    assert ds == pytest.approx(size)
    assert p.volume_maj == pytest.approx(size * 0.99) and p.volume_inv_maj == pytest.approx(1 / (size * 0.99))
    assert p.volume_density_scale == pytest.approx(size * 0.99)
    lo, hi = -ext / 2 / size, ext / 2 / size
    assert np.allclose(p.volume_aabb_min[:], lo + (hi - lo) * [0.25, 0, 0], atol=1e-6)
    assert np.allclose(p.volume_aabb_max[:], lo + (hi - lo) * [1, 1, 0.75], atol=1e-6)
    dti = np.array(p.density_transform_inv[:]).reshape(4, 4).T
    # world centre -> index centre
    assert np.allclose(dti @ [0, 0, 0, 1], [32, 32, 32, 1], atol=1e-4)
    ref = _ref_invert(p.density_transform[:])
    assert np.allclose(p.density_transform_inv[:], ref, rtol=1e-5, atol=1e-5)
    assert np.allclose(p.camera_view_inv[:], _ref_invert(p.camera_view[:]), rtol=1e-5, atol=1e-6)
    assert np.allclose(p.camera_proj_inv[:], _ref_invert(p.camera_proj[:]), rtol=1e-4, atol=1e-5)
    # K = albedo*mis*f_p*Le (fragment.frag:94-97): 0.9 * 1/(1+fp^2) * 1/(4pi) * 4.01
    fp = 1 / (4 * math.pi)
    assert p.dvr_gain[0] == pytest.approx(0.9 / (1 + fp * fp) * fp * 4.01, rel=1e-6)


def test_sample_weight_running_mean():
    from volxel_amd import sample_weight
    assert [sample_weight(f) for f in range(6)] == [0, 0, 0, 0, 0, 0.0]
    assert sample_weight(6) == 0.5 and sample_weight(8) == pytest.approx(3 / 4)


def test_transfer_function_generation():
    from volxel_amd import default_transfer_function, generate_transfer_function, parse_transfer_function
    data, n = default_transfer_function()
    t = data.reshape(-1, 4)
    assert n == 128 and np.all(t[:, :3] == 1) and np.allclose(t[:, 3], np.arange(128) / 128)
    # a single stop: data.ts:35 is true at i = 0 (stop >= 0), so every entry is that colour
    data, n = generate_transfer_function([{"color": [1, 0, 0, 1], "stop": 0.5}], 8)
    assert np.all(data.reshape(-1, 4) == [1, 0, 0, 1])
    # two stops: linear ramp between them, last colour held after the final stop (data.ts:42-53)
    data, n = generate_transfer_function([{"color": [0, 0, 0, 0], "stop": 0.0}, {"color": [1, 1, 1, 1], "stop": 0.5}], 8)
    t = data.reshape(-1, 4)
    assert np.allclose(t[:, 3], [0, 0.25, 0.5, 0.75, 1, 1, 1, 1])
    with pytest.raises(ValueError):
        generate_transfer_function([])
    with pytest.raises(ValueError):
        generate_transfer_function([{"color": [0, 0, 0, 0], "stop": 1.5}])
    d, n, rows = parse_transfer_function("1 0 0 0.5\n0 1 0 1\nbad line\n")
    assert n == 2 and d.tolist() == [1, 0, 0, 0.5, 0, 1, 0, 1]


def test_settings_v3_schema():
    import copy
    from volxel_amd import BENCHMARK_SETTINGS, verify_settings
    verify_settings(copy.deepcopy(BENCHMARK_SETTINGS))
    bad = copy.deepcopy(BENCHMARK_SETTINGS); bad["version"] = "v2"
    with pytest.raises(ValueError):
        verify_settings(bad)
    bad = copy.deepcopy(BENCHMARK_SETTINGS); bad["display"]["renderMode"] = "fancy"
    with pytest.raises(ValueError):
        verify_settings(bad)
    bad = copy.deepcopy(BENCHMARK_SETTINGS); bad["other"]["clipMin"] = [0, 0]
    with pytest.raises(ValueError):
        verify_settings(bad)


def test_unorm8_newton_step_is_exact_division():
    """the device decodes R8 unorm as q=c*r; e=fma(-q,255,c); fma(e,r,q); must equal RN(c/255)"""
    src = r'''
#include <math.h>
#include <stdio.h>
int main(void){ int bad=0; const float r=1.0f/255.0f;
 for(int c=0;c<256;++c){ float fc=(float)c; float q=fc*r; float e=fmaf(-q,255.0f,fc); float v=fmaf(e,r,q);
   if(v!=fc/255.0f) ++bad; }
 printf("%d\n",bad); return 0; }'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-O0", "-ffp-contract=off", os.path.join(d, "t.c"), "-o", os.path.join(d, "t"), "-lm"])
        assert subprocess.check_output([os.path.join(d, "t")]).strip() == b"0"


@pytest.mark.parametrize("case", ["noise_bench_tf", "sphere_default_tf", "ct_narrow_range"])
def test_skip_mask_matches_oracle_and_is_exact(native_lib, oracle, case):
    """the library's host-built empty-space mask equals the oracle's restatement of the rule, and
    skipping never changes a pixel (the skipped samples have alpha == 0 exactly)"""
    from tests.common import make_scene, benchmark_tf, BENCH_CAM
    from volxel_amd import default_transfer_function
    if case == "noise_bench_tf":
        inner, sp = synth.value_noise(64, seed=5, zero_quantile=0.6)
        vox = np.zeros((128, 128, 128), dtype=np.uint16)      # noise block inside empty space
        vox[20:84, 30:94, 40:104] = inner
        tf, L = benchmark_tf()
        kw = dict(sample_range=(0.05645751953125, 1.0), **BENCH_CAM)
    elif case == "sphere_default_tf":
        vox, sp = synth.sphere(64)
        tf, L = default_transfer_function()
        kw = dict(sample_range=(0.1, 0.9))
    else:
        vox, sp = synth.ct_phantom(64)
        tf, L = benchmark_tf()
        kw = dict(sample_range=(0.3, 0.5), density_multiplier=0.99, **BENCH_CAM)
    g = oracle.BrickGrid(vox, sp)
    s, cam, vol, ds, p = make_scene(g, 64, 48, "dvr", **kw)
    level, dims, bits = oracle.skip_mask(p, g, tf, L)
    bc = (C.c_uint32 * 3)(*g.brick_count)
    lvl, md = C.c_uint32(), (C.c_uint32 * 3)()
    got = np.zeros_like(bits)
    rp = np.ascontiguousarray(g.range_packed, dtype=np.uint32)
    tfa = np.ascontiguousarray(tf, dtype=np.float32)
    pp = _abi.VxParams()
    C.memmove(C.byref(pp), C.byref(p), C.sizeof(pp))
    rc = native_lib.vx_debug_build_skip_mask(rp.ctypes.data, bc, tfa.ctypes.data, L, C.byref(pp), got.ctypes.data,
                                             C.byref(lvl), md)
    assert rc == 0 and lvl.value == level and tuple(md) == dims
    assert np.array_equal(got, bits)
    n_empty = int(sum(bin(int(w)).count("1") for w in bits))
    assert 0 < n_empty < dims[0] * dims[1] * dims[2]
    # exactness: identical pixels with and without skipping, fewer evaluated samples
    p.dvr_skip_empty = 1
    a, ca = oracle.render(p, g, tf, L)
    p.dvr_skip_empty = 0
    b, cb = oracle.render(p, g, tf, L)
    assert np.array_equal(a, b)
    assert ca.samples < cb.samples and ca.samples + ca.skip_steps == cb.samples


def test_benchmark_collection_schema():
    """the VolxelBenchmark payload of data-benchmark-url (viewer.ts:72-82, public/benchmark.json)"""
    import copy
    from volxel_amd import BENCHMARK_SETTINGS, BENCHMARK_COLLECTION_MODES, ViewerSettings, verify_benchmark
    coll = {"sharedSettings": [copy.deepcopy(BENCHMARK_SETTINGS)],
            "benchmarks": [{"renderMode": m, "settings": 0} for m in BENCHMARK_COLLECTION_MODES]}
    assert verify_benchmark(coll) is coll
    for bad in ({"benchmarks": []}, {"sharedSettings": [], "benchmarks": [{"settings": 0}]},
                {"sharedSettings": [BENCHMARK_SETTINGS], "benchmarks": [{"settings": 0, "renderMode": "fast"}]},
                {"sharedSettings": [BENCHMARK_SETTINGS], "benchmarks": [{"settings": True}]},
                {"sharedSettings": [{"version": "v2"}], "benchmarks": []}):
        with pytest.raises(ValueError):
            verify_benchmark(bad)
    d = ViewerSettings().to_viewer_dict()
    for k in ("densityMultiplier", "maxSamples", "debugHits", "volumeClipMin", "volumeClipMax", "showEnvironment",
              "useEnv", "lightDir", "syncLightDir", "bounces", "gamma", "exposure", "sampleRange", "renderMode",
              "resolutionFactor"):                      # viewer.ts:147-163
        assert k in d
    assert d["maxSamples"] == 2000 and d["renderMode"] == "default" and d["sampleRange"] == [0.0, 1.0]


def test_transfer_tables_match_committed_fixture():
    """the viewer's default ramp and the benchmark.json stops as 128-entry tables:
    fixture = independent restatement of data.ts:21-60 in tests/golden/make_golden.py"""
    from tests.common import benchmark_tf
    from tests.golden.make_golden import tf_fixture
    from volxel_amd import default_transfer_function
    want = np.load(os.path.join(os.path.dirname(__file__), "golden", "tf_tables.npz"))
    fresh = tf_fixture()
    for k in ("default", "benchmark"):
        assert np.array_equal(fresh[k], want[k])
    assert np.array_equal(default_transfer_function()[0], want["default"])
    assert np.array_equal(benchmark_tf()[0], want["benchmark"])


def test_brick_dumps_fixture(native_lib, oracle):
    """committed brick-encode dumps (16^3 random, 64^3 sphere): the oracle still produces them and the
    native builder produces the same bytes and the same Grid::lookup values"""
    from tests.golden.make_golden import brick_fixture
    from volxel_amd import synth
    want = np.load(os.path.join(os.path.dirname(__file__), "golden", "brick_dumps.npz"))
    fresh = brick_fixture(oracle)
    assert sorted(fresh) == sorted(want.files)
    for k in want.files:
        assert np.array_equal(fresh[k], want[k]), k
    rng = np.random.default_rng(2024)
    for tag, vox in (("r16", rng.integers(0, 4096, size=(16, 16, 16), dtype=np.uint16)), ("s64", synth.sphere(64)[0])):
        m = read_u16_stack_to_grid(vox, (1.0, 1.0, 1.0))
        assert np.array_equal(m.indirection, want[tag + "_indirection"])
        assert np.array_equal(m.range, want[tag + "_range"])
        assert np.array_equal(m.atlas, want[tag + "_atlas"])
        assert tuple(m.atlas_size) == tuple(want[tag + "_atlas_size"])
        for k in range(3):
            assert np.array_equal(m.range_mipmaps[k][0], want[f"{tag}_mip{k}"])


def test_headers_are_plain_c_and_link(native_lib, tmp_path):
    """a C99 host: both headers compile with -pedantic, the program links against libvolxel_hip.so, builds
    a brick grid from DICOM-free input, and gets the no-device error from vx_create on this box"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = r'''
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "volxel_hip.h"
#include "volxel_brick.h"
int main(void) {
  uint16_t* vox = (uint16_t*)calloc(16 * 16 * 16, 2);
  for (int i = 0; i < 16 * 16 * 16; ++i) vox[i] = (uint16_t)(i % 977);
  uint32_t dims[3] = {16, 16, 16};
  float sp[3] = {1.f, 1.f, 2.f};
  VxBrickGrid* g = NULL;
  if (vxb_build_from_u16(vox, dims, sp, 0, 1, &g) != VXB_OK) { printf("build: %s\n", vxb_last_error()); return 1; }
  uint32_t ext[3];
  float t[16];
  vxb_index_extent(g, ext);
  vxb_transform(g, t);
  printf("extent %u %u %u scale_z %g bricks %u lookup %.6f\n", ext[0], ext[1], ext[2], t[10], vxb_brick_counter(g),
         vxb_lookup(g, 3, 2, 1));
  VxParams p;
  memset(&p, 0, sizeof p);
  printf("sizeof(VxParams) %u version %s\n", (unsigned)sizeof p, vx_version());
  VxContext* c = NULL;
  int rc = vx_create(0, &c);
  if (rc == VX_OK) {            /* a GPU box: upload straight from the handle */
    rc = vx_upload_brick_grid(c, g);
    printf("gpu upload rc %d\n", rc);
    vx_destroy(c);
  } else {
    printf("create rc %d: %s\n", rc, vx_last_error(NULL));
  }
  vxb_free(g);
  free(vox);
  return 0;
}
'''
    (tmp_path / "host.c").write_text(src)
    exe = str(tmp_path / "host")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(root, "include"),
                           str(tmp_path / "host.c"), "-o", exe, "-L", os.path.join(root, "volxel_amd"), "-lvolxel_hip",
                           "-Wl,-rpath," + os.path.join(root, "volxel_amd")])
    out = subprocess.check_output([exe], timeout=120).decode()
    assert "extent 64 64 64 scale_z 2" in out and "sizeof(VxParams) %d" % C.sizeof(_abi.VxParams) in out
    assert "gpu upload rc 0" in out or "no HIP device" in out
