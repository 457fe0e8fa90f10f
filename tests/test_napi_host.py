"""The Node N-API shim + JavaScript host (volxel_amd/napi): loads under the image's Node 12,
exports the boundary, its brick builder output equals the oracle's, it refuses to run without a
GPU (CPU test), and one DVR frame rendered from JavaScript matches the oracle (GPU test)."""
import ctypes as C
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAPI = os.path.join(ROOT, "volxel_amd", "napi")

pytestmark = pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")


def _build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "volxel_amd", "csrc"), "-s"])
    subprocess.check_call(["make", "-C", NAPI, "-s"])


def _run(tmp_path, *extra):
    _build()
    subprocess.check_call(["node", os.path.join(NAPI, "smoke.js"), str(tmp_path), *extra], timeout=300)


def _check_grid(oracle, tmp_path):
    from volxel_amd import synth
    vox, sp = synth.sphere(32)
    g = oracle.BrickGrid(vox, sp)
    meta = json.load(open(tmp_path / "meta.json"))
    assert tuple(meta["atlasSize"]) == g.atlas_size and tuple(meta["indexExtent"]) == g.index_extent
    assert meta["brickCounter"] == g.brick_counter
    assert np.array_equal(np.fromfile(tmp_path / "atlas.bin", dtype=np.uint8), g.atlas)
    assert np.array_equal(np.fromfile(tmp_path / "indirection.bin", dtype=np.uint32), g.indirection)
    assert np.array_equal(np.fromfile(tmp_path / "range.bin", dtype=np.uint16), g.range)
    from volxel_amd import default_transfer_function
    assert np.array_equal(np.fromfile(tmp_path / "tf.bin", dtype=np.float32), default_transfer_function()[0])
    return g


def _write_dicoms(tmp_path):
    from tests.dicom_writer import write_slice
    from volxel_amd import synth
    vox, _ = synth.sphere(32)
    os.makedirs(tmp_path / "dicom")
    for z in range(vox.shape[0]):
        with open(tmp_path / "dicom" / ("%04d.dcm" % z), "wb") as f:
            f.write(write_slice(vox[z], spacing=(1.0, 1.0), thickness=2.0, syntax=("1.2.840.10008.1.2", "1.2.840.10008.1.2.1")[z & 1]))


def _check_dicom_path(tmp_path):
    assert np.array_equal(np.fromfile(tmp_path / "atlas_dcm.bin", dtype=np.uint8),
                          np.fromfile(tmp_path / "atlas.bin", dtype=np.uint8))
    m = json.load(open(tmp_path / "meta_dcm.json"))
    assert m["histogramLength"] == 4096 and m["transform"][0] == 1.0 and m["transform"][10] == 2.0
    assert "DICM" in json.load(open(tmp_path / "dcm_error.json"))["msg"]


def test_node_host_cpu(oracle, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    _write_dicoms(tmp_path)
    _run(tmp_path)
    _check_grid(oracle, tmp_path)
    _check_dicom_path(tmp_path)
    assert json.load(open(tmp_path / "nogpu.json"))["threw"] is True   # no JavaScript/CPU fallback
    keys = subprocess.check_output(["node", "-e", "console.log(Object.keys(require('%s')).join(','))" % NAPI]).decode()
    for k in ("Volxel3DDicomRenderer", "VolxelRenderMode", "generateTransferFunction"):
        assert k in keys


@pytest.mark.gpu
def test_node_host_renders_like_the_oracle(oracle, tmp_path):
    import copy
    from volxel_amd import BENCHMARK_SETTINGS
    shared = copy.deepcopy(BENCHMARK_SETTINGS)
    shared["display"]["samples"] = 6
    coll = {"sharedSettings": [shared], "benchmarks": [{"renderMode": m, "settings": 0, "name": m}
                                                      for m in ("default", "no_dda", "raymarch")]}
    json.dump(coll, open(tmp_path / "bench.json", "w"))
    _run(tmp_path, "gpu", str(tmp_path / "bench.json"))
    assert json.load(open(tmp_path / "pipelined.json")) == {"same": True, "frameIndex": 13}
    tl = json.load(open(tmp_path / "tiles.json"))
    assert tl["tiles"] == 2 and tl["nonzero"] >= 1 and "gfx950" in tl["device"]["name"] and tl["device"]["computeUnits"] == 256
    res = json.load(open(tmp_path / "benchmark_results.json"))
    assert [x["name"] for x in res] == ["default", "no_dda", "raymarch"]
    for x in res:                                                     # frames 0..6
        assert x["totalTime"] > 0 and abs(x["timePerSample"] - x["totalTime"] / 7) < 1e-9
        assert x["viewport"] == [0, 0, 0.8 * 96, 0.8 * 64] and x["settings"]["renderMode"] == x["name"]
    g = _check_grid(oracle, tmp_path)
    p = oracle.VxParams()
    raw = open(tmp_path / "params.bin", "rb").read()
    assert len(raw) == C.sizeof(p)
    C.memmove(C.byref(p), raw, len(raw))
    from volxel_amd import default_transfer_function
    tf, L = default_transfer_function()
    from tests.common import default_environment
    assert p.use_env == 1                         # the JS host uploads the default map like the viewer
    want, oc = oracle.render(p, g, tf, L, env=default_environment(oracle))
    img = np.fromfile(tmp_path / "accum.bin", dtype=np.float32).reshape(p.res[1], p.res[0], 4)
    assert np.abs(img - want).max() <= 1e-5       # map background: atan/acos from two math libraries
    assert json.load(open(tmp_path / "counters.json"))["samples"] == oc.samples
    # the JS host's uniforms agree with the Python host's (both restate viewer.ts:1295-1357)
    from tests.common import make_scene
    s, cam, vol, ds, pp = make_scene(g, p.res[0], p.res[1], "dvr", env=True)
    a = np.frombuffer(raw, dtype=np.float32)
    b = np.frombuffer(bytes(pp), dtype=np.float32)
    assert np.allclose(a[:140], b[:140], rtol=1e-6, atol=1e-7)
