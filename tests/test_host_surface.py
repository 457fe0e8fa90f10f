"""The public method surface of the component that lies around the hot path (viewer.ts:963-1040,1442-1462,789-795):
both hosts (volxel_amd.Volxel3DRenderer, napi/viewer.js) carry the reference's names; the container adapters follow
zip.rs / hdr.rs; the orbit camera follows scene.ts:15-52; the synced light follows viewer.ts:789-795.  CPU only."""
import io
import json
import math
import os
import shutil
import struct
import subprocess
import zipfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAPI = os.path.join(ROOT, "volxel_amd", "napi")

REFERENCE_METHODS = {   # viewer.ts line -> (python name, javascript name)
    963: ("restart_from_files", "restartFromFiles"), 977: ("restart_from_zip", "restartFromZip"),
    991: ("restart_from_zip_url", "restartFromZipUrl"), 1005: ("restart_from_urls", "restartFromURLs"),
    1019: ("load_env", "loadEnv"), 1035: ("load_env_from_url", "loadEnvFromUrl"), 864: ("start_benchmark", "startBenchmark"),
    1442: ("render_mode", "renderMode"), 789: ("maybe_sync_light", "maybeSyncLight"),
}


def _zip(entries, method=zipfile.ZIP_DEFLATED):
    buf = io.BytesIO()
    with zipfile.ZipFile(buf, "w", method) as z:
        for name, data in entries:
            if name.endswith("/"):
                z.writestr(zipfile.ZipInfo(name), b"")
            else:
                z.writestr(name, data)
    return buf.getvalue()


def _rgbe_file(w, h, rle):
    """a Radiance file with known pixels + the floats it decodes to"""
    rng = np.random.default_rng(5)
    px = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    px[..., 3] = rng.integers(120, 136, size=(h, w))
    px[0, 0] = (0, 0, 0, 0)                                    # exponent 0: black
    out = b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n" + f"-Y {h} +X {w}\n".encode()
    for y in range(h):
        if rle:
            out += bytes([2, 2, w >> 8, w & 255])
            for ch in range(4):
                row = px[y, :, ch]
                x = 0
                while x < w:
                    run = 1
                    while x + run < w and run < 127 and row[x + run] == row[x]:
                        run += 1
                    if run >= 3:
                        out += bytes([128 + run, int(row[x])]); x += run
                    else:
                        n = min(w - x, 5)
                        out += bytes([n]) + row[x:x + n].tobytes(); x += n
        else:
            out += px[y].tobytes()
    e = px[..., 3].astype(np.int32)
    scale = np.where(e > 0, np.ldexp(1.0, e - 136), 0.0)
    want = np.ones((h, w, 4), dtype=np.float32)
    want[..., :3] = (px[..., :3].astype(np.float64) * scale[..., None]).astype(np.float32)
    return out, want


def test_zip_rules_follow_zip_rs():
    from volxel_amd import ZipReadError, read_zip_slices
    ok = _zip([("study/", b""), ("study/b.dcm", b"BBBB"), ("study/a.dcm", b"AA")])
    assert read_zip_slices(ok) == [b"BBBB", b"AA"]                 # archive order, not sorted (zip.rs:54-97)
    assert read_zip_slices(_zip([("x.dcm", b"1"), ("y.dcm", b"22")], zipfile.ZIP_STORED)) == [b"1", b"22"]
    with pytest.raises(ZipReadError, match="MoreThanOneFolder"):
        read_zip_slices(_zip([("a/", b""), ("b/", b"")]))
    with pytest.raises(ZipReadError, match="MoreThanOneFolder"):
        read_zip_slices(_zip([("a/", b""), ("a/x.dcm", b"1"), ("other/y.dcm", b"2")]))
    with pytest.raises(ZipReadError, match="NoFiles"):
        read_zip_slices(_zip([]))
    with pytest.raises(ZipReadError, match="ExtractFailed"):
        read_zip_slices(b"this is not a zip archive")
    with pytest.raises(ZipReadError, match="NoFiles: No dicom data collected"):
        read_zip_slices(_zip([("only/", b"")]))
    assert read_zip_slices(_zip([("..foo.dcm", b"7")])) == [b"7"]      # a legal enclosed name that merely starts with dots
    with pytest.raises(ZipReadError, match="ExtractFailed"):
        read_zip_slices(_zip([("../escape.dcm", b"7")]))


@pytest.mark.parametrize("rle", [False, True])
def test_radiance_environment_decode(rle):
    from volxel_amd import decode_environment
    data, want = _rgbe_file(16, 5, rle)
    floats, w, h = decode_environment(data)
    assert (w, h) == (16, 5) and np.array_equal(floats.reshape(5, 16, 4), want)
    with pytest.raises(ValueError, match="OpenEXR decode is outside"):
        decode_environment(b"\x76\x2f\x31\x01" + b"\0" * 32)
    with pytest.raises(ValueError, match="unrecognised"):
        decode_environment(b"PNG....")
    if rle:
        # malformed run-length data is an error, never a hang or a read past the buffer (the image crate errors too)
        head = b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 1 +X 16\n" + bytes([2, 2, 0, 16])
        for body in (bytes([0, 0, 0]),                 # zero-length literal: x would never advance
                     bytes([128, 7]),                   # zero-length run
                     bytes([128 + 17, 7]),              # run past the scanline
                     bytes([16, 1, 2, 3]),              # literal past the end of the buffer
                     bytes([128 + 16])):                # the run's value byte is missing
            with pytest.raises(ValueError, match="Radiance map"):
                decode_environment(head + body)
        with pytest.raises(ValueError, match="ends early"):
            decode_environment(data[:-3] if not rle else b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 2 +X 4\n" + bytes(20))


def _bare_renderer():
    """the host object without a device context (vx_create needs a GPU): enough for the camera / light logic"""
    from volxel_amd import Camera, ViewerSettings, Volxel3DRenderer
    r = object.__new__(Volxel3DRenderer)
    r.settings, r.camera = ViewerSettings(), Camera(1)
    r.low_res_preview, r.frame_index, r._ctx = False, 7, None
    return r


def test_python_host_has_the_reference_surface_and_syncs_the_light():
    from volxel_amd import Volxel3DRenderer, register_volxel_components
    for line, (py, _) in REFERENCE_METHODS.items():
        assert hasattr(Volxel3DRenderer, py), f"viewer.ts:{line} -> {py}"
    assert register_volxel_components(lambda: None)["volxel-3d-viewer"] is Volxel3DRenderer   # viewer.ts:1455-1462
    r = _bare_renderer()
    before = r.settings.light_dir
    r.rotate_camera((0.3, 0.1))                       # flag off: the light stays (viewer.ts:790)
    assert r.settings.light_dir == before and r.frame_index == 0
    r.sync_light_dir = True                           # the toggle re-aims at once (viewer.ts:546-549)
    d = r.camera.pos - r.camera.view
    assert np.allclose(r.settings.light_dir, d / np.linalg.norm(d), atol=1e-15)
    r.rotate_camera((-1.1, 0.4))
    d = r.camera.pos - r.camera.view
    assert np.allclose(r.settings.light_dir, d / np.linalg.norm(d), atol=1e-15)
    assert abs(np.linalg.norm(d) - 1.0) < 1e-12       # the orbit keeps the distance (scene.ts:30)


def test_orbit_camera_follows_scene_ts():
    from volxel_amd import Camera
    c = Camera(2.0)
    c.rotate_around_view((-math.pi / 2, 0.0))         # yaw += pi/2 about +y: (0,0,-1) -> (-1,0,0)
    assert np.allclose(c.pos, [-2.0, 0.0, 0.0], atol=1e-12)
    c.rotate_around_view((0.0, 10.0))                 # pitch clamps at pi/2 - 0.01 (scene.ts:19-21)
    assert abs(c.pitch - (math.pi / 2 - 0.01)) < 1e-15 and abs(np.linalg.norm(c.pos) - 2.0) < 1e-12
    c2 = Camera(1.0)
    assert c2.zoom(0.05) is False and c2.zoom(20.0) is False and c2.zoom(2.0) is True   # scene.ts:37
    assert np.allclose(c2.pos, [0, 0, -2.0])
    c2.translate_on_plane((0.1, 0.0))                 # right = dir x up = (0,0,-2)x(0,1,0) = (2,0,0) -> +x
    assert np.allclose(c2.pos - c2.view, [0, 0, -2.0]) and np.allclose(c2.view, [0.5, 0, 0])


@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_javascript_host_has_the_same_surface(tmp_path, native_lib):
    subprocess.check_call(["make", "-C", NAPI, "-s"])
    from volxel_amd import Camera
    z = _zip([("study/", b""), ("study/b.dcm", b"BBBB"), ("study/a.dcm", b"AA")])
    (tmp_path / "ok.zip").write_bytes(z)
    (tmp_path / "two.zip").write_bytes(_zip([("a/", b""), ("b/", b"")]))
    data, want = _rgbe_file(16, 5, True)
    (tmp_path / "env.hdr").write_bytes(data)
    script = r"""
const v = require(process.argv[2]); const fs = require('fs'); const dir = process.argv[3];
const out = {};
out.methods = Object.getOwnPropertyNames(v.Volxel3DDicomRenderer.prototype);
out.reg = Object.keys(v.registerVolxelComponents(() => 42)); out.factory = v.getWorkerFactory()();
out.zip = v.readZipSlices(fs.readFileSync(dir + '/ok.zip')).map(b => Buffer.from(b).toString());
try { v.readZipSlices(fs.readFileSync(dir + '/two.zip')); } catch (e) { out.two = e.message; }
try { v.readZipSlices(Buffer.from('nonsense')); } catch (e) { out.bad = e.message; }
const env = v.decodeEnvironment(fs.readFileSync(dir + '/env.hdr'));
out.env = { w: env.width, h: env.height }; fs.writeFileSync(dir + '/env.f32', Buffer.from(env.floats.buffer));
try { v.decodeEnvironment(Buffer.from([0x76, 0x2f, 0x31, 0x01, 0, 0, 0, 0])); } catch (e) { out.exr = e.message; }
const c = new v.Camera(1); c.rotateAroundView([0.3, 0.1]); c.rotateAroundView([-1.1, 0.4]); out.cam = c.pos;
const r = Object.create(v.Volxel3DDicomRenderer.prototype);
r.settings = { syncLightDir: false, lightDir: [1, 0, 0] }; r.camera = c; r.lowResPreview = false;
r.rotateCamera([0.2, 0]); out.unsynced = r.settings.lightDir; r.syncLightDir = true; out.synced = r.settings.lightDir; out.pos = c.pos;
console.log(JSON.stringify(out));
"""
    (tmp_path / "s.js").write_text(script)
    out = json.loads(subprocess.check_output(["node", str(tmp_path / "s.js"), NAPI, str(tmp_path)], timeout=120))
    for line, (_, js) in REFERENCE_METHODS.items():
        assert js in out["methods"], f"viewer.ts:{line} -> {js}"
    assert out["reg"] == ["volxel-3d-viewer"] and out["factory"] == 42
    assert out["zip"] == ["BBBB", "AA"] and out["two"].startswith("MoreThanOneFolder") and out["bad"].startswith("ExtractFailed")
    assert out["env"] == {"w": 16, "h": 5} and "OpenEXR" in out["exr"]
    assert np.array_equal(np.fromfile(tmp_path / "env.f32", dtype=np.float32).reshape(5, 16, 4), want)
    cam = Camera(1)
    cam.rotate_around_view((0.3, 0.1)); cam.rotate_around_view((-1.1, 0.4))
    assert np.allclose(out["cam"], cam.pos, atol=1e-14)           # both hosts restate scene.ts:15-33
    assert out["unsynced"] == [1, 0, 0]
    p = np.asarray(out["pos"])
    assert np.allclose(out["synced"], p / np.linalg.norm(p), atol=1e-15)
