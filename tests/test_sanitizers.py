"""AddressSanitizer + UBSan run of the host-side native code (DICOM reader, brick builder) over truncated and
corrupted inputs.  CPU build only (GPU ASan is not available on the pool)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not installed")
def test_dicom_reader_and_brick_builder_under_asan_ubsan(tmp_path):
    from tests.dicom_writer import EXPLICIT, IMPLICIT, write_slice
    rng = np.random.default_rng(1)
    v = rng.integers(0, 4000, size=(3, 16, 24), dtype=np.uint16)
    for i, (syn, seq) in enumerate(((EXPLICIT, True), (IMPLICIT, True), (EXPLICIT, False))):
        (tmp_path / f"s{i}.dcm").write_bytes(write_slice(v[i], syntax=syn, with_sequence=seq))
    (tmp_path / "multi.dcm").write_bytes(write_slice(v, bits_stored=12, thickness=None))
    (tmp_path / "crafted.dcm").write_bytes(write_slice(v[0], frames_text="2147483648"))
    src = open(os.path.join(ROOT, "tests", "sanitize_harness.cpp.in")).read().replace("%DIR%", str(tmp_path))
    (tmp_path / "harness.cpp").write_text(src)
    exe = str(tmp_path / "harness")
    csrc = os.path.join(ROOT, "volxel_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-fno-omit-frame-pointer", "-I", os.path.join(ROOT, "include"), str(tmp_path / "harness.cpp"),
                           os.path.join(csrc, "brick_builder.cpp"), os.path.join(csrc, "dicom_reader.cpp"), "-o", exe,
                           "-lpthread"])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "full: rc 0" in p.stdout and "multi: rc 0" in p.stdout and "rejected" in p.stdout
    assert "crafted: rc 1" in p.stdout or "crafted: rc 2" in p.stdout, p.stdout
    assert "ERROR" not in p.stderr and "runtime error" not in p.stderr
