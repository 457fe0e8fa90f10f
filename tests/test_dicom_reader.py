"""N1: native DICOM slice reader == the stacked-u16 path on the same voxels (CPU only).

Reference behaviour followed: dicom_preprocessor/src/lib.rs:47-202.  The reference's DICOM parsing
lives in the un-vendored dicom-object 0.9 crate and it ships no DICOM fixtures: parity unpinned,
slices are synthesised by tests/dicom_writer.py.
"""
import numpy as np
import pytest

from tests.dicom_writer import EXPLICIT, IMPLICIT, write_slice

import volxel_amd as vx


def _stack(seed=3, dims=(40, 24, 11), hi=4000):
    rng = np.random.default_rng(seed)
    v = rng.integers(0, hi, size=(dims[2], dims[1], dims[0]), dtype=np.uint16)
    v[:, :8, :8] = 0
    return v


def _same(a, b, hist_prefix=None):
    for f in ("indirection", "range", "atlas", "range_mipmaps", "transform", "index_extent",
              "indirection_size", "range_size", "atlas_size"):
        x, y = getattr(a, f), getattr(b, f)
        if f == "range_mipmaps":
            assert len(x) == len(y) == 3
            assert all(np.array_equal(p[0], q[0]) and tuple(p[1]) == tuple(q[1]) for p, q in zip(x, y)), f
        else:
            assert np.array_equal(np.asarray(x), np.asarray(y)), f
    n = hist_prefix or len(a.histogram)
    assert np.array_equal(np.asarray(a.histogram)[:n], np.asarray(b.histogram)[:n])


@pytest.mark.parametrize("syntax", [EXPLICIT, IMPLICIT])
def test_slices_match_u16_stack(native_lib, syntax):
    v = _stack()
    files = [write_slice(v[z], spacing=(0.5, 0.75), thickness=1.25, syntax=syntax) for z in range(v.shape[0])]
    g = vx.read_dicoms_to_grid(files)
    ref = vx.read_u16_stack_to_grid(v, spacing=(0.5, 0.75, 1.25))
    _same(g, ref)
    assert len(g.histogram) == 4096                       # 2^BitsStored, lib.rs:87-89
    t = np.asarray(g.transform, dtype=np.float32).reshape(4, 4)
    assert np.array_equal(np.diag(t), np.float32([0.5, 0.75, 1.25, 1.0]))


def test_multiframe_and_missing_thickness(native_lib):
    v = _stack(seed=5, dims=(16, 16, 6), hi=1000)
    f = write_slice(v, spacing=(0.7, 0.4), thickness=None, bits_stored=10, with_sequence=False)
    g = vx.read_dicoms_to_grid([f])
    ref = vx.read_u16_stack_to_grid(v, spacing=(0.7, 0.4, 0.4))   # min(x, y), lib.rs:124
    _same(g, ref, hist_prefix=1024)
    assert len(g.histogram) == 1024


def test_sixteen_bit_values(native_lib):
    v = _stack(seed=7, dims=(24, 16, 3), hi=65535)
    files = [write_slice(v[z], bits_stored=16, syntax=IMPLICIT) for z in range(3)]
    g = vx.read_dicoms_to_grid(files)
    _same(g, vx.read_u16_stack_to_grid(v, spacing=(0.5, 0.75, 1.25)))
    assert len(g.histogram) == 65536


def test_last_file_transform_wins(native_lib):
    v = _stack(seed=9, dims=(16, 8, 2))
    files = [write_slice(v[0], spacing=(1.0, 1.0), thickness=1.0), write_slice(v[1], spacing=(0.25, 0.5), thickness=3.0)]
    t = np.asarray(vx.read_dicoms_to_grid(files).transform, dtype=np.float32).reshape(4, 4)
    assert np.array_equal(np.diag(t), np.float32([0.25, 0.5, 3.0, 1.0]))   # lib.rs:153-154


@pytest.mark.parametrize("kwargs,msg", [
    (dict(samples=3), "More than one sample per pixel"),
    (dict(bits_allocated=8), "only 16bit samples"),
    (dict(signed=True), "only unsigned samples"),
    (dict(omit=("spacing",)), "did not contain pixel spacing"),
    (dict(spacing="0.5"), "did not contain two values"),
    (dict(spacing="a\\0.5"), "parse x spacing"),
    (dict(omit=("pixels",)), "PixelData"),
    (dict(bits_stored=8), "exceeds 2^BitsStored"),
])
def test_reference_panics_become_errors(native_lib, kwargs, msg):
    v = _stack(dims=(16, 8, 1))
    with pytest.raises(RuntimeError, match=msg.replace("^", r"\^")):
        vx.read_dicoms_to_grid([write_slice(v[0], **kwargs)])


def test_malformed_inputs(native_lib):
    v = _stack(dims=(16, 8, 2))
    good = write_slice(v[0])
    with pytest.raises(RuntimeError, match="No dicom data"):
        vx.read_dicoms_to_grid([])
    with pytest.raises(RuntimeError, match="DICM"):
        vx.read_dicoms_to_grid([b"\0" * 200])
    with pytest.raises(RuntimeError):
        vx.read_dicoms_to_grid([good[:-40]])              # truncated PixelData
    with pytest.raises(RuntimeError, match="rows/columns"):
        vx.read_dicoms_to_grid([good, write_slice(v[1][:, :8])])
    with pytest.raises(RuntimeError, match="transfer syntax"):
        vx.read_dicoms_to_grid([write_slice(v[0], syntax="1.2.840.10008.1.2.4.70")])
    # every truncation point must produce an error or a grid, never a crash
    for cut in range(132, len(good) - 16 * 8 * 2, 7):
        try:
            vx.read_dicoms_to_grid([good[:cut]])
        except RuntimeError:
            pass


@pytest.mark.parametrize("frames_text", ["2147483648", "4294967297", "65536", "99999999999999999999"])
def test_crafted_frame_count_is_an_error_not_an_abort(native_lib, frames_text):
    """NumberOfFrames far beyond the PixelData: rows*columns*frames*2 must not wrap the length check, and no
    C++ exception (length_error / bad_alloc of the stack) may cross the C ABI -- ADVICE round 1."""
    v = _stack(dims=(16, 8, 1))
    with pytest.raises(RuntimeError, match="NumberOfFrames|PixelData"):
        vx.read_dicoms_to_grid([write_slice(v[0], frames_text=frames_text)])


def test_stack_deeper_than_the_brick_limit_is_an_error(native_lib):
    """8128 slices is the deepest stack the 10-bit pointers allow (brick.rs:77-81)"""
    v = np.zeros((1, 8, 8), dtype=np.uint16)
    one = write_slice(np.repeat(v, 4100, axis=0))
    with pytest.raises(RuntimeError, match="8128"):
        vx.read_dicoms_to_grid([one, one])
