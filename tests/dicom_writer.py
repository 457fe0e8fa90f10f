"""Tiny DICOM Part-10 writer for the N1 tests (DICOM PS3.5 / PS3.10 encoding rules).

Test data generator only: the reference holds no DICOM fixtures, so slices are
synthesised here.  Writes uncompressed 16-bit monochrome slices in either little-endian syntax.
"""
import struct

import numpy as np

IMPLICIT = "1.2.840.10008.1.2"
EXPLICIT = "1.2.840.10008.1.2.1"
_LONG = {b"OB", b"OW", b"OF", b"SQ", b"UT", b"UN"}


def _pad(v: bytes, pad=b" "):
    return v + pad if len(v) & 1 else v


def _el(tag, vr: bytes, value: bytes, explicit: bool, undefined=False):
    g, e = tag
    ln = 0xFFFFFFFF if undefined else len(value)
    if not explicit:
        return struct.pack("<HHI", g, e, ln) + value
    if vr in _LONG:
        return struct.pack("<HH2sHI", g, e, vr, 0, ln) + value
    return struct.pack("<HH2sH", g, e, vr, ln) + value


def write_slice(pixels: np.ndarray, spacing=(0.5, 0.75), thickness=1.25, bits_stored=12,
                syntax=EXPLICIT, bits_allocated=16, samples=1, signed=False, frames=None,
                with_sequence=True, omit=(), frames_text=None) -> bytes:
    """pixels: [rows, cols] (or [frames, rows, cols]) uint16.  frames_text: write this NumberOfFrames
    string instead of the true count (crafted-file tests)."""
    px = np.ascontiguousarray(pixels, dtype="<u2")
    if px.ndim == 2:
        px = px[None]
    nf, rows, cols = px.shape
    explicit = syntax == EXPLICIT
    meta_body = (_el((2, 1), b"OB", b"\x00\x01", True)
                 + _el((2, 2), b"UI", _pad(b"1.2.840.10008.5.1.4.1.1.2", b"\0"), True)
                 + _el((2, 0x10), b"UI", _pad(syntax.encode(), b"\0"), True))
    meta = _el((2, 0), b"UL", struct.pack("<I", len(meta_body)), True) + meta_body
    ds = b""
    ds += _el((8, 0x60), b"CS", _pad(b"CT"), explicit)
    if with_sequence:
        # a sequence of undefined length with one undefined-length item holding a nested element and
        # one defined-length item: exercises the skip logic
        inner = _el((8, 0x100), b"SH", _pad(b"CODE"), explicit)
        item1 = struct.pack("<HHI", 0xFFFE, 0xE000, 0xFFFFFFFF) + inner + struct.pack("<HHI", 0xFFFE, 0xE00D, 0)
        item2 = struct.pack("<HHI", 0xFFFE, 0xE000, len(inner)) + inner
        seq = item1 + item2 + struct.pack("<HHI", 0xFFFE, 0xE0DD, 0)
        ds += _el((8, 0x1032), b"SQ", seq, explicit, undefined=True)
    if "thickness" not in omit and thickness is not None:
        ds += _el((0x18, 0x50), b"DS", _pad(repr(float(thickness)).encode()), explicit)
    ds += _el((0x28, 2), b"US", struct.pack("<H", samples), explicit)
    ds += _el((0x28, 4), b"CS", _pad(b"MONOCHROME2"), explicit)
    if frames or nf > 1 or frames_text is not None:
        ds += _el((0x28, 8), b"IS", _pad((frames_text if frames_text is not None else str(nf)).encode()), explicit)
    ds += _el((0x28, 0x10), b"US", struct.pack("<H", rows), explicit)
    ds += _el((0x28, 0x11), b"US", struct.pack("<H", cols), explicit)
    if "spacing" not in omit:
        sp = spacing if isinstance(spacing, (bytes, str)) else "\\".join(repr(float(s)) for s in spacing)
        sp = sp.encode() if isinstance(sp, str) else sp
        ds += _el((0x28, 0x30), b"DS", _pad(sp), explicit)
    ds += _el((0x28, 0x100), b"US", struct.pack("<H", bits_allocated), explicit)
    ds += _el((0x28, 0x101), b"US", struct.pack("<H", bits_stored), explicit)
    ds += _el((0x28, 0x102), b"US", struct.pack("<H", bits_stored - 1), explicit)
    ds += _el((0x28, 0x103), b"US", struct.pack("<H", 1 if signed else 0), explicit)
    if "pixels" not in omit:
        ds += _el((0x7FE0, 0x10), b"OW", px.tobytes(), explicit)
    return b"\0" * 128 + b"DICM" + meta + ds
