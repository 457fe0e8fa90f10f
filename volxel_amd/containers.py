"""Host-side container I/O behind the viewer's load methods (viewer.ts:977-1040): ZIP -> DICOM slices with the rules
of zip.rs:36-115, URL -> bytes (worker.ts:115-126), environment bytes -> RGBA floats (hdr.rs:23-36).

None of this is on the hot path (C15 / C16 are container formats).  What ships here is what the
Python standard library can decode: ZIP (stored / deflate) and Radiance RGBE.  OpenEXR -- the other format the
reference's `image` crate decodes -- is refused with a message that names the decoded-floats entry point
(`Volxel3DRenderer.setup_env`), never guessed at."""
from __future__ import annotations

import io
import os
import posixpath
import zipfile
from urllib.parse import urlparse
from urllib.request import urlopen

import numpy as np


class ZipReadError(RuntimeError):
    """zip.rs:11-28: ExtractFailed / MoreThanOneFolder / NoFiles; str() is the reference's `message` getter"""

    def __init__(self, kind: str, detail: str | None = None):
        self.kind = kind
        super().__init__(f"{kind}: {detail if detail is not None else 'No Message Specified'}")


def read_zip_slices(zip_bytes) -> list[bytes]:
    """The files of a ZIP archive in archive order (the reference stacks them in that order, zip.rs:54-97; no sorting),
    with its folder rule: at most one directory entry, and every file that follows it lies directly inside it."""
    try:
        archive = zipfile.ZipFile(io.BytesIO(bytes(zip_bytes)))
    except Exception as e:                       # zip.rs:46-47
        raise ZipReadError("ExtractFailed", str(e)) from None
    infos = archive.infolist()
    if len(infos) < 1:                           # zip.rs:49-51
        raise ZipReadError("NoFiles")
    directory = None
    slices = []
    for info in infos:
        name = info.filename
        norm = posixpath.normpath(name)
        if name.startswith("/") or norm == ".." or norm.startswith("../"):        # enclosed_name(), zip.rs:56 ("..foo" is a legal name)
            raise ZipReadError("ExtractFailed", "No enclosed name was able to be found")
        if info.is_dir():                        # zip.rs:57-63
            if directory is not None:
                raise ZipReadError("MoreThanOneFolder")
            directory = norm
            continue
        if directory is not None and posixpath.dirname(norm) != directory:   # zip.rs:64-70
            raise ZipReadError("MoreThanOneFolder")
        try:
            slices.append(archive.read(info))
        except Exception as e:
            raise ZipReadError("ExtractFailed", str(e)) from None
    if not slices:                               # `result.expect("No dicom data collected")`, zip.rs:104
        raise ZipReadError("NoFiles", "No dicom data collected")
    return slices


def fetch_bytes(url: str) -> bytes:
    """worker.ts:115-126 `fetch(url)` + util.ts exportResponseBytes: a plain path, a file:// URL, or whatever urllib
    can reach (the build and test machines have no network)."""
    parsed = urlparse(url)
    if parsed.scheme in ("", "file") or (len(parsed.scheme) == 1 and os.name == "nt"):
        path = parsed.path if parsed.scheme == "file" else url
        with open(path, "rb") as fh:
            return fh.read()
    with urlopen(url) as resp:                   # noqa: S310 (the caller names the URL)
        status = getattr(resp, "status", 200)
        if status >= 400:
            raise RuntimeError("Environment fetch responded with error response")   # viewer.ts:1037
        return resp.read()


def _rgbe_to_float(rgbe: np.ndarray) -> np.ndarray:
    e = rgbe[..., 3].astype(np.int32)
    scale = np.where(e > 0, np.ldexp(np.float32(1.0), e - (128 + 8)), np.float32(0.0)).astype(np.float32)
    out = np.empty(rgbe.shape[:-1] + (4,), dtype=np.float32)
    out[..., :3] = rgbe[..., :3].astype(np.float32) * scale[..., None]
    out[..., 3] = 1.0
    return out


def decode_environment(data) -> tuple[np.ndarray, int, int]:
    """hdr.rs:23-36 `ExrImage::decode_from_bytes` -> (RGBA32F floats with row 0 = top, width, height).
    Radiance RGBE (.hdr) is decoded; OpenEXR is refused (container decode outside the path)."""
    b = bytes(data)
    if b[:4] == b"\x76\x2f\x31\x01":
        raise ValueError("OpenEXR decode is outside this build's scope (hdr.rs uses the `image` crate's exr decoder): "
                         "decode the map elsewhere and pass {width, height, floats} to setup_env()")
    if not (b.startswith(b"#?RADIANCE") or b.startswith(b"#?RGBE")):
        raise ValueError("unrecognised environment map format (expected Radiance RGBE; decoded floats go to setup_env())")
    pos = b.index(b"\n\n") + 2
    header = b[:pos].decode("latin-1")
    if "FORMAT=32-bit_rle_rgbe" not in header:
        raise ValueError("Radiance map: only FORMAT=32-bit_rle_rgbe is supported")
    eol = b.index(b"\n", pos)
    dims = b[pos:eol].decode("latin-1").split()
    if len(dims) != 4 or dims[0] != "-Y" or dims[2] != "+X":
        raise ValueError("Radiance map: only the standard -Y h +X w orientation is supported")
    h, w = int(dims[1]), int(dims[3])
    p = eol + 1
    img = np.zeros((h, w, 4), dtype=np.uint8)
    for y in range(h):
        if 8 <= w < 32768 and b[p] == 2 and b[p + 1] == 2 and ((b[p + 2] << 8) | b[p + 3]) == w:   # new RLE scanline
            p += 4
            for ch in range(4):
                x = 0
                while x < w:
                    if p + 1 >= len(b):
                        raise ValueError("Radiance map: scanline data ends inside a run")
                    n = b[p]; p += 1
                    run = n > 128
                    n = n - 128 if run else n
                    # a zero-length run never advances (the hang of ADVICE round 3); a run past the scanline or the
                    # buffer is a corrupt file -- the reference's decoder (image crate) errors out in these cases too
                    if n == 0 or x + n > w or (not run and p + n > len(b)):
                        raise ValueError("Radiance map: corrupt run-length data")
                    if run:
                        img[y, x:x + n, ch] = b[p]; p += 1
                    else:
                        img[y, x:x + n, ch] = np.frombuffer(b, dtype=np.uint8, count=n, offset=p); p += n
                    x += n
        else:                                                                                     # flat pixels
            if p + 4 * w > len(b):
                raise ValueError("Radiance map: pixel data ends early")
            img[y] = np.frombuffer(b, dtype=np.uint8, count=4 * w, offset=p).reshape(w, 4); p += 4 * w
    return _rgbe_to_float(img).reshape(-1), w, h
