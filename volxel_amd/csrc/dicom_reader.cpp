// dicom_reader.cpp -- minimal native DICOM slice reader.
//
// Stands in for read_dicom / read_dicoms_internal / read_dicoms_to_grid of the reference's Rust
// preprocessor (dicom_preprocessor/src/lib.rs:47-202), whose parsing lives in the dicom-object /
// dicom-pixeldata 0.9.0 crates (not vendored: parity unpinned; restated from DICOM PS3.5/PS3.10).
// Supported: Part-10 files ("DICM" after the 128-byte preamble), transfer syntaxes Implicit VR
// Little Endian (1.2.840.10008.1.2) and Explicit VR Little Endian (1.2.840.10008.1.2.1), native
// (uncompressed) PixelData.  Enforced exactly like the reference's panics: one sample per pixel,
// 16 bits allocated, unsigned (lib.rs:77-85); PixelSpacing required, SliceThickness defaults to
// min(spacing x, y) (lib.rs:105-124); slices are stacked in the order given, the transform of the
// last file wins, histograms (2^BitsStored bins) are summed (lib.rs:150-176).
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/volxel_brick.h"

extern "C" void vxb_set_error(const char* msg);
int vxb_build_internal(const uint16_t* vox, const uint32_t dims[3], const float spacing[3], uint16_t max_value,
                       uint32_t hist_bins, int n_threads, VxBrickGrid** out);

namespace {

struct Slice {
  uint32_t cols = 0, rows = 0, frames = 1;
  uint32_t bits_alloc = 0, bits_stored = 0, pixel_repr = 0, samples = 1;
  bool have_spacing = false, have_thickness = false;
  float sx = 1, sy = 1, thickness = 1;
  const uint8_t* pixels = nullptr;
  size_t pixel_bytes = 0;
  bool is_dicomdir = false;
};

struct Cursor {
  const uint8_t* p;
  size_t n, pos = 0;
  bool ok(size_t k) const { return pos + k <= n; }
  uint16_t u16() { uint16_t v = (uint16_t)(p[pos] | (p[pos + 1] << 8)); pos += 2; return v; }
  uint32_t u32() { uint32_t v = p[pos] | (p[pos + 1] << 8) | (p[pos + 2] << 16) | ((uint32_t)p[pos + 3] << 24); pos += 4; return v; }
};

bool long_vr(const uint8_t* vr) {
  static const char* L[] = {"OB", "OW", "OF", "SQ", "UT", "UN", "OD", "OL", "UC", "UR", "OV", "SV", "UV"};
  for (auto s : L)
    if (vr[0] == (uint8_t)s[0] && vr[1] == (uint8_t)s[1]) return true;
  return false;
}

// skip an undefined-length sequence / item: scan items until the matching delimiter
bool skip_undefined(Cursor& c, bool explicit_vr, int depth);

bool skip_items(Cursor& c, bool explicit_vr, int depth) {  // inside a sequence of undefined length
  while (c.ok(8)) {
    uint16_t g = c.u16(), e = c.u16();
    uint32_t len = c.u32();
    if (g == 0xFFFE && e == 0xE0DD) return true;             // sequence delimiter
    if (g == 0xFFFE && e == 0xE000) {                        // item
      if (len != 0xFFFFFFFFu) {
        if (!c.ok(len)) return false;
        c.pos += len;
      } else if (!skip_undefined(c, explicit_vr, depth + 1)) return false;
    } else return false;
  }
  return false;
}

bool skip_undefined(Cursor& c, bool explicit_vr, int depth) {  // inside an item of undefined length
  if (depth > 16) return false;
  while (c.ok(8)) {
    uint16_t g = c.u16(), e = c.u16();
    if (g == 0xFFFE && e == 0xE00D) { c.pos += 4; return true; }  // item delimiter
    uint32_t len;
    bool is_sq = false;
    if (explicit_vr && g != 0xFFFE) {
      const uint8_t* vr = c.p + c.pos;
      c.pos += 2;
      is_sq = vr[0] == 'S' && vr[1] == 'Q';
      if (long_vr(vr)) { c.pos += 2; if (!c.ok(4)) return false; len = c.u32(); }
      else { if (!c.ok(2)) return false; len = c.u16(); }
    } else len = c.u32();
    if (len == 0xFFFFFFFFu) {
      (void)is_sq;
      if (!skip_items(c, explicit_vr, depth + 1)) return false;
    } else {
      if (!c.ok(len)) return false;
      c.pos += len;
    }
  }
  return false;
}

std::string str_value(const uint8_t* p, uint32_t len) {
  std::string s((const char*)p, len);
  while (!s.empty() && (s.back() == ' ' || s.back() == '\0')) s.pop_back();
  size_t b = 0;
  while (b < s.size() && s[b] == ' ') ++b;
  return s.substr(b);
}

const char* parse_file(const uint8_t* data, size_t size, Slice& out) {
  if (size < 132 || memcmp(data + 128, "DICM", 4) != 0) return "not a DICOM Part-10 file (no DICM prefix)";
  Cursor c{data, size, 132};
  bool explicit_vr = true;       // group 0002 is always explicit VR little endian
  bool dataset_explicit = true;
  bool in_meta = true;
  while (c.ok(8)) {
    size_t start = c.pos;
    uint16_t g = c.u16(), e = c.u16();
    if (in_meta && g != 0x0002) {  // first element of the data set
      in_meta = false;
      explicit_vr = dataset_explicit;
    }
    uint32_t len;
    bool is_sq = false;
    if (explicit_vr) {
      const uint8_t* vr = c.p + c.pos;
      c.pos += 2;
      is_sq = vr[0] == 'S' && vr[1] == 'Q';
      if (long_vr(vr)) { c.pos += 2; if (!c.ok(4)) return "truncated element"; len = c.u32(); }
      else { if (!c.ok(2)) return "truncated element"; len = c.u16(); }
    } else len = c.u32();
    (void)start;
    if (len == 0xFFFFFFFFu) {
      if (g == 0x7FE0 && e == 0x0010) return "encapsulated (compressed) PixelData is not supported";
      if (g == 0x0004 && e == 0x1220) out.is_dicomdir = true;   // lib.rs:49-72
      if (!skip_items(c, explicit_vr, 0)) return "malformed sequence of undefined length";
      continue;
    }
    if (!c.ok(len)) return "element length exceeds the file";
    const uint8_t* v = c.p + c.pos;
    auto us = [&]() -> uint32_t { return len >= 2 ? (uint32_t)(v[0] | (v[1] << 8)) : 0u; };
    if (g == 0x0002 && e == 0x0010) {
      std::string ts = str_value(v, len);
      if (ts == "1.2.840.10008.1.2") dataset_explicit = false;
      else if (ts == "1.2.840.10008.1.2.1") dataset_explicit = true;
      else return "unsupported transfer syntax (only uncompressed little endian)";
    } else if (g == 0x0004 && e == 0x1220) out.is_dicomdir = true;
    else if (g == 0x0028 && e == 0x0002) out.samples = us();
    else if (g == 0x0028 && e == 0x0008) {
      // NumberOfFrames is an IS string: anything outside 1..65535 is rejected by the caller (a crafted count
      // must not wrap the PixelData length check or ask for an absurd allocation)
      long nf = atol(str_value(v, len).c_str());
      out.frames = nf < 1 ? 1u : (nf > 65536 ? 65536u : (uint32_t)nf);
    }
    else if (g == 0x0028 && e == 0x0010) out.rows = us();
    else if (g == 0x0028 && e == 0x0011) out.cols = us();
    else if (g == 0x0028 && e == 0x0100) out.bits_alloc = us();
    else if (g == 0x0028 && e == 0x0101) out.bits_stored = us();
    else if (g == 0x0028 && e == 0x0103) out.pixel_repr = us();
    else if (g == 0x0028 && e == 0x0030) {  // PixelSpacing "a\b": first value -> x (lib.rs:108-115)
      std::string s = str_value(v, len);
      size_t bs = s.find('\\');
      if (bs == std::string::npos || s.find('\\', bs + 1) != std::string::npos)
        return "Pixel spacing did not contain two values x and y";
      char* end = nullptr;
      std::string a = s.substr(0, bs), b = s.substr(bs + 1);
      out.sx = strtof(a.c_str(), &end);
      if (end == a.c_str()) return "Couldn't parse x spacing to float";
      out.sy = strtof(b.c_str(), &end);
      if (end == b.c_str()) return "Couldn't parse y spacing to float";
      out.have_spacing = true;
    } else if (g == 0x0018 && e == 0x0050) {
      std::string s = str_value(v, len);
      size_t bs = s.find('\\');
      if (bs != std::string::npos) s = s.substr(0, bs);
      char* end = nullptr;
      out.thickness = strtof(s.c_str(), &end);
      if (end == s.c_str()) return "Couldn't parse slice thickness to float";
      out.have_thickness = true;
    } else if (g == 0x7FE0 && e == 0x0010) {
      out.pixels = v;
      out.pixel_bytes = len;
    }
    c.pos += len;
    (void)is_sq;
  }
  return nullptr;
}

}  // namespace

static int read_dicoms_to_grid_impl(const uint8_t* const* files, const uint64_t* sizes, uint32_t n_files,
                                    int n_threads, VxBrickGrid** out);

// No C++ exception may cross the C ABI (a Node or Python host would be std::terminate'd): allocation failures of a
// legitimately huge -- or crafted -- stack come back as an error code like every other failure.
extern "C" int vxb_read_dicoms_to_grid(const uint8_t* const* files, const uint64_t* sizes, uint32_t n_files,
                                       int n_threads, VxBrickGrid** out) {
  if (!out) return VXB_ERR_INVALID;
  *out = nullptr;
  try {
    return read_dicoms_to_grid_impl(files, sizes, n_files, n_threads, out);
  } catch (const std::bad_alloc&) {
    vxb_set_error("out of memory while stacking the slices");
  } catch (const std::exception& e) {
    vxb_set_error(e.what());
  } catch (...) {
    vxb_set_error("unexpected failure while reading the slices");
  }
  if (*out) { vxb_free(*out); *out = nullptr; }
  return VXB_ERR_INVALID;
}

static int read_dicoms_to_grid_impl(const uint8_t* const* files, const uint64_t* sizes, uint32_t n_files,
                                    int n_threads, VxBrickGrid** out) {
  if (!files || !sizes || n_files == 0) {
    vxb_set_error("No dicom data collected");  // lib.rs:181
    return VXB_ERR_INVALID;
  }
  std::vector<uint16_t> stack;
  uint32_t cols = 0, rows = 0, depth = 0, bits_stored = 0;
  float spacing[3] = {1, 1, 1};
  for (uint32_t i = 0; i < n_files; ++i) {
    Slice s;
    const char* err = files[i] ? parse_file(files[i], (size_t)sizes[i], s) : "null file buffer";
    if (err) { vxb_set_error(err); return VXB_ERR_INVALID; }
    if (s.is_dicomdir) {  // lib.rs:65-71 returns an empty Buf3D, which fails the stride assert_eq of buf3d.rs:35-36
      vxb_set_error("DICOMDIR file among the slices (empty Buf3D, buf3d.rs:35)");
      return VXB_ERR_INVALID;
    }
    if (!s.pixels) { vxb_set_error("file has no PixelData"); return VXB_ERR_INVALID; }
    if (s.samples != 1) { vxb_set_error("More than one sample per pixel not currently supported"); return VXB_ERR_INVALID; }
    if (s.bits_alloc != 16) { vxb_set_error("Currently only 16bit samples are supported"); return VXB_ERR_INVALID; }
    if (s.pixel_repr != 0) { vxb_set_error("Currently only unsigned samples are supported"); return VXB_ERR_INVALID; }
    if (!s.have_spacing) { vxb_set_error("Image did not contain pixel spacing information"); return VXB_ERR_INVALID; }
    if (s.bits_stored == 0 || s.bits_stored > 16) { vxb_set_error("bad BitsStored"); return VXB_ERR_INVALID; }
    if (s.frames > 65535u) { vxb_set_error("NumberOfFrames out of range (1..65535)"); return VXB_ERR_INVALID; }
    if (s.cols > 65535u || s.rows > 65535u) { vxb_set_error("Rows / Columns out of range"); return VXB_ERR_INVALID; }
    // cols, rows, frames <= 65535 each: the product is < 2^48 and * 2 cannot wrap a 64-bit size_t
    const uint64_t want = (uint64_t)s.cols * s.rows * s.frames * 2u;
    if ((uint64_t)depth + s.frames > 8128u) {   // brick.rs:77-81: at most 1016 bricks = 8128 voxels per axis
      vxb_set_error("stack deeper than 8128 slices (brick.rs:77-81 brick-count limit)");
      return VXB_ERR_INVALID;
    }
    if (s.cols == 0 || s.rows == 0 || (uint64_t)s.pixel_bytes < want) { vxb_set_error("PixelData shorter than rows*columns*frames"); return VXB_ERR_INVALID; }
    if (depth == 0) { cols = s.cols; rows = s.rows; bits_stored = s.bits_stored; }
    else if (s.cols != cols || s.rows != rows) {                     // buf3d.rs:35-36 assert_eq
      vxb_set_error("slices differ in rows/columns");
      return VXB_ERR_INVALID;
    } else if (s.bits_stored < bits_stored) {                        // lib.rs:160 dicom.histogram[i] out of bounds
      vxb_set_error("a later slice has fewer BitsStored than the first (histogram index out of bounds, lib.rs:160)");
      return VXB_ERR_INVALID;
    }
    size_t n = (size_t)cols * rows * s.frames, base = stack.size();
    stack.resize(base + n);
    memcpy(stack.data() + base, s.pixels, n * 2);                    // little endian host, LE transfer syntax
    if (s.bits_stored < 16) {                                        // lib.rs:95 histogram[*short] bounds check
      const uint16_t lim = (uint16_t)(1u << s.bits_stored);
      for (size_t k = base; k < base + n; ++k)
        if (stack[k] >= lim) {
          vxb_set_error("pixel value exceeds 2^BitsStored (histogram index out of range, lib.rs:95)");
          return VXB_ERR_INVALID;
        }
    }
    depth += s.frames;
    spacing[0] = s.sx;                                               // lib.rs:154: last file wins
    spacing[1] = s.sy;
    spacing[2] = s.have_thickness ? s.thickness : std::min(s.sx, s.sy);
  }
  if (depth == 0) { vxb_set_error("No dicom data collected"); return VXB_ERR_INVALID; }
  uint32_t dims[3] = {cols, rows, depth};
  return vxb_build_internal(stack.data(), dims, spacing, 0, 1u << bits_stored, n_threads, out);
}
