// brick_builder.cpp -- native brick-layout producer (C ABI: include/volxel_brick.h).
//
// C++ stand-in for the reference's Rust/wasm preprocessor: BrickGrid::construct
// (dicom_preprocessor/src/brick.rs:76-204) over a stacked u16 volume
// (dicom_preprocessor/src/lib.rs:142-191, dicom.rs:7-21).  The reference builds serially
// in wasm ("load times ... in excess of 2 minutes"); here the work is split differently:
//   pass 1 (parallel over bricks): integer min/max of the +-2 dilated 12^3 window --
//          raw/max is monotone, so the density range is the range of the raw codes;
//   pass 2 (serial, trivial): allocation order = exclusive scan of "non-constant" flags,
//          which reproduces the reference's z,y,x scan order exactly;
//   pass 3 (parallel over bricks): f16 range, 10-10-10 pointer, u8 quantisation of the
//          8^3 voxels into the atlas slot;
//   pass 4 (parallel): three range mips; histogram by per-thread partials.
// Output is byte-identical to the reference algorithm (checked against oracle/ in tests/).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <memory>
#include <new>
#include <stdexcept>
#include <thread>
#include <vector>

#include "../../include/volxel_brick.h"

namespace {

constexpr uint32_t BRICK = 8, BITS = 10, MAXB = 1u << BITS, NMIPS = 3;

thread_local std::string g_err;

// IEEE binary16, round to nearest even (half::f16::from_f32 / to_f32)
uint16_t f32_to_f16(float f) {
  uint32_t x;
  std::memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u, exp = (x >> 23) & 0xffu, man = x & 0x7fffffu;
  if (exp == 0xff) return (uint16_t)(sign | 0x7c00u | (man ? (0x200u | (man >> 13)) : 0u));
  int e = (int)exp - 112;
  if (e >= 31) return (uint16_t)(sign | 0x7c00u);
  if (e <= 0) {
    if (e < -10) return (uint16_t)sign;
    man |= 0x800000u;
    uint32_t sh = (uint32_t)(14 - e), h = man >> sh, rem = man & ((1u << sh) - 1u), half = 1u << (sh - 1);
    if (rem > half || (rem == half && (h & 1u))) ++h;
    return (uint16_t)(sign | h);
  }
  uint32_t h = ((uint32_t)e << 10) | (man >> 13), rem = man & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;
  return (uint16_t)(sign | h);
}
float f16_to_f32(uint16_t h) {
  uint32_t sign = ((uint32_t)h & 0x8000u) << 16, exp = (h >> 10) & 0x1fu, man = h & 0x3ffu, x;
  if (exp == 0) {
    if (!man) x = sign;
    else {
      int e = -1;
      do { man <<= 1; ++e; } while (!(man & 0x400u));
      x = sign | ((uint32_t)(112 - e) << 23) | ((man & 0x3ffu) << 13);
    }
  } else if (exp == 31) x = sign | 0x7f800000u | (man << 13);
  else x = sign | ((exp + 112) << 23) | (man << 13);
  float f;
  std::memcpy(&f, &x, 4);
  return f;
}
inline uint32_t encode_range(float mn, float mx) {  // brick.rs:19-23
  return ((uint32_t)f32_to_f16(mn) << 16) | f32_to_f16(mx);
}
inline uint8_t encode_voxel(float v, float rx, float ry) {  // brick.rs:45-48
  float n = (v - rx) / (ry - rx);
  if (n < 0.0f) n = 0.0f;
  if (n > 1.0f) n = 1.0f;
  float r = std::round(255.0f * n);
  if (!(r == r) || r <= 0.0f) return 0;
  return r >= 255.0f ? 255 : (uint8_t)r;
}

template <class F>
void parallel_for(size_t n, int threads, F&& f) {
  if (threads <= 1 || n < 2) {
    for (size_t i = 0; i < n; ++i) f(i, 0);
    return;
  }
  std::atomic<size_t> next{0};
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t)
    pool.emplace_back([&, t] {
      for (;;) {
        size_t i = next.fetch_add(1);
        if (i >= n) break;
        f(i, t);
      }
    });
  for (auto& th : pool) th.join();
}

}  // namespace

struct VxBrickGrid {
  uint32_t bc[3]{}, atlas_size[3]{}, extent[3]{};
  std::vector<uint32_t> indirection, range;
  std::vector<uint8_t> atlas;
  std::vector<uint32_t> mips[NMIPS];
  uint32_t mip_size[NMIPS][3]{};
  float transform[16]{};
  std::vector<uint32_t> histogram;
  std::vector<int32_t> histogram_gradient;
  uint32_t grad_min = 0, grad_max = 0;
  uint32_t brick_counter = 0;
  uint16_t max_value = 0;
};

extern "C" {

const char* vxb_last_error(void) { return g_err.c_str(); }

}  // extern "C"

static int build_internal_impl(const uint16_t* vox, const uint32_t dims[3], const float spacing[3], uint16_t max_value,
                               uint32_t hist_bins, int n_threads, VxBrickGrid** out);

// shared by vxb_build_from_u16 and the DICOM reader; hist_bins == 0: 4096 / 65536 by max value.
// Exceptions (allocation failure, the brick-count limit) stop here: nothing crosses the C ABI.
int vxb_build_internal(const uint16_t* vox, const uint32_t dims[3], const float spacing[3], uint16_t max_value,
                       uint32_t hist_bins, int n_threads, VxBrickGrid** out) {
  if (out) *out = nullptr;
  try {
    return build_internal_impl(vox, dims, spacing, max_value, hist_bins, n_threads, out);
  } catch (const std::bad_alloc&) {
    g_err = "out of memory while building the brick grid";
  } catch (const std::exception& e) {
    g_err = e.what();
  } catch (...) {
    g_err = "unexpected failure while building the brick grid";
  }
  if (out) *out = nullptr;
  return VXB_ERR_INVALID;
}

static int build_internal_impl(const uint16_t* vox, const uint32_t dims[3], const float spacing[3], uint16_t max_value,
                               uint32_t hist_bins, int n_threads, VxBrickGrid** out) {
  if (!vox || !dims || !spacing || !out || !dims[0] || !dims[1] || !dims[2]) {
    g_err = "vxb_build_from_u16: null or empty input";
    return VXB_ERR_INVALID;
  }
  *out = nullptr;
  if (n_threads <= 0) n_threads = (int)std::max(1u, std::thread::hardware_concurrency());
  n_threads = std::min(n_threads, 64);
  const size_t nvox = (size_t)dims[0] * dims[1] * dims[2];
  std::unique_ptr<VxBrickGrid> owner(new VxBrickGrid());   // released to *out at the end; freed on any throw
  VxBrickGrid* g = owner.get();
  // brick.rs:77 (the f32 division + ceil of div_round_up is exact for these magnitudes)
  for (int i = 0; i < 3; ++i) {
    uint32_t b = (uint32_t)std::ceil((float)dims[i] / (float)BRICK);
    g->bc[i] = (uint32_t)std::ceil((float)b / 8.0f) * 8u;
    g->extent[i] = g->bc[i] * BRICK;
  }
  if (g->bc[0] >= MAXB || g->bc[1] >= MAXB || g->bc[2] >= MAXB) {  // brick.rs:79-81
    owner.reset();
    g_err = "Exceeded max brick count";
    return VXB_ERR_TOO_MANY_BRICKS;
  }
  // histogram + max (lib.rs:87-102): one pass, per-thread partial histograms over z slabs
  const size_t slab = (size_t)dims[0] * dims[1];
  std::vector<std::vector<uint32_t>> part(n_threads, std::vector<uint32_t>(65536, 0));
  parallel_for(dims[2], n_threads, [&](size_t z, int t) {
    const uint16_t* p = vox + z * slab;
    uint32_t* h = part[t].data();
    for (size_t i = 0; i < slab; ++i) ++h[p[i]];
  });
  std::vector<uint32_t> hist(65536, 0);
  for (auto& ph : part)
    for (int i = 0; i < 65536; ++i) hist[i] += ph[i];
  uint32_t data_max = 0;
  for (int i = 65535; i >= 0; --i)
    if (hist[i]) { data_max = (uint32_t)i; break; }
  if (max_value == 0) max_value = (uint16_t)data_max;
  if (max_value == 0) {
    owner.reset();
    g_err = "vxb_build_from_u16: volume is all zero (raw/max is undefined, dicom.rs:16)";
    return VXB_ERR_INVALID;
  }
  g->max_value = max_value;
  const uint32_t bins = hist_bins ? hist_bins : (max_value < 4096 ? 4096u : 65536u);
  // explicit hist_bins (DICOM reader): the per-file range check already ran there, and counts of a
  // later file beyond the first file's bins are dropped by the summation at lib.rs:159-161
  g->histogram.assign(hist.begin(), hist.begin() + bins);
  {  // dicom.rs:39-66
    std::vector<int32_t> grad(bins);
    uint32_t last = 0, mn = UINT32_MAX, mx = 0;
    for (uint32_t i = 0; i < bins; ++i) {
      int32_t s = (int32_t)g->histogram[i] - (int32_t)last;
      uint32_t a = s < 0 ? (uint32_t)(-(int64_t)s) : (uint32_t)s;
      mx = std::max(mx, a);
      mn = std::min(mn, a);
      grad[i] = s;
      last = g->histogram[i];
    }
    g->histogram_gradient.resize(bins);
    g->histogram_gradient[0] = grad[0];
    for (uint32_t i = 1; i + 1 < bins; ++i) g->histogram_gradient[i] = (grad[i - 1] + grad[i] + grad[i + 1]) / 3;
    g->histogram_gradient[bins - 1] = grad[bins - 1];
    g->grad_min = mn;
    g->grad_max = mx;
  }
  const float fmax = (float)max_value;
  const uint32_t BX = g->bc[0], BY = g->bc[1], BZ = g->bc[2];
  const size_t nb = (size_t)BX * BY * BZ;
  g->indirection.assign(nb, 0);
  g->range.assign(nb, 0);
  std::vector<uint16_t> rmin(nb), rmax(nb);

  // pass 1: raw-code range of the dilated window [-2, 10) (brick.rs:99-112); coordinates
  // outside the data (including the wrapped negatives) read as 0 (dicom.rs:8-10)
  parallel_for((size_t)BZ * BY, n_threads, [&](size_t row, int) {
    uint32_t bz = (uint32_t)(row / BY), by = (uint32_t)(row % BY);
    for (uint32_t bx = 0; bx < BX; ++bx) {
      int lo[3] = {(int)(bx * 8) - 2, (int)(by * 8) - 2, (int)(bz * 8) - 2};
      int hi[3] = {lo[0] + 12, lo[1] + 12, lo[2] + 12};
      bool outside = false;
      int c0[3], c1[3];
      for (int a = 0; a < 3; ++a) {
        c0[a] = std::max(lo[a], 0);
        c1[a] = std::min(hi[a], (int)dims[a]);
        if (lo[a] < 0 || hi[a] > (int)dims[a]) outside = true;
      }
      uint16_t mn = 65535, mx = 0;
      bool any = c0[0] < c1[0] && c0[1] < c1[1] && c0[2] < c1[2];
      if (any) {
        for (int z = c0[2]; z < c1[2]; ++z)
          for (int y = c0[1]; y < c1[1]; ++y) {
            const uint16_t* p = vox + ((size_t)z * dims[1] + y) * dims[0];
            for (int x = c0[0]; x < c1[0]; ++x) {
              uint16_t r = p[x];
              mn = std::min(mn, r);
              mx = std::max(mx, r);
            }
          }
      }
      if (outside || !any) {
        mn = any ? std::min<uint16_t>(mn, 0) : 0;
        mx = any ? mx : 0;
      }
      size_t bi = ((size_t)bz * BY + by) * BX + bx;
      rmin[bi] = mn;
      rmax[bi] = mx;
    }
  });
  // pass 2: allocation order (brick.rs:120-129)
  std::vector<uint32_t> slot(nb);
  uint32_t counter = 0;
  for (size_t bi = 0; bi < nb; ++bi) slot[bi] = (rmin[bi] != rmax[bi]) ? counter++ : UINT32_MAX;
  g->brick_counter = counter;
  // brick.rs:85,151: atlas dims, z pruned to the used slices
  g->atlas_size[0] = BX * 8;
  g->atlas_size[1] = BY * 8;
  g->atlas_size[2] = (uint32_t)(8.0f * std::round(std::ceil((float)counter / (float)(BX * BY))));
  g->atlas.assign((size_t)g->atlas_size[0] * g->atlas_size[1] * g->atlas_size[2], 0);
  const size_t asx = g->atlas_size[0], asy = g->atlas_size[1];
  std::atomic<bool> ptr_overflow{false};
  // pass 3
  parallel_for((size_t)BZ * BY, n_threads, [&](size_t row, int) {
    uint32_t bz = (uint32_t)(row / BY), by = (uint32_t)(row % BY);
    for (uint32_t bx = 0; bx < BX; ++bx) {
      size_t bi = ((size_t)bz * BY + by) * BX + bx;
      float fmn = (float)rmin[bi] / fmax, fmx = (float)rmax[bi] / fmax;
      uint32_t packed = encode_range(fmn, fmx);
      g->range[bi] = packed;
      uint32_t s = slot[bi];
      if (s == UINT32_MAX) continue;
      uint32_t px = s % BX, py = (s / BX) % BY, pz = s / (BX * BY);  // buf3d.rs:29-32
      if (px >= MAXB || py >= MAXB || pz >= MAXB) { ptr_overflow = true; continue; }
      g->indirection[bi] = px | (py << BITS) | (pz << (2 * BITS));
      float rx = f16_to_f32((uint16_t)(packed >> 16)), ry = f16_to_f32((uint16_t)packed);  // :136
      for (uint32_t lz = 0; lz < 8; ++lz) {
        uint32_t z = bz * 8 + lz;
        for (uint32_t ly = 0; ly < 8; ++ly) {
          uint32_t y = by * 8 + ly;
          uint8_t* dst = g->atlas.data() + ((size_t)(pz * 8 + lz) * asy + (py * 8 + ly)) * asx + px * 8;
          const bool row_in = z < dims[2] && y < dims[1];
          const uint16_t* src = row_in ? vox + ((size_t)z * dims[1] + y) * dims[0] : nullptr;
          for (uint32_t lx = 0; lx < 8; ++lx) {
            uint32_t x = bx * 8 + lx;
            float v = (row_in && x < dims[0]) ? (float)src[x] / fmax : 0.0f;
            dst[lx] = encode_voxel(v, rx, ry);
          }
        }
      }
    }
  });
  if (ptr_overflow) {
    owner.reset();
    g_err = "atlas pointer exceeds 10 bits (brick.rs:31)";
    return VXB_ERR_TOO_MANY_BRICKS;
  }
  // pass 4: range mips (brick.rs:153-190)
  for (uint32_t level = 0; level < NMIPS; ++level) {
    uint32_t ms[3] = {BX >> (level + 1), BY >> (level + 1), BZ >> (level + 1)};
    uint32_t ss[2] = {BX >> level, BY >> level};
    const std::vector<uint32_t>& src = level == 0 ? g->range : g->mips[level - 1];
    std::vector<uint32_t>& dst = g->mips[level];
    dst.assign((size_t)ms[0] * ms[1] * ms[2], 0);
    for (int i = 0; i < 3; ++i) g->mip_size[level][i] = ms[i];
    parallel_for(ms[2], n_threads, [&](size_t z, int) {
      for (uint32_t y = 0; y < ms[1]; ++y)
        for (uint32_t x = 0; x < ms[0]; ++x) {
          float mn = 3.40282347e+38f, mx = -3.40282347e+38f;
          for (uint32_t d = 0; d < 8; ++d) {
            uint32_t sx = x * 2 + (d & 1), sy = y * 2 + ((d >> 1) & 1), sz = (uint32_t)z * 2 + (d >> 2);
            uint32_t pk = src[((size_t)sz * ss[1] + sy) * ss[0] + sx];
            mn = std::fmin(mn, f16_to_f32((uint16_t)(pk >> 16)));
            mx = std::fmax(mx, f16_to_f32((uint16_t)pk));
          }
          dst[((size_t)z * ms[1] + y) * ms[0] + x] = encode_range(mn, mx);
        }
    });
  }
  // lib.rs:138: Mat4::from_scale(spacing), column major
  std::memset(g->transform, 0, sizeof g->transform);
  g->transform[0] = spacing[0];
  g->transform[5] = spacing[1];
  g->transform[10] = spacing[2];
  g->transform[15] = 1.0f;
  (void)nvox;
  *out = owner.release();
  return VXB_OK;
}

extern "C" {

int vxb_build_from_u16(const uint16_t* vox, const uint32_t dims[3], const float spacing[3],
                       uint16_t max_value, int n_threads, VxBrickGrid** out) {
  return vxb_build_internal(vox, dims, spacing, max_value, 0, n_threads, out);
}

void vxb_set_error(const char* msg) { g_err = msg ? msg : ""; }

void vxb_free(VxBrickGrid* g) { delete g; }

static void put3(const uint32_t s[3], uint32_t out[3]) { out[0] = s[0]; out[1] = s[1]; out[2] = s[2]; }
void vxb_indirection_size(const VxBrickGrid* g, uint32_t out[3]) { put3(g->bc, out); }
void vxb_range_size(const VxBrickGrid* g, uint32_t out[3]) { put3(g->bc, out); }
void vxb_atlas_size(const VxBrickGrid* g, uint32_t out[3]) { put3(g->atlas_size, out); }
const uint32_t* vxb_indirection_data(const VxBrickGrid* g) { return g->indirection.data(); }
const uint16_t* vxb_range_data(const VxBrickGrid* g) { return (const uint16_t*)g->range.data(); }
const uint8_t* vxb_atlas_data(const VxBrickGrid* g) { return g->atlas.data(); }
uint32_t vxb_range_mipmaps(const VxBrickGrid*) { return NMIPS; }
const uint16_t* vxb_range_mipmap(const VxBrickGrid* g, uint32_t i) {
  return i < NMIPS ? (const uint16_t*)g->mips[i].data() : nullptr;
}
void vxb_range_mipmap_stride(const VxBrickGrid* g, uint32_t i, uint32_t out[3]) {
  if (i < NMIPS) put3(g->mip_size[i], out);
}
void vxb_transform(const VxBrickGrid* g, float out16[16]) { std::memcpy(out16, g->transform, 64); }
float vxb_minorant(const VxBrickGrid*) { return 0.0f; }  // dicom.rs:19-21
float vxb_majorant(const VxBrickGrid*) { return 1.0f; }
void vxb_index_extent(const VxBrickGrid* g, uint32_t out[3]) { put3(g->extent, out); }
uint32_t vxb_histogram_len(const VxBrickGrid* g) { return (uint32_t)g->histogram.size(); }
const uint32_t* vxb_histogram(const VxBrickGrid* g) { return g->histogram.data(); }
const int32_t* vxb_histogram_gradient(const VxBrickGrid* g) { return g->histogram_gradient.data(); }
uint32_t vxb_histogram_gradient_min(const VxBrickGrid* g) { return g->grad_min; }
uint32_t vxb_histogram_gradient_max(const VxBrickGrid* g) { return g->grad_max; }
uint32_t vxb_brick_counter(const VxBrickGrid* g) { return g->brick_counter; }

// Grid::lookup, brick.rs:208-230
float vxb_lookup(const VxBrickGrid* g, uint32_t x, uint32_t y, uint32_t z) {
  uint32_t bx = x >> 3, by = y >> 3, bz = z >> 3;
  if (bx >= g->bc[0] || by >= g->bc[1] || bz >= g->bc[2]) return 0.0f;
  size_t bi = ((size_t)bz * g->bc[1] + by) * g->bc[0] + bx;
  uint32_t p = g->indirection[bi], pk = g->range[bi];
  float mn = f16_to_f32((uint16_t)(pk >> 16)), mx = f16_to_f32((uint16_t)pk);
  uint32_t ax = ((p & 1023u) << 3) + (x & 7), ay = (((p >> 10) & 1023u) << 3) + (y & 7),
           az = (((p >> 20) & 1023u) << 3) + (z & 7);
  if (az >= g->atlas_size[2]) return mn;
  uint8_t d = g->atlas[((size_t)az * g->atlas_size[1] + ay) * g->atlas_size[0] + ax];
  return mn + (float)d * (1.0f / 255.0f) * (mx - mn);  // brick.rs:50-52
}

}  // extern "C"
