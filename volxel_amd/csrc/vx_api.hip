// vx_api.hip -- host side of libvolxel_hip.so: the C ABI of include/volxel_hip.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/volxel_brick.h"
#include "vx_dvr.hpp"
#include "vx_dvr_lds.hpp"
#include "vx_kernels.hpp"
#include "vx_paths.hpp"
#include "vx_events.hpp"

using namespace vx;

namespace {

thread_local std::string g_create_error;

struct EventPair {
  hipEvent_t a, b;
  uint32_t launches = 1;   // frames covered by this interval (pipelined batches cover several)
  bool merge = false;      // interval of a merge_results launch (reported apart, VxCounters.merge_ms)
};

}  // namespace

struct VxContext {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::string err;
  hipDeviceProp_t prop;

  // volume
  bool has_volume = false;
  DevVolume dv{};
  std::vector<void*> vol_allocs;
  void* cq_alloc = nullptr;
  void* bf_alloc = nullptr;
  void* bu_alloc = nullptr;    // bricku8 codes (+ one zero unit)
  void* bur_alloc = nullptr;   // bricku8 per-brick {min, max - min} (+ the {0, 0} entry of the zero unit)
  int layout = VX_LAYOUT_AUTO;       // what the host asked for (vx_set_layout); eff_layout() is what a launch samples
  bool auto_no_cq = false;           // AUTO: the volume is too large for the cellquad layout (path modes use REFERENCE)
  bool auto_no_bf = false;           // AUTO: too large for brickf32 as well (everything uses REFERENCE)

  // transfer function
  float4* tf = nullptr;
  uint32_t tf_len = 0;
  std::vector<float> tf_host;

  // exact empty-space skipping (DVR): macro-cell bitmask, rebuilt when its inputs change
  std::vector<uint32_t> range_host;   // packed (min16<<16)|max16 per brick
  uint32_t* skip_dev = nullptr;
  bool skip_dirty = true;
  float skip_key[4] = {0, 0, 0, 0};   // density_scale, inv_maj, sample_range

  // default mode: local-majorant table (DevVolume::lmaj), rebuilt when its inputs change
  float* lmaj_dev = nullptr;
  bool lmaj_dirty = true;
  float lmaj_key[5] = {0, 0, 0, 0, 0};   // density_scale, inv_maj, maj, sample_range

  unsigned long long* fold_dev = nullptr;   // eight totals of fold_records

  // params
  VxParams params{};
  bool has_params = false;

  // framebuffers
  uint32_t W = 0, H = 0;
  TileMap tm{};
  float4* slab = nullptr;
  size_t slab_quads = 0;
  float4* image = nullptr;
  float4* env_tex = nullptr;   // environment map, GL row order
  float* env_imp = nullptr;    // importance pyramid
  float4* env_impq = nullptr;  // the pyramid as sibling quads (sample_environment)
  float env_avg_w = 0.0f;
  uint32_t env_w = 0, env_h = 0;
  uchar4* display = nullptr;
  uint32_t display_cap = 0;  // pixels
  uint32_t* tile_perm = nullptr;  // vx_set_tile_order: position -> tile, tile -> position (2 * n_tiles)
  uint32_t tile_perm_n = 0;
  size_t slab_cap = 0, image_cap = 0;

  // counters / timing
  DevCounters* dc = nullptr;   // one record per wave of the largest launch grid
  size_t dc_waves = 0;
  uint32_t* order = nullptr;   // launch permutation of the DVR kernel (build_order), dc_waves/4 entries
  bool use_order = true;
  int tex_checked_res[2] = {-1, -1};  // DevVolume::ray_flags: the resolution (pixel + 0.5) / res was last tried against its reciprocal form
  bool tex_by_reciprocal[2] = {false, false};
  bool dvr_fuse = true;              // VX_DVR_FUSE=0: multi-frame DVR launches write per-frame results and merge_results blends them
  bool dvr_shared_window = false;    // VX_DVR_WG=1: one LDS window per workgroup in launches of a multiple of 32 frames (vx_dvr_lds.hpp, WG)
  int order_builds_left = 2;   // rebuild the order after the first frames that follow a change
  VxCounters base{};           // totals folded in when the record array is reallocated
  std::vector<EventPair> free_events, pending_events;
  double kernel_ms = 0.0, last_kernel_ms = 0.0, merge_ms = 0.0;
  uint64_t launches = 0, frames = 0;
  uint32_t min_launch_frames = 0, max_launch_frames = 0;   // what the launches since the last reset covered
  hipStream_t aux_stream = nullptr;   // layout builds of an upload, overlapped with the atlas copy
  double upload_seconds = 0.0;        // wall time of the last vx_upload_volume (copies + layout build)
  uint64_t upload_host_bytes = 0;     // host bytes it moved over PCIe
  int upload_pinned = 0;              // whether the atlas could be pinned in place
  void note_launch(uint32_t n) {
    launches += 1;
    frames += n;
    min_launch_frames = (min_launch_frames == 0 || n < min_launch_frames) ? n : min_launch_frames;
    max_launch_frames = n > max_launch_frames ? n : max_launch_frames;
  }
  int dvr_variant = -1;  // -1: tuned kernel; 0: generic
  int paths_variant = 0;   // 0 / 2: one pixel per lane (render_generic); 1: path segments re-packed through LDS
                           // (vx_paths.hpp, VX_PATHS_KERNEL=packed) -- same bits, measured 3-11 % slower; 3: the
                           // wave-persistent event-batched form (vx_events.hpp, VX_PATHS_KERNEL=events) -- same
                           // bits, denser lanes, measured 1.4-1.8x slower

  // frame pipelining (vx_render_frames): independent accumulation frames in flight on their own
  // streams, each into its own result slab + counter records; blended in order afterwards
  struct Pipe {
    hipStream_t stream = nullptr;
    float4* result = nullptr;
    DevCounters* dc = nullptr;
    hipEvent_t done = nullptr;
    hipEvent_t merged = nullptr;  // recorded on the main stream after this slot's result was blended
    bool has_merged = false;
  };
  std::vector<Pipe> pipes;
  // the result slabs and counter records of all slots are ONE allocation each (slot i at i * pipe_quads /
  // i * pipe_waves): the event kernel addresses a frame slot by base + stride instead of by a pointer table
  float4* pipe_result_pool = nullptr;
  DevCounters* pipe_dc_pool = nullptr;
  size_t pipe_quads = 0, pipe_waves = 0;
  uint32_t pipe_next = 0;
  int dp_env = -1;       // VX_DVR_DP=1: depth-parallel waves (experiment, see vx_dvr.hpp)
  bool dp_active() const { return dp_env == 1; }
};

#define VX_FAIL(ctx, code, ...)                       \
  do {                                                \
    char buf_[512];                                   \
    snprintf(buf_, sizeof buf_, __VA_ARGS__);         \
    (ctx)->err = buf_;                                \
    return (code);                                    \
  } while (0)

#define VX_HIP(ctx, expr)                                                                   \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) VX_FAIL(ctx, VX_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

// every entry point that touches the device first makes the context's device current for the
// calling thread (a host with several contexts / devices must not depend on its own hipSetDevice)
#define VX_DEV(ctx) VX_HIP(ctx, hipSetDevice((ctx)->device))


static void free_volume(VxContext* c) {
  for (void* p : c->vol_allocs) (void)hipFree(p);
  c->vol_allocs.clear();
  if (c->cq_alloc) (void)hipFree(c->cq_alloc);
  if (c->bf_alloc) (void)hipFree(c->bf_alloc);
  if (c->bu_alloc) (void)hipFree(c->bu_alloc);
  if (c->bur_alloc) (void)hipFree(c->bur_alloc);
  c->cq_alloc = nullptr;
  c->bf_alloc = nullptr;
  c->bu_alloc = c->bur_alloc = nullptr;
  c->dv = DevVolume{};
  c->has_volume = false;
  c->skip_dirty = true;
  if (c->lmaj_dev) (void)hipFree(c->lmaj_dev);
  c->lmaj_dev = nullptr;
  c->lmaj_dirty = true;
}

static void drain_events(VxContext* c) {
  for (auto& e : c->pending_events) {
    (void)hipEventSynchronize(e.b);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
      if (e.merge) {
        c->merge_ms += ms;
      } else {
        c->kernel_ms += ms;
        c->last_kernel_ms = ms / (float)(e.launches ? e.launches : 1u);
      }
    }
    e.merge = false;
    c->free_events.push_back(e);
  }
  c->pending_events.clear();
}

static void update_tilemap(VxContext* c) {
  TileMap& t = c->tm;
  t.W = c->W;
  t.H = c->H;
  t.tiles_x = (c->W + VX_SHARD_TILE - 1) / VX_SHARD_TILE;
  t.tiles_y = (c->H + VX_SHARD_TILE - 1) / VX_SHARD_TILE;
  t.n_tiles = t.tiles_x * t.tiles_y;
  t.shard_count = c->has_params && c->params.shard_count > 0 ? (uint32_t)c->params.shard_count : 1u;
  t.shard_rank = c->has_params ? (uint32_t)c->params.shard_rank : 0u;
  t.tiles_per_shard = (t.n_tiles + t.shard_count - 1) / t.shard_count;
  if (c->tile_perm && c->tile_perm_n != t.n_tiles) {  // the order belongs to another tile grid
    (void)hipFree(c->tile_perm);
    c->tile_perm = nullptr;
    c->tile_perm_n = 0;
  }
  t.perm = c->tile_perm;
  t.inv = c->tile_perm ? c->tile_perm + t.n_tiles : nullptr;
}

static int alloc_framebuffers(VxContext* c) {
  if (c->order && c->dc_waves) {  // the launch permutation belongs to the old grid: back to identity
    std::vector<uint32_t> ident(c->dc_waves / 4);
    for (size_t i = 0; i < ident.size(); ++i) ident[i] = (uint32_t)i;
    VX_HIP(c, hipMemcpy(c->order, ident.data(), ident.size() * 4, hipMemcpyHostToDevice));
    c->order_builds_left = 2;
  }
  // grow-only: the low-resolution preview (viewer.ts:1167-1188) resizes twice per restart
  update_tilemap(c);
  c->slab_quads = (size_t)c->tm.tiles_per_shard * 4096u;
  if (c->slab_quads == 0) return VX_OK;
  size_t px = (size_t)c->W * c->H;
  if (c->slab_quads > c->slab_cap) {
    if (c->slab) (void)hipFree(c->slab);
    c->slab = nullptr;
    c->slab_cap = 0;
    VX_HIP(c, hipMalloc(&c->slab, c->slab_quads * sizeof(float4)));
    c->slab_cap = c->slab_quads;
  }
  if (px > c->image_cap) {
    if (c->image) (void)hipFree(c->image);
    c->image = nullptr;
    c->image_cap = 0;
    VX_HIP(c, hipMalloc(&c->image, px * sizeof(float4)));
    c->image_cap = px;
  }
  if (px > c->display_cap) {
    if (c->display) (void)hipFree(c->display);
    c->display = nullptr;
    c->display_cap = 0;
    VX_HIP(c, hipMalloc(&c->display, px * sizeof(uchar4)));
    c->display_cap = (uint32_t)px;
  }
  VX_HIP(c, hipMemsetAsync(c->slab, 0, c->slab_quads * sizeof(float4), c->stream));
  return VX_OK;
}


// ---- exact empty-space skipping: host-side construction of the macro-cell bitmask -------------
// Rule (restated independently by the oracle): TF bin i is dead when its alpha
// is 0 or it lies wholly outside the sample range (one-bin margin); a brick is transparent when
// every bin from I(min)-1 to I(max)+1 is dead, I(x) = floor(x*density_scale*inv_maj*L); a macro
// cell of w = 2^level bricks per axis is empty when the w+1 bricks per axis that can hold a tap of
// its cells (bricks m*w-1 .. m*w+w-1; value 0 outside the grid) are all transparent.
static float f16_bits_to_float(uint16_t h) {
  _Float16 v;
  memcpy(&v, &h, 2);
  return (float)v;
}
static int skip_level_for(const uint32_t extent[3]) {
  for (int g = 1; g <= 3; ++g) {
    uint64_t n = 1;
    for (int a = 0; a < 3; ++a) n *= (uint64_t)(extent[a] >> (3 + g)) + 1u;
    if (n <= 65536u) return g;
  }
  return 3;
}
static void compute_skip_mask(const VxParams& p, const uint32_t* range_packed, const uint32_t bc[3],
                              const uint32_t extent[3], const float* tf_rgba, uint32_t L,
                              std::vector<uint32_t>& bits, int& level_out, uint32_t md[3]) {
  const float lf = (float)L;
  // prefix count of live bins -> O(1) "any live bin in [a, b]"
  std::vector<uint32_t> live(L + 1, 0);
  for (uint32_t i = 0; i < L; ++i) {
    bool dead = tf_rgba[4 * (size_t)i + 3] == 0.0f || (float)((int)i + 2) / lf < p.sample_range[0] ||
                (float)((int)i - 1) / lf > p.sample_range[1];
    live[i + 1] = live[i] + (dead ? 0u : 1u);
  }
  auto transparent = [&](float lo, float hi) {
    float fa = floorf(((lo * p.volume_density_scale) * p.volume_inv_maj) * lf);
    float fb = floorf(((hi * p.volume_density_scale) * p.volume_inv_maj) * lf);
    // v_cvt_i32_f32 semantics: NaN -> 0, saturating
    auto f2i = [](float x) -> int64_t { return x != x ? 0 : (x >= 2147483648.0f ? 2147483647ll : (x <= -2147483648.0f ? -2147483648ll : (int64_t)x)); };
    int64_t a = f2i(fa) - 1, b = f2i(fb) + 1;
    if (a < 0) a = 0;
    if (b > (int64_t)L - 1) b = (int64_t)L - 1;
    if (b < a) return true;
    return live[(size_t)b + 1] - live[(size_t)a] == 0u;
  };
  const size_t nb = (size_t)bc[0] * bc[1] * bc[2];
  std::vector<uint8_t> opaque(nb);
  for (size_t i = 0; i < nb; ++i) {
    uint32_t pk = range_packed[i];
    opaque[i] = transparent(f16_bits_to_float((uint16_t)(pk >> 16)), f16_bits_to_float((uint16_t)pk)) ? 0 : 1;
  }
  const uint8_t zero_opaque = transparent(0.0f, 0.0f) ? 0 : 1;
  const int level = skip_level_for(extent);
  level_out = level;
  const int w = 1 << level;
  for (int a = 0; a < 3; ++a) md[a] = (extent[a] >> (3 + level)) + 1u;
  // separable OR over the window [m*w-1, m*w+w-1] per axis (out-of-grid bricks count as value 0)
  std::vector<uint8_t> ax((size_t)md[0] * bc[1] * bc[2]), ay((size_t)md[0] * md[1] * bc[2]);
  for (uint32_t z = 0; z < bc[2]; ++z)
    for (uint32_t y = 0; y < bc[1]; ++y)
      for (uint32_t m = 0; m < md[0]; ++m) {
        uint8_t o = 0;
        for (int b = (int)m * w - 1; b <= (int)m * w + w - 1; ++b)
          o |= (b < 0 || (uint32_t)b >= bc[0]) ? zero_opaque : opaque[((size_t)z * bc[1] + y) * bc[0] + b];
        ax[((size_t)z * bc[1] + y) * md[0] + m] = o;
      }
  for (uint32_t z = 0; z < bc[2]; ++z)
    for (uint32_t m = 0; m < md[1]; ++m)
      for (uint32_t x = 0; x < md[0]; ++x) {
        uint8_t o = 0;
        for (int b = (int)m * w - 1; b <= (int)m * w + w - 1; ++b)
          o |= (b < 0 || (uint32_t)b >= bc[1]) ? zero_opaque : ax[((size_t)z * bc[1] + b) * md[0] + x];
        ay[((size_t)z * md[1] + m) * md[0] + x] = o;
      }
  const size_t n = (size_t)md[0] * md[1] * md[2];
  bits.assign((n + 31) / 32, 0u);
  for (uint32_t m = 0; m < md[2]; ++m)
    for (uint32_t y = 0; y < md[1]; ++y)
      for (uint32_t x = 0; x < md[0]; ++x) {
        uint8_t o = 0;
        for (int b = (int)m * w - 1; b <= (int)m * w + w - 1; ++b)
          o |= (b < 0 || (uint32_t)b >= bc[2]) ? zero_opaque : ay[((size_t)b * md[1] + y) * md[0] + x];
        if (!o) {
          size_t i = ((size_t)m * md[1] + y) * md[0] + x;
          bits[i >> 5] |= 1u << (i & 31);
        }
      }
}

// the local majorants of the default mode, tabulated on the device with the operations of Frame::local_majorant
static int rebuild_local_majorants(VxContext* c) {
  const VxParams& p = c->params;
  const size_t n = 4 * (size_t)c->dv.bc[0] * c->dv.bc[1] * c->dv.bc[2] + 1;
  if (!c->lmaj_dev) VX_HIP(c, hipMalloc(&c->lmaj_dev, n * sizeof(float)));
  DevVolume dv = c->dv;
  dv.lmaj = nullptr;
  hipLaunchKernelGGL(build_local_majorants, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, p, dv, c->tf,
                     c->tf_len, c->lmaj_dev);
  VX_HIP(c, hipGetLastError());
  c->dv.lmaj = c->lmaj_dev;
  c->dv.lmaj_cells = (uint32_t)(n - 1);
  c->lmaj_dirty = false;
  c->lmaj_key[0] = p.volume_density_scale;
  c->lmaj_key[1] = p.volume_inv_maj;
  c->lmaj_key[2] = p.volume_maj;
  c->lmaj_key[3] = p.sample_range[0];
  c->lmaj_key[4] = p.sample_range[1];
  return VX_OK;
}

static int rebuild_skip_mask(VxContext* c) {
  const VxParams& p = c->params;
  std::vector<uint32_t> bits;
  int level = 1;
  uint32_t md[3];
  compute_skip_mask(p, c->range_host.data(), c->dv.bc, c->dv.extent, c->tf_host.data(), c->tf_len, bits, level, md);
  if (c->skip_dev) (void)hipFree(c->skip_dev);
  c->skip_dev = nullptr;
  VX_HIP(c, hipMalloc(&c->skip_dev, bits.size() * 4));
  VX_HIP(c, hipMemcpyAsync(c->skip_dev, bits.data(), bits.size() * 4, hipMemcpyHostToDevice, c->stream));
  VX_HIP(c, hipStreamSynchronize(c->stream));
  c->dv.skip_bits = c->skip_dev;
  c->dv.skip_level = (uint32_t)level;
  c->dv.skip_words = (uint32_t)bits.size();
  for (int a = 0; a < 3; ++a) c->dv.skip_dims[a] = md[a];
  c->skip_dirty = false;
  c->skip_key[0] = p.volume_density_scale;
  c->skip_key[1] = p.volume_inv_maj;
  c->skip_key[2] = p.sample_range[0];
  c->skip_key[3] = p.sample_range[1];
  return VX_OK;
}


// fold every record array (accumulator slot and pipeline slots) into c->base on the device and zero it
static int fold_counters(VxContext* c) {
  if (!c->dc || !c->dc_waves) return VX_OK;
  if (!c->fold_dev) VX_HIP(c, hipMalloc(&c->fold_dev, 10 * sizeof(unsigned long long)));
  hipLaunchKernelGGL(zero_totals, dim3(1), dim3(10), 0, c->stream, c->fold_dev);
  auto fold = [&](DevCounters* recs, size_t n) {
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(fold_records, dim3(blocks), dim3(256), 0, c->stream, recs, n, c->fold_dev);
  };
  fold(c->dc, c->dc_waves);
  for (auto& p : c->pipes) {
    if (!p.dc) continue;
    if (p.stream) VX_HIP(c, hipStreamSynchronize(p.stream));
    fold(p.dc, c->pipe_waves);
  }
  VX_HIP(c, hipGetLastError());
  VX_HIP(c, hipStreamSynchronize(c->stream));
  unsigned long long h[10];
  VX_HIP(c, hipMemcpy(h, c->fold_dev, sizeof h, hipMemcpyDeviceToHost));
  c->base.samples += h[0];
  c->base.lane_slots += h[1];
  c->base.rays += h[2];
  c->base.pixels += h[3];
  c->base.skip_steps += h[4];
  c->base.grad_samples += h[5];
  c->base.gathers += h[6];
  c->base.lds_reads += h[7];
  c->base.tf_samples += h[8];
  c->base.active_lane_slots += h[9];
  return VX_OK;
}

static int ensure_counters(VxContext* c, size_t waves) {
  if (waves <= c->dc_waves) return VX_OK;
  VX_HIP(c, hipStreamSynchronize(c->stream));
  int rc = fold_counters(c);
  if (rc) return rc;
  if (c->dc) (void)hipFree(c->dc);
  c->dc = nullptr;
  c->dc_waves = 0;
  VX_HIP(c, hipMalloc(&c->dc, waves * sizeof(DevCounters)));
  // on the context's stream: it is a non-blocking stream, a fill on the null stream is not ordered with the launches
  // that follow (seen under rocprofv3's counter collection: the late fill wiped the records of a launch)
  VX_HIP(c, hipMemsetAsync(c->dc, 0, waves * sizeof(DevCounters), c->stream));
  VX_HIP(c, hipStreamSynchronize(c->stream));
  c->dc_waves = waves;
  if (c->order) (void)hipFree(c->order);
  c->order = nullptr;
  {
    std::vector<uint32_t> ident(waves / 4);
    for (size_t i = 0; i < ident.size(); ++i) ident[i] = (uint32_t)i;
    VX_HIP(c, hipMalloc(&c->order, ident.size() * 4));
    VX_HIP(c, hipMemcpy(c->order, ident.data(), ident.size() * 4, hipMemcpyHostToDevice));
    c->order_builds_left = 2;
  }
  return VX_OK;
}

// The device layout the kernels of the current render mode sample.  VX_LAYOUT_AUTO (default): the DVR modes march
// the brickf32 layout through LDS windows (vx_dvr_lds.hpp: fastest, 4 bytes per voxel), `default` and `no_dda`
// gather from cellquad (two 16-byte loads per trilinear look-up instead of eight bounds-checked taps), `raymarch`
// reads its single nearest tap from brickf32.
static int primary_layout(const VxContext* c) {
  if (c->layout != VX_LAYOUT_AUTO) return c->layout;
  return c->auto_no_bf ? VX_LAYOUT_REFERENCE : VX_LAYOUT_BRICKF32;
}
static int eff_layout(const VxContext* c) {
  if (c->layout != VX_LAYOUT_AUTO) return c->layout;
  const int m = c->has_params ? c->params.render_mode : VX_MODE_DVR;
  // raymarch takes ONE nearest tap per sample (common.glsl:72-76): the 4-byte-per-voxel bricks serve it better than
  // the 18-byte-per-voxel quads, and the layout is resident already
  if (m == VX_MODE_DVR || m == VX_MODE_DVR_PHONG || m == VX_MODE_RAYMARCH) return primary_layout(c);
  // `default` / `no_dda`: cellquad (18 B / voxel) while the volume is inside its index range and the build fits the device
  // memory budget (ensure_cellquad); beyond that the fp32 bricks that are resident anyway -- eight taps per look-up, measured
  // 1.4x / 2.0x slower than cellquad and 2.0x / 2.2x faster than the reference textures on the 1024^3 volume at 3840x2160
  // (profiles/r04_layouts_1024.txt) -- and the reference textures only when neither native layout can index the volume
  return c->auto_no_cq ? primary_layout(c) : VX_LAYOUT_CELLQUAD;
}

template <int MODE>
static void launch_generic(VxContext* c, const MultiOut& mo, float weight, dim3 grid, size_t lds, hipStream_t stream) {
  grid.x *= mo.count ? mo.count : 1u;
  const int lay = eff_layout(c);
  if (lay == VX_LAYOUT_BRICKF32)
    hipLaunchKernelGGL((render_generic<MODE, LAYOUT_BF>), grid, dim3(256), lds, stream, c->params,
                       c->dv, c->tf, c->tf_len, mo, weight, c->tm);
  else if (lay == VX_LAYOUT_CELLQUAD)
    hipLaunchKernelGGL((render_generic<MODE, LAYOUT_CQ>), grid, dim3(256), lds, stream, c->params,
                       c->dv, c->tf, c->tf_len, mo, weight, c->tm);
  else
    hipLaunchKernelGGL((render_generic<MODE, LAYOUT_REF>), grid, dim3(256), lds, stream, c->params,
                       c->dv, c->tf, c->tf_len, mo, weight, c->tm);
}

// the three reference modes with the path segments re-packed through LDS (vx_paths.hpp)
template <int MODE>
static void launch_paths(VxContext* c, const MultiOut& mo, float weight, dim3 grid, size_t lds, hipStream_t stream) {
  grid.x *= mo.count ? mo.count : 1u;
  const int lay = eff_layout(c);
  if (lay == VX_LAYOUT_BRICKF32)
    hipLaunchKernelGGL((render_paths<MODE, LAYOUT_BF>), grid, dim3(256), lds, stream, c->params, c->dv, c->tf, c->tf_len,
                       mo, weight, c->tm);
  else if (lay == VX_LAYOUT_CELLQUAD)
    hipLaunchKernelGGL((render_paths<MODE, LAYOUT_CQ>), grid, dim3(256), lds, stream, c->params, c->dv, c->tf, c->tf_len,
                       mo, weight, c->tm);
  else
    hipLaunchKernelGGL((render_paths<MODE, LAYOUT_REF>), grid, dim3(256), lds, stream, c->params, c->dv, c->tf, c->tf_len,
                       mo, weight, c->tm);
}

// default / no_dda as the wave-persistent, event-batched path tracer (vx_events.hpp).  The frame slots of `mo` must be
// consecutive frames at constant strides (vx_render_frames allocates them that way).
template <int MODE>
static void launch_events(VxContext* c, const MultiOut& mo, float weight, dim3 grid, hipStream_t stream) {
  const uint32_t n = mo.count ? mo.count : 1u;
  const uint64_t out_stride = n > 1 ? (uint64_t)(mo.out[1] - mo.out[0]) : 0u, dc_stride = n > 1 ? (uint64_t)(mo.dc[1] - mo.dc[0]) : 0u;
  const uint32_t groups = (n + VX_EV_FRAMES - 1u) / VX_EV_FRAMES;
  grid.x *= groups;
  const size_t lds = (size_t)c->tf_len * sizeof(float4) + 4u * PF_COUNT * 64u * sizeof(float);
  const int lay = eff_layout(c);
#define VX_LAUNCH_EV(LAY) \
  hipLaunchKernelGGL((render_events<MODE, LAY>), grid, dim3(256), lds, stream, c->params, c->dv, c->tf, c->tf_len, mo.out[0], \
                     out_stride, mo.dc[0], dc_stride, mo.frame[0], n, weight, c->tm)
  if (lay == VX_LAYOUT_BRICKF32) VX_LAUNCH_EV(LAYOUT_BF);
  else if (lay == VX_LAYOUT_CELLQUAD) VX_LAUNCH_EV(LAYOUT_CQ);
  else VX_LAUNCH_EV(LAYOUT_REF);
#undef VX_LAUNCH_EV
}
static bool events_possible(const VxContext* c, const MultiOut& mo) {
  if (c->paths_variant != 3 || c->params.debug_hits || c->tf_len > TF_LDS_MAX) return false;
  if (c->params.render_mode != VX_MODE_DEFAULT && c->params.render_mode != VX_MODE_NO_DDA) return false;
  const uint32_t n = mo.count ? mo.count : 1u;
  for (uint32_t i = 1; i < n; ++i)   // consecutive frames, constant strides, slab slots below 2^26
    if (mo.frame[i] != mo.frame[0] + i || mo.out[i] - mo.out[0] != (ptrdiff_t)i * (mo.out[1] - mo.out[0]) ||
        mo.dc[i] - mo.dc[0] != (ptrdiff_t)i * (mo.dc[1] - mo.dc[0]))
      return false;
  return c->slab_quads < (1u << 26) && n <= 64u;
}

static void launch_generic_mode(VxContext* c, const MultiOut& mo, float weight, dim3 grid, hipStream_t stream) {
  if (events_possible(c, mo)) {
    if (c->params.render_mode == VX_MODE_DEFAULT) launch_events<VX_MODE_DEFAULT>(c, mo, weight, grid, stream);
    else launch_events<VX_MODE_NO_DDA>(c, mo, weight, grid, stream);
    return;
  }
  size_t lds = c->tf_len <= TF_LDS_MAX ? (size_t)c->tf_len * sizeof(float4) : 0;
  if (mo.fuse) lds += 4u * 320u * sizeof(float);   // fold_frames' scratch, one per wave (generic_fusable)
  // (bounces < 1: fragment.frag:86-101 still traces the primary segment and one light sample before it tests the
  // count; render_paths loops on `n_paths < bounces` and would leave the slab unwritten -- render_generic serves it)
  if (c->paths_variant == 1 && !c->params.debug_hits && c->params.render_mode <= VX_MODE_RAYMARCH && c->params.bounces >= 1) {
    switch (c->params.render_mode) {
      case VX_MODE_DEFAULT: launch_paths<VX_MODE_DEFAULT>(c, mo, weight, grid, lds, stream); break;
      case VX_MODE_NO_DDA: launch_paths<VX_MODE_NO_DDA>(c, mo, weight, grid, lds, stream); break;
      default: launch_paths<VX_MODE_RAYMARCH>(c, mo, weight, grid, lds, stream); break;
    }
    return;
  }
  switch (c->params.render_mode) {
    case VX_MODE_DEFAULT: launch_generic<VX_MODE_DEFAULT>(c, mo, weight, grid, lds, stream); break;
    case VX_MODE_NO_DDA: launch_generic<VX_MODE_NO_DDA>(c, mo, weight, grid, lds, stream); break;
    case VX_MODE_RAYMARCH: launch_generic<VX_MODE_RAYMARCH>(c, mo, weight, grid, lds, stream); break;
    case VX_MODE_DVR: launch_generic<VX_MODE_DVR>(c, mo, weight, grid, lds, stream); break;
    default: launch_generic<VX_MODE_DVR_PHONG>(c, mo, weight, grid, lds, stream); break;
  }
}

// will launch_generic_mode run render_generic<MODE <= RAYMARCH> for this context?  (Only that kernel applies the running mean of
// a 32-frame launch itself, MultiOut::fuse; the re-packed and event-batched path kernels and the DVR modes on the generic
// kernel keep the result slabs and merge_results.)
static bool generic_fusable(const VxContext* c, const MultiOut& mo) {
  if (c->params.render_mode > VX_MODE_RAYMARCH) return false;
  if (events_possible(c, mo)) return false;
  if (c->paths_variant == 1 && !c->params.debug_hits && c->params.bounces >= 1) return false;
  return true;
}

extern "C" {

const char* vx_version(void) { return "volxel_hip 0.1 (gfx950)"; }

int vx_create(int device_id, VxContext** out) {
  if (!out) return VX_ERR_INVALID;
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    g_create_error = "vx_create: no HIP device visible (libvolxel_hip has no CPU fallback)";
    return VX_ERR_NO_DEVICE;
  }
  if (device_id < 0 || device_id >= n) {
    g_create_error = "vx_create: device ordinal out of range";
    return VX_ERR_INVALID;
  }
  VxContext* c = new VxContext();
  c->device = device_id;
  if (hipSetDevice(device_id) != hipSuccess || hipGetDeviceProperties(&c->prop, device_id) != hipSuccess ||
      hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
    g_create_error = "vx_create: device initialisation failed";
    delete c;
    return VX_ERR_DEVICE;
  }
  c->stream = c->own_stream;
  {
    // lanes = pixels x frames (vx_kernels.hpp frame_group) reads per-lane frame slots from the kernel-argument segment
    // at the offsets of struct KArgs: check once per context that the compiler lays the arguments out that way
    VxParams tp{};
    DevVolume tv{};
    MultiOut tm{};
    tp.res[0] = 0x1234;
    tv.extent[0] = 0x5678u;
    for (uint32_t i = 0; i < (uint32_t)MERGE_MAX; ++i) {
      tm.out[i] = reinterpret_cast<float4*>((uintptr_t)0x100000000ull * (i + 3u) + 16u * i);
      tm.frame[i] = 0xabc00000u + 7u * i;
    }
    tm.count = MERGE_MAX;
    uint32_t* bad = nullptr;
    uint32_t hbad = 1;
    hipError_t ce = hipMalloc(&bad, 4);
    if (ce == hipSuccess) {
      hipLaunchKernelGGL(check_lane_frame_slot, dim3(1), dim3(64), 0, c->stream, tp, tv, (const float4*)nullptr, 0x9abcu, tm, bad);
      ce = hipGetLastError();
    }
    if (ce == hipSuccess) ce = hipMemcpyAsync(&hbad, bad, 4, hipMemcpyDeviceToHost, c->stream);
    if (ce == hipSuccess) ce = hipStreamSynchronize(c->stream);
    if (bad) (void)hipFree(bad);
    if (ce != hipSuccess || hbad != 0u) {
      g_create_error = ce != hipSuccess ? "vx_create: device self-check failed to run"
                                        : "vx_create: kernel-argument layout differs from struct KArgs (lane_frame_slot)";
      (void)hipStreamDestroy(c->own_stream);
      delete c;
      return VX_ERR_DEVICE;
    }
  }
  const char* v = getenv("VX_DVR_KERNEL");
  if (v && !strcmp(v, "generic")) c->dvr_variant = 0;
  const char* pk = getenv("VX_PATHS_KERNEL");
  if (pk && !strcmp(pk, "packed")) c->paths_variant = 1;     // path segments re-packed through LDS (vx_paths.hpp)
  if (pk && !strcmp(pk, "generic")) c->paths_variant = 2;    // one pixel per lane for the whole path (render_generic): the default
  if (pk && !strcmp(pk, "events")) c->paths_variant = 3;     // wave-persistent, event-batched (vx_events.hpp): measured slower
  const char* dpe = getenv("VX_DVR_DP");
  if (dpe) c->dp_env = atoi(dpe);
  const char* fu = getenv("VX_DVR_FUSE");
  if (fu) c->dvr_fuse = atoi(fu) != 0;
  const char* wg = getenv("VX_DVR_WG");
  if (wg) c->dvr_shared_window = atoi(wg) != 0;
  const char* o = getenv("VX_DVR_ORDER");
  if (o && !strcmp(o, "0")) c->use_order = false;
  *out = c;
  return VX_OK;
}

void vx_destroy(VxContext* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  drain_events(c);
  for (auto& e : c->free_events) {
    (void)hipEventDestroy(e.a);
    (void)hipEventDestroy(e.b);
  }
  free_volume(c);
  if (c->tf) (void)hipFree(c->tf);
  if (c->env_tex) (void)hipFree(c->env_tex);
  if (c->env_imp) (void)hipFree(c->env_imp);
  if (c->env_impq) (void)hipFree(c->env_impq);
  if (c->skip_dev) (void)hipFree(c->skip_dev);
  if (c->fold_dev) (void)hipFree(c->fold_dev);
  if (c->slab) (void)hipFree(c->slab);
  if (c->image) (void)hipFree(c->image);
  if (c->display) (void)hipFree(c->display);
  if (c->tile_perm) (void)hipFree(c->tile_perm);
  if (c->dc) (void)hipFree(c->dc);
  if (c->order) (void)hipFree(c->order);
  for (auto& p : c->pipes) {
    if (p.stream) { (void)hipStreamSynchronize(p.stream); (void)hipStreamDestroy(p.stream); }
    if (p.done) (void)hipEventDestroy(p.done);
    if (p.merged) (void)hipEventDestroy(p.merged);
  }
  if (c->pipe_result_pool) (void)hipFree(c->pipe_result_pool);
  if (c->pipe_dc_pool) (void)hipFree(c->pipe_dc_pool);
  if (c->aux_stream) (void)hipStreamDestroy(c->aux_stream);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

const char* vx_last_error(const VxContext* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int vx_set_stream(VxContext* c, void* s) {
  if (!c) return VX_ERR_INVALID;
  VX_DEV(c);
  (void)hipStreamSynchronize(c->stream);
  c->stream = s ? (hipStream_t)s : c->own_stream;
  return VX_OK;
}

// Allocate the device layout the trilinear modes sample (cellquad / brickf32); returns its number of z layers
// (apron-brick layers / brick layers) through n_layers.  The contents are filled by build_layout_layers.
static int alloc_layout(VxContext* c, uint32_t& n_layers) {
  if (c->cq_alloc) {
    (void)hipFree(c->cq_alloc);
    c->cq_alloc = nullptr;
    c->dv.cq = nullptr;
  }
  if (c->bf_alloc) {
    (void)hipFree(c->bf_alloc);
    c->bf_alloc = nullptr;
    c->dv.bf = nullptr;
    c->dv.bf_zero = 0;
  }
  if (c->bu_alloc) (void)hipFree(c->bu_alloc);
  if (c->bur_alloc) (void)hipFree(c->bur_alloc);
  c->bu_alloc = c->bur_alloc = nullptr;
  c->dv.bu = nullptr;
  c->dv.bu_range = nullptr;
  n_layers = 0;
  c->auto_no_cq = c->auto_no_bf = false;
  if (c->layout == VX_LAYOUT_AUTO) {   // what the volume's size allows
    const uint64_t nv = (uint64_t)c->dv.bc[0] * c->dv.bc[1] * c->dv.bc[2] * 512u;
    const uint64_t nq = (uint64_t)(c->dv.bc[0] + 1) * (c->dv.bc[1] + 1) * (c->dv.bc[2] + 1) * CQ_BRICK_QUADS;
    c->auto_no_bf = nv / 4u > 0xffffffffull;
    c->auto_no_cq = nq > 0xffffffffull;
  }
  if (primary_layout(c) == VX_LAYOUT_BRICKF32) {
    uint64_t n_vox = (uint64_t)c->dv.bc[0] * c->dv.bc[1] * c->dv.bc[2] * 512u;
    // the staging loads index the layout in 16-byte units with 32 bits: 64 GiB, about 2500^3 voxels
    if (n_vox / 4u > 0xffffffffull)
      VX_FAIL(c, VX_ERR_INVALID, "volume too large for the brickf32 layout (%llu voxels): select VX_LAYOUT_REFERENCE "
              "with vx_set_layout", (unsigned long long)n_vox);
    // + one zero 16-byte chunk behind the last brick: the window staging of the LDS kernel reads it for rows and
    // chunks outside the volume (one select per load instead of a branch and a zero fill)
    VX_HIP(c, hipMalloc(&c->bf_alloc, n_vox * sizeof(float) + 16));
    VX_HIP(c, hipMemsetAsync((char*)c->bf_alloc + n_vox * sizeof(float), 0, 16, c->stream));
    c->dv.bf = (const float*)c->bf_alloc;
    c->dv.bf_zero = n_vox <= 0xfffffff0ull ? (uint32_t)n_vox : 0u;
    n_layers = c->dv.bc[2];
    return VX_OK;
  }
  if (primary_layout(c) == VX_LAYOUT_BRICKU8) {
    const uint64_t n_bricks = (uint64_t)c->dv.bc[0] * c->dv.bc[1] * c->dv.bc[2];
    const uint64_t n_units = n_bricks * 128u;   // dwords of four codes; the staging indexes them with 32 bits
    if (n_units > 0xfffffff0ull)
      VX_FAIL(c, VX_ERR_INVALID, "volume too large for the bricku8 layout (%llu bricks): select VX_LAYOUT_REFERENCE "
              "with vx_set_layout", (unsigned long long)n_bricks);
    // + one zero unit behind the last brick and its {0, 0} range: rows and chunks outside the volume decode to +0
    VX_HIP(c, hipMalloc(&c->bu_alloc, (n_units + 4u) * sizeof(uint32_t)));
    VX_HIP(c, hipMalloc(&c->bur_alloc, (n_bricks + 1u) * sizeof(float2)));
    VX_HIP(c, hipMemsetAsync((char*)c->bu_alloc + n_units * sizeof(uint32_t), 0, 4u * sizeof(uint32_t), c->stream));
    VX_HIP(c, hipMemsetAsync((char*)c->bur_alloc + n_bricks * sizeof(float2), 0, sizeof(float2), c->stream));
    c->dv.bu = (const uint32_t*)c->bu_alloc;
    c->dv.bu_range = (const float2*)c->bur_alloc;
    n_layers = c->dv.bc[2];
    return VX_OK;
  }
  if (primary_layout(c) != VX_LAYOUT_CELLQUAD) return VX_OK;
  for (int i = 0; i < 3; ++i) c->dv.cq_bc[i] = c->dv.bc[i] + 1;
  uint64_t n_quads = (uint64_t)c->dv.cq_bc[0] * c->dv.cq_bc[1] * c->dv.cq_bc[2] * CQ_BRICK_QUADS;
  // the march indexes quads with 32 bits (and bricks with 24-bit multiplies): 64 GiB, about 1550^3 voxels
  if (n_quads > 0xffffffffull)
    VX_FAIL(c, VX_ERR_INVALID,
            "volume too large for the cellquad layout (%llu quads > 2^32): select VX_LAYOUT_BRICKF32 or "
            "VX_LAYOUT_REFERENCE with vx_set_layout", (unsigned long long)n_quads);
  VX_HIP(c, hipMalloc(&c->cq_alloc, n_quads * sizeof(float4)));
  c->dv.cq = (const float4*)c->cq_alloc;
  n_layers = c->dv.cq_bc[2];
  return VX_OK;
}

// fill z layers [z0, z1) of the layout on `st` (one thread per quad / voxel; layers are contiguous in both layouts)
static int build_layout_layers(VxContext* c, uint32_t z0, uint32_t z1, hipStream_t st) {
  if (z1 <= z0) return VX_OK;
  if (primary_layout(c) == VX_LAYOUT_BRICKF32) {
    const uint64_t per = (uint64_t)c->dv.bc[0] * c->dv.bc[1] * 512u;
    const uint64_t first = per * z0, end = per * z1;
    for (uint64_t at = first; at < end;) {   // <= 2^31 threads per launch
      uint64_t n = end - at < (1ull << 31) ? end - at : (1ull << 31);
      hipLaunchKernelGGL(build_brickf32, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, c->dv,
                         (float*)c->bf_alloc, at, at + n);
      at += n;
    }
  } else if (primary_layout(c) == VX_LAYOUT_BRICKU8) {
    const uint64_t bricks_per = (uint64_t)c->dv.bc[0] * c->dv.bc[1];
    const uint64_t first = bricks_per * 128u * z0, end = bricks_per * 128u * z1;   // dword units, < 2^32 (alloc_layout)
    hipLaunchKernelGGL(build_bricku8, dim3((uint32_t)((end - first + 255) / 256)), dim3(256), 0, st, c->dv,
                       (uint32_t*)c->bu_alloc, first, end);
    const uint32_t b0 = (uint32_t)(bricks_per * z0), b1 = (uint32_t)(bricks_per * z1);
    hipLaunchKernelGGL(build_bricku8_range, dim3((b1 - b0 + 255) / 256), dim3(256), 0, st, c->dv, (float2*)c->bur_alloc,
                       b0, b1);
  } else if (primary_layout(c) == VX_LAYOUT_CELLQUAD) {
    const uint64_t per = (uint64_t)c->dv.cq_bc[0] * c->dv.cq_bc[1] * CQ_BRICK_QUADS;
    const uint64_t first = per * z0, end = per * z1;
    for (uint64_t at = first; at < end;) {
      uint64_t n = end - at < (1ull << 31) ? end - at : (1ull << 31);
      hipLaunchKernelGGL(build_cellquad, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, c->dv,
                         (float4*)c->cq_alloc, at, at + n);
      at += n;
    }
  }
  VX_HIP(c, hipGetLastError());
  return VX_OK;
}

// dvr_phong runs on the LDS-window kernel, which samples the brickf32 layout: a context whose primary layout is
// cellquad gets the 4-byte-per-voxel brick layout built beside it the first time Phong is rendered
static int ensure_brickf32(VxContext* c) {
  if (c->dv.bf) return VX_OK;
  const uint64_t n_vox = (uint64_t)c->dv.bc[0] * c->dv.bc[1] * c->dv.bc[2] * 512u;
  if (n_vox / 4u > 0xffffffffull) return VX_OK;   // too large: the generic kernel serves Phong
  VX_HIP(c, hipMalloc(&c->bf_alloc, n_vox * sizeof(float) + 16));   // + the zero chunk (alloc_layout)
  VX_HIP(c, hipMemsetAsync((char*)c->bf_alloc + n_vox * sizeof(float), 0, 16, c->stream));
  c->dv.bf = (const float*)c->bf_alloc;
  c->dv.bf_zero = n_vox <= 0xfffffff0ull ? (uint32_t)n_vox : 0u;
  for (uint64_t at = 0; at < n_vox;) {
    uint64_t n = n_vox - at < (1ull << 31) ? n_vox - at : (1ull << 31);
    hipLaunchKernelGGL(build_brickf32, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, c->stream, c->dv,
                       (float*)c->bf_alloc, at, at + n);
    at += n;
  }
  VX_HIP(c, hipGetLastError());
  return VX_OK;
}

// the cellquad layout built beside the primary one the first time a mode that gathers from it is rendered
// (VX_LAYOUT_AUTO: the path-traced reference modes)
static int ensure_cellquad(VxContext* c) {
  if (c->dv.cq) return VX_OK;
  for (int i = 0; i < 3; ++i) c->dv.cq_bc[i] = c->dv.bc[i] + 1;
  const uint64_t n_quads = (uint64_t)c->dv.cq_bc[0] * c->dv.cq_bc[1] * c->dv.cq_bc[2] * CQ_BRICK_QUADS;
  if (c->layout == VX_LAYOUT_AUTO) {
    // AUTO builds this layout on demand, beside what is resident: only when it leaves half of the free device memory to
    // the rest of the process (19.8 GB for 1024^3 on a 288 GB MI355X: always; a volume near the layout's 64 GiB index limit
    // on a device that other contexts share: not necessarily).  Otherwise `default` / `no_dda` take the resident bricks.
    // VX_AUTO_CELLQUAD_MAX_BYTES (environment, read here) overrides the budget -- 0 keeps AUTO off this layout.
    size_t free_b = 0, total_b = 0;
    VX_HIP(c, hipMemGetInfo(&free_b, &total_b));
    uint64_t budget = (uint64_t)free_b / 2u;
    if (const char* e = getenv("VX_AUTO_CELLQUAD_MAX_BYTES")) budget = strtoull(e, nullptr, 10);
    if (n_quads * sizeof(float4) > budget) {
      c->auto_no_cq = true;
      return VX_OK;
    }
  }
  VX_HIP(c, hipMalloc(&c->cq_alloc, n_quads * sizeof(float4)));
  c->dv.cq = (const float4*)c->cq_alloc;
  for (uint64_t at = 0; at < n_quads;) {
    uint64_t n = n_quads - at < (1ull << 31) ? n_quads - at : (1ull << 31);
    hipLaunchKernelGGL(build_cellquad, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, c->stream, c->dv,
                       (float4*)c->cq_alloc, at, at + n);
    at += n;
  }
  VX_HIP(c, hipGetLastError());
  return VX_OK;
}

static int build_layout(VxContext* c) {   // whole layout on the context's stream (vx_set_layout)
  uint32_t n = 0;
  int rc = alloc_layout(c, n);
  if (rc) return rc;
  return build_layout_layers(c, 0, n, c->stream);
}

// Pin a caller-owned host range for the duration of an upload so that the copy engine reads it directly at
// PCIe rate ("pin/upload volumes to HBM", BASELINE north star).  Pageable memory would be staged through the
// runtime's bounce buffers at a fraction of that.  Failing to pin (already registered, locked-memory limit)
// is not an error: the copies then go the pageable way.
struct PinnedRange {
  void* p = nullptr;
  bool pinned = false;
  PinnedRange(const void* ptr, size_t bytes) {
    if (!ptr || bytes < (1u << 20)) return;
    p = const_cast<void*>(ptr);
    hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
    pinned = (e == hipSuccess);
    if (!pinned) (void)hipGetLastError();   // clear the sticky error
  }
  ~PinnedRange() {
    if (pinned) (void)hipHostUnregister(p);
  }
};

int vx_upload_volume(VxContext* c, const uint32_t* indirection, const uint32_t ind_size[3],
                     const uint16_t* range, const uint32_t range_size[3], const uint8_t* atlas,
                     const uint32_t atlas_size[3], int n_mips, const uint16_t* const* mip_data,
                     const uint32_t (*mip_size)[3], const uint32_t index_extent[3]) {
  if (!c) return VX_ERR_INVALID;
  VX_DEV(c);
  if (!indirection || !range || !ind_size || !range_size || !atlas_size || !index_extent)
    VX_FAIL(c, VX_ERR_INVALID, "vx_upload_volume: null argument");
  if (n_mips != 3 || !mip_data || !mip_size)
    VX_FAIL(c, VX_ERR_INVALID, "vx_upload_volume: expected 3 range mipmaps (brick.rs:13)");
  for (int i = 0; i < 3; ++i) {
    if (ind_size[i] != range_size[i] || ind_size[i] == 0 || ind_size[i] >= 1024u)
      VX_FAIL(c, VX_ERR_INVALID, "vx_upload_volume: bad brick grid dimensions");
    if (index_extent[i] != ind_size[i] * 8u)
      VX_FAIL(c, VX_ERR_INVALID, "vx_upload_volume: index_extent must be brick_count*8 (brick.rs:236-238)");
    if (i < 2 && atlas_size[i] != ind_size[i] * 8u)
      VX_FAIL(c, VX_ERR_INVALID, "vx_upload_volume: atlas x/y dims must be brick_count*8 (brick.rs:85)");
  }
  if (atlas_size[2] % 8u != 0 || atlas_size[2] > ind_size[2] * 8u)
    VX_FAIL(c, VX_ERR_INVALID, "vx_upload_volume: bad atlas depth");
  size_t atlas_bytes = (size_t)atlas_size[0] * atlas_size[1] * atlas_size[2];
  if (atlas_bytes && !atlas) VX_FAIL(c, VX_ERR_INVALID, "vx_upload_volume: null atlas");
  for (int k = 0; k < 3; ++k) {
    if (!mip_data[k] && (size_t)mip_size[k][0] * mip_size[k][1] * mip_size[k][2])
      VX_FAIL(c, VX_ERR_INVALID, "vx_upload_volume: null mip %d", k);
    if (mip_size[k][0] != (ind_size[0] >> (k + 1)) || mip_size[k][1] != (ind_size[1] >> (k + 1)) ||
        mip_size[k][2] != (ind_size[2] >> (k + 1)))
      VX_FAIL(c, VX_ERR_INVALID, "vx_upload_volume: mip %d has wrong dimensions (brick.rs:156)", k);
  }
  const auto t_begin = std::chrono::steady_clock::now();
  VX_HIP(c, hipStreamSynchronize(c->stream));
  free_volume(c);
  const size_t nb = (size_t)ind_size[0] * ind_size[1] * ind_size[2];
  const size_t per_layer = (size_t)ind_size[0] * ind_size[1];
  // every pointer must address an allocated atlas brick; while scanning, note how many 8-slice atlas layers
  // the bricks of each brick z layer reach into (the builder allocates slots in scan order, brick.rs:127-129,
  // so this grows with z and the layout of early layers can be built while the rest of the atlas still copies)
  std::vector<uint32_t> reach(ind_size[2], 0u);
  {
    const uint32_t max_slot = atlas_size[2] / 8u;
    for (size_t i = 0; i < nb; ++i) {
      uint32_t p = indirection[i];
      uint32_t az = (p >> 20) & 1023u;
      if ((p & 1023u) >= ind_size[0] || ((p >> 10) & 1023u) >= ind_size[1] || (az >= max_slot && p != 0))
        VX_FAIL(c, VX_ERR_INVALID, "vx_upload_volume: indirection pointer outside the atlas");
      uint32_t& r = reach[i / per_layer];
      if (max_slot && az + 1u > r) r = az + 1u;   // constant bricks alias slot 0 (quirk Q6): also fine
    }
    for (uint32_t z = 1; z < ind_size[2]; ++z) reach[z] = reach[z] > reach[z - 1] ? reach[z] : reach[z - 1];
  }
  auto alloc = [&](size_t bytes, void** dst) -> int {
    *dst = nullptr;
    if (bytes == 0) return VX_OK;
    VX_HIP(c, hipMalloc(dst, bytes));
    c->vol_allocs.push_back(*dst);
    return VX_OK;
  };
  void *d_ind = nullptr, *d_range = nullptr, *d_atlas = nullptr, *d_mip[3] = {nullptr, nullptr, nullptr};
  int rc;
  // the atlas allocation is never empty: a tap that points outside the pruned atlas reads byte 0 and selects 0
  // (lookup_density_brick is straight-line code)
  if ((rc = alloc(nb * 4, &d_ind)) || (rc = alloc(nb * 4, &d_range)) || (rc = alloc(atlas_bytes ? atlas_bytes : 16, &d_atlas))) { free_volume(c); return rc; }
  if (!atlas_bytes) VX_HIP(c, hipMemsetAsync(d_atlas, 0, 16, c->stream));
  size_t mip_n[3];
  for (int k = 0; k < 3; ++k) {
    mip_n[k] = (size_t)mip_size[k][0] * mip_size[k][1] * mip_size[k][2];
    if ((rc = alloc(mip_n[k] * 4, &d_mip[k]))) { free_volume(c); return rc; }
    c->dv.mips[k] = (const uint32_t*)d_mip[k];
    for (int i = 0; i < 3; ++i) c->dv.mip_size[k][i] = mip_size[k][i];
  }
  c->dv.indirection = (const uint32_t*)d_ind;
  c->dv.range = (const uint32_t*)d_range;   // u16 stream [max,min] == LE u32 (min<<16)|max
  c->dv.atlas = (const uint8_t*)d_atlas;
  for (int i = 0; i < 3; ++i) {
    c->dv.bc[i] = ind_size[i];
    c->dv.atlas_size[i] = atlas_size[i];
    c->dv.extent[i] = index_extent[i];
  }
  c->range_host.assign((const uint32_t*)range, (const uint32_t*)range + nb);
  c->skip_dirty = true;
  c->order_builds_left = 2;
  uint32_t n_layers = 0;
  if ((rc = alloc_layout(c, n_layers))) { free_volume(c); return rc; }
  if (!c->aux_stream) VX_HIP(c, hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking));

  // ---- copies: metadata first, then the atlas in chunks of whole 8-slice layers from pinned memory;
  //      the layout layers whose bricks are complete are built on the aux stream behind each chunk
  // Every failure from here on goes through ONE exit (below): both streams are synchronised before the pinned
  // ranges are unregistered and the partial volume is freed -- an early return would unpin pages an earlier
  // asynchronous copy may still be reading.
  PinnedRange pin_atlas(atlas, atlas_bytes), pin_ind(indirection, nb * 4), pin_range(range, nb * 4);
  hipError_t le = hipMemcpyAsync(d_ind, indirection, nb * 4, hipMemcpyHostToDevice, c->stream);
  if (le == hipSuccess) le = hipMemcpyAsync(d_range, range, nb * 4, hipMemcpyHostToDevice, c->stream);
  for (int k = 0; k < 3 && le == hipSuccess; ++k)
    if (mip_n[k]) le = hipMemcpyAsync(d_mip[k], mip_data[k], mip_n[k] * 4, hipMemcpyHostToDevice, c->stream);
  const uint32_t atlas_layers = atlas_size[2] / 8u;
  const size_t layer_bytes = (size_t)atlas_size[0] * atlas_size[1] * 8u;
  const uint32_t chunk_layers = layer_bytes ? (uint32_t)std::max<size_t>(1, (16u << 20) / layer_bytes) : 1u;
  std::vector<hipEvent_t> evs;
  uint32_t built = 0;     // layout layers launched so far
  auto buildable = [&](uint32_t copied) {   // layout layers whose source bricks lie in the copied atlas prefix
    uint32_t z = built;
    while (z < n_layers) {
      // cellquad apron layer z reads brick layers z-1 and z; brickf32 layer z reads brick layer z
      uint32_t top = z < ind_size[2] ? z : ind_size[2] - 1u;
      if (reach[top] > copied) break;
      ++z;
    }
    return z;
  };
  for (uint32_t l0 = 0; l0 < atlas_layers && le == hipSuccess; l0 += chunk_layers) {
    uint32_t l1 = l0 + chunk_layers < atlas_layers ? l0 + chunk_layers : atlas_layers;
    le = hipMemcpyAsync((char*)d_atlas + l0 * layer_bytes, atlas + l0 * layer_bytes, (l1 - l0) * layer_bytes,
                        hipMemcpyHostToDevice, c->stream);
    if (le != hipSuccess) break;
    uint32_t z1 = buildable(l1);
    if (z1 > built && l1 < atlas_layers) {   // the last chunk's layers go with the final build below
      hipEvent_t e;
      if ((le = hipEventCreateWithFlags(&e, hipEventDisableTiming)) != hipSuccess) break;
      evs.push_back(e);
      if ((le = hipEventRecord(e, c->stream)) != hipSuccess) break;
      if ((le = hipStreamWaitEvent(c->aux_stream, e, 0)) != hipSuccess) break;
      if ((rc = build_layout_layers(c, built, z1, c->aux_stream))) { le = hipErrorUnknown; break; }
      built = z1;
    }
  }
  if (le == hipSuccess && built < n_layers) {
    hipEvent_t e;
    if ((le = hipEventCreateWithFlags(&e, hipEventDisableTiming)) == hipSuccess) {
      evs.push_back(e);
      le = hipEventRecord(e, c->stream);
      if (le == hipSuccess) le = hipStreamWaitEvent(c->aux_stream, e, 0);
      if (le == hipSuccess && (rc = build_layout_layers(c, built, n_layers, c->aux_stream))) le = hipErrorUnknown;
    }
  }
  hipError_t s1 = hipStreamSynchronize(c->stream);    // host buffers may be dropped on return
  hipError_t s2 = hipStreamSynchronize(c->aux_stream);
  for (hipEvent_t e : evs) (void)hipEventDestroy(e);
  if (le != hipSuccess || s1 != hipSuccess || s2 != hipSuccess) {
    hipError_t bad = le != hipSuccess ? le : (s1 != hipSuccess ? s1 : s2);
    free_volume(c);
    VX_FAIL(c, VX_ERR_DEVICE, "vx_upload_volume: %s", hipGetErrorString(bad));
  }
  c->has_volume = true;
  c->upload_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
  c->upload_host_bytes = (uint64_t)atlas_bytes + (uint64_t)nb * 8u + (uint64_t)(mip_n[0] + mip_n[1] + mip_n[2]) * 4u;
  c->upload_pinned = pin_atlas.pinned ? 1 : 0;
  return VX_OK;
}

int vx_upload_stats(VxContext* c, double* seconds, uint64_t* host_bytes, int* pinned) {
  if (!c) return VX_ERR_INVALID;
  if (seconds) *seconds = c->upload_seconds;
  if (host_bytes) *host_bytes = c->upload_host_bytes;
  if (pinned) *pinned = c->upload_pinned;
  return VX_OK;
}

int vx_upload_brick_grid(VxContext* c, const VxBrickGrid* g) {
  if (!c) return VX_ERR_INVALID;
  if (!g) VX_FAIL(c, VX_ERR_INVALID, "vx_upload_brick_grid: null grid");
  uint32_t is[3], rs[3], as[3], ext[3], ms[3][3];
  vxb_indirection_size(g, is);
  vxb_range_size(g, rs);
  vxb_atlas_size(g, as);
  vxb_index_extent(g, ext);
  const uint32_t n = vxb_range_mipmaps(g);
  if (n != 3) VX_FAIL(c, VX_ERR_INVALID, "vx_upload_brick_grid: grid has %u range mips, expected 3", n);
  const uint16_t* mips[3];
  for (uint32_t i = 0; i < 3; ++i) {
    mips[i] = vxb_range_mipmap(g, i);
    vxb_range_mipmap_stride(g, i, ms[i]);
  }
  return vx_upload_volume(c, vxb_indirection_data(g), is, vxb_range_data(g), rs, vxb_atlas_data(g), as, 3, mips,
                          ms, ext);
}

int vx_set_layout(VxContext* c, int layout) {
  if (!c) return VX_ERR_INVALID;
  VX_DEV(c);
  if (layout != VX_LAYOUT_REFERENCE && layout != VX_LAYOUT_CELLQUAD && layout != VX_LAYOUT_BRICKF32 &&
      layout != VX_LAYOUT_AUTO && layout != VX_LAYOUT_BRICKU8)
    VX_FAIL(c, VX_ERR_INVALID, "vx_set_layout: unknown layout %d", layout);
  if (layout == c->layout) return VX_OK;
  c->layout = layout;
  if (c->has_volume) {
    VX_HIP(c, hipStreamSynchronize(c->stream));
    int rc = build_layout(c);
    if (rc) return rc;
    VX_HIP(c, hipStreamSynchronize(c->stream));
  }
  return VX_OK;
}

int vx_upload_transfer(VxContext* c, const float* rgba, uint32_t length) {
  if (!c) return VX_ERR_INVALID;
  VX_DEV(c);
  if (!rgba || length == 0) VX_FAIL(c, VX_ERR_INVALID, "vx_upload_transfer: empty transfer function");
  // the DVR composite runs straight-line for every lane of a wave step (a lane that does not contribute adds
  // dT * rgb with dT = +0): an Inf / NaN entry would turn that 0 into NaN for lanes that never sampled it
  for (size_t i = 0; i < (size_t)length * 4; ++i)
    if (!std::isfinite(rgba[i]))
      VX_FAIL(c, VX_ERR_INVALID, "vx_upload_transfer: entry %zu component %zu is not finite", i / 4, i % 4);
  VX_HIP(c, hipStreamSynchronize(c->stream));
  if (c->tf) (void)hipFree(c->tf);
  c->tf = nullptr;
  VX_HIP(c, hipMalloc(&c->tf, (size_t)length * sizeof(float4)));
  VX_HIP(c, hipMemcpy(c->tf, rgba, (size_t)length * sizeof(float4), hipMemcpyHostToDevice));
  c->tf_len = length;
  c->tf_host.assign(rgba, rgba + (size_t)length * 4);
  c->skip_dirty = true;
  c->lmaj_dirty = true;
  c->order_builds_left = 2;
  return VX_OK;
}

int vx_upload_environment(VxContext* c, const float* rgba, uint32_t w, uint32_t h) {
  if (!c) return VX_ERR_INVALID;
  VX_HIP(c, hipSetDevice(c->device));
  VX_HIP(c, hipStreamSynchronize(c->stream));
  for (auto& p : c->pipes)
    if (p.stream) VX_HIP(c, hipStreamSynchronize(p.stream));
  if (c->env_tex) (void)hipFree(c->env_tex);
  if (c->env_imp) (void)hipFree(c->env_imp);
  if (c->env_impq) (void)hipFree(c->env_impq);
  c->env_tex = nullptr;
  c->env_imp = nullptr;
  c->env_impq = nullptr;
  c->env_w = c->env_h = 0;
  if (!rgba) return VX_OK;
  if (w == 0 || h == 0 || w > 16384 || h > 16384)
    VX_FAIL(c, VX_ERR_INVALID, "vx_upload_environment: bad size %ux%u", w, h);
  std::vector<float> flipped((size_t)w * h * 4);  // UNPACK_FLIP_Y_WEBGL, environment.ts:30-32
  for (uint32_t y = 0; y < h; ++y)
    memcpy(flipped.data() + (size_t)(h - 1 - y) * w * 4, rgba + (size_t)y * w * 4, (size_t)w * 16);
  VX_HIP(c, hipMalloc(&c->env_tex, flipped.size() * sizeof(float)));
  VX_HIP(c, hipMalloc(&c->env_imp, (size_t)IMP_FLOATS * sizeof(float)));
  VX_HIP(c, hipMalloc(&c->env_impq, (size_t)IMPQ_QUADS * sizeof(float4)));
  VX_HIP(c, hipMemcpy(c->env_tex, flipped.data(), flipped.size() * sizeof(float), hipMemcpyHostToDevice));
  c->env_w = w;
  c->env_h = h;
  hipLaunchKernelGGL(build_importance, dim3(IMP_DIM * IMP_DIM / 256), dim3(256), 0, c->stream, c->env_tex, w, h,
                     c->env_imp);
  for (uint32_t k = 1; k < IMP_LEVELS; ++k) {
    uint32_t n = (IMP_DIM >> k) * (IMP_DIM >> k);
    hipLaunchKernelGGL(build_importance_mip, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->env_imp, k);
  }
  for (uint32_t k = 0; k + 1 < IMP_LEVELS; ++k) {
    uint32_t n = (IMP_DIM >> (k + 1)) * (IMP_DIM >> (k + 1));
    hipLaunchKernelGGL(build_importance_quads, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->env_imp,
                       c->env_impq, k);
  }
  VX_HIP(c, hipGetLastError());
  VX_HIP(c, hipMemcpyAsync(&c->env_avg_w, c->env_imp + (IMP_FLOATS - 1), sizeof(float),
                           hipMemcpyDeviceToHost, c->stream));
  VX_HIP(c, hipStreamSynchronize(c->stream));
  return VX_OK;
}

int vx_debug_read_importance(VxContext* c, float* out) {
  if (!c || !out) return VX_ERR_INVALID;
  VX_DEV(c);
  if (!c->env_imp) VX_FAIL(c, VX_ERR_INVALID, "vx_debug_read_importance: no environment uploaded");
  VX_HIP(c, hipStreamSynchronize(c->stream));
  VX_HIP(c, hipMemcpy(out, c->env_imp, (size_t)IMP_FLOATS * sizeof(float), hipMemcpyDeviceToHost));
  return VX_OK;
}

int vx_set_params(VxContext* c, const VxParams* p) {
  if (!c || !p) return VX_ERR_INVALID;
  VX_DEV(c);
  if (p->render_mode < VX_MODE_DEFAULT || p->render_mode > VX_MODE_DVR_PHONG)
    VX_FAIL(c, VX_ERR_INVALID, "vx_set_params: unknown render mode %d", p->render_mode);
  if (p->shard_count < 1 || p->shard_rank < 0 || p->shard_rank >= p->shard_count)
    VX_FAIL(c, VX_ERR_INVALID, "vx_set_params: bad shard %d/%d", p->shard_rank, p->shard_count);
  if (p->use_env != 0 && p->use_env != 1) VX_FAIL(c, VX_ERR_INVALID, "vx_set_params: use_env must be 0 or 1");
  if (c->W && ((uint32_t)p->res[0] != c->W || (uint32_t)p->res[1] != c->H))
    VX_FAIL(c, VX_ERR_INVALID, "vx_set_params: res %dx%d differs from vx_resize %ux%u", p->res[0],
            p->res[1], c->W, c->H);
  if ((p->render_mode == VX_MODE_DVR || p->render_mode == VX_MODE_DVR_PHONG) &&
      !(p->dvr_step_voxels > 0.0f))
    VX_FAIL(c, VX_ERR_INVALID, "vx_set_params: dvr_step_voxels must be > 0");
  // the kernels count steps in fp32 (t_k = fma(k, dt, t0)): beyond 2^24 the count stops advancing
  if ((p->render_mode == VX_MODE_DVR || p->render_mode == VX_MODE_DVR_PHONG) &&
      (p->dvr_max_steps < 0 || p->dvr_max_steps > (1 << 24)))
    VX_FAIL(c, VX_ERR_INVALID, "vx_set_params: dvr_max_steps %d outside [0, 2^24]", p->dvr_max_steps);
  bool reshard = !c->has_params || p->shard_count != c->params.shard_count ||
                 p->shard_rank != c->params.shard_rank;
  if (!c->has_params || memcmp(&c->params, p, sizeof(VxParams)) != 0) c->order_builds_left = 2;
  c->params = *p;
  c->has_params = true;
  if (reshard && c->W) {
    VX_HIP(c, hipStreamSynchronize(c->stream));
    return alloc_framebuffers(c);
  }
  return VX_OK;
}

int vx_resize(VxContext* c, uint32_t w, uint32_t h) {
  if (!c) return VX_ERR_INVALID;
  if (w == 0 || h == 0 || w > 16384 || h > 16384) VX_FAIL(c, VX_ERR_INVALID, "vx_resize: bad size %ux%u", w, h);
  VX_HIP(c, hipSetDevice(c->device));
  VX_HIP(c, hipStreamSynchronize(c->stream));
  if ((w != c->W || h != c->H) && c->tile_perm) {  // a dealing order belongs to one image size
    (void)hipFree(c->tile_perm);
    c->tile_perm = nullptr;
    c->tile_perm_n = 0;
  }
  c->W = w;
  c->H = h;
  return alloc_framebuffers(c);
}

static bool tuned_possible(const VxContext* c) {
  // (an early-termination threshold <= 0 -- an epsilon >= 1: every ray ends at its first contributing sample -- is
  // served by render_generic, whose Frame::dvr spells the test as the oracle does; the tuned kernels assume tau >= ert
  // implies a contributing sample)
  return c->dvr_variant != 0 && !c->params.debug_hits && c->tf_len <= TF_LDS_MAX && c->params.dvr_ert_tau > 0.0f;
}
// the LDS-window kernel (vx_dvr_lds.hpp): DVR on the brickf32 layout, Phong wherever brickf32 data is resident
static bool use_lds_kernel(const VxContext* c) {
  if (!tuned_possible(c)) return false;
  if (eff_layout(c) == VX_LAYOUT_BRICKU8)   // the same kernel, staging from the 8-bit bricks
    return (c->params.render_mode == VX_MODE_DVR || c->params.render_mode == VX_MODE_DVR_PHONG) && c->dv.bu != nullptr;
  if (c->params.render_mode == VX_MODE_DVR) return eff_layout(c) == VX_LAYOUT_BRICKF32;
  return c->params.render_mode == VX_MODE_DVR_PHONG && c->dv.bf != nullptr;
}
static bool is_tuned(const VxContext* c) {
  const bool dvr_cq = c->params.render_mode == VX_MODE_DVR && eff_layout(c) == VX_LAYOUT_CELLQUAD;
  return tuned_possible(c) && (dvr_cq || use_lds_kernel(c));
}

static int prepare_render(VxContext* c, dim3& grid) {
  if (!c->has_volume) VX_FAIL(c, VX_ERR_NO_VOLUME, "vx_render_frame: no volume uploaded");
  if (!c->has_params) VX_FAIL(c, VX_ERR_INVALID, "vx_render_frame: vx_set_params not called");
  if (!c->tf) VX_FAIL(c, VX_ERR_INVALID, "vx_render_frame: no transfer function");
  if (!c->slab) VX_FAIL(c, VX_ERR_INVALID, "vx_render_frame: vx_resize not called");
  if ((uint32_t)c->params.res[0] != c->W || (uint32_t)c->params.res[1] != c->H)
    VX_FAIL(c, VX_ERR_INVALID, "vx_render_frame: params.res differs from the framebuffer size");
  if (c->params.use_env > 0 && !c->env_tex)
    VX_FAIL(c, VX_ERR_INVALID, "vx_render_frame: use_env = 1 without vx_upload_environment");
  c->dv.env_tex = c->env_tex;
  c->dv.env_imp = c->env_imp;
  c->dv.env_impq = c->env_impq;
  c->dv.env_avg_w = c->env_avg_w;
  c->dv.env_w = c->env_w;
  c->dv.env_h = c->env_h;
  c->dv.bu_active = (eff_layout(c) == VX_LAYOUT_BRICKU8 && c->dv.bu) ? 1u : 0u;
  {
    // The box the rays are clipped to must lie inside the volume (volume.ts:32-37 clips the volume's own box, so the
    // viewer cannot ask for anything else): the trilinear look-up of the cellquad layout relies on every sample's cell
    // lying in the apron lattice (Frame::trilinear, IN_LATTICE), i.e. on index positions within [-1/2, extent + 1/2).
    // A quarter voxel of that margin is left to the rounding of ray positions near the faces.
    const VxParams& p = c->params;
    for (int corner = 0; corner < 8; ++corner) {
      const float w[3] = {(corner & 1) ? p.volume_aabb_max[0] : p.volume_aabb_min[0],
                          (corner & 2) ? p.volume_aabb_max[1] : p.volume_aabb_min[1],
                          (corner & 4) ? p.volume_aabb_max[2] : p.volume_aabb_min[2]};
      for (int i = 0; i < 3; ++i) {
        const float* m = p.density_transform_inv;
        const float q = fmaf(m[12 + i], 1.0f, fmaf(m[8 + i], w[2], fmaf(m[4 + i], w[1], m[i] * w[0])));
        if (!(q >= -0.25f && q <= (float)c->dv.extent[i] + 0.25f))
          VX_FAIL(c, VX_ERR_INVALID, "vx_render_frame: volume_aabb reaches index %.3f on axis %d, outside the volume [0, %u]: "
                  "the clip box must lie inside the volume's own box (volume.ts:32-37)", (double)q, i, c->dv.extent[i]);
      }
    }
  }
  {
    // the wave-uniform terms of the primary ray (DevVolume::cam_o ...), with the operations of setup_world_ray /
    // to_index (vx_device.hpp): fma chains in the same order, IEEE divisions -- the same bits as on the device
    const VxParams& p = c->params;
    auto mat4 = [](const float* m, float x, float y, float z, float w, float out[4]) {
      for (int i = 0; i < 4; ++i) out[i] = fmaf(m[12 + i], w, fmaf(m[8 + i], z, fmaf(m[4 + i], y, m[i] * x)));
    };
    float cw[4], a[4];
    mat4(p.camera_view_inv, 0.0f, 0.0f, 0.0f, 1.0f, cw);
    for (int i = 0; i < 3; ++i) c->dv.cam_o[i] = cw[i] / cw[3];
    mat4(p.density_transform_inv, c->dv.cam_o[0], c->dv.cam_o[1], c->dv.cam_o[2], 1.0f, a);
    for (int i = 0; i < 3; ++i) c->dv.cam_ipos[i] = a[i];
    c->dv.inv_res[0] = 1.0f / (float)p.res[0];
    c->dv.inv_res[1] = 1.0f / (float)p.res[1];
    // the per-ray divisions a launch constant decides (DevVolume::ray_flags): both shortcuts are exact or not taken
    uint32_t flags = 0;
    const float* vi = p.camera_view_inv;
    if (vi[3] == 0.0f && vi[7] == 0.0f && vi[11] == 0.0f && vi[15] == 1.0f) flags |= RAY_AFFINE_VIEW;
    for (int axis = 0; axis < 2; ++axis) {
      if (c->tex_checked_res[axis] != p.res[axis]) {   // tried once per resolution, not per launch
        const float res = (float)p.res[axis], y = c->dv.inv_res[axis];
        bool same = true;
        for (int px = 0; px < p.res[axis] && same; ++px) {
          const float a = (float)px + 0.5f, q0 = a * y;
          same = fmaf(fmaf(-res, q0, a), y, q0) == a / res;
        }
        c->tex_checked_res[axis] = p.res[axis];
        c->tex_by_reciprocal[axis] = same;
      }
      if (c->tex_by_reciprocal[axis]) flags |= (axis == 0 ? RAY_TEX_BY_RECIPROCAL_X : RAY_TEX_BY_RECIPROCAL_Y);
    }
    if (getenv("VX_RAY_SHORTCUTS") && atoi(getenv("VX_RAY_SHORTCUTS")) == 0) flags = 0;   // diagnostic: the divisions themselves
    c->dv.ray_flags = flags;
  }
  {
    const VxParams& p = c->params;
    bool dvr = p.render_mode == VX_MODE_DVR || p.render_mode == VX_MODE_DVR_PHONG;
    if (dvr && p.dvr_skip_empty && !p.debug_hits) {
      if (c->skip_dirty || !c->dv.skip_bits || c->skip_key[0] != p.volume_density_scale ||
          c->skip_key[1] != p.volume_inv_maj || c->skip_key[2] != p.sample_range[0] ||
          c->skip_key[3] != p.sample_range[1]) {
        int rc = rebuild_skip_mask(c);
        if (rc) return rc;
      }
    }
    if (p.render_mode == VX_MODE_DEFAULT && !p.debug_hits &&
        (c->lmaj_dirty || !c->dv.lmaj || c->lmaj_key[0] != p.volume_density_scale || c->lmaj_key[1] != p.volume_inv_maj ||
         c->lmaj_key[2] != p.volume_maj || c->lmaj_key[3] != p.sample_range[0] || c->lmaj_key[4] != p.sample_range[1])) {
      int rc = rebuild_local_majorants(c);
      if (rc) return rc;
    }
  }
  {
    // the layouts this launch samples, built on first use beside the one the upload built
    int lay = eff_layout(c);
    int rc = VX_OK;
    if (lay == VX_LAYOUT_CELLQUAD) {
      rc = ensure_cellquad(c);
      lay = eff_layout(c);   // AUTO may have stepped down to the resident bricks (memory budget)
    }
    if (!rc && (lay == VX_LAYOUT_BRICKF32 ||
                (c->params.render_mode == VX_MODE_DVR_PHONG && lay == VX_LAYOUT_CELLQUAD && tuned_possible(c))))
      rc = ensure_brickf32(c);
    if (rc) return rc;
  }
  uint32_t groups = (c->tm.tiles_per_shard + 7u) / 8u;
  grid = dim3(groups * 128u);
  return ensure_counters(c, (size_t)grid.x * 4u * 8u);  // x8: the depth-parallel DVR grid
}

// one render-kernel launch into `out` (accumulator or a pipeline result slab)
static hipError_t launch_render(VxContext* c, uint32_t frame_index, float weight, dim3 grid, float4* out,
                                DevCounters* dc, hipStream_t stream) {
  bool tuned = is_tuned(c);
  if (tuned && use_lds_kernel(c)) {
    MultiOut mo{};
    mo.count = 1;
    mo.out[0] = out;
    mo.dc[0] = dc;
    mo.frame[0] = frame_index;
    launch_dvr_lds(c->params, c->dv, c->tf, c->tf_len, mo, weight, c->tm, stream, c->use_order ? c->order : nullptr);
  } else if (tuned) {
    launch_dvr_cq(c->params, c->dv, c->tf, c->tf_len, out, frame_index, weight, c->tm, dc, stream,
                  (c->use_order && !c->dp_active()) ? c->order : nullptr);
  } else {
    MultiOut mo{};
    mo.count = 1;
    mo.out[0] = out;
    mo.dc[0] = dc;
    mo.frame[0] = frame_index;
    launch_generic_mode(c, mo, weight, grid, stream);
  }
  return hipGetLastError();
}

static int take_events(VxContext* c, EventPair& ev) {
  if (c->free_events.empty()) {
    if (c->pending_events.size() >= 4096) drain_events(c);
    if (c->free_events.empty()) {
      VX_HIP(c, hipEventCreate(&ev.a));
      VX_HIP(c, hipEventCreate(&ev.b));
      ev.launches = 1;
      return VX_OK;
    }
  }
  ev = c->free_events.back();
  c->free_events.pop_back();
  ev.launches = 1;
  return VX_OK;
}

int vx_render_frame(VxContext* c, uint32_t frame_index, float sample_weight) {
  if (!c) return VX_ERR_INVALID;
  VX_DEV(c);
  dim3 grid;
  int rc = prepare_render(c, grid);
  if (rc) return rc;
  EventPair ev;
  if ((rc = take_events(c, ev))) return rc;
  VX_HIP(c, hipEventRecord(ev.a, c->stream));
  hipError_t le = launch_render(c, frame_index, sample_weight, grid, c->slab, c->dc, c->stream);
  VX_HIP(c, hipEventRecord(ev.b, c->stream));
  const bool ordered = is_tuned(c) && c->use_order && !(c->dp_active() && !use_lds_kernel(c));
  if (!ordered) c->order_builds_left = 0;
  if (ordered && le == hipSuccess && c->order_builds_left > 0) {
    c->order_builds_left--;
    hipLaunchKernelGGL(build_order, dim3(1), dim3(1024), 0, c->stream, c->dc, c->order, grid.x);
    le = hipGetLastError();
  }
  c->pending_events.push_back(ev);
  c->note_launch(1);
  if (le != hipSuccess) VX_FAIL(c, VX_ERR_DEVICE, "render kernel launch: %s", hipGetErrorString(le));
  return VX_OK;
}

static int ensure_pipes(VxContext* c, int n) {
  size_t waves = c->dc_waves;
  if ((int)c->pipes.size() >= n && c->pipe_quads == c->slab_quads && c->pipe_waves == waves) return VX_OK;
  VX_HIP(c, hipStreamSynchronize(c->stream));
  {
    int rc = fold_counters(c);   // keep what the records of the old slots have counted
    if (rc) return rc;
  }
  for (auto& p : c->pipes) {
    if (p.stream) (void)hipStreamSynchronize(p.stream);
    p.result = nullptr;
    p.dc = nullptr;
  }
  if (c->pipe_result_pool) (void)hipFree(c->pipe_result_pool);
  if (c->pipe_dc_pool) (void)hipFree(c->pipe_dc_pool);
  c->pipe_result_pool = nullptr;
  c->pipe_dc_pool = nullptr;
  if ((int)c->pipes.size() < n) c->pipes.resize(n);
  const size_t ns = c->pipes.size();
  VX_HIP(c, hipMalloc(&c->pipe_result_pool, ns * c->slab_quads * sizeof(float4)));
  VX_HIP(c, hipMalloc(&c->pipe_dc_pool, ns * waves * sizeof(DevCounters)));
  VX_HIP(c, hipMemsetAsync(c->pipe_dc_pool, 0, ns * waves * sizeof(DevCounters), c->stream));   // ordered with the launches
  for (size_t i = 0; i < ns; ++i) {
    auto& p = c->pipes[i];
    // streams only for the rolling-window path of the depth-parallel experiment (<= 8 slots); the multi-frame kernels
    // need none
    if (i < 8 && !p.stream && c->dp_active()) VX_HIP(c, hipStreamCreateWithFlags(&p.stream, hipStreamNonBlocking));
    if (!p.done) VX_HIP(c, hipEventCreateWithFlags(&p.done, hipEventDisableTiming));
    if (!p.merged) VX_HIP(c, hipEventCreateWithFlags(&p.merged, hipEventDisableTiming));
    p.has_merged = false;
    p.result = c->pipe_result_pool + i * c->slab_quads;
    p.dc = c->pipe_dc_pool + i * waves;
  }
  VX_HIP(c, hipStreamSynchronize(c->stream));   // the fills are done before any stream launches into the new slots
  c->pipe_quads = c->slab_quads;
  c->pipe_waves = waves;
  return VX_OK;
}

// `count` accumulation frames first_frame.. with their sample weights, up to `in_flight` of them
// concurrently (frames are independent given their index; only the running mean is ordered, and
// it is applied afterwards, in order, by merge_results -> bit-identical to count vx_render_frame
// calls).  Hides the latency-bound tail of one frame behind the bulk of the next ones.
int vx_render_frames(VxContext* c, uint32_t first_frame, uint32_t count, const float* weights, int in_flight) {
  if (!c || (!weights && count)) return VX_ERR_INVALID;
  VX_DEV(c);
  if (in_flight > MERGE_MAX) in_flight = MERGE_MAX;
  uint32_t done = 0;
  // frames that still refresh the launch order, and the degenerate cases, go one by one
  while (done < count && (in_flight <= 1 || c->order_builds_left > 0 || !c->has_params || c->dc_waves == 0)) {
    int rc = vx_render_frame(c, first_frame + done, weights[done]);
    if (rc) return rc;
    ++done;
  }
  if (done == count) return VX_OK;
  dim3 grid;
  int rc = prepare_render(c, grid);
  if (rc) return rc;
  if ((rc = ensure_pipes(c, in_flight))) return rc;
  EventPair ev;
  if ((rc = take_events(c, ev))) return rc;
  ev.launches = count - done;
  hipError_t le = hipSuccess;
  const bool tuned_lds = is_tuned(c) && use_lds_kernel(c);
  const bool tuned_cq = is_tuned(c) && !tuned_lds && !c->dp_active();
  if (tuned_cq || tuned_lds || !is_tuned(c)) {
    // several frames per launch (see MultiOut): one kernel for up to in_flight frames, then the
    // ordered blend of their results (the tuned cellquad DVR kernel and every render_generic mode)
    const uint32_t nqm = (uint32_t)c->slab_quads;
    c->free_events.push_back(ev);   // per-launch intervals instead of one batch interval
    while (done < count && le == hipSuccess) {
      uint32_t n = count - done < (uint32_t)in_flight ? count - done : (uint32_t)in_flight;
      MultiOut mo{};
      MergeArgs ma{};
      mo.count = n;
      ma.count = n;
      for (uint32_t i = 0; i < n; ++i) {
        mo.out[i] = c->pipes[i].result;
        mo.dc[i] = c->pipes[i].dc;
        mo.frame[i] = first_frame + done + i;
        ma.result[i] = c->pipes[i].result;
        ma.weight[i] = weights[done + i];
      }
      // the LDS-window kernel applies the running mean itself when one wave holds every frame of its pixels: a launch of
      // exactly 8, 16, 32 or 64 frames (MultiOut::fuse; VX_DVR_FUSE=0 keeps the result slabs and the blend kernel)
      const bool fused = c->dvr_fuse &&
                         ((tuned_lds && !c->dvr_shared_window && (n == 8u || n == 16u || n == 32u || n == 64u)) ||
                          (!is_tuned(c) && n == 32u && generic_fusable(c, mo)));
      if (fused) {
        bool zero = false;
        for (uint32_t i = 0; i < n; ++i) {
          mo.weight[i] = ma.weight[i];
          zero = zero || ma.weight[i] == 0.0f;
        }
        mo.fuse = zero ? 2u : 1u;
        mo.accum = c->slab;
      }
      EventPair e2;
      if ((rc = take_events(c, e2))) return rc;
      VX_HIP(c, hipEventRecord(e2.a, c->stream));
      if (tuned_cq)
        launch_dvr_cq_multi(c->params, c->dv, c->tf, c->tf_len, mo, 0.0f, c->tm, c->stream,
                            c->use_order ? c->order : nullptr);
      else if (tuned_lds)
        launch_dvr_lds(c->params, c->dv, c->tf, c->tf_len, mo, 0.0f, c->tm, c->stream,
                       c->use_order ? c->order : nullptr, c->dvr_shared_window);
      else
        launch_generic_mode(c, mo, 0.0f, grid, c->stream);
      le = hipGetLastError();
      VX_HIP(c, hipEventRecord(e2.b, c->stream));   // the render kernel alone; the blend is outside
      c->pending_events.push_back(e2);
      if (le == hipSuccess && !fused) {
        EventPair e3;
        if ((rc = take_events(c, e3))) return rc;
        e3.merge = true;
        VX_HIP(c, hipEventRecord(e3.a, c->stream));
        hipLaunchKernelGGL(merge_results, dim3((nqm + 255) / 256), dim3(256), 0, c->stream, c->slab, ma, nqm);
        le = hipGetLastError();
        VX_HIP(c, hipEventRecord(e3.b, c->stream));
        c->pending_events.push_back(e3);
      }
      done += n;
      c->note_launch(n);
    }
    if (le != hipSuccess) VX_FAIL(c, VX_ERR_DEVICE, "render kernel launch: %s", hipGetErrorString(le));
    return VX_OK;
  }
  // the depth-parallel experiment: rolling window of frames on separate streams
  VX_HIP(c, hipEventRecord(ev.a, c->stream));
  // rolling window: frame f renders on slot f % in_flight as soon as that slot's previous result has
  // been blended; the blends happen on the main stream, in frame order
  const uint32_t P = (uint32_t)(in_flight > 8 ? 8 : in_flight);  // streams; more only costs queue slots
  const uint32_t nq = (uint32_t)c->slab_quads;
  for (; done < count && le == hipSuccess; ++done) {
    auto& p = c->pipes[c->pipe_next % P];
    c->pipe_next = (c->pipe_next + 1) % P;
    if (p.has_merged) VX_HIP(c, hipStreamWaitEvent(p.stream, p.merged, 0));
    le = launch_render(c, first_frame + done, 0.0f, grid, p.result, p.dc, p.stream);  // weight 0: raw result
    VX_HIP(c, hipEventRecord(p.done, p.stream));
    VX_HIP(c, hipStreamWaitEvent(c->stream, p.done, 0));
    if (le == hipSuccess) {
      MergeArgs ma{};
      ma.count = 1;
      ma.result[0] = p.result;
      ma.weight[0] = weights[done];
      hipLaunchKernelGGL(merge_results, dim3((nq + 255) / 256), dim3(256), 0, c->stream, c->slab, ma, nq);
      le = hipGetLastError();
    }
    VX_HIP(c, hipEventRecord(p.merged, c->stream));
    p.has_merged = true;
    c->note_launch(1);
  }
  VX_HIP(c, hipEventRecord(ev.b, c->stream));
  c->pending_events.push_back(ev);
  if (le != hipSuccess) VX_FAIL(c, VX_ERR_DEVICE, "render kernel launch: %s", hipGetErrorString(le));
  return VX_OK;
}

int vx_finish(VxContext* c) {
  if (!c) return VX_ERR_INVALID;
  VX_DEV(c);
  VX_HIP(c, hipStreamSynchronize(c->stream));
  return VX_OK;
}

int vx_detile(VxContext* c, const void* gathered, void* image_out) {
  if (!c || !gathered || !image_out) return VX_ERR_INVALID;
  VX_DEV(c);
  dim3 grid((c->W + 15) / 16, (c->H + 15) / 16);
  hipLaunchKernelGGL(detile, grid, dim3(256), 0, c->stream, (const float4*)gathered, (float4*)image_out, c->tm);
  VX_HIP(c, hipGetLastError());
  return VX_OK;
}

// own slab -> row-major image; for a sharded context the tiles of other shards read as zero
static int own_image(VxContext* c) {
  if (!c->slab) VX_FAIL(c, VX_ERR_INVALID, "no framebuffer (vx_resize not called)");
  if (c->tm.shard_count == 1) return vx_detile(c, c->slab, c->image);
  // build a temporary gathered buffer with only this shard's slab filled
  size_t slab_bytes = c->slab_quads * sizeof(float4);
  void* tmp = nullptr;
  VX_HIP(c, hipMalloc(&tmp, slab_bytes * c->tm.shard_count));
  VX_HIP(c, hipMemsetAsync(tmp, 0, slab_bytes * c->tm.shard_count, c->stream));
  VX_HIP(c, hipMemcpyAsync((char*)tmp + slab_bytes * c->tm.shard_rank, c->slab, slab_bytes,
                           hipMemcpyDeviceToDevice, c->stream));
  int rc = vx_detile(c, tmp, c->image);
  (void)hipStreamSynchronize(c->stream);
  (void)hipFree(tmp);
  return rc;
}

int vx_read_accum(VxContext* c, float* out) {
  if (!c || !out) return VX_ERR_INVALID;
  VX_DEV(c);
  int rc = own_image(c);
  if (rc) return rc;
  VX_HIP(c, hipMemcpyAsync(out, c->image, (size_t)c->W * c->H * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
  VX_HIP(c, hipStreamSynchronize(c->stream));
  return VX_OK;
}

int vx_read_display_scaled(VxContext* c, uint8_t* out, uint32_t ow, uint32_t oh, float exposure, float gamma) {
  if (!c || !out) return VX_ERR_INVALID;
  VX_DEV(c);
  if (ow == 0 || oh == 0 || (uint64_t)ow * oh > (1ull << 28)) {
    c->err = "vx_read_display_scaled: bad canvas size";
    return VX_ERR_INVALID;
  }
  int rc = own_image(c);
  if (rc) return rc;
  uint32_t n = ow * oh;
  if (n > c->display_cap) {
    if (c->display) (void)hipFree(c->display);
    c->display = nullptr;
    c->display_cap = 0;
    VX_HIP(c, hipMalloc(&c->display, (size_t)n * sizeof(uchar4)));
    c->display_cap = n;
  }
  hipLaunchKernelGGL(blit_rgba8, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->image, c->display, c->W, c->H,
                     ow, oh, exposure, gamma);
  VX_HIP(c, hipGetLastError());
  VX_HIP(c, hipMemcpyAsync(out, c->display, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
  VX_HIP(c, hipStreamSynchronize(c->stream));
  return VX_OK;
}

int vx_read_display(VxContext* c, uint8_t* out, float exposure, float gamma) {
  if (!c) return VX_ERR_INVALID;
  return vx_read_display_scaled(c, out, c->W, c->H, exposure, gamma);
}

int vx_probe_tile_costs(VxContext* c, uint32_t* costs, uint32_t n) {
  if (!c || !costs) return VX_ERR_INVALID;
  VX_DEV(c);
  if (!c->has_volume) VX_FAIL(c, VX_ERR_NO_VOLUME, "vx_probe_tile_costs: no volume uploaded");
  if (!c->has_params || !c->tf || !c->W) VX_FAIL(c, VX_ERR_INVALID, "vx_probe_tile_costs: params, transfer function and size first");
  if (n != c->tm.n_tiles) VX_FAIL(c, VX_ERR_INVALID, "vx_probe_tile_costs: the image has %u tiles, not %u", c->tm.n_tiles, n);
  dim3 grid;
  int rc = prepare_render(c, grid);   // skip mask, environment pointers
  if (rc) return rc;
  uint32_t* d = nullptr;
  VX_HIP(c, hipMalloc(&d, (size_t)n * 4));
  const int lay = eff_layout(c);
  if (lay == VX_LAYOUT_BRICKF32)
    hipLaunchKernelGGL((probe_tile_costs<LAYOUT_BF>), dim3(n), dim3(64), 0, c->stream, c->params, c->dv, c->tf, c->tf_len, c->tm, d);
  else if (lay == VX_LAYOUT_CELLQUAD)
    hipLaunchKernelGGL((probe_tile_costs<LAYOUT_CQ>), dim3(n), dim3(64), 0, c->stream, c->params, c->dv, c->tf, c->tf_len, c->tm, d);
  else
    hipLaunchKernelGGL((probe_tile_costs<LAYOUT_REF>), dim3(n), dim3(64), 0, c->stream, c->params, c->dv, c->tf, c->tf_len, c->tm, d);
  hipError_t le = hipGetLastError();
  if (le == hipSuccess) le = hipMemcpyAsync(costs, d, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream);
  if (le == hipSuccess) le = hipStreamSynchronize(c->stream);
  (void)hipFree(d);
  if (le != hipSuccess) VX_FAIL(c, VX_ERR_DEVICE, "vx_probe_tile_costs: %s", hipGetErrorString(le));
  return VX_OK;
}

int vx_set_tile_order(VxContext* c, const uint32_t* perm, uint32_t n) {
  if (!c) return VX_ERR_INVALID;
  VX_DEV(c);
  if (!c->W) VX_FAIL(c, VX_ERR_INVALID, "vx_set_tile_order: vx_resize first");
  VX_HIP(c, hipStreamSynchronize(c->stream));
  for (auto& p : c->pipes)
    if (p.stream) VX_HIP(c, hipStreamSynchronize(p.stream));
  if (perm) {
    if (n != c->tm.n_tiles) VX_FAIL(c, VX_ERR_INVALID, "vx_set_tile_order: the image has %u tiles, not %u", c->tm.n_tiles, n);
    std::vector<uint32_t> both(2 * (size_t)n, 0xffffffffu);
    for (uint32_t pos = 0; pos < n; ++pos) {
      uint32_t t = perm[pos];
      if (t >= n || both[n + t] != 0xffffffffu) VX_FAIL(c, VX_ERR_INVALID, "vx_set_tile_order: not a permutation of the tiles");
      both[pos] = t;
      both[n + t] = pos;
    }
    if (c->tile_perm) (void)hipFree(c->tile_perm);
    c->tile_perm = nullptr;
    c->tile_perm_n = 0;
    VX_HIP(c, hipMalloc(&c->tile_perm, both.size() * 4));
    VX_HIP(c, hipMemcpy(c->tile_perm, both.data(), both.size() * 4, hipMemcpyHostToDevice));
    c->tile_perm_n = n;
  } else {
    if (c->tile_perm) (void)hipFree(c->tile_perm);
    c->tile_perm = nullptr;
    c->tile_perm_n = 0;
  }
  update_tilemap(c);
  if (c->slab) VX_HIP(c, hipMemsetAsync(c->slab, 0, c->slab_quads * sizeof(float4), c->stream));  // other tiles now
  c->order_builds_left = 2;
  return VX_OK;
}

int vx_render_size(VxContext* c, uint32_t* w, uint32_t* h) {
  if (!c) return VX_ERR_INVALID;
  if (w) *w = c->W;
  if (h) *h = c->H;
  return VX_OK;
}

int vx_slab_info(VxContext* c, uint64_t* slab_floats, uint32_t* tiles_per_shard) {
  if (!c) return VX_ERR_INVALID;
  if (slab_floats) *slab_floats = (uint64_t)c->slab_quads * 4u;
  if (tiles_per_shard) *tiles_per_shard = c->tm.tiles_per_shard;
  return VX_OK;
}

int vx_slab_device_ptr(VxContext* c, void** p) {
  if (!c || !p) return VX_ERR_INVALID;
  *p = c->slab;
  return VX_OK;
}

int vx_get_counters(VxContext* c, VxCounters* out) {
  if (!c || !out) return VX_ERR_INVALID;
  VX_DEV(c);
  VX_HIP(c, hipStreamSynchronize(c->stream));
  drain_events(c);
  int rc = fold_counters(c);
  if (rc) return rc;
  out->samples = c->base.samples;
  out->rays = c->base.rays;
  out->pixels = c->base.pixels;
  out->skip_steps = c->base.skip_steps;
  out->grad_samples = c->base.grad_samples;
  out->lane_slots = c->base.lane_slots;
  out->launches = c->launches;
  out->frames = c->frames;
  out->kernel_ms = c->kernel_ms;
  out->last_kernel_ms = c->last_kernel_ms;
  out->gathers = c->base.gathers;
  out->lds_reads = c->base.lds_reads;
  out->merge_ms = c->merge_ms;
  out->min_launch_frames = c->min_launch_frames;
  out->max_launch_frames = c->max_launch_frames;
  out->tf_samples = c->base.tf_samples;
  out->active_lane_slots = c->base.active_lane_slots;
  return VX_OK;
}

int vx_reset_counters(VxContext* c) {
  if (!c) return VX_ERR_INVALID;
  VX_DEV(c);
  VX_HIP(c, hipStreamSynchronize(c->stream));
  drain_events(c);
  {
    int rc = fold_counters(c);   // zeroes every record array (accumulator and pipeline slots)
    if (rc) return rc;
  }
  c->base = VxCounters{};
  c->kernel_ms = c->last_kernel_ms = c->merge_ms = 0.0;
  c->launches = 0;
  c->frames = 0;
  c->min_launch_frames = c->max_launch_frames = 0;
  return VX_OK;
}

int vx_device_info(VxContext* c, char* name, uint32_t cap, uint32_t* cus, uint64_t* hbm) {
  if (!c) return VX_ERR_INVALID;
  if (name && cap) {
    snprintf(name, cap, "%s (%s)", c->prop.name, c->prop.gcnArchName);
  }
  if (cus) *cus = (uint32_t)c->prop.multiProcessorCount;
  if (hbm) *hbm = (uint64_t)c->prop.totalGlobalMem;
  return VX_OK;
}

// test hook: the host-side skip mask for a brick grid / TF / uniforms (pure CPU, no context).
// bits_out must hold ((prod(dims)+31)/32) words; call with bits_out == NULL to get level / dims.
int vx_debug_build_skip_mask(const uint32_t* range_packed, const uint32_t brick_count[3], const float* tf_rgba,
                             uint32_t tf_len, const VxParams* p, uint32_t* bits_out, uint32_t* level_out,
                             uint32_t dims_out[3]) {
  if (!range_packed || !brick_count || !tf_rgba || !tf_len || !p) return VX_ERR_INVALID;
  uint32_t extent[3] = {brick_count[0] * 8u, brick_count[1] * 8u, brick_count[2] * 8u};
  std::vector<uint32_t> bits;
  int level = 1;
  uint32_t md[3];
  compute_skip_mask(*p, range_packed, brick_count, extent, tf_rgba, tf_len, bits, level, md);
  if (level_out) *level_out = (uint32_t)level;
  if (dims_out) { dims_out[0] = md[0]; dims_out[1] = md[1]; dims_out[2] = md[2]; }
  if (bits_out) memcpy(bits_out, bits.data(), bits.size() * 4);
  return VX_OK;
}

// test hook: random.glsl on the device
int vx_debug_rng(VxContext* c, int op, const uint32_t* a, const uint32_t* b, uint32_t n, uint32_t* out) {
  if (!c || !a || !out || n == 0 || n > 65536u || op < 0 || op > 4 || (op == 0 && !b)) return VX_ERR_INVALID;
  VX_DEV(c);
  const uint32_t n_in = op < 2 ? n : 1u;
  uint32_t *da = nullptr, *db = nullptr, *dout = nullptr;
  hipError_t e = hipMalloc(&da, (size_t)n_in * 4);
  if (e == hipSuccess) e = hipMalloc(&db, (size_t)n_in * 4);
  if (e == hipSuccess) e = hipMalloc(&dout, (size_t)n * 4);
  if (e == hipSuccess) e = hipMemcpyAsync(da, a, (size_t)n_in * 4, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess && op == 0) e = hipMemcpyAsync(db, b, (size_t)n_in * 4, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(debug_rng, dim3((n + 255) / 256), dim3(256), 0, c->stream, op, da, db, n, dout);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(out, dout, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(da);
  (void)hipFree(db);
  (void)hipFree(dout);
  if (e != hipSuccess) VX_FAIL(c, VX_ERR_DEVICE, "vx_debug_rng: %s", hipGetErrorString(e));
  return VX_OK;
}

// measurement hook: what the vector ALUs of this device sustain -- clocks (nominal) per wave64 VALU instruction per SIMD
int vx_probe_valu_rate(VxContext* c, double* clk_out, uint32_t* clock_khz_out) {
  if (!c || !clk_out) return VX_ERR_INVALID;
  VX_DEV(c);
  float* sink = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t e = hipMalloc(&sink, 4);
  if (e == hipSuccess) e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  const int cus = c->prop.multiProcessorCount, iters = 4096, blocks = cus * 8;   // 32 waves per CU = 8 per SIMD
  float ms = 0.f;
  for (int rep = 0; rep < 3 && e == hipSuccess; ++rep) {
    e = hipEventRecord(e0, c->stream);
    hipLaunchKernelGGL(probe_valu_rate, dim3(blocks), dim3(256), 0, c->stream, sink, iters, 1.0001f, 0.5f);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipEventRecord(e1, c->stream);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(sink);
  if (e != hipSuccess) VX_FAIL(c, VX_ERR_DEVICE, "vx_probe_valu_rate: %s", hipGetErrorString(e));
  const double khz = (double)c->prop.clockRate;
  const double inst_per_simd = (double)iters * 16.0 * 8.0;   // 8 waves per SIMD, 16 FMAs per iteration each
  *clk_out = (ms * 1e-3 * khz * 1e3) / inst_per_simd;
  if (clock_khz_out) *clock_khz_out = (uint32_t)c->prop.clockRate;
  return VX_OK;
}

// measurement hook: L1 gather rate for a given number of distinct lines per gather instruction
int vx_probe_gather_rate(VxContext* c, uint32_t lines, uint32_t distinct, double* clk_out, uint32_t* clock_khz_out) {
  if (!c || !clk_out || lines < 1u || lines > 64u || distinct < 1u || distinct > lines) return VX_ERR_INVALID;
  VX_DEV(c);
  float4* table = nullptr;
  float* sink = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t e = hipMalloc(&table, 128u * 128u);   // 128 lines = 16 KiB: L1 resident
  if (e == hipSuccess) e = hipMalloc(&sink, 4);
  if (e == hipSuccess) e = hipMemsetAsync(table, 0, 128u * 128u, c->stream);
  if (e == hipSuccess) e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  const int cus = c->prop.multiProcessorCount, iters = 512, blocks = cus * 5;   // 20 waves per CU
  float ms = 0.f;
  for (int rep = 0; rep < 3 && e == hipSuccess; ++rep) {   // the last repetition is the one reported
    e = hipEventRecord(e0, c->stream);
    hipLaunchKernelGGL(probe_gather_rate, dim3(blocks), dim3(256), 0, c->stream, table, lines, distinct, iters, sink);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipEventRecord(e1, c->stream);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(table);
  (void)hipFree(sink);
  if (e != hipSuccess) VX_FAIL(c, VX_ERR_DEVICE, "vx_probe_gather_rate: %s", hipGetErrorString(e));
  const double khz = (double)c->prop.clockRate;
  const double gathers_per_cu = (double)blocks * 4.0 * iters * 8.0 / cus;
  *clk_out = (ms * 1e-3 * khz * 1e3) / gathers_per_cu;
  if (clock_khz_out) *clock_khz_out = (uint32_t)c->prop.clockRate;
  return VX_OK;
}

// measurement hook: distinct lines per gather of the tuned cellquad DVR march (one frame, nothing is stored)
int vx_probe_gather_spread(VxContext* c, uint32_t frame_index, uint64_t out3[3]) {
  if (!c || !out3) return VX_ERR_INVALID;
  VX_DEV(c);
  dim3 grid;
  int rc = prepare_render(c, grid);
  if (rc) return rc;
  if (!(is_tuned(c) && eff_layout(c) == VX_LAYOUT_CELLQUAD && c->params.render_mode == VX_MODE_DVR))
    VX_FAIL(c, VX_ERR_INVALID, "vx_probe_gather_spread: needs render_mode dvr on the cellquad layout");
  const size_t waves = (size_t)grid.x * 4u;
  DevCounters* d = nullptr;
  VX_HIP(c, hipMalloc(&d, waves * sizeof(DevCounters)));
  hipError_t e = hipMemsetAsync(d, 0, waves * sizeof(DevCounters), c->stream);
  MultiOut mo{};
  mo.count = 1;
  mo.out[0] = c->slab;   // never written by the probe build
  mo.dc[0] = d;
  mo.frame[0] = frame_index;
  if (e == hipSuccess) {
    launch_dvr_cq_multi(c->params, c->dv, c->tf, c->tf_len, mo, 0.0f, c->tm, c->stream, nullptr, true);
    e = hipGetLastError();
  }
  std::vector<DevCounters> h(waves);
  if (e == hipSuccess) e = hipMemcpyAsync(h.data(), d, waves * sizeof(DevCounters), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(d);
  if (e != hipSuccess) VX_FAIL(c, VX_ERR_DEVICE, "vx_probe_gather_spread: %s", hipGetErrorString(e));
  out3[0] = out3[1] = out3[2] = 0;
  for (const auto& w : h) {
    out3[0] += w.gathers / 2u;   // q0 gathers (q1 repeats the pattern one slice further)
    out3[1] += w.rays;           // wave-wide distinct lines, summed over the q0 gathers
    out3[2] += w.pixels;         // look-ups of the 16 lane quads, summed over the q0 gathers
  }
  return VX_OK;
}

// test hook (not part of the reference boundary): the device's unorm8 decode table
int vx_debug_unorm_table(VxContext* c, float* out256) {
  if (!c || !out256) return VX_ERR_INVALID;
  VX_DEV(c);
  float* d = nullptr;
  VX_HIP(c, hipMalloc(&d, 256 * sizeof(float)));
  hipLaunchKernelGGL(unorm_table, dim3(1), dim3(256), 0, c->stream, d);
  VX_HIP(c, hipMemcpyAsync(out256, d, 256 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  VX_HIP(c, hipStreamSynchronize(c->stream));
  (void)hipFree(d);
  return VX_OK;
}

}  // extern "C"
