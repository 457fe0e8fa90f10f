// vx_kernels.hpp -- kernels: generic per-pixel driver (all modes, both layouts), layout
// conversion, de-tiling, display pass.
#pragma once
#include "vx_modes.hpp"

namespace vx {

// image <-> slab mapping: 64x64-pixel sharding tiles dealt
// round-robin to shards; inside a tile 64 wave-tiles of 8x8 pixels in Morton order; the
// accumulator ("slab") is tile-major so that one wave owns 1 KiB of contiguous pixels.
struct TileMap {
  uint32_t W, H;
  uint32_t tiles_x, tiles_y, n_tiles;
  uint32_t shard_rank, shard_count, tiles_per_shard;
  // optional dealing order of the tiles (vx_set_tile_order): position pos = lt * shard_count + rank holds
  // tile perm[pos]; inv is the inverse.  nullptr: pos == tile id (tiles dealt round-robin in row-major order).
  const uint32_t* perm;
  const uint32_t* inv;
};
VXD uint32_t tile_at(const TileMap& tm, uint32_t lt) {  // tile id of local tile lt, >= n_tiles: padding
  uint32_t pos = lt * tm.shard_count + tm.shard_rank;
  return (tm.perm && pos < tm.n_tiles) ? tm.perm[pos] : pos;
}

VXD uint32_t morton_x(uint32_t m) {  // compact even bits of a 6-bit code
  return (m & 1u) | ((m >> 1) & 2u) | ((m >> 2) & 4u);
}

// block b -> (local tile, first wave-tile).  Blocks are dealt to the 8 XCDs round-robin
// (b % 8 = group sharing one XCD's L2): keep the 16 blocks of one 64x64 tile on one XCD and
// hand whole tiles round-robin to the XCDs so that each L2 sees a compact piece of the
// volume and the XCDs stay balanced.
VXD bool block_to_tile(uint32_t b, const TileMap& tm, uint32_t& lt, uint32_t& sub) {
  lt = (b >> 7) * 8u + (b & 7u);
  sub = (b >> 3) & 15u;
  return lt < tm.tiles_per_shard;
}

VXD bool wave_pixel(const TileMap& tm, uint32_t lt, uint32_t wt, uint32_t lane, int& px, int& py,
                    uint32_t& slab_index) {
  uint32_t t = tile_at(tm, lt);
  // hardware lane -> pixel of the 8x8 wave tile in Morton order: the address coalescer merges the
  // gathers of 4 consecutive lanes, and a 2x2 pixel block touches fewer cache lines than a 4x1 row.
  // The slab keeps its row-major slot order (slot = y*8 + x), so only the lane that owns a pixel moves.
  uint32_t lx = morton_x(lane), ly = morton_x(lane >> 1);
  slab_index = (lt * 64u + wt) * 64u + ly * 8u + lx;
  if (t >= tm.n_tiles) return false;
  uint32_t tx = t % tm.tiles_x, ty = t / tm.tiles_x;
  px = (int)(tx * 64u + morton_x(wt) * 8u + lx);
  py = (int)(ty * 64u + morton_x(wt >> 1) * 8u + ly);
  return (uint32_t)px < tm.W && (uint32_t)py < tm.H;
}

// sum over the 64 lanes of the wave, in every lane: a DPP scan inside the rows of 16 lanes, two row broadcasts, one
// v_readlane (6 vector instructions; the __shfl_down form went through ds_bpermute: ~30 instructions and six LDS round
// trips per sum).  Every launch of this library runs whole waves (256-thread workgroups, no lane returns early).
VXD uint32_t wave_sum(uint32_t x) {
  int v = (int)x;
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8  -> lane 15 of each row = row sum
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
  return (uint32_t)__builtin_amdgcn_readlane(v, 63);
}

// Work counters: one record per wave of the launch grid, owned by that wave and updated with a
// plain read-modify-write (launches on a stream are ordered, so no atomics are needed).
// Atomics on shared words were measured to cost ~1.1 ms per 1080p frame (32768 waves x 4
// same-line atomics at ~88 per microsecond) -- three times the march itself.
struct DevCounters {
  unsigned long long samples, slots;
  uint32_t rays, pixels, skips, grads;
  uint32_t last_slots;  // lane slots of the most recent launch: the cost fed back to build_order
  uint32_t gathers;     // 16-byte-per-lane gather wave instructions issued (tuned DVR kernels)
  uint32_t lds_reads;   // LDS tap-read wave instructions (LDS-tile kernels)
  uint32_t tf;          // samples inside the sample range (LUT fetched)
  uint32_t active;      // lane slots that did work (event-batched path kernels)
  uint32_t pad;
};

__global__ void zero_totals(unsigned long long* sums) { sums[threadIdx.x] = 0ull; }   // 10 totals (fold_records)

// sums the per-wave records of one launch slot into eight 64-bit totals and zeroes them (vx_get_counters /
// vx_reset_counters: 64 bytes cross PCIe instead of every record)
__global__ __launch_bounds__(256) void fold_records(DevCounters* __restrict__ recs, size_t n,
                                                    unsigned long long* __restrict__ sums) {
  unsigned long long s[10] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
  for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u) {
    const DevCounters r = recs[i];
    s[0] += r.samples; s[1] += r.slots; s[2] += r.rays; s[3] += r.pixels;
    s[4] += r.skips; s[5] += r.grads; s[6] += r.gathers; s[7] += r.lds_reads; s[8] += r.tf; s[9] += r.active;
    recs[i] = DevCounters{};
  }
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    unsigned long long x = s[k];
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
    if ((threadIdx.x & 63u) == 0u && x != 0ull) atomicAdd(&sums[k], x);
  }
}

// `block` = logical block id (identical to blockIdx.x unless the launch is permuted by `order`)
VXD void add_counts(DevCounters* dc, uint32_t samples, uint32_t rays, uint32_t pixels, uint32_t skips,
                    uint32_t grads, uint32_t slots, uint32_t block = 0xffffffffu, uint32_t gathers = 0u,
                    uint32_t lds_reads = 0u, uint32_t tf = 0u, uint32_t active = 0u) {
  if ((threadIdx.x & 63u) == 0) {
    if (block == 0xffffffffu) block = blockIdx.x;
    DevCounters* w = dc + (block * (blockDim.x >> 6) + (threadIdx.x >> 6));
    DevCounters c = *w;
    c.last_slots = slots;
    c.gathers += gathers;
    c.lds_reads += lds_reads;
    c.tf += tf;
    c.active += active;
    c.samples += samples;
    c.slots += slots;
    c.rays += rays;
    c.pixels += pixels;
    c.skips += skips;
    c.grads += grads;
    *w = c;
  }
}

// max over the 64 lanes of the wave, in every lane (whole waves, like wave_sum)
VXD uint32_t wave_max_u32(uint32_t x) {
  int v = (int)x;   // trip counts: far below 2^31
  auto mx = [](int a, int b) { return a > b ? a : b; };
  v = mx(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false));
  v = mx(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false));
  v = mx(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false));
  v = mx(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false));
  v = mx(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false));
  v = mx(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false));
  return (uint32_t)__builtin_amdgcn_readlane(v, 63);
}

// The path-traced modes also report their march lane slots (VxCounters.lane_slots / active_lane_slots): a wave runs each of its
// march loops until its slowest lane is through, so with one primary and one shadow segment per pixel (bounces 1) the loops
// take max(it_p) + max(it_s) trips of 64 lane slots, of which sum(it_p + it_s) do work; with more bounces the maxima of the
// per-lane totals bound the trips from below (the utilisation from above).
VXD void flush_counts(DevCounters* dc, const Counts& c, uint32_t pixels, uint32_t block = 0xffffffffu) {
  uint32_t s = wave_sum(c.samples), r = wave_sum(c.rays), k = wave_sum(c.skips),
           g = wave_sum(c.grads), px = wave_sum(pixels), t = wave_sum(c.tf);
  const uint32_t work = wave_sum(c.it_p + c.it_s);
  const uint32_t slots = work ? 64u * (wave_max_u32(c.it_p) + wave_max_u32(c.it_s)) : 0u;
  add_counts(dc, s, r, px, k, g, slots, block, 0u, 0u, t, work);
}

constexpr uint32_t TF_LDS_MAX = 2048;  // entries staged in LDS (32 KiB); longer LUTs stay in L1/L2

// Several independent accumulation frames in ONE launch: launch slot s renders frame slot s % count
// of logical block s / count (through `order` in the tuned DVR kernel), so the latency-bound tail of a
// frame is filled by the blocks of the other frames and paid once per launch.  count == 1 is the plain
// per-frame launch.  merge_results then applies the running-mean blends in frame order.
constexpr int MERGE_MAX = 64;  // frames per launch
struct MultiOut {
  float4* out[MERGE_MAX];
  DevCounters* dc[MERGE_MAX];
  uint32_t frame[MERGE_MAX];
  uint32_t count;
  // Round 4, behind the fields lane_frame_slot addresses (their offsets do not move): the running mean applied IN the render
  // kernel.  fuse != 0 (the launcher sets it for a launch of exactly 8, 16, 32 or 64 frames of the LDS-window DVR kernel, of 32
  // frames of render_generic): a wave holds every frame of its 8, 4, 2 (or 1) pixels, so it folds their results in frame order into `accum` itself -- fragment.frag:158
  // with weight[k] for frame slot k, exactly what merge_results does -- and neither the per-frame result slabs nor the blend
  // kernel are touched.  fuse == 2: some weight of the launch is 0 (the previous value is then dropped: merge_results'
  // `w != 0 ? acc : 0`), the fold tests each weight; fuse == 1: none is.
  uint32_t fuse;
  float4* accum;
  float weight[MERGE_MAX];
};

// launch slot p of a multi-frame launch -> (frame slot, block slot).  Default: frame slots of one block are consecutive
// launch slots (p % count).  -DVX_SLOT_XCD: launch slots are dealt to the 8 XCDs round-robin (p % 8 names the XCD
// class), so let the slots of one class walk the blocks of that class with all their frame slots back to back -- the
// tiles of a class then stay on one XCD's L2 across frames, as block_to_tile arranges for a single frame.  Measured on
// config 3 (ms per frame at 8 / 16 / 24 frames per launch): default 0.437 / 0.410 / 0.393, VX_SLOT_XCD 0.432 / 0.413 / 0.405.
VXD void multi_slot(uint32_t p, uint32_t count, uint32_t& fslot, uint32_t& bslot) {
  if (count <= 1u) { fslot = 0u; bslot = p; return; }
#ifdef VX_SLOT_XCD
  const uint32_t x = p & 7u, i = p >> 3;
  fslot = i % count;
  bslot = (i / count) * 8u + x;
#else
  fslot = p % count;
  bslot = p / count;
#endif
}

// Lanes = pixels x frames (multi-frame launches).  The frame slots of a launch are cut into groups of 2^sh consecutive
// slots, the largest power of two first (20 slots, MAXSH 3: 8 + 8 + 4); the 2^sh waves that would each render the 64
// pixels of one 8x8 wave tile for one slot of the group render 64 >> sh pixels of it for all slots of the group instead:
// hardware lane l of the wave at slot `fslot` takes frame slot base + (l >> (6 - sh)) and Morton pixel
// ((fslot - base) << (6 - sh)) + (l & ((64 >> sh) - 1)).  Every (pixel, frame) of the launch is rendered exactly once,
// by the same arithmetic as before (its seed is tea(pixel, frame), fragment.frag:143), and stored to its frame's slab:
// results are bit-identical; what changes is that the rays of a wave are 64 >> sh neighbouring pixels seen 2^sh times
// under different jitter -- a beam a few voxels wide that the lanes walk in step.
template <int MAXSH>
VXD uint32_t frame_group(uint32_t fslot, uint32_t count, uint32_t& base) {   // wave uniform; returns sh
  base = 0u;
#pragma unroll
  for (int sh = MAXSH; sh > 0; --sh) {
    const uint32_t full = (count - base) >> sh;
    if (fslot < base + (full << sh)) {
      base += ((fslot - base) >> sh) << sh;
      return (uint32_t)sh;
    }
    base += full << sh;
  }
  base = fslot;
  return 0u;
}
// the slab pointer and frame index of a lane's own frame slot, read from the kernel-argument segment (indexing the
// by-value MultiOut with a lane-varying index would copy all of it to scratch).  Both render kernels that use it start
// their argument list with (VxParams, DevVolume, const float4*, uint32_t, MultiOut).
VXD float4* lane_frame_slot(uint32_t my_fslot, uint32_t& frame) {
  struct KArgs { VxParams p; DevVolume v; const float4* tf; uint32_t tf_len; MultiOut mo; };
  typedef const char __attribute__((address_space(4)))* KPtr;
  const KPtr ka = (KPtr)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(KArgs, mo);
  frame = ((const uint32_t __attribute__((address_space(4)))*)(ka + offsetof(MultiOut, frame)))[my_fslot];
  return ((float4* const __attribute__((address_space(4)))*)(ka + offsetof(MultiOut, out)))[my_fslot];
}

// MultiOut::fuse: the running mean of a launch's frames applied by the wave that holds every frame of its pixels (lane l =
// frame slot l >> psh of pixel l & (npx - 1); sh = 3 .. 6: 8 pixels x 8 frames .. 1 pixel x 64 frames).  The wave parks its
// 64 results and the launch's weights {w, 1 - w} in `fold` (320 floats of wave-private LDS), then lanes 0 .. 3 npx - 1 each
// fold one colour channel of one pixel through the frame slots in order: acc = fma(1 - w, r, w * (w != 0 ? acc : 0)) --
// merge_results, operation for operation, on the value the unfused path writes to the frame's result slab
// (fma(1 - 0, L, 0 * 0)).  fuse == 1: no weight of the launch is 0, the test on w is compiled out of the loop.
VXD void fold_frames(float* fold, uint32_t lane, V3 L, bool in_image, uint32_t si, float4* accum, uint32_t fuse, uint32_t sh) {
  const uint32_t psh = 6u - sh, npx = 1u << psh, nfr = 1u << sh;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // whatever the wave read from this LDS before is done
  __builtin_amdgcn_wave_barrier();
  fold[lane * 3u + 0u] = fma_(1.0f, L.x, 0.0f);
  fold[lane * 3u + 1u] = fma_(1.0f, L.y, 0.0f);
  fold[lane * 3u + 2u] = fma_(1.0f, L.z, 0.0f);
  if (lane < nfr) {
    typedef const char __attribute__((address_space(4)))* KPtr;
    struct KArgs { VxParams p; DevVolume v; const float4* tf; uint32_t tf_len; MultiOut mo; };
    const KPtr ka = (KPtr)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(KArgs, mo) + offsetof(MultiOut, weight);
    const float w = ((const float __attribute__((address_space(4)))*)ka)[lane];
    fold[192u + 2u * lane] = w;
    fold[193u + 2u * lane] = 1.0f - w;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const unsigned long long inimg = ballot(in_image);
  const uint32_t fp = lane / 3u, ch = lane - 3u * fp;                     // pixel and channel this lane folds
  // the accumulator slot of pixel fp: hardware lane fp holds frame slot 0 of that pixel (fp < npx <= 8)
  const uint32_t si_f = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((fp & 7u) << 2), (int)si);
  const bool folds = lane < 3u * npx && ((inimg >> fp) & 1ull);
  if (folds) {
    float* const px4 = reinterpret_cast<float*>(accum + si_f);
    float acc = px4[ch];
    const float* rp = fold + fp * 3u + ch;
    if (fuse == 1u) {
#pragma unroll 4
      for (uint32_t k = 0; k < nfr; ++k) {
        const float w = fold[192u + 2u * k], omw = fold[193u + 2u * k];
        acc = fma_(omw, rp[(k << psh) * 3u], w * acc);
      }
    } else {
      for (uint32_t k = 0; k < nfr; ++k) {
        const float w = fold[192u + 2u * k], omw = fold[193u + 2u * k];
        const float prev = w != 0.0f ? acc : 0.0f;
        acc = fma_(omw, rp[(k << psh) * 3u], w * prev);
      }
    }
    px4[ch] = acc;
    if (ch == 0u) px4[3] = 1.0f;
  }
}

// vx_create's check of lane_frame_slot: a kernel with the render kernels' argument list reads every frame slot through the
// kernel-argument segment and compares with the by-value struct; *bad counts the slots that differ (0 unless the
// compiler's argument layout ever stops matching struct KArgs)
__global__ __launch_bounds__(64) void check_lane_frame_slot(const VxParams p, const DevVolume v, const float4* __restrict__ tf,
                                                            uint32_t tf_len, const MultiOut mo, uint32_t* __restrict__ bad) {
  const uint32_t lane = threadIdx.x;
  uint32_t frame = 0;
  float4* slab = lane_frame_slot(lane, frame);   // lane-varying index: a vector load from the argument segment
  uint32_t differs = 0;
  for (uint32_t i = 0; i < (uint32_t)MERGE_MAX; ++i) {   // uniform index: the compiler's own view of the argument
    const uint32_t f = (uint32_t)__builtin_amdgcn_readlane((int)frame, (int)i);
    const uint64_t s = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)((uint64_t)(uintptr_t)slab >> 32), (int)i) << 32) |
                       (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uintptr_t)slab, (int)i);
    differs += (f != mo.frame[i]) | (s != (uint64_t)(uintptr_t)mo.out[i]);
  }
  if (lane == 0) *bad = differs + (p.res[0] != 0x1234) + (v.extent[0] != 0x5678u) + (tf_len != 0x9abcu) + (tf != nullptr);
}

// Occupancy the register allocator is asked for, per mode.  The path-traced modes are bound by latency and
// divergence (about 37 % of the lane slots of an issued VALU instruction do work), so more resident waves
// pay even at the price of a few spills: measured on config 3 (tools/variant_modes.sh, ms per frame, 32 frames
// per launch, 5 / 6 / 8 waves per SIMD): default 0.705 / 0.635 / 0.568, default with 3 bounces 2.31 / 2.06 /
// 1.79, no_dda 0.804 / 0.744 / 0.726, raymarch 1.070 / 1.062 / 1.048, dvr_phong 0.831 / 0.828 / 0.762.
// (-DVX_W_DEFAULT=.. -DVX_W_NO_DDA=.. -DVX_W_RAYMARCH=.. -DVX_W_PHONG=.. through EXTRA rebuilds a variant.)
#ifndef VX_W_DEFAULT
#define VX_W_DEFAULT 8
#define VX_W_NO_DDA 8
// raymarch (91 % useful lanes, bound by its instruction count) is no longer forced: at 5 waves it needs no scratch and
// runs 0.699 ms per frame against 0.704 / 0.721 / 0.732 at 6 / 7 / 8 (round 3, after the division-free reservoir)
#define VX_W_RAYMARCH 5
// dvr_phong has its tuned kernel (vx_dvr_lds.hpp); the generic form only serves the reference / cellquad layouts.
// Round 2 saw wrong pixels from the REFERENCE-layout build at 8 waves: spill code placed ahead of an exec restore
// (DESIGN.md section 5.3).  The taps are straight-line code now and every build is linted for that placement
// (tools/check_exec_prologue.py, run by the Makefile): 8 waves again.
#define VX_W_PHONG 8
#endif
constexpr int generic_min_waves(int mode) {
  return mode == VX_MODE_DEFAULT ? VX_W_DEFAULT : mode == VX_MODE_DVR_PHONG ? VX_W_PHONG
         : mode == VX_MODE_NO_DDA ? VX_W_NO_DDA : mode == VX_MODE_RAYMARCH ? VX_W_RAYMARCH : 1;
}

// One thread per pixel, one wave per 8x8 pixel tile, 4 waves (16x16 pixels) per block.
template <int MODE, int LAYOUT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(generic_min_waves(MODE), 8))) void render_generic(const VxParams p, const DevVolume v,
                                                       const float4* __restrict__ tf_global,
                                                       uint32_t tf_len, const MultiOut mo, float weight,
                                                       const TileMap tm) {
  extern __shared__ float4 tf_lds[];
  TfView tf;
  tf.len = tf_len;
  tf.lenf = (float)tf_len;
  if (tf_len <= TF_LDS_MAX) {
    for (uint32_t i = threadIdx.x; i < tf_len; i += blockDim.x) tf_lds[i] = tf_global[i];
    __syncthreads();
    tf.lut = tf_lds;
    tf.in_lds = true;
  } else {
    tf.lut = tf_global;
    tf.in_lds = false;
  }
  uint32_t fslot, blk;
  multi_slot(blockIdx.x, mo.count, fslot, blk);
  float4* __restrict__ slab = mo.out[fslot];
  DevCounters* __restrict__ dc = mo.dc[fslot];
  const uint32_t frame = mo.frame[fslot];
  uint32_t lt, sub;
  if (!block_to_tile(blk, tm, lt, sub)) return;
  uint32_t wt = sub * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
  uint32_t my_frame = frame;
  // Path-traced modes in a multi-frame launch: lanes = pixels x frames (frame_group above), 8 pixels x 8 frames per wave.
  // The free-flight samples of one wave then fall along a beam 8 pixels wide instead of 64, and the vector L1 -- whose
  // misses in flight bound these modes -- serves more of them: no_dda 0.684 -> 0.622 ms per frame.
  const uint32_t fuse = (MODE <= VX_MODE_RAYMARCH) ? mo.fuse : 0u;   // wave uniform; the launcher sets it for 32-frame launches only
  if (fuse != 0u) {
    // the running mean applied here (MultiOut::fuse): the workgroup at slot r of the 32 takes wave tile r >> 3 of its block
    // position and pixel octet r & 7 as below, but each of its waves takes 2 of the 8 pixels for ALL 32 frames -- the
    // workgroup's beam is the same 8 pixels x 32 frames, and a wave holds every frame of its pixels
    const uint32_t r = fslot & 31u, w = threadIdx.x >> 6;
    wt = sub * 4u + (r >> 3);
    const uint32_t my_fslot = (fslot - r) + (lane >> 1);
    lane = ((r & 7u) << 3) + 2u * w + (lane & 1u);
    slab = lane_frame_slot(my_fslot, my_frame);
  } else if (MODE <= VX_MODE_RAYMARCH && mo.count > 1u) {   // wave uniform
    uint32_t base;
    const uint32_t sh = frame_group<3>(fslot, mo.count, base);
    if (sh != 0u) {
      const uint32_t psh = 6u - sh;
      uint32_t j = fslot - base;
      uint32_t my_fslot = base + (lane >> psh);
      // with 32 frames to share out, the four waves of a workgroup take the SAME 8 pixels (8 frames each) instead of four
      // pixel groups of four tiles: 8 beams per CU instead of 32 (no_dda 0.623 -> 0.606 ms per frame)
      if ((mo.count & 31u) == 0u) {
        const uint32_t r = fslot & 31u;
        wt = sub * 4u + (r >> 3);
        j = r & 7u;
        my_fslot = (fslot - r) + (threadIdx.x >> 6) * 8u + (lane >> 3);
      }
      lane = (j << psh) + (lane & ((1u << psh) - 1u));
      slab = lane_frame_slot(my_fslot, my_frame);
    }
  }
  int px, py;
  uint32_t si;
  bool active = wave_pixel(tm, lt, wt, lane, px, py, si);
  Counts c{0, 0, 0, 0, 0};
  if (fuse != 0u) {
    V3 L = v3(0.f, 0.f, 0.f);
    if (active) {
      Frame<LAYOUT> f{p, v, tf, c};
      const float4 r = f.template shade_pixel<MODE>(px, py, my_frame);
      L = v3(r.x, r.y, r.z);
    }
    // wave-private scratch behind the transfer function (launch_generic adds it to the LDS size of a fused launch)
    float* const fold = reinterpret_cast<float*>(tf_lds + (tf_len <= TF_LDS_MAX ? tf_len : 0u)) + (threadIdx.x >> 6) * 320u;
    fold_frames(fold, threadIdx.x & 63u, L, active, si, mo.accum, fuse, 5u);
    flush_counts(dc, c, active ? 1u : 0u, blk);
    return;
  }
  if (active) {
    Frame<LAYOUT> f{p, v, tf, c};
    float4 r = f.template shade_pixel<MODE>(px, py, my_frame);
    float4 prev = make_float4(0.f, 0.f, 0.f, 0.f);
    if (weight != 0.0f) prev = slab[si];
    // fragment.frag:158  out = (w*prev + (1-w)*result).rgb, alpha 1
    float4 o;
    o.x = fma_(1.0f - weight, r.x, weight * prev.x);
    o.y = fma_(1.0f - weight, r.y, weight * prev.y);
    o.z = fma_(1.0f - weight, r.z, weight * prev.z);
    o.w = 1.0f;
    slab[si] = o;
  }
  flush_counts(dc, c, active ? 1u : 0u, blk);
}

// Cost probe for the tile dealing order (vx_probe_tile_costs): one wave per 64x64 tile of the WHOLE image
// whatever the shard, its lanes on the 8x8 grid of pixels 8 apart; cost = DVR samples (+ skipped steps / 8) of
// those 64 primary rays at frame 0.  Every rank computes the same numbers, so every rank derives the same order.
template <int LAYOUT>
__global__ __launch_bounds__(64) void probe_tile_costs(const VxParams p, const DevVolume v,
                                                        const float4* __restrict__ tf_global, uint32_t tf_len,
                                                        const TileMap tm, uint32_t* __restrict__ costs) {
  TfView tf;
  tf.lut = tf_global;
  tf.in_lds = false;
  tf.len = tf_len;
  tf.lenf = (float)tf_len;
  const uint32_t t = blockIdx.x, lane = threadIdx.x & 63u;
  const int px = (int)((t % tm.tiles_x) * 64u + (lane & 7u) * 8u + 4u), py = (int)((t / tm.tiles_x) * 64u + (lane >> 3) * 8u + 4u);
  Counts c{0, 0, 0, 0, 0};
  if ((uint32_t)px < tm.W && (uint32_t)py < tm.H) {
    Frame<LAYOUT> f{p, v, tf, c};
    (void)f.template shade_pixel<VX_MODE_DVR>(px, py, 0u);
  }
  // a step in an empty macro cell is tested or jumped over, not evaluated: about an eighth of a sample
  uint32_t s = wave_sum(c.samples) + (wave_sum(c.skips) >> 3);
  if (lane == 0) costs[t] = s;
}

// ---- longest-processing-time-first launch order, fed back from the previous frame ------------
// The march is latency-bound on the longest rays: one wave that walks the whole volume needs
// ~0.35 ms even on an idle GPU, so it must start first.  Progressive rendering repeats the same
// view, so the per-wave step counts of frame f are the cost estimate for frame f+1.  Blocks keep
// their XCD class (b % 8, see block_to_tile): within each class they are ranked by the longest of
// their four waves (counting sort, 64 buckets of 32 steps), and launch slot pos*8+x runs the
// pos-th longest block of class x.  Any permutation yields the same image; a stale order after a
// camera change only costs speed for one frame.  One workgroup, runs after every DVR frame.
__global__ __launch_bounds__(1024) void build_order(const DevCounters* __restrict__ dc,
                                                     uint32_t* __restrict__ order, uint32_t n_blocks) {
  __shared__ uint32_t hist[8][64];
  for (uint32_t i = threadIdx.x; i < 8 * 64; i += blockDim.x) (&hist[0][0])[i] = 0;
  __syncthreads();
  auto bucket_of = [&](uint32_t b) {
    const DevCounters* w = dc + (size_t)b * 4u;
    uint32_t c = max(max(w[0].last_slots, w[1].last_slots), max(w[2].last_slots, w[3].last_slots));
    uint32_t k = c >> 11;  // slots / 64 lanes / 32 steps
    return 63u - (k > 63u ? 63u : k);  // bucket 0 = longest
  };
  for (uint32_t b = threadIdx.x; b < n_blocks; b += blockDim.x) atomicAdd(&hist[b & 7u][bucket_of(b)], 1u);
  __syncthreads();
  if (threadIdx.x < 8) {  // exclusive prefix per class
    uint32_t acc = 0;
    for (int k = 0; k < 64; ++k) {
      uint32_t h = hist[threadIdx.x][k];
      hist[threadIdx.x][k] = acc;
      acc += h;
    }
  }
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < n_blocks; b += blockDim.x) {
    uint32_t x = b & 7u;
    uint32_t pos = atomicAdd(&hist[x][bucket_of(b)], 1u);
    order[pos * 8u + x] = b;
  }
}

// ---- reference layout -> cellquad (runs once per upload) -------------------------------
// one thread per stored quad: brick' b, slice lz in [0,9), cell (ly,lx).
// [first, end) = the quads of a range of apron-brick z layers (the upload builds the layers of an atlas chunk
// while the next chunk is still crossing PCIe)
__global__ __launch_bounds__(256) void build_cellquad(const DevVolume v, float4* __restrict__ out,
                                                       uint64_t first, uint64_t end) {
  uint64_t i = first + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= end) return;
  uint32_t q = (uint32_t)(i % CQ_BRICK_QUADS);
  uint64_t b = i / CQ_BRICK_QUADS;
  uint32_t bx = (uint32_t)(b % v.cq_bc[0]);
  uint32_t by = (uint32_t)((b / v.cq_bc[0]) % v.cq_bc[1]);
  uint32_t bz = (uint32_t)(b / ((uint64_t)v.cq_bc[0] * v.cq_bc[1]));
  // inverse of cq_cell
  uint32_t lz = q >> 6, ly = (((q >> 4) & 3u) << 1) | (q & 1u), lx = (q >> 1) & 7u;
  // voxel of local (l) in apron brick b: 8b - 1 + l
  int x = (int)(bx * 8u + lx) - 1, y = (int)(by * 8u + ly) - 1, z = (int)(bz * 8u + lz) - 1;
  float4 o;
  o.x = lookup_density_brick(v, x, y, z);
  o.y = lookup_density_brick(v, x + 1, y, z);
  o.z = lookup_density_brick(v, x, y + 1, z);
  o.w = lookup_density_brick(v, x + 1, y + 1, z);
  out[i] = o;
}

// reference layout -> brickf32: one thread per voxel of the padded grid
__global__ __launch_bounds__(256) void build_brickf32(const DevVolume v, float* __restrict__ out,
                                                       uint64_t first, uint64_t end) {
  uint64_t i = first + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= end) return;
  uint32_t l = (uint32_t)(i & 511u);
  uint64_t b = i >> 9;
  uint32_t bx = (uint32_t)(b % v.bc[0]);
  uint32_t by = (uint32_t)((b / v.bc[0]) % v.bc[1]);
  uint32_t bz = (uint32_t)(b / ((uint64_t)v.bc[0] * v.bc[1]));
  out[i] = lookup_density_brick(v, (int)(bx * 8u + (l & 7u)), (int)(by * 8u + ((l >> 3) & 7u)),
                                (int)(bz * 8u + (l >> 6)));
}

// reference layout -> bricku8: one thread per dword (4 voxels along x) of the padded grid, and one per brick range
__global__ __launch_bounds__(256) void build_bricku8(const DevVolume v, uint32_t* __restrict__ out, uint64_t first,
                                                      uint64_t end) {
  uint64_t i = first + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;   // dword index
  if (i >= end) return;
  uint32_t l = (uint32_t)(i & 127u);        // z * 16 + y * 2 + (x >> 2)
  uint64_t b = i >> 7;
  uint32_t bx = (uint32_t)(b % v.bc[0]);
  uint32_t by = (uint32_t)((b / v.bc[0]) % v.bc[1]);
  uint32_t bz = (uint32_t)(b / ((uint64_t)v.bc[0] * v.bc[1]));
  const uint32_t x0 = bx * 8u + (l & 1u) * 4u, y = by * 8u + ((l >> 1) & 7u), z = bz * 8u + (l >> 4);
  uint32_t w = 0u;
#pragma unroll
  for (uint32_t j = 0; j < 4u; ++j) w |= lookup_code_brick(v, x0 + j, y, z) << (8u * j);
  out[i] = w;
}
__global__ __launch_bounds__(256) void build_bricku8_range(const DevVolume v, float2* __restrict__ out, uint32_t first,
                                                            uint32_t end) {
  uint32_t b = first + blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= end) return;
  const uint32_t rg = v.range[b];
  const float mn = half_bits_to_float(rg >> 16), mx = half_bits_to_float(rg & 0xffffu);
  out[b] = make_float2(mn, mx - mn);   // lookup_density_brick: fma(un, mx - mn, mn)
}

// ---- ordered running-mean blend of pipelined frame results (fragment.frag:158 applied n times) ----
struct MergeArgs {
  const float4* result[MERGE_MAX];
  float weight[MERGE_MAX];
  uint32_t count;
};
__global__ __launch_bounds__(256) void merge_results(float4* __restrict__ slab, const MergeArgs a, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 acc = slab[i];
  for (uint32_t k = 0; k < a.count; ++k) {
    float w = a.weight[k];
    float4 r = a.result[k][i];
    float px = w != 0.0f ? acc.x : 0.0f, py = w != 0.0f ? acc.y : 0.0f, pz = w != 0.0f ? acc.z : 0.0f;
    acc.x = fma_(1.0f - w, r.x, w * px);
    acc.y = fma_(1.0f - w, r.y, w * py);
    acc.z = fma_(1.0f - w, r.z, w * pz);
    acc.w = 1.0f;
  }
  slab[i] = acc;
}

// ---- slab(s) -> row-major image --------------------------------------------------------
// gathered = shard_count slabs back to back (each tiles_per_shard*4096 float4)
__global__ __launch_bounds__(256) void detile(const float4* __restrict__ gathered,
                                               float4* __restrict__ image, const TileMap tm) {
  uint32_t x = blockIdx.x * 16u + (threadIdx.x & 15u);
  uint32_t y = blockIdx.y * 16u + (threadIdx.x >> 4);
  if (x >= tm.W || y >= tm.H) return;
  uint32_t t = (y >> 6) * tm.tiles_x + (x >> 6);
  uint32_t pos = tm.inv ? tm.inv[t] : t;
  uint32_t shard = pos % tm.shard_count, lt = pos / tm.shard_count;
  uint32_t wx = (x >> 3) & 7u, wy = (y >> 3) & 7u;
  uint32_t wt = (wx & 1u) | ((wy & 1u) << 1) | ((wx & 2u) << 1) | ((wy & 2u) << 2) |
                ((wx & 4u) << 2) | ((wy & 4u) << 3);
  uint32_t lane = (y & 7u) * 8u + (x & 7u);
  size_t si = ((size_t)shard * tm.tiles_per_shard + lt) * 4096u + wt * 64u + lane;
  image[(size_t)y * tm.W + x] = gathered[si];
}

// ---- display pass: blit.frag:17-35 -------------------------------------------------------
VXD float hable(float x) {
  const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
  float num = fma_(D, E, x * fma_(C, B, A * x));
  float den = fma_(D, F, x * (A * x + B));
  return num / den - E / F;
}
// out is ow x oh; the w x h image is sampled NEAREST (ow == w, oh == h: identity)
__global__ __launch_bounds__(256) void blit_rgba8(const float4* __restrict__ image,
                                                   uchar4* __restrict__ out, uint32_t w, uint32_t h,
                                                   uint32_t ow, uint32_t oh, float exposure, float gamma) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ow * oh) return;
  uint32_t x = i % ow, y = i / ow;
  uint32_t sx = (uint32_t)(((uint64_t)(2u * x + 1u) * w) / (2ull * ow));
  uint32_t sy = (uint32_t)(((uint64_t)(2u * y + 1u) * h) / (2ull * oh));
  float4 a = image[sy * w + sx];
  float white = hable(11.2f), ig = 1.0f / gamma;
  float c[4] = {powf(hable(exposure * a.x) / white, ig), powf(hable(exposure * a.y) / white, ig),
                powf(hable(exposure * a.z) / white, ig), a.w};
  unsigned char o[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float f = c[k];
    f = (f != f) ? 0.0f : f;
    f = f < 0.0f ? 0.0f : (f > 1.0f ? 1.0f : f);
    o[k] = (unsigned char)f2i(floorf(fma_(f, 255.0f, 0.5f)));
  }
  out[i] = make_uchar4(o[0], o[1], o[2], o[3]);
}

// envSetup.frag:26-41 over the 512x512 importance map, 8x8 bilinear taps per texel
__global__ __launch_bounds__(256) void build_importance(const float4* __restrict__ tex, uint32_t w, uint32_t h,
                                                         float* __restrict__ pyr) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= IMP_DIM * IMP_DIM) return;
  uint32_t px = i % IMP_DIM, py = i / IMP_DIM;
  const int ns = 8;
  const float out_size = (float)(IMP_DIM * ns), inv_samples = 1.0f / (float)(ns * ns);
  float imp = 0.0f;
  for (int y = 0; y < ns; ++y)
    for (int x = 0; x < ns; ++x) {
      float u = ((float)(px * ns) + ((float)x + 0.5f)) / out_size;
      float v = ((float)(py * ns) + ((float)y + 0.5f)) / out_size;
      imp += env_luma(env_texture(tex, w, h, u, v));
    }
  pyr[i] = imp * inv_samples;
}
// generateMipmap (environment.ts:58-60): 2x2 box, one level per launch
__global__ __launch_bounds__(256) void build_importance_mip(float* __restrict__ pyr, uint32_t level) {
  uint32_t n = IMP_DIM >> level, m = n * 2;
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * n) return;
  uint32_t x = i % n, y = i / n;
  const float* src = pyr + imp_offset(level - 1);
  float a = src[(size_t)(2 * y) * m + 2 * x], b = src[(size_t)(2 * y) * m + 2 * x + 1];
  float c = src[(size_t)(2 * y + 1) * m + 2 * x], d = src[(size_t)(2 * y + 1) * m + 2 * x + 1];
  pyr[imp_offset(level) + i] = (((a + b) + c) + d) * 0.25f;
}

// sibling quads of every level 0..8 for sample_environment: one thread per quad
// local majorants of the default mode (Frame::local_majorant): one thread per (level, cell of level-0 strides);
// the last entry is the value outside the grid (range texel 0)
__global__ __launch_bounds__(256) void build_local_majorants(const VxParams p, const DevVolume v, const float4* __restrict__ tf,
                                                             uint32_t tf_len, float* __restrict__ out) {
  const uint32_t nb = v.bc[0] * v.bc[1] * v.bc[2];
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > 4u * nb) return;
  float r = 0.0f;
  if (i < 4u * nb) {
    const uint32_t mip = i / nb, c = i - mip * nb;
    const uint32_t bz = c / (v.bc[0] * v.bc[1]), rem = c - bz * v.bc[0] * v.bc[1], by = rem / v.bc[0], bx = rem - by * v.bc[0];
    r = range_max_texel(v, (int)mip, bx, by, bz);
  }
  const TfView view{tf, tf_len, (float)tf_len, false};
  const float m = p.volume_density_scale * r;   // lookup_majorant, common.glsl:50-53
  out[i] = p.volume_maj * lookup_transfer(view, p.sample_range[0], p.sample_range[1], m * p.volume_inv_maj).w;
}

__global__ __launch_bounds__(256) void build_importance_quads(const float* __restrict__ pyr, float4* __restrict__ quads,
                                                               uint32_t level) {
  uint32_t half = IMP_DIM >> (level + 1), n = half * 2;
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= half * half) return;
  uint32_t X = i % half, Y = i / half;
  const float* src = pyr + imp_offset(level);
  quads[impq_offset(level) + i] = make_float4(src[(size_t)(2 * Y) * n + 2 * X], src[(size_t)(2 * Y) * n + 2 * X + 1],
                                              src[(size_t)(2 * Y + 1) * n + 2 * X], src[(size_t)(2 * Y + 1) * n + 2 * X + 1]);
}

// test hook: the unorm8 table
__global__ void unorm_table(float* out) { out[threadIdx.x] = unorm8(threadIdx.x); }

// test hook (vx_debug_rng): random.glsl:41-106 evaluated on the device, one thread per output word
__global__ __launch_bounds__(256) void debug_rng(int op, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                                  uint32_t n, uint32_t* __restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (op == 4) {
    // 256 draws r = (256 * (a[0] + i) + j) / 2^24: how many free-flight logarithms differ in any bit from -logf(1 - r)
    uint32_t bad = 0;
    for (uint32_t j = 0; j < 256u; ++j) {
      const float r = (float)(256u * (a[0] + i) + j) * 5.9604644775390625e-08f;
      bad += __builtin_bit_cast(uint32_t, neg_log_one_minus(r)) != __builtin_bit_cast(uint32_t, -logf(1.0f - r));
    }
    out[i] = bad;
  } else if (op == 0) {
    out[i] = tea32(a[i], b[i]);
  } else if (op == 1) {
    out[i] = wang_hash(a[i]);
  } else {
    Rng s = seed_xoshiro(a[0]);
    uint32_t w = 0;
    float f = 0.0f;
    for (uint32_t k = 0; k <= i; ++k) {  // thread i walks the stream to its word (n is small)
      if (op == 2) w = xoshiro_next(s);
      else f = rng(s);
    }
    out[i] = op == 2 ? w : __builtin_bit_cast(uint32_t, f);
  }
}

// measurement hook (vx_probe_valu_rate): nothing but independent v_fma_f32 chains, 8 waves per SIMD
__global__ __launch_bounds__(256) void probe_valu_rate(float* __restrict__ out, int iters, float a, float b) {
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = (float)threadIdx.x + (float)i;
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_fmaf(acc[i], a, b);
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) r += acc[i];
  if (r == 12345.678f) out[0] = r;
}

// measurement hook (vx_probe_gather_rate): nothing but 16-byte-per-lane gathers.  The 64 lanes form `lines` groups of
// consecutive lanes, every group inside one 128-byte line of a 16 KiB (L1-resident) table; the groups use `distinct`
// different lines in turn (distinct == lines: every group its own line; distinct < lines: a line is looked up by
// several groups, as in the march, where neighbouring 4-lane groups straddle the same lines).  8 gathers in flight.
__device__ inline uint32_t probe_hash(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__global__ __launch_bounds__(256) void probe_gather_rate(const float4* __restrict__ base, uint32_t lines, uint32_t distinct,
                                                          int iters, float* __restrict__ out) {
  const uint32_t lane = threadIdx.x & 63u, wave = blockIdx.x * 4u + (threadIdx.x >> 6);
  const uint32_t g = (lane * lines) >> 6;                      // group of lanes sharing one line
  const uint32_t first = (g * 64u + lines - 1u) / lines;       // first lane of the group
  const uint32_t within = (lane - first) & 7u;                 // quad inside the line
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  uint32_t s = probe_hash(wave * 977u + 13u);
  for (int it = 0; it < iters; ++it) {
    float4 q[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      s = s * 1664525u + 1013904223u;                           // wave-uniform stream
      uint32_t line = ((s >> 8) + (g % distinct) * 37u) & 127u; // 37 is odd: `distinct` different lines per gather
      q[u] = base[(line << 3) + within];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) { acc.x += q[u].x; acc.y += q[u].y; acc.z += q[u].z; acc.w += q[u].w; }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

}  // namespace vx
