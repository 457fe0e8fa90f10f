// vx_dvr_lds.hpp -- DVR (VX_MODE_DVR) and DVR + central-difference gradient + Blinn-Phong (VX_MODE_DVR_PHONG,
// BASELINE config 4) on the "brickf32" layout: 8^3 bricks of decoded fp32 voxels in HBM, the active window of
// voxels staged per wave through LDS ("volume laid out in 8^3 bricks, per-workgroup LDS staging of the active
// brick", BASELINE north star).
//
// Why LDS: the cellquad kernel (vx_dvr.hpp) is bound by the vector L1's tag pipe -- the 64 lanes of a gather touch
// ~10 distinct 128-byte lines but the L1 works on groups of 4 lanes and spends ~30 look-ups on them (bench.py
// roofline.l1).  Here a wave fetches every voxel of its window ONCE with row-coalesced 16-byte loads (3 load
// instructions per window of ~16 march steps instead of 2 per march step) and then takes the eight taps of a sample -- and the 24
// further taps of a gradient -- from LDS, which has no tags.
//
// One wave = one 8x8-pixel tile, as everywhere.  Per window:
//   1. anchor: the window is DIMX x DIMY x DIMZ voxels; per axis it starts at the smallest cell any live lane
//      samples next (largest, if the wave marches in the negative direction), x rounded down to a multiple of 4;
//   2. stage: lane = one (y,z) row of the window: X / 4 aligned 16-byte loads from the brick rows it crosses,
//      all issued back to back, then as many ds_write_b128; rows and chunks outside the volume are zeros (A4);
//   3. march: up to S steps; a lane takes a step while the cell of its next sample (and, for Phong, the cells one
//      voxel either side) lies inside the window, otherwise it waits for the next window.  Lanes advance at their
//      own pace -- every ray still evaluates exactly its own sample sequence, so densities, TF bins, sample counts
//      and termination are bit-identical to Frame<>::dvr and to the oracle.
// No barriers: the tile is private to the wave.
#pragma once
#include <type_traits>

#include "vx_dvr.hpp"

namespace vx {

// Window geometry (measured on config 3 / 4, 1x MI355X, 16 frames per launch, ms per frame DVR / Phong:
//   X 16, D 10 / 12, natural strides (16, 160)        0.498 / 0.838   a third of the time in LDS bank conflicts
//   X 16, D 9 / 10, strides (20, 188 / 220)            0.485 / 0.670   (+ hardware transcendentals in the shading)
//   X 16, D 8 / 10                                      0.453 / 0.664   64 rows = ONE staging pass, 6 blocks per CU
//   X 12, D 8 / 10                                      0.438 / 0.614   three chunks per row instead of four
//   X 12, D 8 / 11 (shipped)                            0.438 / 0.591
//   D 10 for DVR 0.534, S 12 / 20 / 24 / 32 steps per window 0.446 / 0.462 / 0.487 / 0.496 (S = 16: 0.438)).
// X voxels per row are staged as aligned 16-byte chunks; a row occupies RS words of LDS and a z slice SS words.
// RS = 4 (mod 8) and SS = 28 (mod 32) spread the lanes of a read over the 32 banks: eight consecutive rows land on
// eight different 4-bank groups, slices step by 28 banks.  (-DVX_LDS_X / _D / _DP / _S rebuild a variant:
// tools/variant_build.sh.)
#ifndef VX_LDS_D
#define VX_LDS_D 8
#endif
#ifndef VX_LDS_DP
#define VX_LDS_DP 10
#endif
#ifndef VX_LDS_X
#define VX_LDS_X 12
#endif
#ifndef VX_LDS_S
#define VX_LDS_S 16
#endif
#ifndef VX_LDS_S_PHONG   // steps per window of the shading kernel: with one frame's 64 pixels per wave 12 / 16 / 20 / 24 gave
#define VX_LDS_S_PHONG 16  // 0.365 / 0.370 / 0.375 / 0.377 ms per frame; with lanes = pixels x frames 12 / 16 / 20: 0.335 / 0.327 / 0.329
#endif
#ifndef VX_LDS_WGD   // y / z size of a window the four waves of a workgroup share (WG builds): 16 x 16 rows = 64 per wave
#define VX_LDS_WGD 16
#endif
template <bool PHONG, bool WG = false>
struct LdsTile {
  static constexpr int X = VX_LDS_X;             // X / 4 chunks of 16 bytes per row
#ifndef VX_LDS_DY   // windows that are not square in (y, z), with lanes = pixels x frames, config 3, ms per frame at 32 frames per
                    // launch: 8x8 0.2098, 6x10 0.2120, 10x6 0.2241, 7x9 0.2115, 9x7 0.2180; steps per window 16 / 20 / 24 / 32:
                    // 0.2110 / 0.2113 / 0.2120 / 0.2119; X = 8: 0.2255 -- the cube stays
#define VX_LDS_DY VX_LDS_D
#endif
#ifndef VX_LDS_DZ
#define VX_LDS_DZ VX_LDS_D
#endif
  static constexpr int Y = WG ? VX_LDS_WGD : (PHONG ? VX_LDS_DP : VX_LDS_DY);
  static constexpr int Z = WG ? VX_LDS_WGD : (PHONG ? VX_LDS_DP : VX_LDS_DZ);
  static constexpr int RS = (X % 8 == 4) ? X : X + 4;             // row stride in words, = 4 mod 8
  static constexpr int SS = (Y * RS + 31 - 28) / 32 * 32 + 28;    // slice stride in words: >= Y * RS, = 28 mod 32
  static constexpr int ROWS = Y * Z;
  static constexpr int FLOATS = SS * Z;          // 3968 B (DVR) / 4960 B (Phong) per wave
  static constexpr int PASSES = WG ? 1 : (ROWS + 63) / 64;   // WG: wave w stages rows 64 w .. 64 w + 63
  static_assert(!WG || ROWS == 256, "a shared window is staged by four waves, 64 rows each");
  static constexpr int LO_MARGIN = PHONG ? 1 : 0;   // cells below the sample's cell that must be resident
  static constexpr int HI_MARGIN = PHONG ? 2 : 1;   // taps above it (x+1; x+2 for the gradient)
  static_assert(SS >= Y * RS && SS % 4 == 0 && RS % 4 == 0 && RS >= X && X % 4 == 0, "tile strides");
};

// ---- wave64 integer min / max with DPP (row scan + row broadcasts), result in every lane ----
template <int CTRL, int ROW_MASK, int BANK_MASK>
VXD int dpp_src(int identity, int v) {
  return __builtin_amdgcn_update_dpp(identity, v, CTRL, ROW_MASK, BANK_MASK, false);
}
template <bool IS_MIN>
VXD int wave_minmax(int v) {
  constexpr int ID = IS_MIN ? 0x7fffffff : (int)0x80000000;
  auto op = [](int a, int b) { return IS_MIN ? (a < b ? a : b) : (a > b ? a : b); };
  v = op(v, dpp_src<0x111, 0xf, 0xf>(ID, v));  // row_shr:1
  v = op(v, dpp_src<0x112, 0xf, 0xf>(ID, v));  // row_shr:2
  v = op(v, dpp_src<0x114, 0xf, 0xf>(ID, v));  // row_shr:4
  v = op(v, dpp_src<0x118, 0xf, 0xf>(ID, v));  // row_shr:8   -> lane 15 of each row = row result
  v = op(v, dpp_src<0x142, 0xa, 0xf>(ID, v));  // row_bcast:15 into rows 1 and 3
  v = op(v, dpp_src<0x143, 0xc, 0xf>(ID, v));  // row_bcast:31 into rows 2 and 3
  return __builtin_amdgcn_readlane(v, 63);
}

// three minima at once, the stages of the three chains interleaved: a DPP instruction must wait two cycles for the
// instruction that wrote its source, and three independent chains fill those slots with work instead of s_nop
VXD void wave_min3(int& a, int& b, int& c) {
  constexpr int ID = 0x7fffffff;
  auto mn = [](int x, int y) { return x < y ? x : y; };
#define VX_STAGE(CTRL, RM)                              \
  {                                                     \
    const int ta = dpp_src<CTRL, RM, 0xf>(ID, a), tb = dpp_src<CTRL, RM, 0xf>(ID, b), tc = dpp_src<CTRL, RM, 0xf>(ID, c); \
    a = mn(a, ta); b = mn(b, tb); c = mn(c, tc);        \
  }
  VX_STAGE(0x111, 0xf) VX_STAGE(0x112, 0xf) VX_STAGE(0x114, 0xf) VX_STAGE(0x118, 0xf) VX_STAGE(0x142, 0xa) VX_STAGE(0x143, 0xc)
#undef VX_STAGE
  a = __builtin_amdgcn_readlane(a, 63);
  b = __builtin_amdgcn_readlane(b, 63);
  c = __builtin_amdgcn_readlane(c, 63);
}

// one trilinear mix of eight taps, common.glsl:62-68 (without the density scale)
VXD float mix8(float v000, float v100, float v010, float v110, float v001, float v101, float v011, float v111,
               float fx, float wx, float fy, float wy, float fz, float wz) {
  float lx0 = fma_(v100, fx, v000 * wx);
  float lx1 = fma_(v110, fx, v010 * wx);
  float hx0 = fma_(v101, fx, v001 * wx);
  float hx1 = fma_(v111, fx, v011 * wx);
  float l = fma_(lx1, fy, lx0 * wy);
  float h = fma_(hx1, fy, hx0 * wy);
  return fma_(h, fz, l * wz);
}

// occupancy asked of the register allocator: the DVR build fits 61 VGPRs without a scratch access and gains from 8 resident
// waves per SIMD (ms per frame at 5 / 6 / 7 / 8 waves: 0.434 / 0.431 / 0.420 / 0.415); the Phong build needs 86
#ifndef VX_W_LDS
#define VX_W_LDS 8
#endif
// with the 8 KB macro-cell mask beside the tiles six workgroups fit a CU: 6 waves per SIMD, 80 VGPRs
#ifndef VX_W_LDS_SKIP
#define VX_W_LDS_SKIP 6
#endif
#ifndef VX_W_LDS_PHONG
#define VX_W_LDS_PHONG 7
#endif
typedef const float __attribute__((address_space(3))) * LdsFloatPtr;

// U8: the window is staged from the bricku8 layout (8-bit codes + a range per brick, decoded here with A4's fma) instead
// of brickf32's fp32 voxels -- the dword index of a 4-voxel chunk is brickf32's 16-byte-unit index, so the row and chunk
// arithmetic is shared; everything after the staging is the same code on the same values.
// WG (round 4): ONE window per workgroup.  In a launch of a multiple of 32 frames the four waves of a workgroup take the SAME 8
// pixels (8 frames each): 256 rays of one narrow beam.  They place one window four times the volume of a wave's (12 x 16 x 16),
// each wave stages a quarter of its rows, and every lane marches in it -- 1.7x the steps per window for the same staging work
// per wave.  The cost is two workgroup barriers per window (the anchor's minima go through LDS; nobody may restage a tile a
// sister wave still reads).  Every barrier sits on a workgroup-uniform path: the loop ends for all four waves together, when
// the minima in LDS say no wave has a live ray.  Which samples a ray evaluates is untouched: same bits
// (tests/test_gpu_parity.py::test_shared_window_kernel_is_bit_identical).
// MEASURED (profiles/r04_shared_window.txt), config 3, 32 frames per launch: wave-windows per frame 235 955 -> 146 521 (0.62x), lane
// utilisation 0.936 -> 0.924 (a longer window has a longer tail of lanes that have left it), and 0.2040-0.2086 ms per frame against
// 0.2045-0.2074 for the wave-private windows on the same boxes: the windows it saves (7 % of the vector work by the listing's
// prices) go into the exchange through LDS (~100 clocks per window and wave), the idle lane slots and the barrier skew.
// Not faster: it stays OPT-IN (VX_DVR_WG=1).
template <int S, bool PHONG, bool SKIP, bool U8 = false, bool WG = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PHONG ? (SKIP ? 1 : VX_W_LDS_PHONG) : (SKIP ? VX_W_LDS_SKIP : VX_W_LDS), 8))) void render_dvr_lds(const VxParams p, const DevVolume v,
                                                       const float4* __restrict__ tf_global, uint32_t tf_len,
                                                       const MultiOut mo, float weight, const TileMap tm,
                                                       const uint32_t* __restrict__ order) {
  static_assert(!WG || (!SKIP && !PHONG), "the shared window serves the plain DVR march");
  using TL = LdsTile<PHONG, WG>;
  constexpr int DX = TL::X, DY = TL::Y, DZ = TL::Z, RS = TL::RS, SS = TL::SS;
  extern __shared__ float4 lds_raw[];
  float4* tf_lds = lds_raw;
  uint32_t* mask_lds = reinterpret_cast<uint32_t*>(lds_raw + tf_len);
  float* tile = reinterpret_cast<float*>(mask_lds + (SKIP ? ((v.skip_words + 3u) & ~3u) : 0u)) + (WG ? 0u : (threadIdx.x >> 6) * TL::FLOATS);
  // WG: behind the shared tile, per wave {min x, min y, min z, first live lane's cell x, y, z, live | direction bits, -}
  int* const wg_box = reinterpret_cast<int*>(tile + TL::FLOATS);
  const uint32_t wave = threadIdx.x >> 6;
  for (uint32_t i = threadIdx.x; i < tf_len; i += blockDim.x) tf_lds[i] = tf_global[i];
  if (SKIP)
    for (uint32_t i = threadIdx.x; i < v.skip_words; i += blockDim.x) mask_lds[i] = v.skip_bits[i];
  __syncthreads();
  uint32_t fslot, bslot;
  multi_slot(blockIdx.x, mo.count, fslot, bslot);
  const uint32_t blk = order ? order[bslot] : bslot;
  float4* __restrict__ slab = mo.out[fslot];
  DevCounters* __restrict__ dc = mo.dc[fslot];
  const uint32_t frame = mo.frame[fslot];
  uint32_t lt, sub;
  if (!block_to_tile(blk, tm, lt, sub)) return;
  uint32_t wt = sub * 4u + (threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t plane = lane, my_frame = frame;
  const uint32_t fuse = WG ? 0u : mo.fuse;   // the running mean of the launch's frames applied here (wave uniform)
#ifndef VX_DVR_FL_MAXSH   // groups of up to 2^3 frame slots: 8 pixels x 8 frames per wave (groups of 16 / 32 measured within 1 %:
#define VX_DVR_FL_MAXSH 3 // ms per frame at 20 / 32 frames per launch 0.2244 / 0.2169 against 0.2216-0.2245 / 0.2150-0.2185)
#endif
  if (WG) {   // (the launcher only takes this build for a multiple of 32 frames) the workgroup at slot r of a group of 32 takes
              // wave tile r >> 3 of its block position and pixel octet r & 7; its four waves take 8 frames of the group each
    const uint32_t r = fslot & 31u;
    wt = sub * 4u + (r >> 3);
    plane = ((r & 7u) << 3) + (lane & 7u);
    slab = lane_frame_slot((fslot - r) + wave * 8u + (lane >> 3), my_frame);
  } else if (mo.count > 1u) {   // lanes = pixels x frames (vx_kernels.hpp frame_group)
    uint32_t base;
    // fused running mean (MultiOut::fuse): ONE group of all 32 / 64 frames of the launch, 2 pixels / 1 pixel per wave
    const uint32_t sh = fuse ? frame_group<6>(fslot, mo.count, base) : frame_group<VX_DVR_FL_MAXSH>(fslot, mo.count, base);
    if (sh != 0u) {
      const uint32_t psh = 6u - sh;
      plane = ((fslot - base) << psh) + (lane & ((1u << psh) - 1u));
      slab = lane_frame_slot(base + (lane >> psh), my_frame);
    }
  }
  int px, py;
  uint32_t si;
  const bool in_image = wave_pixel(tm, lt, wt, plane, px, py, si);

  DvrRay r{};
  if (in_image) r = dvr_setup(p, v, px, py, my_frame);
  const bool hit0 = in_image && r.hit;
  const uint32_t n_rays = (uint32_t)__builtin_popcountll(ballot(hit0));
  // samples the lane's ray still has: k = 0 .. nray - 1 (march contract, vx_dvr.hpp); a ray that terminates early
  // gets nray = -1, so liveness is `kf < nray` -- one compare on registers the loop holds anyway
  float nray = hit0 ? r.n : 0.0f;

  float scale = p.volume_density_scale;
  asm volatile("" : "+v"(scale));   // keep it in a VGPR: out of SGPRs the allocator re-loaded it from the kernel
                                    // arguments inside the march loop, a scalar-memory wait per step
  const float inv_maj = p.volume_inv_maj, maj = p.volume_maj;
  const float sr0 = p.sample_range[0], sr1 = p.sample_range[1];
  const float lenf = (float)tf_len;
  const int last = (int)tf_len - 1;
  const float ert = p.dvr_ert_tau;   // > 0: use_lds_kernel (vx_api.hip)
  const uint32_t ex = v.extent[0], ey = v.extent[1], ez = v.extent[2];
  const uint32_t bcx = v.bc[0], bcy = v.bc[1];
  const float4* __restrict__ bf4 = reinterpret_cast<const float4*>(v.bf);   // 16-byte units: 64 GiB of layout in 32 bits
  const uint32_t* __restrict__ bu = v.bu;                                   // U8: one dword of four codes per unit
  const float2* __restrict__ bur = v.bu_range;
  const uint32_t zero_chunk = bcx * bcy * v.bc[2] * 128u;   // the all-zero chunk behind the last brick (vx_api alloc_layout)
  const uint32_t sh = 3u + v.skip_level, md0 = v.skip_dims[0], md1 = v.skip_dims[1];
  const uint32_t cmaxx = ex + 7u, cmaxy = ey + 7u, cmaxz = ez + 7u;
  // Phong terms (vx_modes.hpp Frame::dvr<true>)
  V3 nl = v3(-p.light_dir[0], -p.light_dir[1], -p.light_dir[2]);
  V3 hv = v3(0.f, 0.f, 0.f);
  if (PHONG) hv = normalize3(sub3(nl, r.wdir));
  const float gsx = p.density_transform_inv[0], gsy = p.density_transform_inv[5], gsz = p.density_transform_inv[10];
  // steps per index unit along each axis, for the number of steps a lane can take inside a window; an axis the ray
  // does not move along gets a huge factor: any distance to a face times it exceeds every step count
  // (v_rcp_f32: the quotient only feeds estimates that an exact test confirms or that are conservative by a step)
  const float ivx = r.dq.x != 0.0f ? __builtin_amdgcn_rcpf(r.dq.x) : 3.0e38f;
  const float ivy = r.dq.y != 0.0f ? __builtin_amdgcn_rcpf(r.dq.y) : 3.0e38f;
  const float ivz = r.dq.z != 0.0f ? __builtin_amdgcn_rcpf(r.dq.z) : 3.0e38f;

  float Cx = 0.f, Cy = 0.f, Cz = 0.f, T = 1.0f, tau = 0.0f, kf = 0.0f;   // kf: per-lane step index
  uint32_t n_samples = 0, n_slots = 0, n_skipped = 0, n_grads = 0, n_loads = 0, n_reads = 0, n_tf = 0;   // wave-uniform

  // cell-frame position of the lane's next sample and its floor (the cell), as floats: the march needs no integer
  // cell -- the tile offset is formed in floating point (exact: small integers) and converted once
  float qx = 0.f, qy = 0.f, qz = 0.f, flx = 0.f, fly = 0.f, flz = 0.f;
  auto is_alive = [&]() { return kf < nray; };
  auto next_sample = [&]() {
    qx = fma_(kf, r.dq.x, r.q0.x);
    qy = fma_(kf, r.dq.y, r.q0.y);
    qz = fma_(kf, r.dq.z, r.q0.z);
    flx = floorf(qx); fly = floorf(qy); flz = floorf(qz);
  };
  next_sample();

  // The wave marches in the direction of its first live lane (rays of a wave are nearly parallel and keep their
  // direction; any choice is correct, it only decides which end of the cell range a window hugs).
  bool fwx = true, fwy = true, fwz = true;
  {
    const unsigned long long live0 = ballot(is_alive());
    if (live0 != 0ull) {
      const int first = (int)__builtin_ctzll(live0);
      fwx = __builtin_amdgcn_readlane(__builtin_bit_cast(int, r.dq.x), first) >= 0;   // sign bit clear
      fwy = __builtin_amdgcn_readlane(__builtin_bit_cast(int, r.dq.y), first) >= 0;
      fwz = __builtin_amdgcn_readlane(__builtin_bit_cast(int, r.dq.z), first) >= 0;
    }
    if (WG) {   // one direction for the workgroup: that of the lowest wave with a live ray
      if (lane == 0u) wg_box[wave * 8u + 6u] = live0 != 0ull ? (1 | (fwx ? 2 : 0) | (fwy ? 4 : 0) | (fwz ? 8 : 0)) : 0;
      __syncthreads();
      const int d0 = wg_box[6], d1 = wg_box[14], d2 = wg_box[22], d3 = wg_box[30];
      const int d = __builtin_amdgcn_readfirstlane(d0 ? d0 : (d1 ? d1 : (d2 ? d2 : d3)));
      fwx = (d & 2) != 0 || d == 0; fwy = (d & 4) != 0 || d == 0; fwz = (d & 8) != 0 || d == 0;
      __syncthreads();   // the slots are written again by the first window
    }
  }
  int LOx = 0, LOy = 0, LOz = 0;   // origin of the resident window (wave uniform)
  // window origin from the extreme cell of the lanes in `mask` along each axis: a forward window starts at the
  // smallest cell, a backward window ends at the largest; `slack` cells of extra room behind the anchor.
  // x is staged in aligned 16-byte chunks: a forward window starts at or below its anchor cell (round down), a
  // backward window ends at or above it (round up) -- the anchor stays inside, up to 3 columns are unused.
  auto anchor = [&](bool mask, int cx, int cy, int cz, int slack, int& ox, int& oy, int& oz) {
    int ex_ = mask ? (fwx ? cx : -cx) : 0x7fffffff;   // min of c, or -(max of c)
    int ey_ = mask ? (fwy ? cy : -cy) : 0x7fffffff;
    int ez_ = mask ? (fwz ? cz : -cz) : 0x7fffffff;
    wave_min3(ex_, ey_, ez_);
    ox = fwx ? ex_ - TL::LO_MARGIN - slack : -ex_ + TL::HI_MARGIN - (DX - 1) + slack;
    oy = fwy ? ey_ - TL::LO_MARGIN - slack : -ey_ + TL::HI_MARGIN - (DY - 1) + slack;
    oz = fwz ? ez_ - TL::LO_MARGIN - slack : -ez_ + TL::HI_MARGIN - (DZ - 1) + slack;
    ox = fwx ? (ox & ~3) : ((ox + 3) & ~3);
  };
  // a lane can step while its sample's cell, with the margins, is inside [origin, origin + D)
  auto inside_of = [&](int ox, int oy, int oz, int cx, int cy, int cz) {
    uint32_t rx = (uint32_t)(cx - ox - TL::LO_MARGIN), ry = (uint32_t)(cy - oy - TL::LO_MARGIN),
             rz = (uint32_t)(cz - oz - TL::LO_MARGIN);
    return (rx < (uint32_t)(DX - TL::LO_MARGIN - TL::HI_MARGIN)) & (ry < (uint32_t)(DY - TL::LO_MARGIN - TL::HI_MARGIN)) &
           (rz < (uint32_t)(DZ - TL::LO_MARGIN - TL::HI_MARGIN));
  };
  // ---- per window and lane: klim = the step index up to which the lane stays inside the resident window, so that the
  // march tests ONE compare (kf < klim) per step instead of three cell ranges and two liveness compares.
  // Along an axis the positions q(k) = fma(k, dq, q0) are monotone in k, so the steps a lane can take are the k below
  // the first one at or beyond the face it moves towards.  c = ceil((face - q0) / dq) estimates that index; the exact
  // test of q(c - 1) -- the very fma the march evaluates -- confirms that every sample below c is inside.  If the
  // estimate was a step too high the lane falls back to the single step its current sample (tested on integer cells)
  // allows; a step too low only ends the lane's run in this window one sample early.  Either way the lane never reads
  // outside the tile and the window sequence only affects speed, never which samples are evaluated.
  float klim = 0.0f;
  // LDS byte address of cell (0,0,0) of the index grid in the resident tile (wave uniform), as a float in a VECTOR
  // register: the address of a sample's cell is three full-rate fmas on it and one conversion (an fma with a scalar
  // source issues at half rate, profiles/r03_op_rates.txt)
  float tile_base = 0.0f;
  const int tile_addr = (int)(uint32_t)(uintptr_t)(LdsFloatPtr)tile;   // LDS byte address of the wave's tile
  [[maybe_unused]] auto axis_limit = [&](float q0a, float dqa, float iva, int lo_cell, int n_cells) {
    // cells [lo_cell, lo_cell + n_cells) are steppable: lo <= q < hi
    const float lo = (float)lo_cell, hi = lo + (float)n_cells;
    const bool bw = dqa < 0.0f;
    const float face = bw ? lo : hi;
    // never beyond the ray's own last sample: q(n - 1) inside means every remaining sample is (monotone), and the
    // estimate of a ray that barely moves along this axis (|dq| -> 0: c in the millions or infinite) stays a sample
    // index the test below can evaluate -- clamped at 2^24 such a ray failed the test and crawled one step per
    // window (found as a 0.6 ms single-frame launch: two such lanes per frame hold their waves for milliseconds).
    // fminf also turns a NaN estimate into n.
    float c = fminf(ceilf((face - q0a) * iva), nray);
    const float q1 = fma_(c - 1.0f, dqa, q0a);
    const bool in1 = (q1 < face) != bw;              // forward: q1 < hi; backward: q1 >= lo
    return in1 ? c : -1.0f;
  };
#ifndef VX_LIMITS_PER_AXIS
  // Round 4: ONE estimate for the three axes.  ceil is monotone, so min over the axes of ceil(x_a) is ceil(min x_a): the
  // three quotients are formed (a subtraction and a multiplication each), one min3, one ceil, one clamp at the ray's own
  // count -- and the candidate c is confirmed on all three axes at the same index c - 1, with the very fmas the march
  // evaluates there.  Monotone positions make every k < c inside once q(c - 1) is (the current sample is inside: `now`).
  // Against the per-axis form (kept under -DVX_LIMITS_PER_AXIS: three ceil / clamp / select chains, 44 vector instructions
  // per window) this is 24; it accepts a superset of the per-axis form's candidates (an axis whose own estimate was a step
  // too high no longer vetoes a c another axis keeps below it).  Which samples are evaluated does not depend on it.
  // the faces the ray moves towards, as offsets from the window origin: n cells ahead when it moves forwards, 0 backwards
  const float fcx = r.dq.x < 0.0f ? 0.0f : (float)(DX - TL::LO_MARGIN - TL::HI_MARGIN);
  const float fcy = r.dq.y < 0.0f ? 0.0f : (float)(DY - TL::LO_MARGIN - TL::HI_MARGIN);
  const float fcz = r.dq.z < 0.0f ? 0.0f : (float)(DZ - TL::LO_MARGIN - TL::HI_MARGIN);
  const unsigned long long bwx = ballot(r.dq.x < 0.0f), bwy = ballot(r.dq.y < 0.0f), bwz = ballot(r.dq.z < 0.0f);
  auto set_limits = [&](bool now) {
    const float facex = (float)(LOx + TL::LO_MARGIN) + fcx, facey = (float)(LOy + TL::LO_MARGIN) + fcy,
                facez = (float)(LOz + TL::LO_MARGIN) + fcz;
    const float xx = (facex - r.q0.x) * ivx, xy = (facey - r.q0.y) * ivy, xz = (facez - r.q0.z) * ivz;
    const float c = fminf(ceilf(fminf(xx, fminf(xy, xz))), nray);     // a NaN estimate becomes n
    const float kc = c - 1.0f;
    // forward: q(c - 1) < hi; backward: q(c - 1) >= lo -- the compare against the face, flipped for the lanes that move backwards
    const unsigned long long okx = ballot(fma_(kc, r.dq.x, r.q0.x) < facex) ^ bwx;
    const unsigned long long oky = ballot(fma_(kc, r.dq.y, r.q0.y) < facey) ^ bwy;
    const unsigned long long okz = ballot(fma_(kc, r.dq.z, r.q0.z) < facez) ^ bwz;
    const unsigned long long ok = okx & oky & okz;
    float k;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(k) : "v"(-1.0f), "v"(c), "s"(ok));
    k = fmaxf(k, kf + 1.0f);                         // the current sample is inside (integer test): one step at least
    k = fminf(k, nray);
    klim = now ? k : 0.0f;
    tile_base = (float)(tile_addr - 4 * ((LOz * SS) + (LOy * RS) + LOx));   // |.| < 2^23: exact
    asm volatile("" : "+v"(tile_base));
  };
#else
  // `now`: the lane is alive and its current sample's cell is inside the window (exact_window has just tested it)
  auto set_limits = [&](bool now) {
    const float kx = axis_limit(r.q0.x, r.dq.x, ivx, LOx + TL::LO_MARGIN, DX - TL::LO_MARGIN - TL::HI_MARGIN);
    const float ky = axis_limit(r.q0.y, r.dq.y, ivy, LOy + TL::LO_MARGIN, DY - TL::LO_MARGIN - TL::HI_MARGIN);
    const float kz = axis_limit(r.q0.z, r.dq.z, ivz, LOz + TL::LO_MARGIN, DZ - TL::LO_MARGIN - TL::HI_MARGIN);
    float k = fminf(kx, fminf(ky, kz));
    k = fmaxf(k, kf + 1.0f);                         // the current sample is inside (integer test): one step at least
    k = fminf(k, nray);
    klim = now ? k : 0.0f;
    tile_base = (float)(tile_addr - 4 * ((LOz * SS) + (LOy * RS) + LOx));   // |.| < 2^23: exact
    asm volatile("" : "+v"(tile_base));
  };
#endif
  // exact window for the lanes as they stand; lanes too far apart for one window (a wave astride two entry faces of
  // the clip box): serve the first live lane
  // returns, per lane: alive and the current sample's cell inside the window placed
  bool wg_any = true;   // WG: some wave of the workgroup has a live ray (workgroup uniform, from the minima in LDS)
  auto exact_window = [&](unsigned long long live) -> bool {
    const bool alive = is_alive();
    const int cxi = (int)flx, cyi = (int)fly, czi = (int)flz;
    if (WG) {
      // the wave's extreme cells and its first live lane's cell go through LDS; the barrier behind the writes is also the
      // point after which no wave reads the resident tile any more (every wave has left its march)
      int ex_ = alive ? (fwx ? cxi : -cxi) : 0x7fffffff, ey_ = alive ? (fwy ? cyi : -cyi) : 0x7fffffff,
          ez_ = alive ? (fwz ? czi : -czi) : 0x7fffffff;
      wave_min3(ex_, ey_, ez_);
      const int first = live != 0ull ? (int)__builtin_ctzll(live) : 0;
      const int fx_ = __builtin_amdgcn_readlane(cxi, first), fy_ = __builtin_amdgcn_readlane(cyi, first),
                fz_ = __builtin_amdgcn_readlane(czi, first);
      if (lane == 0u) {
        int* b = wg_box + wave * 8u;
        b[0] = ex_; b[1] = ey_; b[2] = ez_; b[3] = fx_; b[4] = fy_; b[5] = fz_; b[6] = live != 0ull ? 1 : 0;
      }
      __syncthreads();
      auto mn = [](int a, int b) { return a < b ? a : b; };
      const int gx = __builtin_amdgcn_readfirstlane(mn(mn(wg_box[0], wg_box[8]), mn(wg_box[16], wg_box[24])));
      const int gy = __builtin_amdgcn_readfirstlane(mn(mn(wg_box[1], wg_box[9]), mn(wg_box[17], wg_box[25])));
      const int gz = __builtin_amdgcn_readfirstlane(mn(mn(wg_box[2], wg_box[10]), mn(wg_box[18], wg_box[26])));
      const int l0 = wg_box[6], l1 = wg_box[14], l2 = wg_box[22], l3 = wg_box[30];
      const int fw = __builtin_amdgcn_readfirstlane(l0 ? 0 : (l1 ? 1 : (l2 ? 2 : (l3 ? 3 : -1))));   // lowest wave with a live ray
      wg_any = fw >= 0;
      if (!wg_any) return false;
      LOx = fwx ? gx - TL::LO_MARGIN : -gx + TL::HI_MARGIN - (DX - 1);
      LOy = fwy ? gy - TL::LO_MARGIN : -gy + TL::HI_MARGIN - (DY - 1);
      LOz = fwz ? gz - TL::LO_MARGIN : -gz + TL::HI_MARGIN - (DZ - 1);
      LOx = fwx ? (LOx & ~3) : ((LOx + 3) & ~3);
      // progress: the window must hold the first live lane of that wave (rays too far apart for one window -- a workgroup
      // astride two entry faces of the clip box -- are served one neighbourhood at a time); every wave decides alike
      const int ax = __builtin_amdgcn_readfirstlane(wg_box[fw * 8 + 3]), ay = __builtin_amdgcn_readfirstlane(wg_box[fw * 8 + 4]),
                az = __builtin_amdgcn_readfirstlane(wg_box[fw * 8 + 5]);
      const uint32_t rx = (uint32_t)(ax - LOx - TL::LO_MARGIN), ry = (uint32_t)(ay - LOy - TL::LO_MARGIN),
                     rz = (uint32_t)(az - LOz - TL::LO_MARGIN);
      if (!((rx < (uint32_t)(DX - TL::LO_MARGIN - TL::HI_MARGIN)) & (ry < (uint32_t)(DY - TL::LO_MARGIN - TL::HI_MARGIN)) &
            (rz < (uint32_t)(DZ - TL::LO_MARGIN - TL::HI_MARGIN)))) {
        LOx = ax - TL::LO_MARGIN - (fwx ? 0 : DX - 1 - TL::LO_MARGIN - TL::HI_MARGIN);
        LOx = fwx ? (LOx & ~3) : ((LOx + 3) & ~3);
        LOy = ay - TL::LO_MARGIN - (fwy ? 0 : DY - 1 - TL::LO_MARGIN - TL::HI_MARGIN);
        LOz = az - TL::LO_MARGIN - (fwz ? 0 : DZ - 1 - TL::LO_MARGIN - TL::HI_MARGIN);
      }
      return (bool)(alive & inside_of(LOx, LOy, LOz, cxi, cyi, czi));
    }
    anchor(alive, cxi, cyi, czi, 0, LOx, LOy, LOz);
    bool now = alive & inside_of(LOx, LOy, LOz, cxi, cyi, czi);
    if (ballot(now) == 0ull) {
      const int first = (int)__builtin_ctzll(live);
      LOx = __builtin_amdgcn_readlane(cxi, first) - TL::LO_MARGIN - (fwx ? 0 : DX - 1 - TL::LO_MARGIN - TL::HI_MARGIN);
      LOx = fwx ? (LOx & ~3) : ((LOx + 3) & ~3);
      LOy = __builtin_amdgcn_readlane(cyi, first) - TL::LO_MARGIN - (fwy ? 0 : DY - 1 - TL::LO_MARGIN - TL::HI_MARGIN);
      LOz = __builtin_amdgcn_readlane(czi, first) - TL::LO_MARGIN - (fwz ? 0 : DZ - 1 - TL::LO_MARGIN - TL::HI_MARGIN);
      now = alive & inside_of(LOx, LOy, LOz, cxi, cyi, czi);
    }
    return now;
  };
  // stage, part 1: lane = (y,z) row of the window at (ox,oy,oz): X / 4 aligned 16-byte loads into registers, issued
  // back to back; rows and chunks outside the volume are zeros (A4)
  constexpr int NC = DX / 4;
  auto issue_loads = [&](int ox, int oy, int oz, float4 (&vals)[TL::PASSES][NC]) {
    uint32_t xoff[NC];
    bool xin[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {   // chunk c covers x = ox + 4c .. +3: brick column and half row are wave uniform
      const int gx = ox + 4 * c;
      xin[c] = (uint32_t)gx < ex;
      xoff[c] = (((uint32_t)gx >> 3) << 7) + (((uint32_t)gx & 7u) >> 2);   // brick x * 128 + half row (16-byte units)
    }
#pragma unroll
    for (int ps = 0; ps < TL::PASSES; ++ps) {
      const uint32_t row = lane + 64u * (WG ? wave : (uint32_t)ps);
      const uint32_t zz = row / (uint32_t)DY, yy = row - zz * (uint32_t)DY;
      const int gy = oy + (int)yy, gz = oz + (int)zz;
      const bool rin = row < (uint32_t)TL::ROWS && (uint32_t)gy < ey && (uint32_t)gz < ez;
      // 16-byte units from the start of the layout to the brick row (y,z) of brick column 0
      // (24-bit multiplies: brick coordinates are below 2^10, their products below 2^24 for every volume the layout
      // can index; hipcc otherwise picks the quarter-rate v_mad_u64_u32 / v_mul_lo_u32.  Rows outside the volume are
      // selected away, whatever their index came to.)
      const uint32_t rowbase = (mad24(mad24((uint32_t)gz >> 3, bcy, (uint32_t)gy >> 3) & 0xffffffu, bcx, 0u) << 7) +
                               ((((uint32_t)gz & 7u) << 4) | (((uint32_t)gy & 7u) << 1));
      if (U8) {
        // codes and brick ranges of the row's chunks first (all loads in flight), then the decode; the unit behind the
        // last brick is a zero dword under the range {0, 0}: fma(0, 0, 0) = +0, as A4 asks for outside the volume
        uint32_t code[NC];
        float2 rg[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const uint32_t at = (rin && xin[c]) ? rowbase + xoff[c] : zero_chunk;
          const uint32_t cw = bu[at];
          const float2 rr = bur[at >> 7];
          code[c] = cw;
          rg[c] = rr;
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const float4 val = decode_codes4(code[c], rg[c]);
          vals[ps][c] = val;
        }
      } else {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const uint32_t at = (rin && xin[c]) ? rowbase + xoff[c] : zero_chunk;
          const float4 val = bf4[at];
          vals[ps][c] = val;
        }
      }
    }
    n_loads += (U8 ? 2u : 1u) * (uint32_t)NC * (uint32_t)TL::PASSES;   // U8: a dword of codes and a brick range per chunk
  };
  // stage, part 2: the rows into the wave's tile
  auto write_tile = [&](float4 (&vals)[TL::PASSES][NC]) {
    if (!WG) {   // (WG: the barrier of exact_window already separates the old tile's reads from these writes)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // earlier tile reads are done
      __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int ps = 0; ps < TL::PASSES; ++ps) {
      const uint32_t row = lane + 64u * (WG ? wave : (uint32_t)ps);
      if (row < (uint32_t)TL::ROWS) {
        const uint32_t zz = row / (uint32_t)DY, yy = row - zz * (uint32_t)DY;
        float4* dst = reinterpret_cast<float4*>(tile + zz * (uint32_t)SS + yy * (uint32_t)RS);
#pragma unroll
        for (int c = 0; c < NC; ++c) dst[c] = vals[ps][c];
      }
    }
    if (WG) {
      __syncthreads();   // the four quarters of the tile are in place
    } else {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  };

  // ---- SKIP: exact empty-space skipping (A12 note: a sample in a macro cell that can only see TF-transparent bricks
  // has alpha == 0 exactly) -------------------------------------------------------------------------------------------
  // is the lane's next sample in an empty macro cell?  (defined for every lane: the clamps keep the index in the mask)
  auto in_empty_cell = [&]() {
    uint32_t cx = (uint32_t)((int)flx + 1), cy = (uint32_t)((int)fly + 1), cz = (uint32_t)((int)flz + 1);
    cx = cx < cmaxx ? cx : cmaxx; cy = cy < cmaxy ? cy : cmaxy; cz = cz < cmaxz ? cz : cmaxz;
    const uint32_t mi = mad24(mad24(cz >> sh, md1, cy >> sh), md0, cx >> sh);   // at most 65536 macro cells
    return (bool)((mask_lds[mi >> 5] >> (mi & 31u)) & 1u);
  };
  // further steps a lane in an empty macro cell may pass over: it moves on to about one step short of the exit face of
  // the cell (vx_dvr.hpp does the same with the last sample of a batch); every sample passed over lies inside the same
  // empty macro cell, so the set of evaluated samples is unchanged.  (iv* = steps per index unit; a jump only has to
  // be conservative: one whole step is kept in hand against the rounding of the three products.  An axis the ray does
  // not move along has a huge factor: its distance to the far face becomes huge and drops out of the minimum.)
  // Only with steps well above the rounding of a sample position (2^-14 voxel at coordinate 1024); finer marches
  // skip sample by sample.
  const bool jumps = SKIP && p.dvr_step_voxels >= 0.015625f;
  auto jump_of = [&](bool emp) {
    const float Sf1 = (float)(1u << sh) - 1.0f;
    uint32_t cx = (uint32_t)((int)flx + 1), cy = (uint32_t)((int)fly + 1), cz = (uint32_t)((int)flz + 1);
    cx = cx < cmaxx ? cx : cmaxx; cy = cy < cmaxy ? cy : cmaxy; cz = cz < cmaxz ? cz : cmaxz;
    // q in [b - 1, b - 1 + S) inside the macro cell at b = (c >> sh) << sh; exit face along the ray
    const float fx_ = (float)((cx >> sh) << sh) + (r.dq.x < 0.0f ? -1.0f : Sf1);
    const float fy_ = (float)((cy >> sh) << sh) + (r.dq.y < 0.0f ? -1.0f : Sf1);
    const float fz_ = (float)((cz >> sh) << sh) + (r.dq.z < 0.0f ? -1.0f : Sf1);
    const float dmin = fminf((fx_ - qx) * ivx, fminf((fy_ - qy) * ivy, (fz_ - qz) * ivz));
    // samples k+1 .. k+n are passed over unseen: n < (steps to the exit face) keeps them inside the cell -- one whole
    // step in hand against the rounding of dmin; the sample the lane lands on is tested like any
    float n = floorf(dmin) - 1.0f;
    n = fminf(n, 1048576.0f);
    return (jumps && emp && n >= 1.0f) ? n : 0.0f;
  };
  // free flight: a sample in an empty macro cell needs no taps, hence no window -- before a window is placed the lanes
  // that stand in empty cells pass over them (jump, then step by step to the exit face) while the others wait, so
  // that windows are only staged where something can be seen.  Bounded: a lane still in empty space after FLY rounds
  // goes on inside the next window (the march tests the mask per step wherever the window touches an empty cell).
  constexpr int FLY = 48;
  auto free_flight = [&]() {
#pragma unroll 1
    for (int it = 0; it < FLY; ++it) {
      const bool emp = is_alive() & in_empty_cell();
      const unsigned long long em = ballot(emp);
      if (em == 0ull) break;
      n_skipped += (uint32_t)__builtin_popcountll(em);
      n_slots += 64u;
      kf = emp ? kf + 1.0f + jump_of(emp) : kf;
      next_sample();
    }
  };
  // does the resident window touch an empty macro cell?  It is at most 12 cells wide and a macro cell at least 16, so
  // the eight corner cells name every macro cell under it (lanes 0..7 test one each, with the clamps of the per-step test)
  auto touches_empty = [&]() {
    const int ax = (lane & 1u) ? LOx + DX - 1 : LOx, ay = (lane & 2u) ? LOy + DY - 1 : LOy,
              az = (lane & 4u) ? LOz + DZ - 1 : LOz;
    auto clampi = [](int x, uint32_t hi) { return (uint32_t)(x < 0 ? 0 : (x > (int)hi ? (int)hi : x)); };
    const uint32_t cx = clampi(ax + 1, cmaxx), cy = clampi(ay + 1, cmaxy), cz = clampi(az + 1, cmaxz);
    const uint32_t mi = mad24(mad24(cz >> sh, md1, cy >> sh), md0, cx >> sh);
    const bool empty = (mask_lds[mi >> 5] >> (mi & 31u)) & 1u;
    return (ballot(empty) & 0xffull) != 0ull;
  };
  bool wtest = false;   // SKIP: the resident window touches an empty macro cell: the march tests the mask per step
  // place and stage the next window for the lanes in `live`; false: free flight ended every ray
  auto next_window = [&](unsigned long long live) {
    if (SKIP) {
      free_flight();
      live = ballot(is_alive());
      if (live == 0ull) return false;
    }
    const bool now = exact_window(live);
    if (WG && !wg_any) return false;   // workgroup uniform: no wave has a live ray left
    if (SKIP) wtest = touches_empty();
    set_limits(now);
    float4 vals[TL::PASSES][NC];
    issue_loads(LOx, LOy, LOz, vals);
    write_tile(vals);
    return true;
  };

  bool wg_go = true;
  {
    const unsigned long long live = ballot(is_alive());
    if (WG) wg_go = next_window(live);   // every wave takes part in every window of its workgroup
    else if (live != 0ull) (void)next_window(live);
  }
  // ---- 3. march: up to S steps out of LDS.  TEST (SKIP builds): the window touches an empty macro cell, the mask is
  // tested per step; in the other windows the test is compiled out.
  // Per step and stepping lane (52 -> 36 vector instructions against the first form of this loop): one compare; the
  // tile offset as two fmas on the float cells, one conversion, one shift-add; eight taps (four ds_read2_b32 at
  // immediate offsets of one address); the 14-instruction mix; scale; range test; then index + 1, the next position
  // (three fmas) and its floors.  (The body under `if (go)` -- EXEC = the stepping lanes -- was tried: the compiler
  // answered the divergent region with 22 register copies per step for the values it carries round the loop.)
  auto march = [&](auto test_tag) {
    constexpr bool TEST = decltype(test_tag)::value;
    // One exit, tested at the bottom on the updated registers (a window is only placed where a lane can step, so the
    // first trip always has one): with the test at the top the loop had two exits that leave different versions of
    // every carried value live, and the compiler paid for the merge with 20 register copies per step.
    int s = 0;
    bool more;
    bool go = kf < klim;     // carried: the bottom test of one trip is the lane mask of the next
    unsigned long long gom = ballot(go);   // the same as a scalar (the ballot of a carried bool would be materialised)
    // one step of the wave.  ALL: every lane of the wave steps (the common case since a wave's lanes are a few pixels
    // under many frames' jitter): no select on the tile address or on the step count
    auto step = [&](auto all_tag) {
      constexpr bool ALL = decltype(all_tag)::value;
      n_slots += 64u;
      bool eval = go;
      float jump = 0.0f;   // SKIP: further steps this lane may pass over (all inside the same empty macro cell)
      if (TEST) {
        const bool empty = in_empty_cell();
        eval = go & !empty;
        const bool emp = go & empty;
        const unsigned long long em = ballot(emp);
        n_skipped += (uint32_t)__builtin_popcountll(em);
        if (em != 0ull) jump = jump_of(emp);      // wave uniform
      }
      // (without skipping a lane evaluates one sample per step it takes: its count is kf after the march)
      if (SKIP) n_samples += (uint32_t)__builtin_popcountll(ballot(eval));
      // byte address of the sample's cell in the tile: (z * SS + y * RS + x) * 4 + tile_base, every partial sum an
      // integer below 2^23, exact in fp32 (cells are below 2^13, the byte strides below 2^11).  A lane that does not step (its pending sample
      // lies outside this window) reads the tile's first cell instead: every read stays inside the wave's tile.
      const int cell_addr = (int)fma_(flz, (float)(4 * SS), fma_(fly, (float)(4 * RS), fma_(flx, 4.0f, tile_base)));
      const int addr = ALL ? cell_addr : (go ? cell_addr : tile_addr);
      const LdsFloatPtr tp = (LdsFloatPtr)(uintptr_t)(uint32_t)addr;
      const LdsFloatPtr tq = tp + SS;             // slice z + 1
      const float v000 = tp[0], v100 = tp[1], v010 = tp[RS], v110 = tp[RS + 1];
      const float v001 = tq[0], v101 = tq[1], v011 = tq[RS], v111 = tq[RS + 1];
      n_reads += 4u;
      const float fx = qx - flx, fy = qy - fly, fz = qz - flz;
      const float wx = 1.0f - fx, wy = 1.0f - fy, wz = 1.0f - fz;
      // common.glsl:62-68: x lerps of the four rows, y lerps of the two slices, z lerp
      const float lx0 = fma_(v100, fx, v000 * wx), lx1 = fma_(v110, fx, v010 * wx);
      const float hx0 = fma_(v101, fx, v001 * wx), hx1 = fma_(v111, fx, v011 * wx);
      const float ml = fma_(lx1, fy, lx0 * wy), mh = fma_(hx1, fy, hx0 * wy);
      const float d = scale * fma_(mh, fz, ml * wz);
      const float dn = d * inv_maj;
      // A7 / A12 only where they can matter: a sample outside the sample range, or whose TF entry has alpha 0, leaves
      // tau, T and C exactly as they are, and on this kind of data most wave steps have no lane inside the range at
      // all (config 3: 82 %), so the LUT fetch, the classification and the composite sit behind one wave-uniform branch
      // (the lane mask as the AND of three compare results: the ballot of a conjunction is materialised with a select
      // and a second compare, two half-rate vector instructions per step -- profiles/r03_op_rates.txt)
      // (the upper bound is only tested where the lower one holds for some lane: on this kind of data the lower bound
      // alone turns 80 % of the wave steps away)
      const unsigned long long rlo = (TEST ? ballot(eval) : gom) & ballot(!(dn < sr0));
      if (rlo != 0ull) {
        const unsigned long long rm = rlo & ballot(!(dn > sr1));
        n_tf += (uint32_t)__builtin_popcountll(rm);
#ifdef VX_COUNT_INRANGE   // diagnostic build: skip_steps counts the wave steps that enter this block
        if (!SKIP) n_skipped += 1u;
#endif
        const int ti = clamp0_i32((int)(dn * lenf), last);   // dn >= 0: truncation == floor
        float4 rgba = tf_lds[ti];
        // alpha of the lanes in `rm`, 0 elsewhere: the select takes the scalar mask as it stands (spelled on the bool, the
        // compiler evaluates both range compares a second time)
        float alpha;
        asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(alpha) : "v"(rgba.w), "s"(rm));
        const bool contrib = alpha > 0.0f;
        if (PHONG) {
          const unsigned long long cm = ballot(contrib);
          if (cm != 0ull) {   // wave-uniform: the 24 further taps only when some lane shades
            n_grads += (uint32_t)__builtin_popcountll(cm);
            n_reads += 12u;
            if (contrib) {
              // central differences one voxel either side, in the sample's cell frame (cells c +- e, the sample's
              // fractions): T(c + e) - T(c - e) per axis, each T a full common.glsl:62-68 mix times the density scale
              // x: T(c + ex) mixes the taps x+1, x+2 and T(c - ex) the taps x-1, x of the same four rows -- no lerp is shared
              const float xm0 = tp[-1], xp0 = tp[2], xm1 = tp[RS - 1], xp1 = tp[RS + 2];
              const float xm2 = tq[-1], xp2 = tq[2], xm3 = tq[RS - 1], xp3 = tq[RS + 2];
              const float gx = scale * mix8(v100, xp0, v110, xp1, v101, xp2, v111, xp3, fx, wx, fy, wy, fz, wz) -
                               scale * mix8(xm0, v000, xm1, v010, xm2, v001, xm3, v011, fx, wx, fy, wy, fz, wz);
              // y: T(c + ey) mixes rows y+1, y+2 -- the x lerp of row y+1 is the centre's lx1 / hx1 (same operands, same
              // operation: same bits); likewise T(c - ey) reuses lx0 / hx0
              const float ym0 = tp[-RS], ym1 = tp[-RS + 1], yp0 = tp[2 * RS], yp1 = tp[2 * RS + 1];
              const float ym2 = tq[-RS], ym3 = tq[-RS + 1], yp2 = tq[2 * RS], yp3 = tq[2 * RS + 1];
              const float lxp = fma_(yp1, fx, yp0 * wx), hxp = fma_(yp3, fx, yp2 * wx);
              const float lxm = fma_(ym1, fx, ym0 * wx), hxm = fma_(ym3, fx, ym2 * wx);
              const float gy = scale * fma_(fma_(hxp, fy, hx1 * wy), fz, fma_(lxp, fy, lx1 * wy) * wz) -
                               scale * fma_(fma_(hx0, fy, hxm * wy), fz, fma_(lx0, fy, lxm * wy) * wz);
              // z: T(c + ez) mixes slices z+1, z+2 -- the y lerp of slice z+1 is the centre's mh; T(c - ez) reuses ml
              const LdsFloatPtr tzm = tp - SS;
              const LdsFloatPtr tzp = tq + SS;
              const float zm0 = tzm[0], zm1 = tzm[1], zm2 = tzm[RS], zm3 = tzm[RS + 1];
              const float zp0 = tzp[0], zp1 = tzp[1], zp2 = tzp[RS], zp3 = tzp[RS + 1];
              const float mp = fma_(fma_(zp3, fx, zp2 * wx), fy, fma_(zp1, fx, zp0 * wx) * wy);
              const float mm = fma_(fma_(zm3, fx, zm2 * wx), fy, fma_(zm1, fx, zm0 * wx) * wy);
              const float gz = scale * fma_(mp, fz, mh * wz) - scale * fma_(ml, fz, mm * wz);
              const V3 g = v3(gx * gsx, gy * gsy, gz * gsz);
              phong_shade(p, g, nl, hv, rgba);
            }
          }
        }
        // tau += a*maj*dt; C += (T_prev - T) * rgb   (raymarch.glsl:43 / SURVEY A12) -- straight line, as vx_dvr.hpp
        // No select on `contrib`: T is always exp2(-tau * log2 e) of the lane's current tau (1 at tau = 0; a lane that is
        // never sampled again keeps both), so a sample with alpha = 0 -- tau unchanged: fma(0, dt, tau) == tau --
        // recomputes the same T, dT = T - T = +0 and C += 0 * rgb leaves C as it is (vx_upload_transfer refuses
        // non-finite entries, so 0 * rgb is 0)
        tau = fma_(alpha * maj, r.dt, tau);
        const float Tn = __builtin_amdgcn_exp2f(tau * -1.4426950408889634f);
        const float dT = T - Tn;
        Cx = fma_(dT, rgba.x, Cx);
        Cy = fma_(dT, rgba.y, Cy);
        Cz = fma_(dT, rgba.z, Cz);
        T = Tn;
        // early ray termination (vx_oracle.c dvr_pixel: contrib && tau >= ert): the ray has no further samples; its T
        // stays exp2(-tau log2 e), so that nothing later in the march touches its C, and becomes 0 after the march
        // ert > 0 (the launcher sends an epsilon >= 1 to render_generic): only a contributing sample can carry tau over
        // it, and the ray ends right there, so `tau >= ert` alone says "terminated" for every lane, now and later -- no
        // compare on alpha here, and nray follows once per window, after the march
        klim = tau >= ert ? 0.0f : klim;
      }
      // the lanes that stepped move on to their next sample (the others recompute the position they already hold)
      kf = ALL ? kf + 1.0f : (go ? kf + 1.0f + jump : kf);
      next_sample();
      ++s;
      go = kf < klim;
      gom = ballot(go);
    };
#ifndef VX_NO_ALL_LANES_LOOP
    if (!TEST) {
      const unsigned long long full = ballot(true);   // the wave's lanes (all 64 unless the grid's last wave is ragged)
      bool fast = gom == full;
      if (fast) {
#pragma unroll 1
        do {
          step(std::true_type{});
          fast = (gom == full) & (s < S);
        } while (fast);
      }
    }
#endif
    // The window is left when fewer than VX_LDS_MIN_ACTIVE lanes can still step in it (1: when nobody can): the stragglers'
    // steps are taken in the next window, which is anchored at them anyway -- which samples are evaluated does not change.
#ifndef VX_LDS_MIN_ACTIVE
#define VX_LDS_MIN_ACTIVE 1
#endif
    auto enough = [&]() {
      return VX_LDS_MIN_ACTIVE <= 1 ? (gom != 0ull) : (__builtin_popcountll(gom) >= VX_LDS_MIN_ACTIVE);
    };
    more = (s < S) & ((s == 0) ? (gom != 0ull) : enough());   // the first trip of a window always has a lane that steps
    if (more) {
#pragma unroll 1
      do {
        step(std::false_type{});
        more = (s < S) & enough();
      } while (more);
    }
  };
  while (WG ? wg_go : true) {
    if (!WG && ballot(is_alive()) == 0ull) break;
    if (SKIP && wtest) march(std::integral_constant<bool, SKIP>{});
    else march(std::false_type{});
    nray = tau >= ert ? -1.0f : nray;   // the rays the march terminated
    // ---- next window ------------------------------------------------------------------------------------------------
    const unsigned long long live = ballot(is_alive());
    if (!WG && live == 0ull) break;
    if (!next_window(live)) break;      // WG: false for the four waves together
  }

  // (kf counts the samples of a lane: without skipping every step it takes evaluates one; a terminated ray stopped at kf)
  if (!SKIP) n_samples = wave_sum((uint32_t)kf);   // kf <= 2^24: exact; lane 0 holds the sum (add_counts reads it there)
  // a ray that terminated early is opaque: T = 0 (vx_modes.hpp Frame::dvr)
  if (nray < 0.0f) T = 0.0f;
  if (!WG && fuse != 0u) {
    // ---- the running mean of the launch, in the wave that holds every frame of its pixels (MultiOut::fuse) -------------
    // lane l holds frame slot l >> psh of pixel l & (npx - 1); its result goes to the wave's tile (the march is over); lanes
    // 0 .. 3 npx - 1 then each fold one colour channel of one pixel through the frame slots in order:
    // acc = fma(1 - w, r, w * (w != 0 ? acc : 0)) -- merge_results, operation for operation, on the value dvr_store would
    // have written to the frame's result slab (fma(1, L, 0 * 0)).
    V3 L = v3(0.f, 0.f, 0.f);
    if (in_image) L = dvr_radiance(p, v, r, Cx, Cy, Cz, T);
    fold_frames(tile, lane, L, in_image, si, mo.accum, fuse, 31u - (uint32_t)__builtin_clz(mo.count));   // count = 8, 16, 32 or 64
  } else if (in_image) dvr_store(p, v, r, Cx, Cy, Cz, T, weight, slab, si);
  const uint32_t n_px = (uint32_t)__builtin_popcountll(ballot(in_image));
  add_counts(dc, n_samples, n_rays, n_px, n_skipped, n_grads, n_slots, blk, n_loads, n_reads, n_tf);
}

#ifndef VX_LDS_S_WG   // steps per shared window (its geometry ends a lane's run long before)
#define VX_LDS_S_WG 32   // 24 / 28 / 32 / 36 / 40 / 48 measured: 0.2070 / 0.2062 / 0.2040 / 0.2057 / 0.2075 / 0.2044 ms per frame (lane utilisation 0.939 ... 0.912)
#endif
inline void launch_dvr_lds(const VxParams& p, const DevVolume& v, const float4* tf, uint32_t tf_len, const MultiOut& mo,
                           float weight, const TileMap& tm, hipStream_t stream, const uint32_t* order, bool shared_window = false) {
  const uint32_t groups = (tm.tiles_per_shard + 7u) / 8u;
  const dim3 grid(groups * 128u * (mo.count ? mo.count : 1u)), block(256);
  const bool skip = p.dvr_skip_empty && v.skip_bits;
  const bool phong = p.render_mode == VX_MODE_DVR_PHONG;
  const bool u8 = v.bu_active != 0u;
  // one window per workgroup: the plain DVR march on brickf32 in a launch of a multiple of 32 frames (render_dvr_lds, WG)
  if (shared_window && !skip && !phong && !u8 && mo.count >= 32u && (mo.count & 31u) == 0u) {
    const size_t lds_wg = (size_t)tf_len * sizeof(float4) + (size_t)LdsTile<false, true>::FLOATS * sizeof(float) + 32u * sizeof(int);
    hipLaunchKernelGGL((render_dvr_lds<VX_LDS_S_WG, false, false, false, true>), grid, block, lds_wg, stream, p, v, tf, tf_len, mo,
                       weight, tm, order);
    return;
  }
  const size_t tile_bytes = 4u * (size_t)(phong ? LdsTile<true>::FLOATS : LdsTile<false>::FLOATS) * sizeof(float);
  const size_t lds = (size_t)tf_len * sizeof(float4) + (skip ? (((size_t)v.skip_words + 3u) & ~(size_t)3u) * 4u : 0u) + tile_bytes;
#define VX_LAUNCH_LDS(PH, SK, U)                                                                                          \
  hipLaunchKernelGGL((render_dvr_lds<(PH ? VX_LDS_S_PHONG : VX_LDS_S), PH, SK, U>), grid, block, lds, stream, p, v, tf, tf_len, \
                     mo, weight, tm, order)
  if (phong) {
    if (skip) { if (u8) VX_LAUNCH_LDS(true, true, true); else VX_LAUNCH_LDS(true, true, false); }
    else      { if (u8) VX_LAUNCH_LDS(true, false, true); else VX_LAUNCH_LDS(true, false, false); }
  } else {
    if (skip) { if (u8) VX_LAUNCH_LDS(false, true, true); else VX_LAUNCH_LDS(false, true, false); }
    else      { if (u8) VX_LAUNCH_LDS(false, false, true); else VX_LAUNCH_LDS(false, false, false); }
  }
#undef VX_LAUNCH_LDS
}

}  // namespace vx
