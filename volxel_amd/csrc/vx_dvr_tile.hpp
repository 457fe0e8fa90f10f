// vx_dvr_tile.hpp -- DVR (VX_MODE_DVR) on the "brickf32" layout with per-wave LDS staging of
// the active tile of voxels.
//
// Why: the cellquad kernel gathers 32 B per sample straight through the vector L1; rocprof
// (profiles/r01_v1_dvr_cellquad.txt) shows the waves 80 % of their life in s_waitcnt with the
// texture-data path (TD/TA) as the busiest unit at ~33 tag look-ups per gather instruction,
// while VALU sits at a third of its rate.  The 64 rays of one 8x8-pixel wave stay within a
// few voxels of each other, and neighbouring rays / consecutive steps re-read the same voxels
// about six times.  So each wave
//   1. takes S march steps at a time ("macro step"), computes the integer bounding box of all
//      trilinear taps its live lanes will touch (DPP wave min/max, no LDS, no barrier),
//   2. copies that box of decoded fp32 voxels from the 8^3-brick layout into its private LDS
//      tile -- one aligned 16-byte chunk (half a brick row, 4 voxels) per lane and load, so a
//      whole box is ~3 load instructions; each voxel is fetched once per wave instead of ~6 times,
//   3. runs the S sample steps out of LDS: 4 ds_read2_b32 per sample at immediate offsets.
// Boxes that do not fit the 12x10x10 tile (grazing / very wide footprints) take the same steps
// with direct global taps.  Arithmetic is identical to Frame<>::dvr: densities, TF bins, sample
// counts and termination are bit-identical to the oracle.
#pragma once
#include "vx_dvr.hpp"

namespace vx {

constexpr int TILE_X = 12, TILE_Y = 10, TILE_Z = 10;  // x: three 16-byte chunks of 4 voxels
constexpr int TILE_FLOATS = TILE_X * TILE_Y * TILE_Z;  // 4000 B per wave
constexpr int DVR_MACRO = 8;                            // march steps per staged tile

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

// (idx / d) for idx < 512, 1 <= d <= 12, as multiply-shift with rcp16 = 65536/d + 1
VXD uint32_t small_div(uint32_t idx, uint32_t rcp16) { return (idx * rcp16) >> 16; }
__constant__ const uint32_t RCP16[13] = {0,     65537, 32769, 21846, 16385, 13108, 10923,
                                         9363,  8193,  7282,  6554,  5958,  5462};

template <int S>
__global__ __launch_bounds__(256) void render_dvr_tile(const VxParams p, const DevVolume v,
                                                        const float4* __restrict__ tf_global,
                                                        uint32_t tf_len, float4* __restrict__ slab,
                                                        uint32_t frame, float weight, const TileMap tm,
                                                        DevCounters* __restrict__ dc) {
  extern __shared__ float4 lds_raw[];
  float4* tf_lds = lds_raw;
  float* tile = reinterpret_cast<float*>(lds_raw + tf_len) + (threadIdx.x >> 6) * TILE_FLOATS;
  for (uint32_t i = threadIdx.x; i < tf_len; i += blockDim.x) tf_lds[i] = tf_global[i];
  __syncthreads();
  uint32_t lt, sub;
  if (!block_to_tile(blockIdx.x, tm, lt, sub)) return;
  const uint32_t wt = sub * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
  int px, py;
  uint32_t si;
  const bool in_image = wave_pixel(tm, lt, wt, lane, px, py, si);

  DvrRay r{};
  if (in_image) r = dvr_setup(p, px, py, frame);
  bool alive = in_image && r.hit;
  const uint32_t n_rays = (uint32_t)__builtin_popcountll(__ballot(alive));

  const float scale = p.volume_density_scale, inv_maj = p.volume_inv_maj, maj = p.volume_maj;
  const float sr0 = p.sample_range[0], sr1 = p.sample_range[1];
  const float lenf = (float)tf_len;
  const int last = (int)tf_len - 1;
  const float ert = p.dvr_ert_tau;
  const int max_steps = p.dvr_max_steps;
  const bool skip = p.dvr_skip_empty && v.skip_bits;

  float Cx = 0.f, Cy = 0.f, Cz = 0.f, T = 1.0f, tau = 0.0f, kf = 0.0f;
  int k = 0;
  uint32_t n_samples = 0, n_slots = 0, n_direct = 0;  // wave-uniform

  // one compositing step given the eight taps of the sample's cell (A5 + A7 + A12)
  auto composite = [&](float v000, float v100, float v010, float v110, float v001, float v101,
                       float v011, float v111, float fx, float fy, float fz) {
    float wx = 1.0f - fx, wy = 1.0f - fy, wz = 1.0f - fz;
    float lx0 = fma_(v100, fx, v000 * wx);
    float lx1 = fma_(v110, fx, v010 * wx);
    float hx0 = fma_(v101, fx, v001 * wx);
    float hx1 = fma_(v111, fx, v011 * wx);
    float l = fma_(lx1, fy, lx0 * wy);
    float h = fma_(hx1, fy, hx0 * wy);
    float d = scale * fma_(h, fz, l * wz);
    float dn = d * inv_maj;
    int ti = (int)(dn * lenf);
    ti = ti > last ? last : ti;
    ti = ti < 0 ? 0 : ti;
    bool in_range = !(dn < sr0 || dn > sr1);
    float4 rgba = tf_lds[ti];
    float alpha = in_range ? rgba.w : 0.0f;
    if (alpha > 0.0f) {
      tau = fma_(alpha * maj, r.dt, tau);
      float Tn = __builtin_amdgcn_exp2f(tau * -1.4426950408889634f);
      float dT = T - Tn;
      Cx = fma_(dT, rgba.x, Cx);
      Cy = fma_(dT, rgba.y, Cy);
      Cz = fma_(dT, rgba.z, Cz);
      T = Tn;
      if (tau >= ert) {
        T = 0.0f;
        alive = false;
      }
    }
  };

  while (true) {
    // ---- bounding box of the taps of the next S steps (end points suffice: fl(fma(t,d,o)) is
    //      monotone in t, so every intermediate sample lies between them on each axis) --------
    float ta = fma_(kf, r.dt, r.t0);
    alive = alive && (ta < r.far) && (k < max_steps);
    if (__ballot(alive) == 0ull) break;
    float tb = fma_(kf + (float)(S - 1), r.dt, r.t0);
    tb = tb < r.far ? tb : r.far;
    int lo[3], hi[3];
    {
      float a0 = floorf(fma_(ta, r.idir.x, r.ipos.x) - 0.5f), b0 = floorf(fma_(tb, r.idir.x, r.ipos.x) - 0.5f);
      float a1 = floorf(fma_(ta, r.idir.y, r.ipos.y) - 0.5f), b1 = floorf(fma_(tb, r.idir.y, r.ipos.y) - 0.5f);
      float a2 = floorf(fma_(ta, r.idir.z, r.ipos.z) - 0.5f), b2 = floorf(fma_(tb, r.idir.z, r.ipos.z) - 0.5f);
      lo[0] = (int)fminf(a0, b0); hi[0] = (int)fmaxf(a0, b0) + 1;
      lo[1] = (int)fminf(a1, b1); hi[1] = (int)fmaxf(a1, b1) + 1;
      lo[2] = (int)fminf(a2, b2); hi[2] = (int)fmaxf(a2, b2) + 1;
    }
    int LO[3], dim[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      int l = wave_minmax<true>(alive ? lo[a] : 0x7fffffff);
      int h = wave_minmax<false>(alive ? hi[a] : (int)0x80000000);
      LO[a] = l;
      dim[a] = h - l + 1;
    }
    // the tile starts on a 4-voxel boundary in x so that a lane can stage one aligned 16-byte
    // chunk (half a brick row) with a single load and a single ds_write_b128
    {
      int hx = LO[0] + dim[0] - 1;
      LO[0] &= ~3;
      dim[0] = hx - LO[0] + 1;
    }
    const bool staged = dim[0] <= TILE_X && dim[1] <= TILE_Y && dim[2] <= TILE_Z;  // uniform

    if (staged) {
      // ---- stage the box: lanes own (x,y) columns, loop over z ------------------------------
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // earlier tile reads are done
      __builtin_amdgcn_wave_barrier();
      const uint32_t nch = ((uint32_t)dim[0] + 3u) >> 2;            // 1..3 chunks per row
      const uint32_t dy = (uint32_t)dim[1];
      const uint32_t n_items = nch * dy * (uint32_t)dim[2];          // <= 300
      const uint32_t rcp_c = RCP16[nch], rcp_y = RCP16[dy];
      const uint32_t bstride = v.bc[0] * v.bc[1];
      constexpr int MAX_PASS = (3 * TILE_Y * TILE_Z + 63) / 64;      // 5
      float4 vals[MAX_PASS];
      uint32_t dsto[MAX_PASS];
      // all loads first (whole passes are skipped by a uniform branch), one wait, then the writes
#pragma unroll
      for (int ps = 0; ps < MAX_PASS; ++ps) {
        if ((uint32_t)ps * 64u < n_items) {
          uint32_t id = lane + (uint32_t)ps * 64u;
          uint32_t rest = small_div(id, rcp_c), c = id - rest * nch;
          uint32_t zz = small_div(rest, rcp_y), yy = rest - zz * dy;
          int gx = LO[0] + (int)(c * 4u), gy = LO[1] + (int)yy, gz = LO[2] + (int)zz;
          float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
          if (id < n_items && (uint32_t)gx < v.extent[0] && (uint32_t)gy < v.extent[1] &&
              (uint32_t)gz < v.extent[2]) {
            uint32_t b = ((uint32_t)gz >> 3) * bstride + ((uint32_t)gy >> 3) * v.bc[0] + ((uint32_t)gx >> 3);
            uint32_t l = (((uint32_t)gz & 7u) << 6) | (((uint32_t)gy & 7u) << 3) | ((uint32_t)gx & 7u);
            val = *reinterpret_cast<const float4*>(v.bf + ((size_t)b * 512u + l));
          }
          vals[ps] = val;
          dsto[ps] = id < n_items ? (zz * TILE_Y + yy) * TILE_X + c * 4u : 0xffffffffu;
        }
      }
#pragma unroll
      for (int ps = 0; ps < MAX_PASS; ++ps)
        if ((uint32_t)ps * 64u < n_items && dsto[ps] != 0xffffffffu)
          *reinterpret_cast<float4*>(tile + dsto[ps]) = vals[ps];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
      n_direct += 1;
    }

    // ---- S sample steps ---------------------------------------------------------------------
    auto step = [&](auto&& taps) {
      float t = fma_(kf, r.dt, r.t0);
      alive = alive && (t < r.far) && (k < max_steps);
      unsigned long long m = __ballot(alive);
      n_slots += (m != 0ull) ? 64u : 0u;
      bool evaluated = false;
      if (alive) {
        float qx = fma_(t, r.idir.x, r.ipos.x) - 0.5f;
        float qy = fma_(t, r.idir.y, r.ipos.y) - 0.5f;
        float qz = fma_(t, r.idir.z, r.ipos.z) - 0.5f;
        float flx = floorf(qx), fly = floorf(qy), flz = floorf(qz);
        bool empty = false;
        if (skip) {
          uint32_t cx = (uint32_t)((int)flx + 1), cy = (uint32_t)((int)fly + 1), cz = (uint32_t)((int)flz + 1);
          cx = cx < v.extent[0] + 7u ? cx : v.extent[0] + 7u;
          cy = cy < v.extent[1] + 7u ? cy : v.extent[1] + 7u;
          cz = cz < v.extent[2] + 7u ? cz : v.extent[2] + 7u;
          empty = skip_test(v.skip_bits, 3u + v.skip_level, v.skip_dims[0], v.skip_dims[1], cx, cy, cz);
        }
        if (!empty) taps((int)flx, (int)fly, (int)flz, qx - flx, qy - fly, qz - flz);
        evaluated = !empty;
      }
      n_samples += (uint32_t)__builtin_popcountll(__ballot(evaluated));
      kf += 1.0f;
      ++k;
    };
    if (staged) {
#pragma unroll
      for (int s = 0; s < S; ++s)
        step([&](int ix, int iy, int iz, float fx, float fy, float fz) {
          const float* tp = tile + ((iz - LO[2]) * TILE_Y + (iy - LO[1])) * TILE_X + (ix - LO[0]);
          composite(tp[0], tp[1], tp[TILE_X], tp[TILE_X + 1], tp[TILE_X * TILE_Y], tp[TILE_X * TILE_Y + 1],
                    tp[TILE_X * TILE_Y + TILE_X], tp[TILE_X * TILE_Y + TILE_X + 1], fx, fy, fz);
        });
    } else {
#pragma unroll 1
      for (int s = 0; s < S; ++s)
        step([&](int ix, int iy, int iz, float fx, float fy, float fz) {
          composite(bf_voxel(v, ix, iy, iz), bf_voxel(v, ix + 1, iy, iz), bf_voxel(v, ix, iy + 1, iz),
                    bf_voxel(v, ix + 1, iy + 1, iz), bf_voxel(v, ix, iy, iz + 1),
                    bf_voxel(v, ix + 1, iy, iz + 1), bf_voxel(v, ix, iy + 1, iz + 1),
                    bf_voxel(v, ix + 1, iy + 1, iz + 1), fx, fy, fz);
        });
    }
  }

  if (in_image) dvr_store(p, v, r, Cx, Cy, Cz, T, weight, slab, si);
  const uint32_t n_px = (uint32_t)__builtin_popcountll(__ballot(in_image));
  add_counts(dc, n_samples, n_rays, n_px, n_direct, 0u, n_slots);
}

inline void launch_dvr_tile(const VxParams& p, const DevVolume& v, const float4* tf, uint32_t tf_len,
                            float4* slab, uint32_t frame, float weight, const TileMap& tm,
                            DevCounters* dc, hipStream_t stream) {
  uint32_t groups = (tm.tiles_per_shard + 7u) / 8u;
  size_t lds = (size_t)tf_len * sizeof(float4) + 4u * TILE_FLOATS * sizeof(float);
  hipLaunchKernelGGL((render_dvr_tile<DVR_MACRO>), dim3(groups * 128u), dim3(256), lds, stream, p, v, tf,
                     tf_len, slab, frame, weight, tm, dc);
}

}  // namespace vx
