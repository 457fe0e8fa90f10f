// vx_dvr.hpp -- the headline kernel: deterministic front-to-back DVR (VX_MODE_DVR) on the
// MI355X "cellquad" layout.  Same arithmetic as Frame<LAYOUT_CQ>::dvr<false> (bit-identical
// densities, TF bins and termination decisions); exp(-tau) uses v_exp_f32.
//
// Shape: one wave = one 8x8-pixel tile, 4 waves (16x16 pixels) per workgroup, TF LUT in LDS,
// two 16-byte loads per sample (the xy-quads of z-slices k and k+1 of one apron brick),
// wave-uniform ballot to skip the compositing block while no lane sees opacity, exact
// per-wave sample counting with s_bcnt1 on the exec mask.
#pragma once
#include <cstdlib>

#include "vx_kernels.hpp"

namespace vx {

// [build] march contract (oracle/vx_oracle.c dvr_pixel): a ray has n samples, sample k sits at
// q = fma(k, dq, q0) in the cell frame of A5 (position - 1/2)
struct DvrRay {
  V3 q0, dq;       // cell-frame position of sample 0, increment per sample (index space, raymarch.glsl:31-32)
  float dt;        // world-space step (optical depth per sample = alpha * maj * dt)
  float n;         // number of samples: min(ceil((far - t0) / dt), max_steps); 0 for a miss
  V3 wdir;         // world-space direction (for the environment term)
  bool hit;
};

VXD DvrRay dvr_setup(const VxParams& p, const DevVolume& v, int px, int py, uint32_t frame) {
  float tex_x = tex_coord(px, p.res[0], &v, 0);
  float tex_y = tex_coord(py, p.res[1], &v, 1);
  float jx = 0.5f, jy = 0.5f, off = 0.5f;
  if (p.dvr_jitter) {  // wave-uniform: the integer RNG (32 TEA rounds) only runs when it is used
    Rng s = seed_xoshiro(tea32(42u * (uint32_t)(py * p.res[0] + px), frame));
    float a0 = rng(s), a1 = rng(s), b0 = rng(s), b1 = rng(s);
    jx = (a0 + b0) / 2.0f;
    jy = (a1 + b1) / 2.0f;
    (void)rng(s);      // tau_target slot of raymarch.glsl:28
    off = rng(s);      // start jitter, raymarch.glsl:30
  }
  Ray ray = setup_world_ray(p, tex_x, tex_y, jx, jy, &v);
  DvrRay r;
  float near, far;
  V3 ipos, idir;
  r.hit = ray_box_intersection(ray, p.volume_aabb_min, p.volume_aabb_max, near, far);
  to_index(p, ray, ipos, idir, &v);
  r.dt = p.dvr_step_voxels / sqrtf(dot3(idir, idir));
  const float t0 = fma_(off, r.dt, near);
  const float x = (far - t0) / r.dt;
  r.n = (r.hit && x > 0.0f) ? fminf(ceilf(x), (float)p.dvr_max_steps) : 0.0f;
  r.dq = v3(r.dt * idir.x, r.dt * idir.y, r.dt * idir.z);
  r.q0 = v3(fma_(t0, idir.x, ipos.x) - 0.5f, fma_(t0, idir.y, ipos.y) - 0.5f, fma_(t0, idir.z, ipos.z) - 0.5f);
  r.wdir = ray.d;
  return r;
}

// write-back shared by the tuned DVR kernels: gain, background, sanitize, running mean
// the radiance of a DVR ray: gain, background, sanitize (the value dvr_store blends)
VXD V3 dvr_radiance(const VxParams& p, const DevVolume& dv, const DvrRay& r, float Cx, float Cy, float Cz, float T) {
  float Lx = Cx * p.dvr_gain[0], Ly = Cy * p.dvr_gain[1], Lz = Cz * p.dvr_gain[2];
  if (p.show_environment > 0 && T > 0.0f) {
    V3 env = lookup_environment(p, dv, r.wdir);
    Lx = fma_(T, env.x, Lx);
    Ly = fma_(T, env.y, Ly);
    Lz = fma_(T, env.z, Lz);
  }
  return v3(sanitize1(Lx), sanitize1(Ly), sanitize1(Lz));
}
VXD void dvr_store(const VxParams& p, const DevVolume& dv, const DvrRay& r, float Cx, float Cy, float Cz, float T,
                   float weight, float4* __restrict__ slab, uint32_t si) {
  const V3 L = dvr_radiance(p, dv, r, Cx, Cy, Cz, T);
  const float Lx = L.x, Ly = L.y, Lz = L.z;
  float4 prev = make_float4(0.f, 0.f, 0.f, 0.f);
  if (weight != 0.0f) prev = slab[si];
  float4 o;
  o.x = fma_(1.0f - weight, Lx, weight * prev.x);
  o.y = fma_(1.0f - weight, Ly, weight * prev.y);
  o.z = fma_(1.0f - weight, Lz, weight * prev.z);
  o.w = 1.0f;
  slab[si] = o;
}

// U = march steps per loop iteration: the 2*U gathers of a batch are issued back to back before
// any of them is consumed, so a wave keeps 2*U loads in flight instead of 2 (the march is
// latency-bound on the longest rays: tools/tail_probe.py).
// SKIP = exact empty-space skipping: the macro-cell bitmask (<= 8 KiB) sits in LDS next to the TF;
// a sample in an empty macro cell is neither gathered nor counted (its alpha is exactly 0), and a
// lane whose last sample of a batch was empty jumps to one step before the exit of that macro
// cell.  Every jumped-over sample lies inside the same empty macro cell, so the set of evaluated
// samples is exactly "samples whose macro cell is not empty" -- what the oracle counts.
// PROBE = measurement build (vx_probe_gather_spread): the same march, but every gather also counts the distinct
// 128-byte lines its 64 lane addresses fall into -- over the whole wave and per group of 4 consecutive lanes (the
// unit the L1 tag pipe works on) -- and nothing is written to the framebuffer.  The sums go to rays / pixels of
// the counter record (spread[0] = wave-wide lines, spread[1] = quad look-ups; per q0 gather, q1 is the same
// pattern one slice further).
template <int U, bool SKIP, bool PROBE = false>
__global__ __launch_bounds__(256) void render_dvr_cq(const VxParams p, const DevVolume v,
                                                      const float4* __restrict__ tf_global,
                                                      uint32_t tf_len, const MultiOut mo, float weight,
                                                      const TileMap tm,
                                                      const uint32_t* __restrict__ order) {
  extern __shared__ float4 tf_lds[];
  uint32_t* mask_lds = reinterpret_cast<uint32_t*>(tf_lds + tf_len);
  for (uint32_t i = threadIdx.x; i < tf_len; i += blockDim.x) tf_lds[i] = tf_global[i];
  if (SKIP)
    for (uint32_t i = threadIdx.x; i < v.skip_words; i += blockDim.x) mask_lds[i] = v.skip_bits[i];
  __syncthreads();
  // launch slot -> (frame slot, logical block); blocks in longest-first order of the previous frame
  // launch index = bslot*count + fslot.  (An XCD-class-preserving interleave was measured slower
  // at 2 and 4 shards: one frame slot per XCD class balances better than one tile set per class.)
  uint32_t fslot, bslot;
  multi_slot(blockIdx.x, mo.count, fslot, bslot);
  const uint32_t blk = order ? order[bslot] : bslot;
  float4* __restrict__ slab = mo.out[fslot];
  DevCounters* __restrict__ dc = mo.dc[fslot];
  const uint32_t frame = mo.frame[fslot];
  uint32_t lt, sub;
  if (!block_to_tile(blk, tm, lt, sub)) return;
  const uint32_t wt = sub * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
  int px, py;
  uint32_t si;
  const bool in_image = wave_pixel(tm, lt, wt, lane, px, py, si);

  DvrRay r{};
  if (in_image) r = dvr_setup(p, v, px, py, frame);
  bool alive = in_image && r.hit;
  const uint32_t n_rays = (uint32_t)__builtin_popcountll(__ballot(alive));

  const float scale = p.volume_density_scale, inv_maj = p.volume_inv_maj, maj = p.volume_maj;
  const float sr0 = p.sample_range[0], sr1 = p.sample_range[1];
  const float lenf = (float)tf_len;
  const int last = (int)tf_len - 1;
  const uint32_t cbx = v.cq_bc[0], cby = v.cq_bc[1];
  const float4* __restrict__ cq = v.cq;
  const uint32_t cmaxx = v.extent[0] + 7u, cmaxy = v.extent[1] + 7u, cmaxz = v.extent[2] + 7u;
  const float ert = p.dvr_ert_tau;
  const uint32_t sh = 3u + v.skip_level, md0 = v.skip_dims[0], md1 = v.skip_dims[1];

  float Cx = 0.f, Cy = 0.f, Cz = 0.f, T = 1.0f, tau = 0.0f, kf = 0.0f;  // kf: per-lane step index
  uint32_t n_samples = 0, n_slots = 0, n_skipped = 0, n_batches = 0, n_tf = 0;    // wave-uniform
  uint32_t spread_wave = 0, spread_quad = 0;                            // PROBE only

  while (true) {
    {
      alive = alive && (kf < r.n);
      if (__ballot(alive) == 0ull) break;
    }
    n_batches += 1u;
    // ---- phase 1: addresses + gathers of U consecutive steps --------------------------------
    float4 q0[U], q1[U];
    float fx[U], fy[U], fz[U];
    bool ok[U];
    bool last_empty = false;
    float lqx = 0.f, lqy = 0.f, lqz = 0.f;
    uint32_t lcx = 0, lcy = 0, lcz = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float ku = kf + (float)u;
      bool in = alive && (ku < r.n);
      // A5 on the cellquad layout: cell (floor(p-0.5)) + 1 -> apron brick / local cell
      float qx = fma_(ku, r.dq.x, r.q0.x);
      float qy = fma_(ku, r.dq.y, r.q0.y);
      float qz = fma_(ku, r.dq.z, r.q0.z);
      float flx = floorf(qx), fly = floorf(qy), flz = floorf(qz);
      fx[u] = qx - flx; fy[u] = qy - fly; fz[u] = qz - flz;
      uint32_t cx = (uint32_t)((int)flx + 1), cy = (uint32_t)((int)fly + 1), cz = (uint32_t)((int)flz + 1);
      // inside the clipped AABB these are always in the lattice; the clamp only makes a stray or
      // already finished ray read defined memory instead of faulting
      cx = cx < cmaxx ? cx : cmaxx; cy = cy < cmaxy ? cy : cmaxy; cz = cz < cmaxz ? cz : cmaxz;
      bool eval = in;
      if (SKIP) {
        uint32_t mi = ((cz >> sh) * md1 + (cy >> sh)) * md0 + (cx >> sh);
        bool empty = (mask_lds[mi >> 5] >> (mi & 31u)) & 1u;
        eval = in && !empty;
        n_skipped += (uint32_t)__builtin_popcountll(__ballot(in && empty));
        if (u == U - 1) {
          last_empty = in && empty;
          lqx = qx; lqy = qy; lqz = qz;
          lcx = cx; lcy = cy; lcz = cz;
        }
      }
      ok[u] = eval;
      // quad index in 32 bits (24-bit multiplies; < 2^32 quads = 64 GiB), one 64-bit shift-add
      uint32_t b = mad24_s(mad24_s(cz >> 3, cby, cy >> 3), cbx, cx >> 3);
      uint32_t cell = cq_cell(cx & 7u, cy & 7u, cz & 7u);
      uint32_t o = mad24_s(b, CQ_BRICK_QUADS, cell);  // b < 2^24 bricks
      asm volatile("" : "+v"(o));  // keep the index math unconditional: a select, not a branch
      // unconditional loads (finished / skipped lanes read quad 0): no control flow between the
      // 2*U gathers, so they are all in flight before the first s_waitcnt
      o = eval ? o : 0u;
      const float4* qp = cq + o;
      q0[u] = qp[0];
      q1[u] = qp[cq_next_slice(cz)];
      if (PROBE) {
        const uint32_t line = o >> 3;  // 8 quads of 16 bytes per 128-byte line
        unsigned long long rem = ~0ull;
        while (rem) {
          uint32_t first = (uint32_t)__builtin_ctzll(rem);
          uint32_t val = (uint32_t)__shfl((int)line, (int)first, 64);
          rem &= ~__ballot(line == val);
          spread_wave += 1u;
        }
        const uint32_t qb = lane & ~3u, kq = lane & 3u;
        uint32_t l0 = (uint32_t)__shfl((int)line, (int)qb, 64), l1 = (uint32_t)__shfl((int)line, (int)(qb + 1u), 64),
                 l2 = (uint32_t)__shfl((int)line, (int)(qb + 2u), 64);
        bool first_in_quad = !((kq > 0u && line == l0) || (kq > 1u && line == l1) || (kq > 2u && line == l2));
        spread_quad += (uint32_t)__builtin_popcountll(__ballot(first_in_quad));
      }
    }
    // ---- phase 2: interpolate, classify, composite in order -- straight-line code: a lane that
    // does not contribute (finished, skipped, alpha == 0, out of the sample range) runs the same
    // instructions with alpha = 0, which leaves tau, T and C bit-for-bit unchanged
    // (fma(0,dt,tau) == tau and its colour increment is selected to 0).
#pragma unroll
    for (int u = 0; u < U; ++u) {
      bool a = ok[u] && alive;  // a lane that terminated earlier in this batch drops out
      unsigned long long m = __ballot(a);
      n_samples += (uint32_t)__builtin_popcountll(m);
      n_slots += m ? 64u : 0u;
      float wx = 1.0f - fx[u], wy = 1.0f - fy[u], wz = 1.0f - fz[u];
      float lx0 = fma_(q0[u].y, fx[u], q0[u].x * wx);
      float lx1 = fma_(q0[u].w, fx[u], q0[u].z * wx);
      float hx0 = fma_(q1[u].y, fx[u], q1[u].x * wx);
      float hx1 = fma_(q1[u].w, fx[u], q1[u].z * wx);
      float l = fma_(lx1, fy[u], lx0 * wy);
      float h = fma_(hx1, fy[u], hx0 * wy);
      float d = scale * fma_(h, fz[u], l * wz);
      float dn = d * inv_maj;
      // A7: NEAREST LUT, range test
      int ti = med3_i32((int)(dn * lenf), 0, last);  // dn >= 0: truncation == floor
      bool in_range = !(dn < sr0 || dn > sr1);
      n_tf += (uint32_t)__builtin_popcountll(__ballot(a && in_range));
      float4 rgba = tf_lds[ti];
      float alpha = (a && in_range) ? rgba.w : 0.0f;
      bool contrib = alpha > 0.0f;
      // tau += a*maj*dt; C += (T_prev - T) * rgb   (raymarch.glsl:43 / SURVEY A12)
      tau = fma_(alpha * maj, r.dt, tau);
      float Tn = __builtin_amdgcn_exp2f(tau * -1.4426950408889634f);
      float dT = contrib ? T - Tn : 0.0f;
      Cx = fma_(dT, rgba.x, Cx);
      Cy = fma_(dT, rgba.y, Cy);
      Cz = fma_(dT, rgba.z, Cz);
      bool done = contrib && (tau >= ert);
      T = contrib ? (done ? 0.0f : Tn) : T;
      alive = alive && !done;
    }
    kf += (float)U;
    if (SKIP && last_empty) {
      // steps from the batch's last sample to the faces of its macro cell, along the ray (dq = index units per step)
      const float Sf = (float)(1u << sh);
      float bx = (float)((lcx >> sh) << sh) - 1.0f, by = (float)((lcy >> sh) << sh) - 1.0f,
            bz = (float)((lcz >> sh) << sh) - 1.0f;  // q in [b, b + S) inside the macro cell
      float dx = r.dq.x > 0.0f ? (bx + Sf - lqx) / r.dq.x : (r.dq.x < 0.0f ? (bx - lqx) / r.dq.x : 3.0e38f);
      float dy = r.dq.y > 0.0f ? (by + Sf - lqy) / r.dq.y : (r.dq.y < 0.0f ? (by - lqy) / r.dq.y : 3.0e38f);
      float dz = r.dq.z > 0.0f ? (bz + Sf - lqz) / r.dq.z : (r.dq.z < 0.0f ? (bz - lqz) / r.dq.z : 3.0e38f);
      float n = floorf(fminf(dx, fminf(dy, dz))) - 2.0f;  // stay >= one whole step short of the exit face
      n = fminf(n, 1048576.0f);
      if (n >= 1.0f) kf += n;
    }
  }

  if (PROBE) {
    add_counts(dc, n_samples, spread_wave, spread_quad, n_skipped, 0u, n_slots, blk, n_batches * (uint32_t)(2 * U));
    return;
  }
  if (in_image) dvr_store(p, v, r, Cx, Cy, Cz, T, weight, slab, si);
  const uint32_t n_px = (uint32_t)__builtin_popcountll(__ballot(in_image));
  add_counts(dc, n_samples, n_rays, n_px, n_skipped, 0u, n_slots, blk, n_batches * (uint32_t)(2 * U), 0u, n_tf);
}

VXD int wave_min_i32(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    int w = __shfl_xor(v, o, 64);
    v = w < v ? w : v;
  }
  return v;
}

// ---------------------------------------------------------------------------------------------
// Depth-parallel variant: one wave = 8 rays (an 8x1 pixel row) x 8 consecutive march steps.
//
// Why: the 8x8-pixel kernel above keeps 64 rays in lock step, so a wave that looks through the
// whole clip box runs ~1500 dependent iterations (~0.35 ms even on an idle GPU).  That chain is the
// floor of the frame time and it does not shrink when the image is sharded over more GPUs
// (tools/shard_probe.py: 1.9x at 8 shards).  Here lane (r, j) evaluates step kb+j of ray r: the 64
// samples of an iteration are 8 steps of 8 rays, the longest chain is 8x shorter and there are 8x
// more waves to spread.  Density, TF and alpha of the 8 steps are independent; only the optical
// depth is a recurrence, tau_k = fma(alpha_k*maj, dt, tau_{k-1}).  It is evaluated in EXACTLY that
// order with an 8-step DPP chain (row_shr:1 + fma), so tau, the termination decision and the
// sample count stay bit-identical to the sequential oracle; the colour sum is re-associated
// (per-lane partial sums, reduced once per ray: differences ~1e-7, no decision depends on it).
VXD float dpp_shr1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, false));
}
template <int SH>
VXD float dpp_shr(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x110 + SH, 0xf, 0xf, false));
}

// block b -> (local tile, wave-tile, half): 128 blocks per 64x64 tile, whole tiles round-robin over XCDs
VXD bool block_to_tile_dp(uint32_t b, const TileMap& tm, uint32_t& lt, uint32_t& wt, uint32_t& half) {
  lt = (b >> 10) * 8u + (b & 7u);
  uint32_t sub = (b >> 3) & 127u;
  wt = sub >> 1;
  half = sub & 1u;
  return lt < tm.tiles_per_shard;
}

template <bool SKIP>
__global__ __launch_bounds__(256) void render_dvr_dp(const VxParams p, const DevVolume v,
                                                      const float4* __restrict__ tf_global,
                                                      uint32_t tf_len, float4* __restrict__ slab,
                                                      uint32_t frame, float weight, const TileMap tm,
                                                      DevCounters* __restrict__ dc) {
  extern __shared__ float4 tf_lds[];
  uint32_t* mask_lds = reinterpret_cast<uint32_t*>(tf_lds + tf_len);
  for (uint32_t i = threadIdx.x; i < tf_len; i += blockDim.x) tf_lds[i] = tf_global[i];
  if (SKIP)
    for (uint32_t i = threadIdx.x; i < v.skip_words; i += blockDim.x) mask_lds[i] = v.skip_bits[i];
  __syncthreads();
  uint32_t lt, wt, half;
  if (!block_to_tile_dp(blockIdx.x, tm, lt, wt, half)) return;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t j = lane & 7u, ray = lane >> 3;            // depth slot, ray of this wave
  const uint32_t row = half * 4u + (threadIdx.x >> 6);      // pixel row inside the 8x8 wave tile
  const bool j0 = j == 0u;
  int px = 0, py = 0;
  uint32_t si = 0;
  bool in_image;
  {
    uint32_t t = tile_at(tm, lt);
    si = (lt * 64u + wt) * 64u + row * 8u + ray;
    in_image = t < tm.n_tiles;
    if (in_image) {
      uint32_t tx = t % tm.tiles_x, ty = t / tm.tiles_x;
      px = (int)(tx * 64u + morton_x(wt) * 8u + ray);
      py = (int)(ty * 64u + morton_x(wt >> 1) * 8u + row);
      in_image = (uint32_t)px < tm.W && (uint32_t)py < tm.H;
    }
  }
  DvrRay r{};
  if (in_image) r = dvr_setup(p, v, px, py, frame);   // the 8 lanes of a ray compute the same setup
  bool ray_alive = in_image && r.hit;
  const uint32_t n_rays = (uint32_t)__builtin_popcountll(__ballot(ray_alive && j0));

  const float scale = p.volume_density_scale, inv_maj = p.volume_inv_maj, maj = p.volume_maj;
  const float sr0 = p.sample_range[0], sr1 = p.sample_range[1];
  const float lenf = (float)tf_len;
  const int last = (int)tf_len - 1;
  const uint32_t cbx = v.cq_bc[0], cby = v.cq_bc[1];
  const float4* __restrict__ cq = v.cq;
  const uint32_t cmaxx = v.extent[0] + 7u, cmaxy = v.extent[1] + 7u, cmaxz = v.extent[2] + 7u;
  const float ert = p.dvr_ert_tau;
  const uint32_t sh = 3u + v.skip_level, md0 = v.skip_dims[0], md1 = v.skip_dims[1];
  const float jf = (float)j;

  float Cx = 0.f, Cy = 0.f, Cz = 0.f;   // per-lane partial colour sums
  float carry = 0.0f;                    // optical depth of the ray before this iteration's 8 steps
  float kb = 0.0f;                       // wave-uniform base step of the iteration
  uint32_t n_samples = 0, n_slots = 0, n_skipped = 0, n_tf = 0;

  while (true) {
    {
      ray_alive = ray_alive && (kb < r.n);   // first step of the group decides whether the ray goes on
      if (__ballot(ray_alive) == 0ull) break;
    }
    const float k = kb + jf;
    const bool in = ray_alive && (k < r.n);
    // A5 on the cellquad layout (as render_dvr_cq)
    float qx = fma_(k, r.dq.x, r.q0.x);
    float qy = fma_(k, r.dq.y, r.q0.y);
    float qz = fma_(k, r.dq.z, r.q0.z);
    float flx = floorf(qx), fly = floorf(qy), flz = floorf(qz);
    float fx = qx - flx, fy = qy - fly, fz = qz - flz;
    uint32_t cx = (uint32_t)((int)flx + 1), cy = (uint32_t)((int)fly + 1), cz = (uint32_t)((int)flz + 1);
    cx = cx < cmaxx ? cx : cmaxx; cy = cy < cmaxy ? cy : cmaxy; cz = cz < cmaxz ? cz : cmaxz;
    bool eval = in, empty = false;
    if (SKIP) {
      uint32_t mi = ((cz >> sh) * md1 + (cy >> sh)) * md0 + (cx >> sh);
      empty = (mask_lds[mi >> 5] >> (mi & 31u)) & 1u;
      eval = in && !empty;
    }
    uint32_t b = mad24(mad24(cz >> 3, cby, cy >> 3), cbx, cx >> 3);
    uint32_t cell = cq_cell(cx & 7u, cy & 7u, cz & 7u);
    uint32_t o = eval ? mad24(b, CQ_BRICK_QUADS, cell) : 0u;
    const float4* qp = cq + o;
    float4 q0 = qp[0];
    float4 q1 = qp[cq_next_slice(cz)];
    float wx = 1.0f - fx, wy = 1.0f - fy, wz = 1.0f - fz;
    float lx0 = fma_(q0.y, fx, q0.x * wx);
    float lx1 = fma_(q0.w, fx, q0.z * wx);
    float hx0 = fma_(q1.y, fx, q1.x * wx);
    float hx1 = fma_(q1.w, fx, q1.z * wx);
    float l = fma_(lx1, fy, lx0 * wy);
    float h = fma_(hx1, fy, hx0 * wy);
    float d = scale * fma_(h, fz, l * wz);
    float dn = d * inv_maj;
    int ti = (int)(dn * lenf);
    ti = ti > last ? last : ti;
    ti = ti < 0 ? 0 : ti;
    bool in_range = !(dn < sr0 || dn > sr1);
    float4 rgba = tf_lds[ti];
    float alpha = (eval && in_range) ? rgba.w : 0.0f;
    // ---- exact sequential optical depth over the ray's 8 steps (DPP chain) --------------------
    const float x = alpha * maj;
    float tau = carry;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      float prev = dpp_shr1(tau);
      prev = j0 ? carry : prev;
      tau = fma_(x, r.dt, prev);       // alpha == 0: tau = prev exactly
    }
    float tau_prev = dpp_shr1(tau);
    tau_prev = j0 ? carry : tau_prev;
    // a step counts (and contributes) while the ray had not terminated before it
    const bool valid = eval && (tau_prev < ert);
    {
      unsigned long long m = __ballot(valid);
      n_samples += (uint32_t)__builtin_popcountll(m);
      n_tf += (uint32_t)__builtin_popcountll(__ballot(valid && in_range));
      n_slots += m ? 64u : 0u;
      if (SKIP) n_skipped += (uint32_t)__builtin_popcountll(__ballot(in && empty && (tau_prev < ert)));
    }
    if (__ballot(valid && alpha > 0.0f) != 0ull) {
      float Tn = __builtin_amdgcn_exp2f(tau * -1.4426950408889634f);
      float Tp = __builtin_amdgcn_exp2f(tau_prev * -1.4426950408889634f);
      float dT = (valid && alpha > 0.0f) ? Tp - Tn : 0.0f;
      Cx = fma_(dT, rgba.x, Cx);
      Cy = fma_(dT, rgba.y, Cy);
      Cz = fma_(dT, rgba.z, Cz);
    }
    // carry = tau of the ray's last slot (monotone, so it also tells whether the ray terminated)
    carry = __shfl(tau, (int)(lane | 7u), 64);
    ray_alive = ray_alive && (carry < ert);
    kb += 8.0f;
    if (SKIP) {
      // all live rays ended this group in an empty macro cell: advance every ray by the smallest
      // safe jump of their last slots (each last slot knows how far its own macro cell extends)
      const bool last_slot = (j == 7u);
      bool can = !ray_alive || !last_slot || (in && empty);
      if (__ballot(!can) == 0ull && __ballot(ray_alive) != 0ull) {
        const float Sf = (float)(1u << sh);
        float bx = (float)((cx >> sh) << sh) - 1.0f, by = (float)((cy >> sh) << sh) - 1.0f,
              bz = (float)((cz >> sh) << sh) - 1.0f;
        float dx = r.dq.x > 0.0f ? (bx + Sf - qx) / r.dq.x : (r.dq.x < 0.0f ? (bx - qx) / r.dq.x : 3.0e38f);
        float dy = r.dq.y > 0.0f ? (by + Sf - qy) / r.dq.y : (r.dq.y < 0.0f ? (by - qy) / r.dq.y : 3.0e38f);
        float dz = r.dq.z > 0.0f ? (bz + Sf - qz) / r.dq.z : (r.dq.z < 0.0f ? (bz - qz) / r.dq.z : 3.0e38f);
        float n = floorf(fminf(dx, fminf(dy, dz))) - 2.0f;
        n = (ray_alive && last_slot) ? fminf(n, 1048576.0f) : 1048576.0f;
        int ni = wave_min_i32((int)fmaxf(n, 0.0f));
        if (ni >= 1) kb += (float)ni;
      }
    }
  }

  // ---- per-ray colour: sum of the 8 lanes' partial sums, lands in slot 7 -----------------------
#pragma unroll
  for (int sft = 1; sft < 8; sft <<= 1) {
    float ax = sft == 1 ? dpp_shr<1>(Cx) : (sft == 2 ? dpp_shr<2>(Cx) : dpp_shr<4>(Cx));
    float ay = sft == 1 ? dpp_shr<1>(Cy) : (sft == 2 ? dpp_shr<2>(Cy) : dpp_shr<4>(Cy));
    float az = sft == 1 ? dpp_shr<1>(Cz) : (sft == 2 ? dpp_shr<2>(Cz) : dpp_shr<4>(Cz));
    bool take = j >= (uint32_t)sft;
    Cx += take ? ax : 0.0f;
    Cy += take ? ay : 0.0f;
    Cz += take ? az : 0.0f;
  }
  const bool writer = in_image && j == 7u;
  if (writer) {
    float T = (carry >= ert) ? 0.0f : __builtin_amdgcn_exp2f(carry * -1.4426950408889634f);
    dvr_store(p, v, r, Cx, Cy, Cz, T, weight, slab, si);
  }
  const uint32_t n_px = (uint32_t)__builtin_popcountll(__ballot(writer));
  add_counts(dc, n_samples, n_rays, n_px, n_skipped, 0u, n_slots, 0xffffffffu, 0u, 0u, n_tf);
}

inline void launch_dvr_cq_multi(const VxParams& p, const DevVolume& v, const float4* tf, uint32_t tf_len,
                                const MultiOut& mo, float weight, const TileMap& tm, hipStream_t stream,
                                const uint32_t* order, bool probe = false);

inline void launch_dvr_cq(const VxParams& p, const DevVolume& v, const float4* tf, uint32_t tf_len,
                          float4* slab, uint32_t frame, float weight, const TileMap& tm,
                          DevCounters* dc, hipStream_t stream, const uint32_t* order) {
  MultiOut mo{};
  mo.count = 1;
  mo.out[0] = slab;
  mo.dc[0] = dc;
  mo.frame[0] = frame;
  launch_dvr_cq_multi(p, v, tf, tf_len, mo, weight, tm, stream, order);
}

inline void launch_dvr_cq_multi(const VxParams& p, const DevVolume& v, const float4* tf, uint32_t tf_len,
                                const MultiOut& mo, float weight, const TileMap& tm, hipStream_t stream,
                                const uint32_t* order, bool probe) {
  float4* slab = mo.out[0];
  DevCounters* dc = mo.dc[0];
  const uint32_t frame = mo.frame[0];
  uint32_t groups = (tm.tiles_per_shard + 7u) / 8u;
  static const int dp = [] { const char* e = getenv("VX_DVR_DP"); return e ? atoi(e) : -1; }();
  // depth-parallel waves only on request (VX_DVR_DP=1): they cut the longest dependent chain 8x
  // (8 shards: 0.19 ms instead of 0.28 ms per GPU) but cost ~1.6x the instructions (0.80 vs 0.51 ms
  // on one GPU) and re-associate the colour sum, which gives up bit-identity across shard counts
  if (dp == 1 && !probe) {
    dim3 grid(groups * 1024u), block(256);
    const bool skip = p.dvr_skip_empty && v.skip_bits;
    size_t lds = (size_t)tf_len * sizeof(float4) + (skip ? (size_t)v.skip_words * 4u : 0u);
    if (skip)
      hipLaunchKernelGGL((render_dvr_dp<true>), grid, block, lds, stream, p, v, tf, tf_len, slab, frame, weight, tm, dc);
    else
      hipLaunchKernelGGL((render_dvr_dp<false>), grid, block, lds, stream, p, v, tf, tf_len, slab, frame, weight, tm, dc);
    return;
  }
  static const int unroll = [] { const char* e = getenv("VX_DVR_UNROLL"); return e ? atoi(e) : 4; }();
  dim3 grid(groups * 128u * mo.count), block(256);
  const bool skip = p.dvr_skip_empty && v.skip_bits;
  size_t lds = (size_t)tf_len * sizeof(float4) + (skip ? (size_t)v.skip_words * 4u : 0u);
#define VX_LAUNCH(UU, SS) \
  hipLaunchKernelGGL((render_dvr_cq<UU, SS>), grid, block, lds, stream, p, v, tf, tf_len, mo, weight, tm, order)
  if (probe) {
    if (skip)
      hipLaunchKernelGGL((render_dvr_cq<4, true, true>), grid, block, lds, stream, p, v, tf, tf_len, mo, weight, tm, order);
    else
      hipLaunchKernelGGL((render_dvr_cq<4, false, true>), grid, block, lds, stream, p, v, tf, tf_len, mo, weight, tm, order);
    return;
  }
  if (skip) {
    switch (unroll) {
      case 1: VX_LAUNCH(1, true); break;
      case 2: VX_LAUNCH(2, true); break;
      default: VX_LAUNCH(4, true); break;
    }
  } else {
    switch (unroll) {
      case 1: VX_LAUNCH(1, false); break;
      case 2: VX_LAUNCH(2, false); break;
      default: VX_LAUNCH(4, false); break;
    }
  }
#undef VX_LAUNCH
}

}  // namespace vx
