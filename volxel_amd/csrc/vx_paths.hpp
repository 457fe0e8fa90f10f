// vx_paths.hpp -- the reference's path-traced render modes (default = DDA, no_dda, raymarch) with the segments of
// a path split and re-packed ("wavefront" form), fragment.frag:79-124.
//
// The one-pixel-per-lane form (render_generic) runs trace_path as the shader does: primary segment, then -- only
// in the lanes whose ray found a collision -- the shadow segment of the next-event estimate, then the next
// bounce.  rocprofv3 on it (profiles/r01_generic_modes.txt): 36 % (default) and 48 % (no_dda) of the VALU lane
// slots do work; the shadow segments are half of the time and run with the lanes that missed switched off.
//
// Here a 256-thread workgroup owns its 16x16 pixels for the whole path, but between segments the live paths are
// re-packed through LDS: every lane whose primary segment ended in a collision writes its path record (slab slot,
// xoshiro state, throughput, position, direction, radiance so far: 17 dwords) to the next free slot -- ballot +
// prefix count inside a wave, one LDS counter per wave across waves, no global atomics, no second launch -- and
// the lanes 0 .. n-1 then run the shadow segments (and, with more bounces, scattering and the next primary
// segment) on dense waves.  A path record carries its pixel's RNG state, so every pixel consumes exactly the
// stream of draws of fragment.frag:79-124 / dda.glsl:21-98 / normal.glsl:6-57 / raymarch.glsl:8-55 whichever
// lane runs it: images and sample counts are bit-identical to render_generic and to the oracle.
#pragma once
#include "vx_kernels.hpp"

namespace vx {

struct PathLds {
  uint32_t si[256];        // slab slot of the path's pixel
  uint32_t rng[4][256];    // xoshiro state
  float f[12][256];        // throughput, origin, direction, radiance
  uint32_t wave_count[4];  // collisions per wave of the current round
};

// occupancy asked of the register allocator: these kernels are latency bound (one dependent chain of look-ups per
// ray), 8 resident waves per SIMD beat 4-5 by 30 % (default mode 0.92 / 0.73 / 0.64 ms at 1 / 6 / 8)
#ifndef VX_W_PATHS
#define VX_W_PATHS 8
#endif

template <int MODE, int LAYOUT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(VX_W_PATHS, 8))) void render_paths(
    const VxParams p, const DevVolume v, const float4* __restrict__ tf_global, uint32_t tf_len, const MultiOut mo,
    float weight, const TileMap tm) {
  extern __shared__ float4 tf_lds[];
  __shared__ PathLds q;
  TfView tf;
  tf.len = tf_len;
  tf.lenf = (float)tf_len;
  if (tf_len <= TF_LDS_MAX) {
    for (uint32_t i = threadIdx.x; i < tf_len; i += blockDim.x) tf_lds[i] = tf_global[i];
    tf.lut = tf_lds;
    tf.in_lds = true;
  } else {
    tf.lut = tf_global;
    tf.in_lds = false;
  }
  __syncthreads();
  uint32_t fslot, blk;
  multi_slot(blockIdx.x, mo.count, fslot, blk);
  float4* __restrict__ slab = mo.out[fslot];
  DevCounters* __restrict__ dc = mo.dc[fslot];
  const uint32_t frame = mo.frame[fslot];
  uint32_t lt, sub;
  if (!block_to_tile(blk, tm, lt, sub)) return;   // block uniform
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const uint32_t wt = sub * 4u + wave;
  int px, py;
  uint32_t si;
  const bool in_image = wave_pixel(tm, lt, wt, lane, px, py, si);
  Counts c{0, 0, 0, 0, 0};
  Frame<LAYOUT> fr{p, v, tf, c};

  // fragment.frag:158 for one finished path: out = w*prev + (1-w)*sanitize(result), alpha 1
  auto finish = [&](uint32_t slot, V3 L) {
    float4 prev = make_float4(0.f, 0.f, 0.f, 0.f);
    if (weight != 0.0f) prev = slab[slot];
    float4 o;
    o.x = fma_(1.0f - weight, sanitize1(L.x), weight * prev.x);
    o.y = fma_(1.0f - weight, sanitize1(L.y), weight * prev.y);
    o.z = fma_(1.0f - weight, sanitize1(L.z), weight * prev.z);
    o.w = 1.0f;
    slab[slot] = o;
  };
  // a path that left the volume: environment radiance with the MIS weight of its last scattering event
  // (fragment.frag:117-121)
  auto escape = [&](V3 L, V3 thr, V3 dir, uint32_t n_paths, float f_p) {
    if (p.show_environment > 0) {
      V3 Le = lookup_environment(p, v, dir);
      float pe = p.use_env > 0 ? pdf_environment(p, v, dir) : 0.0f;
      float mis = n_paths > 0u ? Frame<LAYOUT>::power_heuristic(f_p, pe) : 1.0f;
      L.x = fma_(thr.x * mis, Le.x, L.x);
      L.y = fma_(thr.y * mis, Le.y, L.y);
      L.z = fma_(thr.z * mis, Le.z, L.z);
    }
    return L;
  };

  // ---- set-up: every pixel's primary ray (fragment.frag:128-156) -----------------------------------------------
  Rng s{0, 0, 0, 0};
  Ray ray{v3(0, 0, 0), v3(0, 0, 1)};
  V3 L = v3(0, 0, 0), thr = v3(1, 1, 1);
  float t = 0.0f, f_p = 0.0f;
  uint32_t my_si = si;
  bool marching = in_image;   // this lane holds a path whose next segment is a primary (sample_volume) segment
  if (in_image) {
    s = seed_xoshiro(tea32(42u * (uint32_t)(py * p.res[0] + px), frame));   // :143-144
    float tex_x = ((float)px + 0.5f) / (float)p.res[0];
    float tex_y = ((float)py + 0.5f) / (float)p.res[1];
    float a0 = rng(s), a1 = rng(s), b0 = rng(s), b1 = rng(s);              // :146
    ray = setup_world_ray(p, tex_x, tex_y, (a0 + b0) / 2.0f, (a1 + b1) / 2.0f, &v);
    float near, far;
    if (fr.slab(ray, near, far)) c.rays++;
  }

  // ---- rounds: primary segment -> re-pack the collided paths -> NEE shadow segment -> scatter ------------------
  // (each segment kind appears once in the loop body: one copy of sample_volume and of transmittance in the code)
  const uint32_t bounces = (uint32_t)p.bounces;
  for (uint32_t n_paths = 0; n_paths < bounces; ++n_paths) {
    bool hit = false;
    if (marching) {
      hit = fr.template sample_volume<MODE>(ray, t, thr, s);
      if (!hit) finish(my_si, escape(L, thr, ray.d, n_paths, f_p));   // left the volume: fragment.frag:117-121
    }
    marching = false;
    const unsigned long long m = __ballot(hit);
    if (lane == 0) q.wave_count[wave] = (uint32_t)__builtin_popcountll(m);
    __syncthreads();
    uint32_t base = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < 4; ++w) {
      uint32_t n = q.wave_count[w];
      base += w < wave ? n : 0u;
      total += n;
    }
    if (total == 0u) break;   // workgroup uniform
    if (hit) {
      const uint32_t slot = base + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
      V3 o = madd3(ray.o, t, ray.d);   // fragment.frag:88: the collision point
      q.si[slot] = my_si;
      q.rng[0][slot] = s.x; q.rng[1][slot] = s.y; q.rng[2][slot] = s.z; q.rng[3][slot] = s.w;
      q.f[0][slot] = thr.x; q.f[1][slot] = thr.y; q.f[2][slot] = thr.z;
      q.f[3][slot] = o.x; q.f[4][slot] = o.y; q.f[5][slot] = o.z;
      q.f[6][slot] = ray.d.x; q.f[7][slot] = ray.d.y; q.f[8][slot] = ray.d.z;
      q.f[9][slot] = L.x; q.f[10][slot] = L.y; q.f[11][slot] = L.z;
    }
    __syncthreads();
    if (threadIdx.x < total) {
      const uint32_t i = threadIdx.x;
      my_si = q.si[i];
      s = Rng{q.rng[0][i], q.rng[1][i], q.rng[2][i], q.rng[3][i]};
      thr = v3(q.f[0][i], q.f[1][i], q.f[2][i]);
      ray.o = v3(q.f[3][i], q.f[4][i], q.f[5][i]);
      ray.d = v3(q.f[6][i], q.f[7][i], q.f[8][i]);
      L = v3(q.f[9][i], q.f[10][i], q.f[11][i]);
      // next-event estimate, fragment.frag:90-99
      float e0 = rng(s), e1 = rng(s);
      V3 w_i = v3(-p.light_dir[0], -p.light_dir[1], -p.light_dir[2]);
      float4 Le = make_float4(p.env_strength * 4.01f, p.env_strength * 4.01f, p.env_strength * 4.01f, 1.0f);
      if (p.use_env > 0) Le = sample_environment(p, v, e0, e1, w_i);
      const float pdf = Le.w;
      if (pdf > 0.0f) {
        f_p = Frame<LAYOUT>::phase_hg(dot3(neg3(ray.d), w_i), p.volume_phase_g);
        float mis = p.show_environment > 0 ? Frame<LAYOUT>::power_heuristic(pdf, f_p) : 1.0f;
        float Tr = fr.template transmittance<MODE>(Ray{ray.o, w_i}, s);
        L.x += thr.x * mis * f_p * Tr * Le.x / pdf;
        L.y += thr.y * mis * f_p * Tr * Le.y / pdf;
        L.z += thr.z * mis * f_p * Tr * Le.z / pdf;
      }
      bool ended = n_paths + 1u >= bounces;   // fragment.frag:101
      if (!ended) {
        float rr = Frame<LAYOUT>::luma(thr);   // russian roulette, :103-108
        if (rr < 0.1f) {
          float prob = 1.0f - rr;
          if (rng(s) < prob) ended = true;
          else {
            float qq = 1.0f - prob;
            thr = v3(thr.x / qq, thr.y / qq, thr.z / qq);
          }
        }
      }
      if (ended) {
        finish(my_si, L);
      } else {
        float u0 = rng(s), u1 = rng(s);        // scatter, :111-113
        V3 sd = Frame<LAYOUT>::sample_phase_hg(ray.d, p.volume_phase_g, u0, u1);
        f_p = Frame<LAYOUT>::phase_hg(dot3(neg3(ray.d), sd), p.volume_phase_g);
        ray.d = sd;
        marching = true;   // the next round starts with this path's next primary segment, on this (dense) lane
      }
    }
    __syncthreads();   // the records are consumed: the next round may overwrite them
  }
  flush_counts(dc, c, in_image ? 1u : 0u, blk);
}

}  // namespace vx
