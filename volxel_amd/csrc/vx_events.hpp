// vx_events.hpp -- the reference's `default` (hierarchical DDA + null collisions, dda.glsl) and `no_dda` (delta / ratio
// tracking, normal.glsl) render modes as a WAVE-PERSISTENT, EVENT-BATCHED path tracer (fragment.frag:79-124).
//
// Why: these modes run loops whose trip counts are random per ray (the free-flight distance is exponential: its
// standard deviation equals its mean).  With one pixel per lane for a whole path (render_generic) a wave waits for
// its slowest ray in every loop: rocprofv3 measured 37 % (default) and 48 % (no_dda) of the issued lane slots doing
// work, and no ordering of the pixels can change that -- the spread is the estimator's, not the image's.  Re-packing
// paths only BETWEEN segments (vx_paths.hpp) does not reach the idle slots either: they are inside the loops.
//
// MEASURED OUTCOME (config 3, 1080p, 32 frames per launch, bounces 1): lane utilisation of the
// march 0.36 -> 0.63 (default) and 0.43 -> 0.72 (no_dda), bit-identical images and counts -- and 0.66 / 1.28 ms per
// frame against render_generic's 0.46 / 0.72.  rocprofv3 (tools/events_pmc.sh): the event passes, the unified step's
// branches and the LDS-resident path state cost as many vector instructions as the denser march saves (6.46 -> 6.81 G
// and 8.86 -> 9.68 G per 16-frame launch), and the issue rate falls (0.68 -> 0.54, 0.58 -> 0.38 of the clocks): every
// step ends in a dependent chain rng -> log -> position -> gather -> classify that only waves in flight hide, and this
// kernel needs 80 registers (6 waves per SIMD instead of 8), spills and LDS traffic for its dense lanes.
// It therefore ships as the opt-in form (VX_PATHS_KERNEL=events), tested for bit-identity; render_generic stays the
// default for these modes.
//
// Here a wave keeps its 64 lanes busy by construction:
//   * a wave owns its 8x8-pixel tile for up to VX_EV_FRAMES accumulation frames of the launch (a launch renders up
//     to 64 independent frames, vx_render_frames): 64 x frames paths, handed out from a wave-private counter;
//   * the march loop is ONE loop for both kinds of segment of a mode -- the collision search of sample_volume and
//     the transmittance estimate towards the light differ in a few selects -- so a lane can be in either kind
//     beside its neighbours;
//   * a lane whose segment ends parks (status PEND) and the wave keeps marching with the others; when VX_EV_BATCH
//     lanes are parked or idle (or nobody marches) the wave handles all their events at once, each kind under one
//     wave-uniform branch: collision -> light sample + shadow segment; shadow segment over -> radiance, Russian
//     roulette, scattering, next segment (or the pixel is written); idle -> next path of the tile (seed, camera ray,
//     slab test).  Event code is the expensive, divergent part of a path tracer (TEA seed, inverse matrices, the
//     environment warp, trigonometry): batched, it runs for 16+ lanes per pass instead of 1-2.
// Every path carries its own xoshiro state and consumes exactly the draws of fragment.frag:79-124 in their order, so
// pixels, sample counts and DDA step counts are bit-identical to render_generic -- hence to the oracle -- whichever
// lane runs them and in whatever order (tests/test_gpu_parity.py::test_repacked_path_kernel_is_bit_identical, kernel "events").
// Per-path state that only events touch (radiance, throughput, world ray, pixel) lives in LDS, one dword per lane
// and field: 88 bytes per lane.
#pragma once
#include "vx_kernels.hpp"

namespace vx {

#ifndef VX_EV_FRAMES
#define VX_EV_FRAMES 8     // accumulation frames one wave works through (64 x this many paths per wave)
#endif
#ifndef VX_EV_BATCH
#define VX_EV_BATCH 16     // parked + idle lanes that trigger an event pass
#endif
#ifndef VX_EV_STRIDE
#define VX_EV_STRIDE 8     // march iterations between two looks at the lane census
#endif
#ifndef VX_W_EVENTS
#define VX_W_EVENTS 6
#endif

enum { EV_IDLE = 0, EV_MARCH = 1, EV_PEND = 2 };
// per-lane path state in LDS (field-major, 64 lanes per field per wave)
// L radiance so far, T throughput, O / D the path's current world ray, S = thr * mis * f_p and E = Le of the pending
// light sample (fragment.frag:97 is finished when its shadow segment is), PDF its pdf, FP the phase value of the last
// scattering event (for the MIS weight of an escaping path, :118), NP scattering events so far, PIX slab slot | frame slot << 26
enum { PF_LX = 0, PF_LY, PF_LZ, PF_TX, PF_TY, PF_TZ, PF_OX, PF_OY, PF_OZ, PF_DX, PF_DY, PF_DZ, PF_SX, PF_SY, PF_SZ,
       PF_EX, PF_EY, PF_EZ, PF_PDF, PF_FP, PF_NP, PF_PIX, PF_COUNT };

template <int MODE, int LAYOUT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(VX_W_EVENTS, 8))) void render_events(
    const VxParams p, const DevVolume v, const float4* __restrict__ tf_global, uint32_t tf_len,
    float4* __restrict__ out0, uint64_t out_stride, DevCounters* __restrict__ dc0, uint64_t dc_stride, uint32_t frame0,
    uint32_t n_frames, float weight, const TileMap tm) {
  static_assert(MODE == VX_MODE_DEFAULT || MODE == VX_MODE_NO_DDA, "event kernel: default and no_dda");
  constexpr bool DDA = MODE == VX_MODE_DEFAULT;
  extern __shared__ float4 lds_raw[];
  TfView tf;
  tf.len = tf_len;
  tf.lenf = (float)tf_len;
  tf.lut = lds_raw;
  tf.in_lds = true;
  for (uint32_t i = threadIdx.x; i < tf_len; i += blockDim.x) lds_raw[i] = tf_global[i];
  __syncthreads();
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  float* st = reinterpret_cast<float*>(lds_raw + tf_len) + wave * (PF_COUNT * 64u) + lane;   // field f at st[f * 64]
  auto ld = [&](int f) { return st[f * 64]; };
  auto sto = [&](int f, float x) { st[f * 64] = x; };
  auto ldu = [&](int f) { return __builtin_bit_cast(uint32_t, st[f * 64]); };
  auto stu = [&](int f, uint32_t x) { st[f * 64] = __builtin_bit_cast(float, x); };

  const uint32_t groups = (n_frames + VX_EV_FRAMES - 1u) / VX_EV_FRAMES;
  const uint32_t grp = blockIdx.x % groups, blk = blockIdx.x / groups;
  uint32_t lt, sub;
  if (!block_to_tile(blk, tm, lt, sub)) return;   // block uniform
  const uint32_t wt = sub * 4u + wave;
  const uint32_t fs0 = grp * VX_EV_FRAMES;
  const uint32_t nf = (n_frames - fs0) < (uint32_t)VX_EV_FRAMES ? (n_frames - fs0) : (uint32_t)VX_EV_FRAMES;
  const uint32_t total = 64u * nf;   // paths of this wave: item i = pixel (i & 63) of frame slot fs0 + (i >> 6)
  uint32_t next = 0;                 // wave uniform

  Counts c{0, 0, 0, 0, 0};
  Frame<LAYOUT> fr{p, v, tf, c};
  uint32_t n_slots = 0, n_active = 0;   // wave uniform

  // ---- the segment a lane is marching (registers) ------------------------------------------------------------------
  int status = EV_IDLE;
  bool shadow = false;          // kind of the segment: collision search (sample_volume) or transmittance
  bool ended_hit = false;       // PEND: the collision search found a collision
  Rng s{0, 0, 0, 0};
  V3 ipos = v3(0, 0, 0), idir = v3(0, 0, 1), ri = v3(0, 0, 0);
  float t = 0.f, far = 0.f, Tr = 1.f, tau = 0.f, mip = 3.f;
  uint32_t steps = 0;           // DDA steps of a shadow segment (dda.glsl:18,33) / iteration guard
  V3 hit_rgb = v3(0, 0, 0);     // TF colour at the collision (raymarch-independent: dda.glsl:91-92, normal.glsl:50-51)

  // start a segment along world ray (o, d); false: the ray misses the clip box (or the first free flight leaves it)
  auto begin_segment = [&](V3 o, V3 d, bool is_shadow) {
    shadow = is_shadow;
    Tr = 1.0f;
    steps = 0;
    float near;
    const Ray ray{o, d};
    if (!fr.slab(ray, near, far)) return false;
    to_index(p, ray, ipos, idir);
    if (DDA) {
      ri = v3(1.0f / idir.x, 1.0f / idir.y, 1.0f / idir.z);          // dda.glsl:24,68
      t = near + 1e-6f;
      tau = neg_log_one_minus(rng(s));
      mip = 3.0f;
    } else {
      t = fma_(neg_log_one_minus(rng(s)), p.volume_inv_maj, near);        // normal.glsl:13,40
    }
    return true;
  };

  // ---- one march iteration for the lanes in EV_MARCH ---------------------------------------------------------------
  // no_dda: normal.glsl:14-30 (transmittance_simple) and :41-56 (sample_volume_simple), one iteration
  // default: dda.glsl:33-60 (transmittanceDDA) and :76-97 (sample_volumeDDA), one DDA step
  auto march_step = [&]() {
    // (A straight-line form of this step -- selects on lane predicates, one wave-uniform branch around the sample,
    // conditional RNG draws as selects of the state -- was measured too: 1.55 instead of 1.28 ms per frame on no_dda.)
    const bool on = status == EV_MARCH;
    const unsigned long long om = ballot(on);
    n_slots += 64u;
    n_active += (uint32_t)__builtin_popcountll(om);
    if (on) {
      bool alive = t < far;
      if (DDA && shadow) alive = alive && (steps < 100u);             // dda.glsl:33 (the count advances only past t < far)
      if (!DDA || !shadow) alive = alive && (steps < LOOP_GUARD);     // the reference loops are unbounded
      if (!alive) {
        status = EV_PEND;
        ended_hit = false;
      } else {
        steps++;
        bool candidate = true;
        float majorant = 0.0f;
        if (DDA) {
          const V3 curr = madd3(ipos, t, idir);
          const int m = Frame<LAYOUT>::round_mip(mip);
          const V3 cell = Frame<LAYOUT>::dda_cell(curr, m);
          majorant = fr.local_majorant(cell, m);
          const float dt = Frame<LAYOUT>::step_dda(curr, cell, ri, m);
          c.skips++;
          t += dt;
          tau = fma_(-majorant, dt, tau);
          mip = gl_min(mip + 0.25f, 3.0f);
          candidate = !(tau > 0.0f);
          if (candidate) {
            t += tau / majorant;
            if (t >= far) {               // dda.glsl:43,84: break
              candidate = false;
              status = EV_PEND;
              ended_hit = false;
            }
          }
        }
        if (candidate) {
          const float4 rgba = fr.transfer(fr.trilinear(madd3(ipos, t, idir)) * p.volume_inv_maj);
          c.samples++;
          const float d = p.volume_maj * rgba.w;
          if (!shadow) {
            const bool real = DDA ? (rng(s) * majorant < d) : (rng(s) < d * p.volume_inv_maj);
            if (real) {
              hit_rgb = v3(rgba.x, rgba.y, rgba.z);
              status = EV_PEND;
              ended_hit = true;
            }
          } else {
            bool touched = true;
            if (DDA) {
              touched = rng(s) * majorant < d;
              if (touched) Tr *= gl_max(0.0f, 1.0f - p.volume_maj / majorant);   // quirk Q9
            } else {
              Tr *= fma_(-d, p.volume_inv_maj, 1.0f);
            }
            if (touched && Tr < 0.1f) {                               // dda.glsl:51-56, normal.glsl:22-26
              const float prob = 1.0f - Tr;
              if (rng(s) < prob) {
                Tr = 0.0f;
                status = EV_PEND;
                ended_hit = false;
              } else {
                Tr /= 1.0f - prob;
              }
            }
          }
          if (status == EV_MARCH) {
            if (DDA) {
              tau = neg_log_one_minus(rng(s));                             // dda.glsl:58,94
              mip = gl_max(0.0f, mip - 2.0f);
            } else {
              t = fma_(neg_log_one_minus(rng(s)), p.volume_inv_maj, t);    // normal.glsl:28,54
            }
          }
        }
      }
    }
  };

  // fragment.frag:158 for one finished path: out = w*prev + (1-w)*sanitize(result), alpha 1
  auto finish = [&](V3 L) {
    const uint32_t pix = ldu(PF_PIX);
    float4* __restrict__ slab = out0 + (uint64_t)(pix >> 26) * out_stride;
    const uint32_t slot = pix & 0x3ffffffu;
    float4 prev = make_float4(0.f, 0.f, 0.f, 0.f);
    if (weight != 0.0f) prev = slab[slot];
    float4 o;
    o.x = fma_(1.0f - weight, sanitize1(L.x), weight * prev.x);
    o.y = fma_(1.0f - weight, sanitize1(L.y), weight * prev.y);
    o.z = fma_(1.0f - weight, sanitize1(L.z), weight * prev.z);
    o.w = 1.0f;
    slab[slot] = o;
    status = EV_IDLE;
  };
  // a path that left the volume: environment radiance with the MIS weight of its last scattering event (:117-121)
  auto escape_and_finish = [&]() {
    V3 L = v3(ld(PF_LX), ld(PF_LY), ld(PF_LZ));
    if (p.show_environment > 0) {
      const V3 thr = v3(ld(PF_TX), ld(PF_TY), ld(PF_TZ));
      const V3 dir = v3(ld(PF_DX), ld(PF_DY), ld(PF_DZ));
      const uint32_t n_paths = ldu(PF_NP);
      const V3 Le = lookup_environment(p, v, dir);
      const float pe = p.use_env > 0 ? pdf_environment(p, v, dir) : 0.0f;
      const float mis = n_paths > 0u ? Frame<LAYOUT>::power_heuristic(ld(PF_FP), pe) : 1.0f;
      L.x = fma_(thr.x * mis, Le.x, L.x);
      L.y = fma_(thr.y * mis, Le.y, L.y);
      L.z = fma_(thr.z * mis, Le.z, L.z);
    }
    finish(L);
  };
  // start the collision search along the path's current ray; a ray that misses the box escapes at once
  auto begin_primary = [&]() {
    const V3 o = v3(ld(PF_OX), ld(PF_OY), ld(PF_OZ)), d = v3(ld(PF_DX), ld(PF_DY), ld(PF_DZ));
    if (begin_segment(o, d, false)) status = EV_MARCH;
    else escape_and_finish();
  };

  const uint32_t bounces = (uint32_t)p.bounces;
  // ---- the event pass ------------------------------------------------------------------------------------------------
  auto events = [&]() {
    // (A) collision searches that ended
    const bool a_on = status == EV_PEND && !shadow;
    if (ballot(a_on) != 0ull) {
      if (a_on) {
        if (!ended_hit) {
          escape_and_finish();                                           // fragment.frag:117-121
        } else {
          // throughput *= albedo * TF colour (dda.glsl:91-92 / normal.glsl:50-51: the same products in both orders'
          // own rounding -- dda multiplies by the albedo first, normal by the colour times the albedo)
          V3 thr = v3(ld(PF_TX), ld(PF_TY), ld(PF_TZ));
          if (DDA) {
            thr.x *= p.volume_albedo[0]; thr.y *= p.volume_albedo[1]; thr.z *= p.volume_albedo[2];
            thr.x *= hit_rgb.x; thr.y *= hit_rgb.y; thr.z *= hit_rgb.z;
          } else {
            thr.x *= hit_rgb.x * p.volume_albedo[0];
            thr.y *= hit_rgb.y * p.volume_albedo[1];
            thr.z *= hit_rgb.z * p.volume_albedo[2];
          }
          sto(PF_TX, thr.x); sto(PF_TY, thr.y); sto(PF_TZ, thr.z);
          const V3 d = v3(ld(PF_DX), ld(PF_DY), ld(PF_DZ));
          // the world-space parameter of the collision is the index-space one (to_index is linear in t)
          const V3 o = madd3(v3(ld(PF_OX), ld(PF_OY), ld(PF_OZ)), t, d);  // fragment.frag:89
          sto(PF_OX, o.x); sto(PF_OY, o.y); sto(PF_OZ, o.z);
          const float e0 = rng(s), e1 = rng(s);                          // rng2 of sample_environment, :92
          V3 w_i = v3(-p.light_dir[0], -p.light_dir[1], -p.light_dir[2]);
          float4 Le = make_float4(p.env_strength * 4.01f, p.env_strength * 4.01f, p.env_strength * 4.01f, 1.0f);
          if (p.use_env > 0) Le = sample_environment(p, v, e0, e1, w_i);
          const float pdf = Le.w;
          if (pdf > 0.0f) {
            const float f_p = Frame<LAYOUT>::phase_hg(dot3(neg3(d), w_i), p.volume_phase_g);
            const float mis = p.show_environment > 0 ? Frame<LAYOUT>::power_heuristic(pdf, f_p) : 1.0f;
            // L += thr * mis * f_p * Tr * Le / pdf (:97), evaluated left to right: the factors before Tr now, the rest
            // when the shadow segment has produced Tr
            sto(PF_SX, thr.x * mis * f_p); sto(PF_SY, thr.y * mis * f_p); sto(PF_SZ, thr.z * mis * f_p);
            sto(PF_EX, Le.x); sto(PF_EY, Le.y); sto(PF_EZ, Le.z);
            sto(PF_PDF, pdf);
            // transmittance(Ray(o, w_i)): a ray that misses the box has Tr = 1 and draws nothing (dda.glsl:23, normal.glsl:8)
            status = begin_segment(o, w_i, true) ? EV_MARCH : EV_PEND;
            ended_hit = false;
          } else {
            // no light sample (:93 not taken): the path goes on as after a shadow segment that adds nothing
            shadow = true;
            status = EV_PEND;
            ended_hit = true;               // for (B): nothing to add
          }
        }
      }
    }
    // (B) shadow segments that ended: radiance of the light sample, then the path's next step (:97-113)
    const bool b_on = status == EV_PEND && shadow;
    if (ballot(b_on) != 0ull) {
      if (b_on) {
        V3 L = v3(ld(PF_LX), ld(PF_LY), ld(PF_LZ));
        if (!ended_hit) {
          const float pdf = ld(PF_PDF);
          L.x += ld(PF_SX) * Tr * ld(PF_EX) / pdf;
          L.y += ld(PF_SY) * Tr * ld(PF_EY) / pdf;
          L.z += ld(PF_SZ) * Tr * ld(PF_EZ) / pdf;
        }
        const uint32_t n_paths = ldu(PF_NP) + 1u;
        stu(PF_NP, n_paths);
        bool go_on = n_paths < bounces;                                  // :101
        V3 thr = v3(ld(PF_TX), ld(PF_TY), ld(PF_TZ));
        if (go_on) {
          const float rr = Frame<LAYOUT>::luma(thr);
          if (rr < 0.1f) {                                               // :103-108
            const float prob = 1.0f - rr;
            if (rng(s) < prob) go_on = false;
            else {
              const float q = 1.0f - prob;
              thr = v3(thr.x / q, thr.y / q, thr.z / q);
              sto(PF_TX, thr.x); sto(PF_TY, thr.y); sto(PF_TZ, thr.z);
            }
          }
        }
        if (!go_on) {
          finish(L);                                                     // free_path = false: no environment term
        } else {
          sto(PF_LX, L.x); sto(PF_LY, L.y); sto(PF_LZ, L.z);
          const V3 d = v3(ld(PF_DX), ld(PF_DY), ld(PF_DZ));
          const float u0 = rng(s), u1 = rng(s);                          // :111
          const V3 sd = Frame<LAYOUT>::sample_phase_hg(d, p.volume_phase_g, u0, u1);
          sto(PF_FP, Frame<LAYOUT>::phase_hg(dot3(neg3(d), sd), p.volume_phase_g));
          sto(PF_DX, sd.x); sto(PF_DY, sd.y); sto(PF_DZ, sd.z);
          begin_primary();
        }
      }
    }
    // (C) idle lanes take the wave's next paths (fragment.frag:128-156)
    const bool c_on = status == EV_IDLE;
    const unsigned long long cm = ballot(c_on);
    if (cm != 0ull && next < total) {
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(cm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cm, 0u));
      const uint32_t item = next + rank;
      const uint32_t taken = (uint32_t)__builtin_popcountll(cm);
      next = next + taken < total ? next + taken : total;
      if (c_on && item < total) {
        const uint32_t fs = fs0 + (item >> 6);
        int px, py;
        uint32_t si;
        const bool in_image = wave_pixel(tm, lt, wt, item & 63u, px, py, si);
        if (in_image) {
          stu(PF_PIX, (fs << 26) | si);
          s = seed_xoshiro(tea32(42u * (uint32_t)(py * p.res[0] + px), frame0 + fs));   // :143-144
          const float tex_x = ((float)px + 0.5f) / (float)p.res[0];
          const float tex_y = ((float)py + 0.5f) / (float)p.res[1];
          const float a0 = rng(s), a1 = rng(s), b0 = rng(s), b1 = rng(s);               // :146
          const Ray ray = setup_world_ray(p, tex_x, tex_y, (a0 + b0) / 2.0f, (a1 + b1) / 2.0f, &v);
          sto(PF_LX, 0.f); sto(PF_LY, 0.f); sto(PF_LZ, 0.f);
          sto(PF_TX, 1.f); sto(PF_TY, 1.f); sto(PF_TZ, 1.f);
          sto(PF_OX, ray.o.x); sto(PF_OY, ray.o.y); sto(PF_OZ, ray.o.z);
          sto(PF_DX, ray.d.x); sto(PF_DY, ray.d.y); sto(PF_DZ, ray.d.z);
          sto(PF_FP, 0.f);
          stu(PF_NP, 0u);
          {
            float near_, far_;
            if (fr.slab(ray, near_, far_)) c.rays++;                    // what render_generic counts as a ray
          }
          begin_primary();
        }
      }
    }
  };

  // ---- main loop -------------------------------------------------------------------------------------------------------
  while (true) {
    const unsigned long long mm = ballot(status == EV_MARCH);
    const uint32_t waiting = 64u - (uint32_t)__builtin_popcountll(mm);
    const bool work = next < total;
    const unsigned long long pm = ballot(status == EV_PEND);
    if (mm == 0ull && pm == 0ull && !work) break;
    const bool can_fill = work && ballot(status == EV_IDLE) != 0ull;
    if ((pm != 0ull || can_fill) && (waiting >= (uint32_t)VX_EV_BATCH || mm == 0ull)) events();
#pragma unroll 1
    for (int k = 0; k < VX_EV_STRIDE; ++k) {
      if (ballot(status == EV_MARCH) == 0ull) break;
      march_step();
    }
  }

  // pixels: every item of the wave that lies in the image was written once
  uint32_t n_pixels;
  {
    int px, py;
    uint32_t si;
    n_pixels = wave_sum(wave_pixel(tm, lt, wt, lane, px, py, si) ? nf : 0u);
  }
  DevCounters* __restrict__ dc = dc0 + (uint64_t)fs0 * dc_stride;
  const uint32_t s_ = wave_sum(c.samples), r_ = wave_sum(c.rays), k_ = wave_sum(c.skips), t_ = wave_sum(c.tf);
  add_counts(dc, s_, r_, n_pixels, k_, 0u, n_slots, blk, 0u, 0u, t_, n_active);
}

}  // namespace vx
