// vx_modes.hpp -- the march loops and the per-pixel driver (fragment.frag main) for gfx950.
#pragma once
#include "vx_device.hpp"

namespace vx {

constexpr uint32_t LOOP_GUARD = 1u << 20;

template <int LAYOUT>
struct Frame {
  const VxParams& p;
  const DevVolume& v;
  TfView tf;
  Counts& c;

  VXD float transfer_alpha(float d) const {   // the .a of transfer(d), counted the same
    if (!(d < p.sample_range[0] || d > p.sample_range[1])) c.tf++;
    return lookup_transfer_alpha(tf, p.sample_range[0], p.sample_range[1], d);
  }
  VXD float4 transfer(float d) const {
    if (!(d < p.sample_range[0] || d > p.sample_range[1])) c.tf++;
    return lookup_transfer(tf, p.sample_range[0], p.sample_range[1], d);
  }
  // the samples of the march loops lie inside the clipped volume box (t in [near, far) of the slab test), hence inside
  // the apron lattice of the cellquad layout
  VXD float trilinear(V3 ip) const {
    return lookup_density_trilinear<LAYOUT, true>(v, p.volume_density_scale, ip);
  }
  // lookup_density_stochastic, common.glsl:56-58,72-76
  VXD float density_stochastic(V3 ipos, Rng& s) const {
    int tap[3];
    stochastic_tricubic_filter(ipos, s, tap);
    return p.volume_density_scale * lookup_density_nearest<LAYOUT>(v, tap[0], tap[1], tap[2]);
  }
  VXD bool slab(const Ray& r, float& near, float& far) const {
    return ray_box_intersection(r, p.volume_aabb_min, p.volume_aabb_max, near, far);
  }

  // ---- A10/A11 RAYMARCH: sampling/raymarch.glsl ------------------------------------
  VXD float transmittance_raymarch(const Ray& ray, Rng& s) const {  // raymarch.glsl:8-23
    float near, far;
    if (!slab(ray, near, far)) return 1.0f;
    V3 ipos, idir;
    to_index(p, ray, ipos, idir);
    float dt = (far - near) / 64.0f;
    near = fma_(rng(s), dt, near);
    float tau = 0.0f;
    float kf = 0.0f;
#pragma unroll 1
    for (int i = 0; i < 64; ++i, kf += 1.0f) {
      float t = gl_min(fma_(kf, dt, near), far);
      float d = density_stochastic(madd3(ipos, t, idir), s);
      float a = transfer_alpha(d * p.volume_inv_maj);
      tau = fma_(a * p.volume_maj, dt, tau);
    }
    c.samples += 64u;
    return expf(-tau);
  }
  VXD bool sample_raymarch(const Ray& ray, float& t, V3& thr, Rng& s) const {  // :25-55
    float near, far;
    if (!slab(ray, near, far)) return false;
    V3 ipos, idir;
    to_index(p, ray, ipos, idir);
    float tau_target = neg_log_one_minus(rng(s));
    float dt = (far - near) / 64.0f;
    near = fma_(rng(s), dt, near);
    float tau = 0.0f, kf = 0.0f;
#pragma unroll 1
    for (int i = 0; i < 64; ++i, kf += 1.0f) {
      t = gl_min(fma_(kf, dt, near), far);
      const float dn = density_stochastic(madd3(ipos, t, idir), s) * p.volume_inv_maj;
      const float a = transfer_alpha(dn);
      tau = fma_(a * p.volume_maj, dt, tau);
      if (tau >= tau_target) {
        c.samples += (uint32_t)i + 1u;
        const float4 rgba = lookup_transfer(tf, p.sample_range[0], p.sample_range[1], dn);   // the colour, once
        thr.x *= rgba.x * p.volume_albedo[0];
        thr.y *= rgba.y * p.volume_albedo[1];
        thr.z *= rgba.z * p.volume_albedo[2];
        return true;
      }
    }
    c.samples += 64u;
    return false;
  }

  // ---- A13 default mode: sampling/dda.glsl ------------------------------------------
  // The cell of the range-texture level `mip` that holds `pos` -- floor(pos / dim), dim = 8 << mip -- serves both halves of
  // a DDA step: stepDDA's floor(pos * inv_dim) (dda.glsl:13) and the majorant look-up's ivec3(floor(pos)) >> (3 + mip)
  // (common.glsl:50-53) are the same integer, because dim is a power of two (the product is exact) and
  // floor(floor(x) / 2^k) == floor(x / 2^k).  One floor per axis instead of two, no shifts.
  VXD static V3 dda_cell(V3 pos, int mip) {
    // 1 / dim: dim is a power of two, its reciprocal is the float with exponent -(3 + mip) -- no division sequence
    const float inv_dim = __builtin_bit_cast(float, (uint32_t)(124 - mip) << 23);
    return v3(floorf(pos.x * inv_dim), floorf(pos.y * inv_dim), floorf(pos.z * inv_dim));
  }
  VXD static float step_dda(V3 pos, V3 cell, V3 inv_dir, int mip) {  // dda.glsl:11-16
    float dim = (float)(8 << mip);
    float ox = (inv_dir.x >= 0.0f) ? dim + 0.5f : -0.5f;
    float oy = (inv_dir.y >= 0.0f) ? dim + 0.5f : -0.5f;
    float oz = (inv_dir.z >= 0.0f) ? dim + 0.5f : -0.5f;
    float tx = ((cell.x * dim + ox) - pos.x) * inv_dir.x;
    float ty = ((cell.y * dim + oy) - pos.y) * inv_dir.y;
    float tz = ((cell.z * dim + oz) - pos.z) * inv_dir.z;
    return gl_min(tx, gl_min(ty, tz));
  }
  // dda.glsl:36,78: u_volume_maj * lookup_transfer(lookup_majorant(curr, mip) * inv_maj).a.  The value is a pure
  // function of the range-texture cell, the transfer function and four uniforms: vx_api tabulates it per cell with
  // these same operations (build_local_majorants), so a DDA step costs one load instead of the dependent chain
  // level dimensions -> range texel -> LUT entry.
  VXD float local_majorant(V3 cell, int mip) const {
    const uint32_t bx = (uint32_t)f2i(cell.x), by = (uint32_t)f2i(cell.y), bz = (uint32_t)f2i(cell.z);
    const bool in = bx < v.bc[0] && by < v.bc[1] && bz < v.bc[2];
    // 24-bit multiplies: at most 128 bricks per axis (brick.rs:77-81), four levels
    const uint32_t i = mad24_s(mad24_s(mad24_s((uint32_t)mip, v.bc[2], bz), v.bc[1], by), v.bc[0], bx);
    return v.lmaj[in ? i : v.lmaj_cells];   // the entry after the last level holds the value outside the grid
  }
  // round(mip), half away from zero (quirk Q12): mip is a multiple of 1/4 in [0, 3], so floor(mip + 1/2) is the same
  VXD static int round_mip(float mip) { return f2i(mip + 0.5f); }
  VXD float transmittance_dda(const Ray& ray, Rng& s) const {  // dda.glsl:21-62
    float near, far;
    if (!slab(ray, near, far)) return 1.0f;
    V3 ipos, idir;
    to_index(p, ray, ipos, idir);
    V3 ri = v3(1.0f / idir.x, 1.0f / idir.y, 1.0f / idir.z);
    float t = near + 1e-6f, Tr = 1.0f, tau = neg_log_one_minus(rng(s)), mip = 3.0f;
    uint32_t step = 0;
    while (t < far && (step++ < 100u)) {
      V3 curr = madd3(ipos, t, idir);
      int m = round_mip(mip);
      const V3 cell = dda_cell(curr, m);
      float majorant = local_majorant(cell, m);
      float dt = step_dda(curr, cell, ri, m);
      c.skips++;
      t += dt;
      tau = fma_(-majorant, dt, tau);
      mip = gl_min(mip + 0.25f, 3.0f);
      if (tau > 0.0f) continue;
      t += tau / majorant;
      if (t >= far) break;
      float4 rgba = transfer(trilinear(madd3(ipos, t, idir)) * p.volume_inv_maj);
      c.samples++;
      float d = p.volume_maj * rgba.w;
      if (rng(s) * majorant < d) {
        Tr *= gl_max(0.0f, 1.0f - p.volume_maj / majorant);  // quirk Q9
        if (Tr < 0.1f) {
          float prob = 1.0f - Tr;
          if (rng(s) < prob) return 0.0f;
          Tr /= 1.0f - prob;
        }
      }
      tau = neg_log_one_minus(rng(s));
      mip = gl_max(0.0f, mip - 2.0f);
    }
    return Tr;
  }
  VXD bool sample_dda(const Ray& ray, float& t, V3& thr, Rng& s) const {  // dda.glsl:65-98
    float near, far;
    if (!slab(ray, near, far)) return false;
    V3 ipos, idir;
    to_index(p, ray, ipos, idir);
    V3 ri = v3(1.0f / idir.x, 1.0f / idir.y, 1.0f / idir.z);
    t = near + 1e-6f;
    float tau = neg_log_one_minus(rng(s)), mip = 3.0f;
    uint32_t guard = 0;  // the reference loop is unbounded; a wedged wave would hang the GPU
    while (t < far && guard++ < LOOP_GUARD) {
      V3 curr = madd3(ipos, t, idir);
      int m = round_mip(mip);
      const V3 cell = dda_cell(curr, m);
      float majorant = local_majorant(cell, m);
      float dt = step_dda(curr, cell, ri, m);
      c.skips++;
      t += dt;
      tau = fma_(-majorant, dt, tau);
      mip = gl_min(mip + 0.25f, 3.0f);
      if (tau > 0.0f) continue;
      t += tau / majorant;
      if (t >= far) break;
      float4 rgba = transfer(trilinear(madd3(ipos, t, idir)) * p.volume_inv_maj);
      c.samples++;
      float d = p.volume_maj * rgba.w;
      if (rng(s) * majorant < d) {
        thr.x *= p.volume_albedo[0]; thr.y *= p.volume_albedo[1]; thr.z *= p.volume_albedo[2];
        thr.x *= rgba.x; thr.y *= rgba.y; thr.z *= rgba.z;
        return true;
      }
      tau = neg_log_one_minus(rng(s));
      mip = gl_max(0.0f, mip - 2.0f);
    }
    return false;
  }

  // ---- A14 NO_DDA: sampling/normal.glsl ---------------------------------------------
  VXD float transmittance_simple(const Ray& ray, Rng& s) const {  // normal.glsl:6-31
    float near, far;
    if (!slab(ray, near, far)) return 1.0f;
    V3 ipos, idir;
    to_index(p, ray, ipos, idir);
    float t = fma_(neg_log_one_minus(rng(s)), p.volume_inv_maj, near), Tr = 1.0f;
    uint32_t guard = 0;
    while (t < far && guard++ < LOOP_GUARD) {
      const float a = transfer_alpha(trilinear(madd3(ipos, t, idir)) * p.volume_inv_maj);
      c.samples++;
      float d = p.volume_maj * a;
      Tr *= fma_(-d, p.volume_inv_maj, 1.0f);
      if (Tr < 0.1f) {
        float prob = 1.0f - Tr;
        if (rng(s) < prob) return 0.0f;
        Tr /= 1.0f - prob;
      }
      t = fma_(neg_log_one_minus(rng(s)), p.volume_inv_maj, t);
    }
    return Tr;
  }
  VXD bool sample_simple(const Ray& ray, float& t, V3& thr, Rng& s) const {  // :33-57
    float near, far;
    if (!slab(ray, near, far)) return false;
    V3 ipos, idir;
    to_index(p, ray, ipos, idir);
    t = fma_(neg_log_one_minus(rng(s)), p.volume_inv_maj, near);
    // One exit, at the bottom (the reference's loop leaves from its middle on a collision and from its top at the far
    // face: the compiler answers two exits with a register copy per carried value and trip).  The free flight of :55 is
    // drawn for every lane and kept by the lanes that go on, so a lane that collides leaves with the state and the t
    // of :48-52.
    bool hit = false;
    float dn = 0.0f;
    if (t < far) {
      uint32_t guard = 0;
      bool more;
#pragma unroll 1
      do {
        dn = trilinear(madd3(ipos, t, idir)) * p.volume_inv_maj;
        const float a = transfer_alpha(dn);
        c.samples++;
        const float d = p.volume_maj * a;
        const float p_real = d * p.volume_inv_maj;
        hit = rng(s) < p_real;
        Rng s2 = s;
        const float tn = fma_(neg_log_one_minus(rng(s2)), p.volume_inv_maj, t);
        s.x = hit ? s.x : s2.x; s.y = hit ? s.y : s2.y; s.z = hit ? s.z : s2.z; s.w = hit ? s.w : s2.w;
        t = hit ? t : tn;
        more = !hit & (t < far) & (++guard < LOOP_GUARD);
      } while (more);
    }
    if (!hit) return false;
    const float4 rgba = lookup_transfer(tf, p.sample_range[0], p.sample_range[1], dn);   // the colour, once
    thr.x *= rgba.x * p.volume_albedo[0];
    thr.y *= rgba.y * p.volume_albedo[1];
    thr.z *= rgba.z * p.volume_albedo[2];
    return true;
  }

  // sampling.glsl:11-44
  template <int MODE>
  VXD bool sample_volume(const Ray& ray, float& t, V3& thr, Rng& s) const {
    if (MODE == VX_MODE_NO_DDA) return sample_simple(ray, t, thr, s);
    if (MODE == VX_MODE_RAYMARCH) return sample_raymarch(ray, t, thr, s);
    return sample_dda(ray, t, thr, s);
  }
  template <int MODE>
  VXD float transmittance(const Ray& ray, Rng& s) const {
    if (MODE == VX_MODE_NO_DDA) return transmittance_simple(ray, s);
    if (MODE == VX_MODE_RAYMARCH) return transmittance_raymarch(ray, s);
    return transmittance_dda(ray, s);
  }

  // ---- utils.glsl helpers ---------------------------------------------------------------
  VXD static float sqr(float x) { return x * x; }
  VXD static float luma(V3 col) { return dot3(col, v3(0.212671f, 0.715160f, 0.072169f)); }
  VXD static float power_heuristic(float a, float b) { return sqr(a) / (sqr(a) + sqr(b)); }
  VXD static float phase_hg(float cos_t, float g) {  // utils.glsl:121-124
    const float inv_4pi = 1.0f / (4.0f * 3.14159265358979323846f);
    float denom = fma_(2.0f * g, cos_t, 1.0f + sqr(g));
    return inv_4pi * (1.0f - sqr(g)) / (denom * sqrtf(denom));
  }
  VXD static V3 align3(V3 N, V3 w) {  // utils.glsl:106-114
    V3 T;
    if (fabsf(N.x) > fabsf(N.y)) {
      float l = sqrtf(fma_(N.z, N.z, N.x * N.x));
      T = v3(-N.z / l, 0.0f / l, N.x / l);
    } else {
      float l = sqrtf(fma_(N.z, N.z, N.y * N.y));
      T = v3(0.0f / l, N.z / l, -N.y / l);
    }
    V3 B = cross3(N, T);
    return normalize3(v3(fma_(w.z, N.x, fma_(w.y, B.x, w.x * T.x)),
                         fma_(w.z, N.y, fma_(w.y, B.y, w.x * T.y)),
                         fma_(w.z, N.z, fma_(w.y, B.z, w.x * T.z))));
  }
  VXD static V3 sample_phase_hg(V3 dir, float g, float u0, float u1) {  // utils.glsl:133-139
    float cos_t;
    if (fabsf(g) < 1e-4f) {
      cos_t = fma_(-2.0f, u0, 1.0f);
    } else {
      float q = (1.0f - sqr(g)) / fma_(2.0f * g, u0, 1.0f - g);
      cos_t = ((1.0f + sqr(g)) - sqr(q)) / (2.0f * g);
    }
    float sin_t = sqrtf(gl_max(0.0f, 1.0f - sqr(cos_t)));
    float phi = 2.0f * 3.14159265358979323846f * u1;
    return align3(dir, v3(sin_t * cosf(phi), sin_t * sinf(phi), cos_t));
  }

  // ---- A15 trace_path, fragment.frag:79-124 (directional light, environment.glsl:30-33) --
  template <int MODE>
  VXD float4 trace_path(Ray ray, Rng& s) const {
    V3 L = v3(0, 0, 0), thr = v3(1, 1, 1);
    bool free_path = true;
    uint32_t n_paths = 0;
    float t = 0.0f, f_p = 0.0f;
    // trips of the march loops (Counts::it_p / it_s): DDA steps in the default mode, samples in the others
    auto trips = [&]() { return MODE == VX_MODE_DEFAULT ? c.skips : c.samples; };
    auto primary = [&]() {
      const uint32_t before = trips();
      const bool hit = sample_volume<MODE>(ray, t, thr, s);
      c.it_p += trips() - before;
      return hit;
    };
    while (primary()) {
      ray.o = madd3(ray.o, t, ray.d);
      float e0 = rng(s), e1 = rng(s);  // rng2 argument of sample_environment, fragment.frag:92
      V3 w_i = v3(-p.light_dir[0], -p.light_dir[1], -p.light_dir[2]);
      float4 Le = make_float4(p.env_strength * 4.01f, p.env_strength * 4.01f, p.env_strength * 4.01f, 1.0f);
      if (p.use_env > 0) Le = sample_environment(p, v, e0, e1, w_i);
      const float pdf = Le.w;
      if (pdf > 0.0f) {
        f_p = phase_hg(dot3(neg3(ray.d), w_i), p.volume_phase_g);
        float mis = p.show_environment > 0 ? power_heuristic(pdf, f_p) : 1.0f;
        const uint32_t before = trips();
        float Tr = transmittance<MODE>(Ray{ray.o, w_i}, s);
        c.it_s += trips() - before;
        L.x += thr.x * mis * f_p * Tr * Le.x / pdf;
        L.y += thr.y * mis * f_p * Tr * Le.y / pdf;
        L.z += thr.z * mis * f_p * Tr * Le.z / pdf;
      }
      if (++n_paths >= (uint32_t)p.bounces) { free_path = false; break; }
      float rr = luma(thr);
      if (rr < 0.1f) {
        float prob = 1.0f - rr;
        if (rng(s) < prob) { free_path = false; break; }
        float q = 1.0f - prob;
        thr = v3(thr.x / q, thr.y / q, thr.z / q);
      }
      float u0 = rng(s), u1 = rng(s);
      V3 sd = sample_phase_hg(ray.d, p.volume_phase_g, u0, u1);
      f_p = phase_hg(dot3(neg3(ray.d), sd), p.volume_phase_g);
      ray.d = sd;
    }
    if (free_path && p.show_environment > 0) {
      V3 Le = lookup_environment(p, v, ray.d);
      // pdf_environment is 0 for the delta light ([build]); power_heuristic(f_p, 0) = 1 or NaN(0/0)
      float pe = p.use_env > 0 ? pdf_environment(p, v, ray.d) : 0.0f;
      float mis = n_paths > 0u ? power_heuristic(f_p, pe) : 1.0f;
      L.x = fma_(thr.x * mis, Le.x, L.x);
      L.y = fma_(thr.y * mis, Le.y, L.y);
      L.z = fma_(thr.z * mis, Le.z, L.z);
    }
    return make_float4(L.x, L.y, L.z, gl_clamp((float)n_paths, 0.0f, 1.0f));
  }

  // ---- A12 [build] deterministic DVR (generic form; the tuned kernel is vx_dvr.hpp) ------
  template <bool PHONG>
  VXD float4 dvr(const Ray& ray, float start_offset) const {
    float near, far;
    V3 C = v3(0, 0, 0);
    float T = 1.0f;
    bool hit = slab(ray, near, far);
    if (hit) {
      c.rays++;
      V3 ipos, idir;
      to_index(p, ray, ipos, idir);
      float dt = p.dvr_step_voxels / sqrtf(dot3(idir, idir));
      float t0 = fma_(start_offset, dt, near);
      float tau = 0.0f;
      // [build] march contract: n samples, sample k at q = fma(k, dq, q0) in the cell frame
      const float xq = (far - t0) / dt;
      const float nf = (xq > 0.0f) ? fminf(ceilf(xq), (float)p.dvr_max_steps) : 0.0f;
      const V3 dq = v3(dt * idir.x, dt * idir.y, dt * idir.z);
      const V3 q0 = v3(fma_(t0, idir.x, ipos.x) - 0.5f, fma_(t0, idir.y, ipos.y) - 0.5f, fma_(t0, idir.z, ipos.z) - 0.5f);
      V3 nl = v3(-p.light_dir[0], -p.light_dir[1], -p.light_dir[2]);
      V3 hv = v3(0, 0, 0);
      if (PHONG) hv = normalize3(sub3(nl, ray.d));
      for (float kf = 0.0f; kf < nf; kf += 1.0f) {
        const float qx = fma_(kf, dq.x, q0.x), qy = fma_(kf, dq.y, q0.y), qz = fma_(kf, dq.z, q0.z);
        const float flx = floorf(qx), fly = floorf(qy), flz = floorf(qz);
        const float fx = qx - flx, fy = qy - fly, fz = qz - flz;
        const int cx = f2i(flx), cy = f2i(fly), cz = f2i(flz);
        if (p.dvr_skip_empty && v.skip_bits) {  // exact skipping: alpha would be exactly 0
          const int mx = cx + 1, my = cy + 1, mz = cz + 1;
          uint32_t sh = 3u + v.skip_level;
          if (mx >= 0 && my >= 0 && mz >= 0 && ((uint32_t)mx >> sh) < v.skip_dims[0] &&
              ((uint32_t)my >> sh) < v.skip_dims[1] && ((uint32_t)mz >> sh) < v.skip_dims[2] &&
              skip_test(v.skip_bits, sh, v.skip_dims[0], v.skip_dims[1], (uint32_t)mx, (uint32_t)my, (uint32_t)mz)) {
            c.skips++;
            continue;
          }
        }
        // lookup_density_trilinear, common.glsl:61-69, from the cell and fractions of q
        const float dens = trilinear_cell<LAYOUT>(v, p.volume_density_scale, cx, cy, cz, fx, fy, fz);
        float4 rgba = transfer(dens * p.volume_inv_maj);
        c.samples++;
        if (rgba.w > 0.0f) {
          if (PHONG) {
            c.grads++;
            // [build] central differences one voxel either side, in the sample's own cell frame (cells c +- e, the
            // sample's fractions): DESIGN.md section 2
            const float ds = p.volume_density_scale;
            float gx = trilinear_cell<LAYOUT>(v, ds, cx + 1, cy, cz, fx, fy, fz) - trilinear_cell<LAYOUT>(v, ds, cx - 1, cy, cz, fx, fy, fz);
            float gy = trilinear_cell<LAYOUT>(v, ds, cx, cy + 1, cz, fx, fy, fz) - trilinear_cell<LAYOUT>(v, ds, cx, cy - 1, cz, fx, fy, fz);
            float gz = trilinear_cell<LAYOUT>(v, ds, cx, cy, cz + 1, fx, fy, fz) - trilinear_cell<LAYOUT>(v, ds, cx, cy, cz - 1, fx, fy, fz);
            V3 g = v3(gx * p.density_transform_inv[0], gy * p.density_transform_inv[5],
                      gz * p.density_transform_inv[10]);
            phong_shade(p, g, nl, hv, rgba);
          }
          tau = fma_(rgba.w * p.volume_maj, dt, tau);
          float Tn = expf(-tau);
          float dT = T - Tn;
          C.x = fma_(dT, rgba.x, C.x);
          C.y = fma_(dT, rgba.y, C.y);
          C.z = fma_(dT, rgba.z, C.z);
          T = Tn;
          if (tau >= p.dvr_ert_tau) { T = 0.0f; break; }
        }
      }
    }
    V3 L = v3(C.x * p.dvr_gain[0], C.y * p.dvr_gain[1], C.z * p.dvr_gain[2]);
    if (p.show_environment > 0 && T > 0.0f) {
      V3 Le = lookup_environment(p, v, ray.d);
      L.x = fma_(T, Le.x, L.x);
      L.y = fma_(T, Le.y, L.y);
      L.z = fma_(T, Le.z, L.z);
    }
    return make_float4(L.x, L.y, L.z, hit ? 1.0f : 0.0f);
  }

  // ---- fragment.frag:128-158 for one pixel (without the running-mean blend) -------------
  template <int MODE>
  VXD float4 shade_pixel(int px, int py, uint32_t frame) const {
    Rng s = seed_xoshiro(tea32(42u * (uint32_t)(py * p.res[0] + px), frame));  // :143-144
    float tex_x = tex_coord(px, p.res[0], &v, 0);
    float tex_y = tex_coord(py, p.res[1], &v, 1);
    float a0 = rng(s), a1 = rng(s), b0 = rng(s), b1 = rng(s);  // :146
    float jx = (a0 + b0) / 2.0f, jy = (a1 + b1) / 2.0f;
    constexpr bool DVR = (MODE == VX_MODE_DVR || MODE == VX_MODE_DVR_PHONG);
    if (DVR && !p.dvr_jitter) { jx = 0.5f; jy = 0.5f; }
    Ray ray = setup_world_ray(p, tex_x, tex_y, jx, jy, &v);   // uniform terms from the host (DevVolume::cam_o)
    float4 r;
    if (p.debug_hits) {  // :147-153
      float near, far;
      if (slab(ray, near, far)) {
        V3 h = madd3(ray.o, near, ray.d);
        r = make_float4((h.x - p.volume_aabb_min[0]) / (p.volume_aabb_max[0] - p.volume_aabb_min[0]),
                        (h.y - p.volume_aabb_min[1]) / (p.volume_aabb_max[1] - p.volume_aabb_min[1]),
                        (h.z - p.volume_aabb_min[2]) / (p.volume_aabb_max[2] - p.volume_aabb_min[2]), 1.0f);
        c.rays++;
      } else {
        V3 bg = lookup_environment(p, v, ray.d);
        r = make_float4(bg.x, bg.y, bg.z, 1.0f);
      }
      return r;
    }
    if (DVR) {
      (void)rng(s);  // tau_target slot of raymarch.glsl:28
      float u_start = rng(s);
      r = dvr<MODE == VX_MODE_DVR_PHONG>(ray, p.dvr_jitter ? u_start : 0.5f);
    } else {
      float near, far;
      if (slab(ray, near, far)) c.rays++;
      r = trace_path<MODE>(ray, s);
    }
    return make_float4(sanitize1(r.x), sanitize1(r.y), sanitize1(r.z), sanitize1(r.w));
  }
};

}  // namespace vx
