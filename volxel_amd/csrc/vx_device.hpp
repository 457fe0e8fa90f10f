// vx_device.hpp -- gfx950 device functions of the Volxel ray loop.
//
// Restates, for CDNA4, the shader functions on the hot path:
//   random.glsl (A1/A2), sampling/common.glsl (A4-A7), utils.glsl (A8/A9/A19),
//   sampling/{raymarch,dda,normal}.glsl (A10-A14), fragment.frag (A15/A16).
// Arithmetic contract (DESIGN.md "Arithmetic contract"): binary32, -ffp-contract=off, every
// product term added to a running sum is one v_fma_f32, IEEE division / sqrt, GLSL min/max.
// Written independently of oracle/ (which is test infrastructure and not compiled here).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/volxel_hip.h"

#define VXD __device__ __forceinline__

namespace vx {

// ---------------------------------------------------------------------------------------
// device view of an uploaded volume
struct DevVolume {
  // reference layout: the three textures of viewer.ts:1106-1142 as linear buffers
  const uint32_t* indirection;  // 10-10-10 pointers            (brick.rs:30-35)
  const uint32_t* range;        // (f16 min << 16) | f16 max    (brick.rs:19-23)
  const uint8_t* atlas;         // u8 voxels, x fastest         (buf3d.rs:26-28)
  const uint32_t* mips[3];      // range mips, GL levels 1..3   (brick.rs:153-190)
  uint32_t bc[3];               // bricks per axis (= indirection = range dims)
  uint32_t atlas_size[3];
  uint32_t extent[3];           // padded index extent = bc*8   (brick.rs:236-238)
  uint32_t mip_size[3][3];
  // MI355X layout "cellquad": apron bricks of pre-decoded fp32 xy-quads
  const float4* cq;             // [(bc+1)^3][9 slices][8][8] float4
  uint32_t cq_bc[3];            // bc + 1
  // MI355X layout "brickf32": every 8^3 brick decoded to fp32, 2 KiB contiguous, brick-major
  const float* bf;              // [bc.z][bc.y][bc.x][8][8][8], one all-zero 16-byte chunk behind the last brick
  uint32_t bf_zero;             // index (in floats) of that chunk, or 0 when the layout needs more than 32 index bits
  // MI355X layout "bricku8" (opt-in): the same bricks as the atlas' 8-bit codes, 4 per dword, dword index = the
  // 16-byte-unit index of brickf32 (brick * 128 + z * 16 + y * 2 + (x >> 2)); one zero dword behind the last brick.
  // bu_range[b] = {min, max - min} of brick b (range texture, decoded from f16); the entry behind the last brick is
  // {0, 0}: it serves the zero dword, so rows and chunks outside the volume decode to 0 (A4).
  const uint32_t* bu;
  const float2* bu_range;
  uint32_t bu_active;           // this launch samples bricku8 (set by vx_api prepare_render)
  // exact empty-space skipping (DVR): one bit per macro cell of 8 << skip_level voxels
  const uint32_t* skip_bits;    // nullptr: none
  uint32_t skip_level;
  uint32_t skip_dims[3];        // (extent >> (3 + level)) + 1
  uint32_t skip_words;
  // default mode (A13): the local majorant maj * TF(scale * range max).a of every cell of range-texture levels 0..3
  // (dda.glsl:36,78), levels back to back, each at the strides of level 0; entry lmaj_cells = the value outside
  const float* lmaj;            // built by vx_api before a `default` launch (build_local_majorants)
  uint32_t lmaj_cells;          // 4 * bricks
  // environment map (environment.ts): RGBA32F texels in GL row order + importance mip pyramid
  const float4* env_tex;        // nullptr: none (directional light only)
  uint32_t env_w, env_h;
  const float* env_imp;         // levels 0..9 of the 512^2 map back to back (imp_offset)
  const float4* env_impq;       // the same levels 0..8 as 2x2 sibling quads, one 16-byte load per level
  float env_avg_w;              // level 9 (the mean importance)
  // wave-uniform terms of the primary ray, evaluated once per launch on the host with the device's own operations
  // (IEEE fma chains and divisions: vx_api.hip derive_camera) instead of once per wave on the vector ALUs -- the
  // reference hoists its matrix inverses the same way (quirk Q11).  Perspective camera only (an orthographic ray's
  // origin is per pixel); read by the tuned DVR kernels through dvr_setup.
  float cam_o[3];               // inverse(view) * (0,0,0,1), divided by w          (utils.glsl:25-27)
  float cam_ipos[3];            // density_transform_inv * (cam_o, 1)               (to_index of the origin)
  float inv_res[2];             // 1 / u_res
  // Round 4: two more per-ray divisions decided per launch on the host (vx_api.hip prepare_render), both exact:
  //  * RAY_AFFINE_VIEW: inverse(view) has the bottom row (0,0,0,1), so the w of inverse(view) * (v, 1) is fma(1, 1, 0 * ...)
  //    = 1.0 and the three divisions by it (utils.glsl:35-37) return their numerators;
  //  * RAY_TEX_BY_RECIPROCAL: (pixel + 0.5) / res over the whole image equals the quotient corrected once with the rounded
  //    reciprocal -- q0 = a * y, q = fma(fma(-res, q0, a), y, q0) -- for EVERY pixel coordinate of this resolution (the host
  //    tries all of them against the IEEE division; a resolution for which one differs keeps the division).
  uint32_t ray_flags;
};
constexpr uint32_t RAY_AFFINE_VIEW = 1u, RAY_TEX_BY_RECIPROCAL_X = 2u, RAY_TEX_BY_RECIPROCAL_Y = 4u;

constexpr uint32_t IMP_DIM = 512, IMP_LEVELS = 10, IMP_FLOATS = 349525;
// quads of level k (children of the texels of level k+1): (IMP_DIM >> (k+1))^2 float4, row major
VXD constexpr uint32_t impq_offset(uint32_t level) {
  uint32_t o = 0;
  for (uint32_t k = 0; k < level; ++k) o += (IMP_DIM >> (k + 1)) * (IMP_DIM >> (k + 1));
  return o;
}
constexpr uint32_t IMPQ_QUADS = 87381;
VXD constexpr uint32_t imp_offset(uint32_t level) {
  uint32_t o = 0;
  for (uint32_t k = 0; k < level; ++k) o += (IMP_DIM >> k) * (IMP_DIM >> k);
  return o;
}

// macro cell of trilinear cell c (given as c+1 >= 0): is it flagged empty?
VXD bool skip_test(const uint32_t* bits, uint32_t sh, uint32_t d0, uint32_t d1, uint32_t cx, uint32_t cy,
                   uint32_t cz) {
  uint32_t mi = ((cz >> sh) * d1 + (cy >> sh)) * d0 + (cx >> sh);
  return (bits[mi >> 5] >> (mi & 31u)) & 1u;
}

// cell order inside an apron brick: 9 z slices of 8x8 cells; inside a slice a 128-byte line (8 quads)
// holds 4 x 2 cells (x, y) -- in-line order x*2 + (y & 1), which costs the fewest bit operations --,
// lines x-fastest.  (Tried: 8x1 rows = the plain x-fastest order, and
// 2x2x2 blocks with z padded to 10 slices)
constexpr uint32_t CQ_SLICE_QUADS = 64;
constexpr uint32_t CQ_BRICK_QUADS = 9 * CQ_SLICE_QUADS;  // 576 float4 = 9216 B
VXD uint32_t cq_cell(uint32_t lx, uint32_t ly, uint32_t lz) {  // local cell (0..7, 0..7, 0..8) -> quad index
  return (lz << 6) | ((ly & 6u) << 3) | (lx << 1) | (ly & 1u);
}
VXD uint32_t cq_next_slice(uint32_t) { return CQ_SLICE_QUADS; }

// ---------------------------------------------------------------------------------------
// arithmetic helpers
VXD float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
VXD float gl_min(float x, float y) { return (y < x) ? y : x; }
VXD float gl_max(float x, float y) { return (x < y) ? y : x; }
VXD float gl_clamp(float x, float lo, float hi) { return gl_min(gl_max(x, lo), hi); }
// truncating, saturating, NaN -> 0: v_cvt_i32_f32 (C's (int)x is undefined out of range)
VXD int f2i(float x) {
  int r;
  asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}

// -log(1 - r) for a draw r in [0, 1): the argument 1 - r lies in [2^-24, 1], never denormal, zero, infinite or NaN, so
// logf's scaling of denormal arguments and its pass-through of infinities (seven instructions, 26 clocks at gfx950's
// rates -- tools/op_rate.hip -- per free flight) select nothing.  What is left is logf's own sequence, the hardware log2
// times ln 2 as a two-float product: the same bits for every argument in range.
VXD float neg_log_one_minus(float r) {
  const float y = __builtin_amdgcn_logf(1.0f - r);
  const float hi = y * 0x1.62e42ep-1f;                                             // 0x3f317217
  const float lo = __builtin_fmaf(y, 0x1.efa39ep-25f, __builtin_fmaf(y, 0x1.62e42ep-1f, -hi));   // 0x3377d1cf
  return -(hi + lo);
}

// lane mask of a per-lane condition (HIP's __ballot takes an int: the bool is first materialised as 0 / 1 and compared
// again, two vector instructions per use)
VXD unsigned long long ballot(bool b) { return __builtin_amdgcn_ballot_w64(b); }

// a*b + c on 24-bit unsigned operands: one full-rate v_mad_u32_u24 (hipcc otherwise picks the
// quarter-rate v_mad_u64_u32 for 32-bit index arithmetic)
VXD uint32_t mad24(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t r;
  asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// the same with a wave-uniform multiplier held in an SGPR (no v_mov per use)
VXD uint32_t mad24_s(uint32_t a, uint32_t b_uniform, uint32_t c) {
  uint32_t r;
  asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b_uniform), "v"(c));
  return r;
}
// clamp to [0, hi_uniform] in one instruction, the wave-uniform bound straight from its SGPR.  hi_uniform MUST be
// wave uniform (a kernel argument or derived from one): an "s" operand fed a divergent value is silently read from
// the first active lane.  Every call site passes a function of DevVolume / tf_len.
VXD int clamp0_i32(int x, int hi_uniform) {
  int r;
  asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "s"(hi_uniform));
  return r;
}
// clamp to [lo, hi] in one instruction
VXD int med3_i32(int x, int lo, int hi) {
  int r;
  asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(lo), "v"(hi));
  return r;
}

struct V3 {
  float x, y, z;
};
VXD V3 v3(float x, float y, float z) { return V3{x, y, z}; }
VXD float dot3(V3 a, V3 b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
VXD V3 sub3(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
VXD V3 scale3(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
VXD V3 neg3(V3 a) { return v3(-a.x, -a.y, -a.z); }
VXD V3 normalize3(V3 a) {
  float inv = 1.0f / sqrtf(dot3(a, a));
  return scale3(a, inv);
}
VXD V3 cross3(V3 a, V3 b) {
  return v3(fma_(-b.y, a.z, a.y * b.z), fma_(-b.z, a.x, a.z * b.x), fma_(-b.x, a.y, a.x * b.y));
}
VXD V3 madd3(V3 o, float t, V3 d) {
  return v3(fma_(t, d.x, o.x), fma_(t, d.y, o.y), fma_(t, d.z, o.z));
}
VXD void mat4_mul(const float* m, float x, float y, float z, float w, float out[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
    out[i] = fma_(m[12 + i], w, fma_(m[8 + i], z, fma_(m[4 + i], y, m[i] * x)));
}

// ---------------------------------------------------------------------------------------
// A1/A2: integer RNG, random.glsl:41-106 (bit exact)
VXD uint32_t tea32(uint32_t v0, uint32_t v1) {  // random.glsl:41-51 with N = 32
  uint32_t s0 = 0u;
#ifndef VX_TEA_UNROLL
#define VX_TEA_UNROLL 32   // straight line: the round constants fold into literals (6 instead of 7 vector instructions per half round; DVR -1 %)
#endif
  constexpr int kTeaUnroll = VX_TEA_UNROLL;
#pragma unroll kTeaUnroll
  for (int n = 0; n < 32; ++n) {
    s0 += 0x9e3779b9u;
    v0 += ((v1 << 4) + 0xA341316Cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xC8013EA4u);
    v1 += ((v0 << 4) + 0xAD90777Du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7E95761Eu);
  }
  return v0;
}
VXD uint32_t wang_hash(uint32_t x) {  // random.glsl:59-66
  x = (x ^ 61u) ^ (x >> 16);
  x *= 9u;
  x = x ^ (x >> 4);
  x *= 0x27d4eb2du;
  x = x ^ (x >> 15);
  return x;
}
struct Rng {
  uint32_t x, y, z, w;
};
VXD Rng seed_xoshiro(uint32_t seed) {  // random.glsl:69-76
  return Rng{wang_hash(seed), wang_hash(seed + 1u), wang_hash(seed + 2u), wang_hash(seed + 3u)};
}
VXD uint32_t rotl32(uint32_t v, uint32_t k) { return __builtin_rotateleft32(v, k); }
VXD uint32_t xoshiro_next(Rng& s) {  // random.glsl:80-94 (s.x + s.z: quirk Q1)
  uint32_t result = rotl32(s.x + s.z, 7u) + s.x;
  uint32_t t = s.y << 9;
  s.z ^= s.x;
  s.w ^= s.y;
  s.y ^= s.z;
  s.x ^= s.w;
  s.z ^= t;
  s.w = rotl32(s.w, 11u);
  return result;
}
VXD float rng(Rng& s) {  // random.glsl:103-106: exact u24 -> f32, times 2^-24
  return (float)(xoshiro_next(s) >> 8) * 5.9604644775390625e-08f;
}

// ---------------------------------------------------------------------------------------
// texel access on the reference layout
VXD float half_bits_to_float(uint32_t h) {
  _Float16 v = __builtin_bit_cast(_Float16, (uint16_t)h);
  return (float)v;
}

// R8 unorm -> float, f = c/255 correctly rounded (GL ES 3.0 2.1.6.1) without a divide:
// q = c*r, one Newton correction with the exact residual.  Verified for all 256 codes by
// tests/test_gpu_parity.py::test_unorm_table.
VXD float unorm8(uint32_t c) {
  float fc = (float)c;
  const float r = 1.0f / 255.0f;
  float q = fc * r;
  float e = fma_(-q, 255.0f, fc);
  return fma_(e, r, q);
}

// A4: lookup_density_brick, common.glsl:35-43; out-of-range taps are 0.
// Straight-line: the coordinates are clamped into the grid, every lane runs the range -> pointer -> atlas chain,
// and the bounds tests select the result.  (The first form returned early per tap; eight -- for the Phong gradient
// 56 -- nested divergent ifs per sample serialise the three dependent loads of every tap and gave the register
// allocator 112 join blocks per loop body to place spill code in: DESIGN.md section 5.3.)
VXD float lookup_density_brick(const DevVolume& v, int x, int y, int z) {
  const bool in = (uint32_t)x < v.extent[0] && (uint32_t)y < v.extent[1] && (uint32_t)z < v.extent[2];
  const uint32_t ux = (uint32_t)clamp0_i32(x, (int)v.extent[0] - 1), uy = (uint32_t)clamp0_i32(y, (int)v.extent[1] - 1),
                 uz = (uint32_t)clamp0_i32(z, (int)v.extent[2] - 1);
  const uint32_t bi = ((uz >> 3) * v.bc[1] + (uy >> 3)) * v.bc[0] + (ux >> 3);
  const uint32_t rg = v.range[bi];
  const float mn = half_bits_to_float(rg >> 16), mx = half_bits_to_float(rg & 0xffffu);
  const uint32_t ptr = v.indirection[bi];
  const uint32_t ax = ((ptr & 1023u) << 3) + (ux & 7u);
  const uint32_t ay = (((ptr >> 10) & 1023u) << 3) + (uy & 7u);
  const uint32_t az = (((ptr >> 20) & 1023u) << 3) + (uz & 7u);
  // WebGL2 robust texelFetch: outside the (pruned) atlas -> 0 (a constant brick keeps pointer 0: quirk Q6)
  // (such a lane reads byte 0: the atlas allocation is never empty, vx_upload_volume)
  const bool inz = az < v.atlas_size[2];
  const size_t ao = ((size_t)az * v.atlas_size[1] + ay) * v.atlas_size[0] + ax;
  float un = unorm8(v.atlas[inz ? ao : (size_t)0]);
  un = inz ? un : 0.0f;
  const float r = fma_(un, mx - mn, mn);
  return in ? r : 0.0f;
}

// the 8-bit code lookup_density_brick decodes for an in-extent voxel (0 where the robust fetch returns 0: unorm8(0) == 0),
// for the bricku8 build
VXD uint32_t lookup_code_brick(const DevVolume& v, uint32_t ux, uint32_t uy, uint32_t uz) {
  const uint32_t bi = ((uz >> 3) * v.bc[1] + (uy >> 3)) * v.bc[0] + (ux >> 3);
  const uint32_t ptr = v.indirection[bi];
  const uint32_t ax = ((ptr & 1023u) << 3) + (ux & 7u);
  const uint32_t ay = (((ptr >> 10) & 1023u) << 3) + (uy & 7u);
  const uint32_t az = (((ptr >> 20) & 1023u) << 3) + (uz & 7u);
  const bool inz = az < v.atlas_size[2];
  const size_t ao = ((size_t)az * v.atlas_size[1] + ay) * v.atlas_size[0] + ax;
  const uint32_t code = v.atlas[inz ? ao : (size_t)0];
  return inz ? code : 0u;
}
// four codes of one dword -> densities: A4's decode, fma(unorm8(code), max - min, min)
VXD float4 decode_codes4(uint32_t code4, float2 rg) {
  float4 r;
  r.x = fma_(unorm8(code4 & 255u), rg.y, rg.x);
  r.y = fma_(unorm8((code4 >> 8) & 255u), rg.y, rg.x);
  r.z = fma_(unorm8((code4 >> 16) & 255u), rg.y, rg.x);
  r.w = fma_(unorm8(code4 >> 24), rg.y, rg.x);
  return r;
}

VXD float gl_mix(float x, float y, float a) { return fma_(y, a, x * (1.0f - a)); }

enum { LAYOUT_REF = 0, LAYOUT_CQ = 1, LAYOUT_BF = 2 };

// decoded voxel from the brickf32 layout; out-of-range taps are 0; straight-line as above
VXD float bf_voxel(const DevVolume& v, int x, int y, int z) {
  const bool in = (uint32_t)x < v.extent[0] && (uint32_t)y < v.extent[1] && (uint32_t)z < v.extent[2];
  if (v.bf_zero != 0u) {   // wave uniform; every volume below 2^32 voxels of layout (about 1600^3)
    // a voxel outside the volume reads the zero chunk behind the last brick: one select on the index -- no clamps, no
    // select on the value (its index is formed from the raw coordinates, wraps, and is dropped)
    const uint32_t ux = (uint32_t)x, uy = (uint32_t)y, uz = (uint32_t)z;
    const uint32_t b = mad24_s(mad24_s(uz >> 3, v.bc[1], uy >> 3), v.bc[0], ux >> 3);
    const uint32_t l = ((uz & 7u) << 6) | ((uy & 7u) << 3) | (ux & 7u);
    return v.bf[in ? (b << 9) | l : v.bf_zero];
  }
  const uint32_t ux = (uint32_t)clamp0_i32(x, (int)v.extent[0] - 1), uy = (uint32_t)clamp0_i32(y, (int)v.extent[1] - 1),
                 uz = (uint32_t)clamp0_i32(z, (int)v.extent[2] - 1);
  const uint32_t b = ((uz >> 3) * v.bc[1] + (uy >> 3)) * v.bc[0] + (ux >> 3);
  const uint32_t l = ((uz & 7u) << 6) | ((uy & 7u) << 3) | (ux & 7u);
  const float r = v.bf[(size_t)b * 512u + l];
  return in ? r : 0.0f;
}

// A4 through the device layouts: the decoded value of voxel (x,y,z) is the .x tap of cellquad cell
// (x,y,z) / the brickf32 voxel -- one load instead of the range -> pointer -> atlas chain.  Same bits:
// both layouts were filled by lookup_density_brick.
template <int LAYOUT>
VXD float lookup_density_nearest(const DevVolume& v, int x, int y, int z) {
  if (LAYOUT == LAYOUT_CQ) {
    const bool in = (uint32_t)x < v.extent[0] && (uint32_t)y < v.extent[1] && (uint32_t)z < v.extent[2];
    const uint32_t cx = (uint32_t)clamp0_i32(x, (int)v.extent[0] - 1) + 1u, cy = (uint32_t)clamp0_i32(y, (int)v.extent[1] - 1) + 1u,
                   cz = (uint32_t)clamp0_i32(z, (int)v.extent[2] - 1) + 1u;
    const uint32_t b = ((cz >> 3) * v.cq_bc[1] + (cy >> 3)) * v.cq_bc[0] + (cx >> 3);
    const size_t o = (size_t)b * CQ_BRICK_QUADS + cq_cell(cx & 7u, cy & 7u, cz & 7u);
    const float r = reinterpret_cast<const float*>(v.cq + o)[0];
    return in ? r : 0.0f;
  } else if (LAYOUT == LAYOUT_BF) {
    return bf_voxel(v, x, y, z);
  }
  return lookup_density_brick(v, x, y, z);
}

// the eight taps of cell (ix,iy,iz) mixed x -> y -> z with the fractions given (common.glsl:62-68)
// IN_LATTICE (cellquad only): the caller guarantees a cell inside the apron lattice [-1, extent + 6] -- true for every
// sample position inside the (clipped) volume box, where the path-traced modes take all their samples: floor(p - 1/2)
// of p in [0, extent] is in [-1, extent - 1].  The clamp stays (a stray or NaN position reads defined memory), the
// eight selects that zero an out-of-lattice cell go.
template <int LAYOUT, bool IN_LATTICE = false>
VXD float trilinear_cell(const DevVolume& v, float density_scale, int ix, int iy, int iz, float fx, float fy, float fz) {
  float v000, v100, v010, v110, v001, v101, v011, v111;
  if (LAYOUT == LAYOUT_CQ) {
    // cell (ix,iy,iz) lives in apron brick (i+1)>>3 at local (i+1)&7; both z slices of the
    // xy-quad are one 16-byte load each, already decoded with each tap's own brick range
    // a cell outside the apron lattice [-1, extent + 6] has eight zero taps: the index is clamped into the lattice
    // and the quads selected to 0 (straight-line; mix of zeros = +0, times the scale as before)
    const int mxx = (int)v.extent[0] + 7, mxy = (int)v.extent[1] + 7, mxz = (int)v.extent[2] + 7;
    const bool in = IN_LATTICE || ((uint32_t)(ix + 1) <= (uint32_t)mxx && (uint32_t)(iy + 1) <= (uint32_t)mxy && (uint32_t)(iz + 1) <= (uint32_t)mxz);
    const uint32_t cx = (uint32_t)clamp0_i32(ix + 1, mxx), cy = (uint32_t)clamp0_i32(iy + 1, mxy), cz = (uint32_t)clamp0_i32(iz + 1, mxz);
    const uint32_t b = ((cz >> 3) * v.cq_bc[1] + (cy >> 3)) * v.cq_bc[0] + (cx >> 3);
    const size_t o = (size_t)b * CQ_BRICK_QUADS + cq_cell(cx & 7u, cy & 7u, cz & 7u);
    const float4 q0 = v.cq[o];
    const float4 q1 = v.cq[o + cq_next_slice(cz)];
    v000 = in ? q0.x : 0.0f; v100 = in ? q0.y : 0.0f; v010 = in ? q0.z : 0.0f; v110 = in ? q0.w : 0.0f;
    v001 = in ? q1.x : 0.0f; v101 = in ? q1.y : 0.0f; v011 = in ? q1.z : 0.0f; v111 = in ? q1.w : 0.0f;
  } else if (LAYOUT == LAYOUT_BF) {
    v000 = bf_voxel(v, ix, iy, iz);
    v100 = bf_voxel(v, ix + 1, iy, iz);
    v010 = bf_voxel(v, ix, iy + 1, iz);
    v110 = bf_voxel(v, ix + 1, iy + 1, iz);
    v001 = bf_voxel(v, ix, iy, iz + 1);
    v101 = bf_voxel(v, ix + 1, iy, iz + 1);
    v011 = bf_voxel(v, ix, iy + 1, iz + 1);
    v111 = bf_voxel(v, ix + 1, iy + 1, iz + 1);
  } else {
    v000 = lookup_density_brick(v, ix, iy, iz);
    v100 = lookup_density_brick(v, ix + 1, iy, iz);
    v010 = lookup_density_brick(v, ix, iy + 1, iz);
    v110 = lookup_density_brick(v, ix + 1, iy + 1, iz);
    v001 = lookup_density_brick(v, ix, iy, iz + 1);
    v101 = lookup_density_brick(v, ix + 1, iy, iz + 1);
    v011 = lookup_density_brick(v, ix, iy + 1, iz + 1);
    v111 = lookup_density_brick(v, ix + 1, iy + 1, iz + 1);
  }
  float lx0 = gl_mix(v000, v100, fx);
  float lx1 = gl_mix(v010, v110, fx);
  float hx0 = gl_mix(v001, v101, fx);
  float hx1 = gl_mix(v011, v111, fx);
  return density_scale * gl_mix(gl_mix(lx0, lx1, fy), gl_mix(hx0, hx1, fy), fz);
}

// A5: lookup_density_trilinear, common.glsl:61-69
template <int LAYOUT, bool IN_LATTICE = false>
VXD float lookup_density_trilinear(const DevVolume& v, float density_scale, V3 p) {
  float qx = p.x - 0.5f, qy = p.y - 0.5f, qz = p.z - 0.5f;
  float flx = floorf(qx), fly = floorf(qy), flz = floorf(qz);
  float fx = qx - flx, fy = qy - fly, fz = qz - flz;
  return trilinear_cell<LAYOUT, IN_LATTICE>(v, density_scale, f2i(flx), f2i(fly), f2i(flz), fx, fy, fz);
}

// lookup_majorant, common.glsl:50-53: the range-texture texel (level `mip`, .x = R = max) by cell, 0 outside the level
VXD float range_max_texel(const DevVolume& v, int mip, uint32_t bx, uint32_t by, uint32_t bz) {
  const uint32_t* data = mip == 0 ? v.range : v.mips[mip - 1];
  const uint32_t sx = mip == 0 ? v.bc[0] : v.mip_size[mip - 1][0], sy = mip == 0 ? v.bc[1] : v.mip_size[mip - 1][1],
                 sz = mip == 0 ? v.bc[2] : v.mip_size[mip - 1][2];
  if (bx < sx && by < sy && bz < sz) return half_bits_to_float(data[(bz * sy + by) * sx + bx] & 0xffffu);
  return 0.0f;
}

// A7: lookup_transfer, common.glsl:78-83; NEAREST + CLAMP_TO_EDGE (viewer.ts:386-389)
struct TfView {
  const float4* lut;  // LDS (in_lds) or global
  uint32_t len;
  float lenf;
  bool in_lds;        // wave uniform: the fetch is a ds_read_b128 or a global load, never a flat one
};
typedef const float4 __attribute__((address_space(3))) * LdsFloat4Ptr;
VXD float4 lookup_transfer(const TfView& tf, const float sr0, const float sr1, float d) {
  if (d < sr0 || d > sr1) return make_float4(0.f, 0.f, 0.f, 0.f);
  // floor(d * L) clamped to [0, L - 1]: the conversion truncates toward zero, which is floor for the products >= 0 and
  // lands on entry 0 like floor does for the negative ones
  const int i = clamp0_i32(f2i(d * tf.lenf), (int)tf.len - 1);
  if (tf.in_lds) {
    const LdsFloat4Ptr e = (LdsFloat4Ptr)tf.lut + i;
    return make_float4(e->x, e->y, e->z, e->w);
  }
  return tf.lut[i];
}

// A7 for a march that only needs the extinction (raymarch.glsl:20-21,41-43 read .a; .rgb only at the collision)
VXD float lookup_transfer_alpha(const TfView& tf, const float sr0, const float sr1, float d) {
  const int i = clamp0_i32(f2i(d * tf.lenf), (int)tf.len - 1);
  float a;
  if (tf.in_lds) a = ((LdsFloat4Ptr)tf.lut + i)->w;
  else a = tf.lut[i].w;
  return (d < sr0 || d > sr1) ? 0.0f : a;
}

// A6: stochastic_tricubic_filter, common.glsl:9-32.
// Each of the nine reservoir decisions is `rng < w / max(1e-3, sum)` with an IEEE quotient (common.glsl:21,25,29).
// The quotient q = RN(w / m) is only compared with r, a multiple of 2^-24 in [0, 1): with e = r * m - w,
// |e| > m * 2^-23 means |r - w/m| > 2^-23, more than the 2^-24 by which q can differ from w / m (|w/m| < 2), so
// (r < q) == (e < 0) -- and d = fma(r, m, -w) is e correctly rounded, same sign, |d| > m * 2^-23 only if |e| is.  One
// fma, one compare and one band test replace the ten-instruction division sequence; when some lane of the wave falls
// inside the band (about 3e-7 of the decisions) the stage is re-decided with the quotient itself, so every decision --
// hence every tap, every sample and the RNG stream -- is the one the reference's expression gives.
// (max(1e-3, sum) is sum itself at every decision: the running sum holds the second weight, (3t^3 - 6t^2 + 4) / 6 >= 1/6,
// from the first decision on.  The margin |d| - m * 2^-23 of the band test is returned; the caller tests the smallest of a
// stage's three.  A NaN position makes every compare false here and in the reference expression alike.)
VXD bool reservoir_take(float r, float w, float sum, float& margin) {
  const float d = fma_(r, sum, -w);
  margin = fma_(sum, -1.1920928955078125e-07f, __builtin_fabsf(d));
  return d < 0.0f;
}
VXD void stochastic_tricubic_filter(V3 ipos, Rng& s, int tap[3]) {
  const float q[3] = {ipos.x - 0.5f, ipos.y - 0.5f, ipos.z - 0.5f};
  int ii[3], idx[3] = {0, 0, 0};
  float t[3], t2[3], w[3], sum[3], r[3];
  const float sixth = 1.0f / 6.0f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float fl = floorf(q[c]);
    ii[c] = f2i(fl);
    t[c] = q[c] - fl;   // == q - float(int(floor q)), the conversions cancel
    t2[c] = t[c] * t[c];
    w[c] = sixth * (fma_(-3.0f, t[c], fma_(3.0f, t2[c], -t[c] * t2[c])) + 1.0f);
    sum[c] = w[c];
  }
  // stage k: weight k (common.glsl:19,23,27), running sum, three draws, three decisions
  auto stage = [&](int k) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      if (k == 1) w[c] = sixth * (fma_(-6.0f, t2[c], 3.0f * t[c] * t2[c]) + 4.0f);
      else if (k == 2) w[c] = sixth * (fma_(3.0f, t[c], fma_(3.0f, t2[c], -3.0f * t[c] * t2[c])) + 1.0f);
      else w[c] = sixth * t[c] * t2[c];
      sum[c] = w[c] + sum[c];
    }
    r[0] = rng(s); r[1] = rng(s); r[2] = rng(s);
    bool take[3];
    float margin[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) take[c] = reservoir_take(r[c], w[c], sum[c], margin[c]);
    const bool unsure = !(fminf(fminf(margin[0], margin[1]), margin[2]) > 0.0f);
    if (ballot(unsure) != 0ull) {   // wave uniform, rare: the reference's expression itself
#pragma unroll
      for (int c = 0; c < 3; ++c) take[c] = r[c] < w[c] / gl_max(1e-3f, sum[c]);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) idx[c] = take[c] ? k : idx[c];
  };
  stage(1);
  stage(2);
  stage(3);
#pragma unroll
  for (int c = 0; c < 3; ++c) tap[c] = ii[c] + idx[c] - 1;
}

// ---------------------------------------------------------------------------------------
struct Ray {
  V3 o, d;
};

// A8: ray_box_intersection, utils.glsl:61-69 (the clip box, volume.ts:32-37)
VXD bool ray_box_intersection(const Ray& r, const float* bmin, const float* bmax, float& near,
                              float& far) {
  float ix = 1.0f / r.d.x, iy = 1.0f / r.d.y, iz = 1.0f / r.d.z;
  float lox = (bmin[0] - r.o.x) * ix, loy = (bmin[1] - r.o.y) * iy, loz = (bmin[2] - r.o.z) * iz;
  float hix = (bmax[0] - r.o.x) * ix, hiy = (bmax[1] - r.o.y) * iy, hiz = (bmax[2] - r.o.z) * iz;
  float tminx = gl_min(lox, hix), tminy = gl_min(loy, hiy), tminz = gl_min(loz, hiz);
  float tmaxx = gl_max(lox, hix), tmaxy = gl_max(loy, hiy), tmaxz = gl_max(loz, hiz);
  near = gl_max(0.0f, gl_max(tminx, gl_max(tminy, tminz)));
  far = gl_min(tmaxx, gl_min(tmaxy, tmaxz));
  return near <= far;
}

// fragment.frag:131 `(pixel + 0.5) / u_res` for one axis; with the host's go-ahead the IEEE division sequence (ten vector
// instructions) is the reciprocal product corrected by its own remainder (three), the same bits for every pixel of the image
VXD float tex_coord(int pixel, int res, const DevVolume* hv, int axis) {
  const float a = (float)pixel + 0.5f;
  if (hv && (hv->ray_flags & (axis == 0 ? RAY_TEX_BY_RECIPROCAL_X : RAY_TEX_BY_RECIPROCAL_Y))) {   // wave uniform
    const float y = hv->inv_res[axis], q0 = a * y;
    return fma_(fma_(-(float)res, q0, a), y, q0);
  }
  return a / (float)res;
}

// A9: setup_world_ray, fragment.frag:57-65 + utils.glsl:23-40, inverses hoisted (Q11)
// hv: the host-evaluated uniform terms (DevVolume::cam_o ...), or nullptr to evaluate them here -- the same bits
VXD Ray setup_world_ray(const VxParams& p, float tex_x, float tex_y, float rx, float ry, const DevVolume* hv = nullptr) {
  float x_off = fma_(rx, 2.0f, -1.0f) * (hv ? hv->inv_res[0] : 1.0f / (float)p.res[0]);
  float y_off = fma_(ry, 2.0f, -1.0f) * (hv ? hv->inv_res[1] : 1.0f / (float)p.res[1]);
  float sx = tex_x + x_off, sy = tex_y + y_off;
  float cw[4], vp[4], wp[4];
  if (p.camera_ortho) {  // [build] parallel rays: near-plane point of the pixel, camera -z axis (wave uniform)
    mat4_mul(p.camera_proj_inv, fma_(sx, 2.0f, -1.0f), fma_(sy, 2.0f, -1.0f), -1.0f, 1.0f, vp);
    mat4_mul(p.camera_view_inv, vp[0] / vp[3], vp[1] / vp[3], vp[2] / vp[3], 1.0f, wp);
    mat4_mul(p.camera_view_inv, 0.0f, 0.0f, -1.0f, 0.0f, cw);
    return Ray{v3(wp[0] / wp[3], wp[1] / wp[3], wp[2] / wp[3]), normalize3(v3(cw[0], cw[1], cw[2]))};
  }
  V3 cam;
  if (hv) {
    cam = v3(hv->cam_o[0], hv->cam_o[1], hv->cam_o[2]);
  } else {
    mat4_mul(p.camera_view_inv, 0.0f, 0.0f, 0.0f, 1.0f, cw);
    cam = v3(cw[0] / cw[3], cw[1] / cw[3], cw[2] / cw[3]);
  }
  mat4_mul(p.camera_proj_inv, fma_(sx, 2.0f, -1.0f), fma_(sy, 2.0f, -1.0f), 0.0f, 1.0f, vp);
  float vx_ = vp[0] / vp[3], vy_ = vp[1] / vp[3], vz_ = vp[2] / vp[3];
  mat4_mul(p.camera_view_inv, vx_, vy_, vz_, 1.0f, wp);
  V3 world;
  if (hv && (hv->ray_flags & RAY_AFFINE_VIEW)) world = v3(wp[0], wp[1], wp[2]);   // wp[3] == 1.0: x / 1 = x (wave uniform)
  else world = v3(wp[0] / wp[3], wp[1] / wp[3], wp[2] / wp[3]);
  return Ray{cam, normalize3(sub3(world, cam))};
}

VXD void to_index(const VxParams& p, const Ray& r, V3& ipos, V3& idir, const DevVolume* hv = nullptr) {
  float a[4], b[4];
  mat4_mul(p.density_transform_inv, r.d.x, r.d.y, r.d.z, 0.0f, b);
  idir = v3(b[0], b[1], b[2]);
  if (hv && !p.camera_ortho) {   // the perspective origin is the camera: transformed once on the host
    ipos = v3(hv->cam_ipos[0], hv->cam_ipos[1], hv->cam_ipos[2]);
  } else {
    mat4_mul(p.density_transform_inv, r.o.x, r.o.y, r.o.z, 1.0f, a);
    ipos = v3(a[0], a[1], a[2]);
  }
}

// texture(u_envmap, uv): LINEAR on level 0, REPEAT in s, CLAMP_TO_EDGE in t (environment.ts:22-26);
// [build] exact fp32 fractions as weights
VXD V3 env_texture(const float4* __restrict__ tex, uint32_t w, uint32_t h, float u, float v) {
  float x = fma_(u, (float)w, -0.5f), y = fma_(v, (float)h, -0.5f);
  float fx = floorf(x), fy = floorf(y);
  float a = x - fx, b = y - fy;
  int W = (int)w, H = (int)h;
  int i0 = f2i(fx), j0 = f2i(fy);
  int i1 = i0 + 1, j1 = j0 + 1;
  i0 %= W; if (i0 < 0) i0 += W;
  i1 %= W; if (i1 < 0) i1 += W;
  j0 = j0 < 0 ? 0 : (j0 > H - 1 ? H - 1 : j0);
  j1 = j1 < 0 ? 0 : (j1 > H - 1 ? H - 1 : j1);
  float4 t00 = tex[(size_t)j0 * w + i0], t10 = tex[(size_t)j0 * w + i1];
  float4 t01 = tex[(size_t)j1 * w + i0], t11 = tex[(size_t)j1 * w + i1];
  float na = 1.0f - a, nb = 1.0f - b;
  V3 lo = v3(fma_(t10.x, a, t00.x * na), fma_(t10.y, a, t00.y * na), fma_(t10.z, a, t00.z * na));
  V3 hi = v3(fma_(t11.x, a, t01.x * na), fma_(t11.y, a, t01.y * na), fma_(t11.z, a, t01.z * na));
  return v3(fma_(hi.x, b, lo.x * nb), fma_(hi.y, b, lo.y * nb), fma_(hi.z, b, lo.z * nb));
}
VXD float env_luma(V3 col) { return dot3(col, v3(0.212671f, 0.715160f, 0.072169f)); }
VXD float imp_fetch(const float* __restrict__ imp, int x, int y, int mip) {
  int n = (int)(IMP_DIM >> mip);
  if (x < 0 || y < 0 || x >= n || y >= n) return 0.0f;
  return imp[imp_offset((uint32_t)mip) + (size_t)y * n + x];
}

// environment.glsl:19-27; directional branch: pow base clamped at 0 ([build], quirk Q16)
VXD V3 lookup_environment(const VxParams& p, const DevVolume& dv, V3 dir) {
  if (p.use_env > 0) {
    const float pi = 3.14159265358979323846f;
    float u = atan2f(dir.z, dir.x) / (2.0f * pi) + 0.5f;
    float v = 1.0f - acosf(dir.y) / pi;
    V3 t = env_texture(dv.env_tex, dv.env_w, dv.env_h, u, v);
    return v3(p.env_strength * t.x, p.env_strength * t.y, p.env_strength * t.z);
  }
  V3 nl = v3(-p.light_dir[0], -p.light_dir[1], -p.light_dir[2]);
  float c = gl_max(dot3(dir, nl), 0.0f);
  float s = gl_clamp(powf(c, 300.0f), 0.0f, 1.0f);
  float e = p.env_strength * fma_(s, 4.0f, 0.01f);
  return v3(e, e, e);
}
// environment.glsl:82-86 (u_use_env = 1 only)
VXD float pdf_environment(const VxParams& p, const DevVolume& dv, V3 dir) {
  const float inv_4pi = 1.0f / (4.0f * 3.14159265358979323846f);
  V3 le = lookup_environment(p, dv, dir);
  return env_luma(le) / dv.env_avg_w * inv_4pi;
}
// environment.glsl:35-79 (u_use_env = 1 only): hierarchical warp over the importance mips
VXD float4 sample_environment(const VxParams& p, const DevVolume& dv, float u0, float u1, V3& w_i) {
  const float pi = 3.14159265358979323846f, inv_4pi = 1.0f / (4.0f * 3.14159265358979323846f);
  int px = 0, py = 0;
  float sx = u0, sy = u1, wsel = 0.0f;
  for (int mip = (int)IMP_LEVELS - 2; mip >= 0; --mip) {
    // the four texels pos*2 + {0,1}^2 of level `mip` are one quad, indexed by the parent texel
    const int half = (int)(IMP_DIM >> (mip + 1));
    float4 q = dv.env_impq[impq_offset((uint32_t)mip) + (uint32_t)(py * half + px)];
    px *= 2; py *= 2;
    float w0 = q.x, w1 = q.y, w2 = q.z, w3 = q.w;
    float q0 = w0 + w2, q1 = w1 + w3;
    float d = q0 / gl_max(1e-8f, q0 + q1);
    float wtop, qsel;
    if (sx < d) { sx = sx / d; wsel = w0; wtop = w2; qsel = q0; }
    else { sx = (sx - d) / (1.0f - d); px += 1; wsel = w1; wtop = w3; qsel = q1; }
    float e = wsel / qsel;
    if (sy < e) { sy = sy / e; }
    else { py += 1; sy = (sy - e) / (1.0f - e); wsel = wtop; }
  }
  const float inv_dim = 1.0f / (float)IMP_DIM;
  float uvx = ((float)px + sx) * inv_dim, uvy = ((float)py + sy) * inv_dim;
  float theta = gl_clamp(1.0f - uvy, 0.0f, 1.0f) * pi;
  float phi = (gl_clamp(uvx, 0.0f, 1.0f) * 2.0f - 1.0f) * pi;
  float sin_t = sinf(theta);
  w_i = v3(sin_t * cosf(phi), cosf(theta), sin_t * sinf(phi));
  V3 t = env_texture(dv.env_tex, dv.env_w, dv.env_h, uvx, uvy);
  float pdf = wsel / dv.env_avg_w;  // wsel = texel (px,py) of level 0, env_avg_w = level 9
  return make_float4(p.env_strength * t.x, p.env_strength * t.y, p.env_strength * t.z, pdf * inv_4pi);
}

// [build] Blinn-Phong terms of VX_MODE_DVR_PHONG on the hardware transcendentals (v_rsq_f32, v_log_f32, v_exp_f32:
// 1 ulp each) instead of the device library's correctly rounded sqrt / divide / powf, which cost ~200 instructions
// per shaded sample -- more than the whole march step.  The oracle keeps libm; the images agree within the stated
// tolerance (the terms only scale a colour increment, no decision depends on them).
VXD float rsq_fast(float x) { return __builtin_amdgcn_rsqf(x); }
VXD float pow_fast(float x, float y) {   // x >= 0
  if (y == 0.0f) return 1.0f;
  return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x));   // x == 0: exp2(-inf) = 0 for y > 0
}
// rgb of a sample shaded by the density gradient g (world space): n = -g/|g|, ambient + diffuse + Blinn specular
VXD void phong_shade(const VxParams& p, V3 g, V3 nl, V3 hv, float4& rgba) {
  const float g2 = dot3(g, g);
  if (g2 > 1e-12f) {
    const V3 n = scale3(g, -rsq_fast(g2));
    const float ndl = gl_max(0.0f, dot3(n, nl));
    const float ndh = gl_max(0.0f, dot3(n, hv));
    const float diff = fma_(p.phong_kd, ndl, p.phong_ka);
    const float spec = p.phong_ks * pow_fast(ndh, p.phong_shininess);
    rgba.x = fma_(rgba.x, diff, spec);
    rgba.y = fma_(rgba.y, diff, spec);
    rgba.z = fma_(rgba.z, diff, spec);
  }
}

VXD float sanitize1(float x) { return (x != x || __builtin_isinf(x)) ? 0.0f : x; }

// per-wave work counters, flushed with one atomic per wave
struct Counts {
  uint32_t samples, rays, skips, grads, tf;
  // march-loop trips of this lane in the collision searches (primary segments) and in the transmittance estimates (shadow
  // segments) of trace_path: a few instructions per SEGMENT, none per trip (differences of samples / skips around the
  // calls).  The wave's march lane slots follow at the kernel's end: 64 x (max over the lanes of each), exact at bounces 1
  uint32_t it_p, it_s;
};

}  // namespace vx
