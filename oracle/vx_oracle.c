/*
 * vx_oracle.c -- scalar CPU restatement of the Volxel raymarch hot path.
 * TEST INFRASTRUCTURE ONLY (see vx_oracle.h).  PARITY UNPINNED (see vx_oracle.h).
 *
 * Every function cites the reference lines it restates; paths are relative to
 * /root/reference.  "[build]" marks behaviour the reference does not define and this
 * project fixes (listed in DESIGN.md section "Arithmetic contract").
 *
 * Arithmetic contract (identical in the HIP kernels, which are written separately):
 *   - IEEE binary32, round to nearest even, no flush to zero, compiled -ffp-contract=off;
 *   - a sum is evaluated left to right and every product term that is ADDED to a running
 *     sum is fused into it with one fmaf (GLSL ES 3.00 leaves contraction to the
 *     implementation; CDNA4 executes it as one v_fma_f32).  The first term of a sum and all
 *     other operations are rounded individually;
 *   - mat4*vec4 = fmaf chain over columns 0..3; dot = fmaf chain over x,y,z;
 *   - min(x,y) = (y<x)?y:x, max(x,y) = (x<y)?y:x  (GLSL ES 3.00 section 8.3);
 *   - float->int conversion truncates, saturates and maps NaN to 0 (v_cvt_i32_f32);
 *   - the value returned by a function call (sqr(), dot(), ...) is a rounded value and is
 *     never fused into the caller's expression;
 *   - exp/log/pow/sin/cos are libm's; kernels use the device library's.  Only modes whose
 *     branch decisions do not depend on them are compared bit-for-bit.
 */
#include "vx_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- tolerance-envelope variants (oracle/Makefile; tests/test_tolerance_envelope.py) -----------------------------------
 * GLSL ES 3.00 leaves two things to the implementation that the arithmetic contract above FIXES: whether a*b+c is contracted
 * into one rounding, and how many ulps the transcendental built-ins may be off.  No WebGL2 implementation can run here, so the
 * nearest thing to "how far is the reference's own output from this oracle" is how far the ORACLE moves when those two
 * freedoms are exercised the other way.  The contract build defines neither macro and is bit-for-bit what it was.
 *   VXO_NO_CONTRACTION        every a*b+c the contract fuses is rounded twice (an implementation that never contracts)
 *   VXO_TRANSCENDENTAL_ULPS=n exp / log / pow / sin / cos / atan / acos results are biased by n ulps (relative n * 2^-23):
 *                             an implementation at the far end of a n-ulp error budget, every call off in the same direction */
#ifdef VXO_NO_CONTRACTION
static inline float vxo_unfused(float a, float b, float c) { const float prod = a * b; return prod + c; }
#define fmaf(a, b, c) vxo_unfused((a), (b), (c))
#endif
#ifdef VXO_TRANSCENDENTAL_ULPS
static inline float vxo_biased(float r) { return r * (1.0f + (float)(VXO_TRANSCENDENTAL_ULPS) * 1.1920929e-7f); }
#define expf(x) vxo_biased(expf(x))
#define logf(x) vxo_biased(logf(x))
#define powf(x, y) vxo_biased(powf((x), (y)))
#define sinf(x) vxo_biased(sinf(x))
#define cosf(x) vxo_biased(cosf(x))
#define atan2f(y, x) vxo_biased(atan2f((y), (x)))
#define acosf(x) vxo_biased(acosf(x))
#endif

/* ------------------------------------------------------------------------------------ */
/* small helpers of the arithmetic contract                                              */

static inline float gl_min(float x, float y) { return (y < x) ? y : x; }
static inline float gl_max(float x, float y) { return (x < y) ? y : x; }
static inline float gl_clamp(float x, float lo, float hi) { return gl_min(gl_max(x, lo), hi); }

static inline int32_t f2i(float x) {
  if (x != x) return 0;
  if (x >= 2147483648.0f) return INT32_MAX;
  if (x <= -2147483648.0f) return INT32_MIN;
  return (int32_t)x;
}

typedef struct { float x, y, z; } v3;
static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline float dot3(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline v3 sub3(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 scale3(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 neg3(v3 a) { return V3(-a.x, -a.y, -a.z); }
static inline v3 normalize3(v3 a) {
  float inv = 1.0f / sqrtf(dot3(a, a));
  return scale3(a, inv);
}
static inline v3 cross3(v3 a, v3 b) {
  /* a.y*b.z - b.y*a.z etc.: first product rounded, second fused (subtracted) */
  return V3(fmaf(-b.y, a.z, a.y * b.z), fmaf(-b.z, a.x, a.z * b.x), fmaf(-b.x, a.y, a.x * b.y));
}
/* o + t*d */
static inline v3 madd3(v3 o, float t, v3 d) {
  return V3(fmaf(t, d.x, o.x), fmaf(t, d.y, o.y), fmaf(t, d.z, o.z));
}
/* column-major mat4 * (x,y,z,w) */
static inline void mat4_mul(const float* m, float x, float y, float z, float w, float out[4]) {
  for (int i = 0; i < 4; ++i)
    out[i] = fmaf(m[12 + i], w, fmaf(m[8 + i], z, fmaf(m[4 + i], y, m[i] * x)));
}

/* ------------------------------------------------------------------------------------ */
/* RNG -- shaders/random.glsl                                                            */

/* random.glsl:41-51 */
uint32_t vxo_tea(uint32_t v0, uint32_t v1, uint32_t n) {
  uint32_t s0 = 0u;
  for (uint32_t i = 0; i < n; ++i) {
    s0 += 0x9e3779b9u;
    v0 += ((v1 << 4) + 0xA341316Cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xC8013EA4u);
    v1 += ((v0 << 4) + 0xAD90777Du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7E95761Eu);
  }
  return v0;
}

/* random.glsl:54-56 */
static inline uint32_t rotl32(uint32_t x, uint32_t k) { return (x << k) | (x >> (32u - k)); }

/* random.glsl:59-66 */
uint32_t vxo_wang(uint32_t x) {
  x = (x ^ 61u) ^ (x >> 16);
  x *= 9u;
  x = x ^ (x >> 4);
  x *= 0x27d4eb2du;
  x = x ^ (x >> 15);
  return x;
}

/* random.glsl:69-76 */
void vxo_seed_xoshiro(uint32_t seed, uint32_t s[4]) {
  s[0] = vxo_wang(seed + 0u);
  s[1] = vxo_wang(seed + 1u);
  s[2] = vxo_wang(seed + 2u);
  s[3] = vxo_wang(seed + 3u);
}

/* random.glsl:80-94.  NOTE s.x + s.z (quirk Q1), not the canonical s0 + s3. */
uint32_t vxo_xoshiro_next(uint32_t s[4]) {
  uint32_t result = rotl32(s[0] + s[2], 7u) + s[0];
  uint32_t t = s[1] << 9;
  s[2] ^= s[0];
  s[3] ^= s[1];
  s[1] ^= s[2];
  s[0] ^= s[3];
  s[2] ^= t;
  s[3] = rotl32(s[3], 11u);
  return result;
}

/* random.glsl:103-106 */
float vxo_rng(uint32_t s[4]) {
  uint32_t r = vxo_xoshiro_next(s);
  return (float)(r >> 8) / 16777216.0f;
}

/* fragment.frag:140,143 */
uint32_t vxo_pixel_seed(uint32_t px, uint32_t py, uint32_t res_x, uint32_t frame) {
  return vxo_tea(42u * (py * res_x + px), frame, 32u);
}

/* ------------------------------------------------------------------------------------ */
/* IEEE binary16 <-> binary32, round to nearest even -- what half::f16::from_f32 /        */
/* to_f32 (half 2.7.1, Cargo.lock:636-637) compute at brick.rs:19-28                      */

uint16_t vxo_f32_to_f16(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t exp = (x >> 23) & 0xffu;
  uint32_t man = x & 0x7fffffu;
  if (exp == 0xffu) { /* inf / nan */
    if (man == 0) return (uint16_t)(sign | 0x7c00u);
    return (uint16_t)(sign | 0x7c00u | 0x0200u | (man >> 13));
  }
  int32_t e = (int32_t)exp - 127 + 15;
  if (e >= 0x1f) return (uint16_t)(sign | 0x7c00u); /* overflow -> inf */
  if (e <= 0) {                                     /* subnormal half or zero */
    if (e < -10) return (uint16_t)sign;
    man |= 0x800000u;
    uint32_t shift = (uint32_t)(14 - e); /* 14..24 */
    uint32_t half_man = man >> shift;
    uint32_t rem = man & ((1u << shift) - 1u);
    uint32_t halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (half_man & 1u))) half_man += 1;
    return (uint16_t)(sign | half_man);
  }
  uint32_t half = ((uint32_t)e << 10) | (man >> 13);
  uint32_t rem = man & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (half & 1u))) half += 1; /* may carry into exp */
  return (uint16_t)(sign | half);
}

float vxo_f16_to_f32(uint16_t h) {
  uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1fu;
  uint32_t man = h & 0x3ffu;
  uint32_t x;
  if (exp == 0) {
    if (man == 0) {
      x = sign;
    } else { /* subnormal: normalise */
      int e = -1;
      do {
        man <<= 1;
        ++e;
      } while (!(man & 0x400u));
      man &= 0x3ffu;
      x = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
    }
  } else if (exp == 0x1f) {
    x = sign | 0x7f800000u | (man << 13);
  } else {
    x = sign | ((exp - 15 + 127) << 23) | (man << 13);
  }
  float f;
  memcpy(&f, &x, 4);
  return f;
}

/* ------------------------------------------------------------------------------------ */
/* brick grid -- dicom_preprocessor/src/{buf3d,dicom,brick}.rs                            */

#define BRICK_SIZE 8u       /* brick.rs:9  */
#define BITS_PER_AXIS 10u   /* brick.rs:10 */
#define MAX_BRICKS (1u << BITS_PER_AXIS)
#define NUM_MIPMAPS 3u      /* brick.rs:13 */

/* buf3d.rs:26-28 */
static inline size_t buf_index(const uint32_t s[3], uint32_t x, uint32_t y, uint32_t z) {
  return (size_t)z * s[0] * s[1] + (size_t)y * s[0] + x;
}

/* dicom.rs:7-17 */
float vxo_dicom_lookup(const uint16_t* vox, const uint32_t dims[3], uint16_t max_value, uint32_t x,
                       uint32_t y, uint32_t z) {
  if (z >= dims[2] || y >= dims[1] || x >= dims[0]) return 0.0f;
  uint16_t raw = vox[buf_index(dims, x, y, z)];
  return (float)raw / (float)max_value;
}

/* brick.rs:19-23 */
static inline uint32_t encode_range(float x, float y) {
  return ((uint32_t)vxo_f32_to_f16(x) << 16) | (uint32_t)vxo_f32_to_f16(y);
}
/* brick.rs:24-28 */
static inline void decode_range(uint32_t d, float* x, float* y) {
  *x = vxo_f16_to_f32((uint16_t)(d >> 16));
  *y = vxo_f16_to_f32((uint16_t)d);
}
/* brick.rs:45-48; Rust f32::clamp keeps NaN, `as u8` saturates and maps NaN to 0 */
static inline uint8_t encode_voxel(float value, float rx, float ry) {
  float n = (value - rx) / (ry - rx);
  if (n < 0.0f) n = 0.0f;
  if (n > 1.0f) n = 1.0f;
  float r = roundf(255.0f * n);
  if (r != r) return 0;
  if (r <= 0.0f) return 0;
  if (r >= 255.0f) return 255;
  return (uint8_t)r;
}
/* brick.rs:54-57 (component-wise f32 division then ceil) */
static inline uint32_t div_round_up_f(uint32_t num, uint32_t den) {
  return (uint32_t)ceilf((float)num / (float)den);
}

/* brick.rs:77 */
void vxo_brick_count(const uint32_t dims[3], uint32_t out[3]) {
  for (int i = 0; i < 3; ++i)
    out[i] = div_round_up_f(div_round_up_f(dims[i], BRICK_SIZE), 1u << NUM_MIPMAPS) *
             (1u << NUM_MIPMAPS);
}

/* brick.rs:76-204 */
int64_t vxo_brick_construct(const uint16_t* vox, const uint32_t dims[3], uint16_t max_value,
                            uint32_t* indirection, uint32_t* range, uint8_t* atlas, uint32_t* mip0,
                            uint32_t* mip1, uint32_t* mip2, uint32_t atlas_size_out[3]) {
  uint32_t bc[3];
  vxo_brick_count(dims, bc);
  if (bc[0] >= MAX_BRICKS || bc[1] >= MAX_BRICKS || bc[2] >= MAX_BRICKS) return -1; /* :79-81 */
  uint32_t as[3] = {bc[0] * BRICK_SIZE, bc[1] * BRICK_SIZE, bc[2] * BRICK_SIZE};
  memset(atlas, 0, (size_t)as[0] * as[1] * as[2]);
  uint64_t counter = 0;
  for (uint32_t bz = 0; bz < bc[2]; ++bz)
    for (uint32_t by = 0; by < bc[1]; ++by)
      for (uint32_t bx = 0; bx < bc[0]; ++bx) {
        size_t bi = buf_index(bc, bx, by, bz);
        indirection[bi] = 0; /* :96 */
        float lmin = 3.40282347e+38f, lmax = -3.40282347e+38f; /* f32::MAX / f32::MIN */
        for (int32_t lz = -2; lz < (int32_t)BRICK_SIZE + 2; ++lz)
          for (int32_t ly = -2; ly < (int32_t)BRICK_SIZE + 2; ++ly)
            for (int32_t lx = -2; lx < (int32_t)BRICK_SIZE + 2; ++lx) {
              /* :105-107 negative coordinates wrap to huge u32 -> lookup returns 0 */
              uint32_t ux = (uint32_t)((int32_t)(bx * BRICK_SIZE) + lx);
              uint32_t uy = (uint32_t)((int32_t)(by * BRICK_SIZE) + ly);
              uint32_t uz = (uint32_t)((int32_t)(bz * BRICK_SIZE) + lz);
              float v = vxo_dicom_lookup(vox, dims, max_value, ux, uy, uz);
              lmin = fminf(lmin, v);
              lmax = fmaxf(lmax, v);
            }
        range[bi] = encode_range(lmin, lmax); /* :119 */
        if (lmin == lmax) continue;           /* :120 */
        uint64_t index = counter++;           /* :127-128 */
        /* buf3d.rs:29-32 calculate_coord on the indirection strides */
        uint32_t px = (uint32_t)(index % bc[0]);
        uint32_t py = (uint32_t)((index / bc[0]) % bc[1]);
        uint32_t pz = (uint32_t)(index / ((uint64_t)bc[0] * bc[1]));
        if (px >= MAX_BRICKS || py >= MAX_BRICKS || pz >= MAX_BRICKS) return -1; /* :31 */
        indirection[bi] = px | (py << BITS_PER_AXIS) | (pz << (2 * BITS_PER_AXIS)); /* :32-34 */
        float rx, ry;
        decode_range(range[bi], &rx, &ry); /* :136 */
        for (uint32_t lz = 0; lz < BRICK_SIZE; ++lz)
          for (uint32_t ly = 0; ly < BRICK_SIZE; ++ly)
            for (uint32_t lx = 0; lx < BRICK_SIZE; ++lx) {
              size_t ai = buf_index(as, px * BRICK_SIZE + lx, py * BRICK_SIZE + ly,
                                    pz * BRICK_SIZE + lz);
              float v = vxo_dicom_lookup(vox, dims, max_value, bx * BRICK_SIZE + lx,
                                         by * BRICK_SIZE + ly, bz * BRICK_SIZE + lz);
              atlas[ai] = encode_voxel(v, rx, ry); /* :142 */
            }
      }
  /* :151 prune */
  uint32_t slices =
      (uint32_t)((float)BRICK_SIZE *
                 roundf(ceilf((float)counter / (float)(bc[0] * bc[1]))));
  atlas_size_out[0] = as[0];
  atlas_size_out[1] = as[1];
  atlas_size_out[2] = slices;
  /* :153-190 range mipmaps */
  uint32_t* mips[3] = {mip0, mip1, mip2};
  for (uint32_t level = 0; level < NUM_MIPMAPS; ++level) {
    uint32_t ms[3] = {bc[0] >> (level + 1), bc[1] >> (level + 1), bc[2] >> (level + 1)};
    uint32_t ss[3] = {bc[0] >> level, bc[1] >> level, bc[2] >> level};
    const uint32_t* src = level == 0 ? range : mips[level - 1];
    for (uint32_t z = 0; z < ms[2]; ++z)
      for (uint32_t y = 0; y < ms[1]; ++y)
        for (uint32_t x = 0; x < ms[0]; ++x) {
          float lmin = 3.40282347e+38f, lmax = -3.40282347e+38f;
          for (uint32_t dz = 0; dz < 2; ++dz)
            for (uint32_t dy = 0; dy < 2; ++dy)
              for (uint32_t dx = 0; dx < 2; ++dx) {
                float rx, ry;
                decode_range(src[buf_index(ss, x * 2 + dx, y * 2 + dy, z * 2 + dz)], &rx, &ry);
                lmin = fminf(lmin, rx);
                lmax = fmaxf(lmax, ry);
              }
          mips[level][buf_index(ms, x, y, z)] = encode_range(lmin, lmax);
        }
  }
  return (int64_t)counter;
}

/* dicom.rs:39-66 */
void vxo_histogram_gradient(const uint32_t* hist, uint32_t n, int32_t* smoothed, uint32_t* gmin,
                            uint32_t* gmax) {
  int32_t* grad = (int32_t*)malloc(sizeof(int32_t) * (n ? n : 1));
  uint32_t last = 0, mn = UINT32_MAX, mx = 0;
  for (uint32_t i = 0; i < n; ++i) {
    int32_t step = (int32_t)hist[i] - (int32_t)last;
    uint32_t a = step < 0 ? (uint32_t)(-(int64_t)step) : (uint32_t)step;
    if (a > mx) mx = a;
    if (a < mn) mn = a;
    grad[i] = step;
    last = hist[i];
  }
  if (n > 0) {
    smoothed[0] = grad[0];
    for (uint32_t i = 1; i + 1 < n; ++i) smoothed[i] = (grad[i - 1] + grad[i] + grad[i + 1]) / 3;
    if (n > 1) smoothed[n - 1] = grad[n - 1];
  }
  *gmin = mn;
  *gmax = mx;
  free(grad);
}

/* ---- texel access on the uploaded textures (viewer.ts:1106-1142) -------------------- */

/* RGB10_A2UI / UNSIGNED_INT_2_10_10_10_REV: x = bits 0-9 (brick.rs:37-43) */
static inline void fetch_ptr(const VxoVolume* v, int32_t bx, int32_t by, int32_t bz, uint32_t p[3]) {
  if (bx < 0 || by < 0 || bz < 0 || (uint32_t)bx >= v->ind_size[0] ||
      (uint32_t)by >= v->ind_size[1] || (uint32_t)bz >= v->ind_size[2]) {
    p[0] = p[1] = p[2] = 0; /* WebGL2 robust out-of-range texelFetch returns zero */
    return;
  }
  uint32_t d = v->indirection[buf_index(v->ind_size, (uint32_t)bx, (uint32_t)by, (uint32_t)bz)];
  p[0] = d & (MAX_BRICKS - 1);
  p[1] = (d >> BITS_PER_AXIS) & (MAX_BRICKS - 1);
  p[2] = (d >> (2 * BITS_PER_AXIS)) & (MAX_BRICKS - 1);
}
/* RG16F texel of GL level `mip`: R = max, G = min (brick.rs:19-23,357-359) */
static inline void fetch_range(const VxoVolume* v, int32_t bx, int32_t by, int32_t bz, int32_t mip,
                               float* r, float* g) {
  const uint16_t* data = mip == 0 ? v->range : v->mips[mip - 1];
  const uint32_t* s = mip == 0 ? v->range_size : v->mip_size[mip - 1];
  if (bx < 0 || by < 0 || bz < 0 || (uint32_t)bx >= s[0] || (uint32_t)by >= s[1] ||
      (uint32_t)bz >= s[2]) {
    *r = 0.0f;
    *g = 0.0f;
    return;
  }
  size_t i = buf_index(s, (uint32_t)bx, (uint32_t)by, (uint32_t)bz);
  *r = vxo_f16_to_f32(data[2 * i]);
  *g = vxo_f16_to_f32(data[2 * i + 1]);
}
/* R8 unorm: f = c / (2^8 - 1)  (OpenGL ES 3.0 section 2.1.6.1) */
static inline float fetch_atlas(const VxoVolume* v, uint32_t x, uint32_t y, uint32_t z) {
  if (x >= v->atlas_size[0] || y >= v->atlas_size[1] || z >= v->atlas_size[2]) return 0.0f;
  return (float)v->atlas[buf_index(v->atlas_size, x, y, z)] / 255.0f;
}

/* brick.rs:208-230 (CPU twin; decode_voxel brick.rs:50-52 multiplies by 1/255) */
float vxo_brick_lookup(const VxoVolume* v, uint32_t x, uint32_t y, uint32_t z) {
  uint32_t p[3];
  fetch_ptr(v, (int32_t)(x >> 3), (int32_t)(y >> 3), (int32_t)(z >> 3), p);
  float mx, mn;
  fetch_range(v, (int32_t)(x >> 3), (int32_t)(y >> 3), (int32_t)(z >> 3), 0, &mx, &mn);
  uint8_t data =
      v->atlas[buf_index(v->atlas_size, (p[0] << 3) + (x & 7), (p[1] << 3) + (y & 7),
                         (p[2] << 3) + (z & 7))];
  return mn + (float)data * (1.0f / 255.0f) * (mx - mn);
}

/* ------------------------------------------------------------------------------------ */
/* shaders/sampling/common.glsl                                                          */

/* common.glsl:35-43 */
float vxo_lookup_density_brick(const VxoVolume* v, int32_t x, int32_t y, int32_t z) {
  if (x < 0 || y < 0 || z < 0 || (uint32_t)x >= v->index_extent[0] ||
      (uint32_t)y >= v->index_extent[1] || (uint32_t)z >= v->index_extent[2])
    return 0.0f; /* out-of-range taps contribute 0 */
  int32_t bx = x >> 3, by = y >> 3, bz = z >> 3;
  float r, g;
  fetch_range(v, bx, by, bz, 0, &r, &g);
  float range_x = g, range_y = r; /* .yx */
  uint32_t p[3];
  fetch_ptr(v, bx, by, bz, p);
  float un = fetch_atlas(v, (p[0] << 3) + ((uint32_t)x & 7u), (p[1] << 3) + ((uint32_t)y & 7u),
                         (p[2] << 3) + ((uint32_t)z & 7u));
  return fmaf(un, range_y - range_x, range_x);
}

/* mix(x,y,a) = x*(1-a) + y*a */
static inline float gl_mix(float x, float y, float a) { return fmaf(y, a, x * (1.0f - a)); }

/* common.glsl:61-69 */
/* the eight taps of cell (ix,iy,iz) mixed x -> y -> z with the fractions given (common.glsl:62-68) */
float vxo_trilinear_cell(const VxoVolume* v, float density_scale, int32_t ix, int32_t iy, int32_t iz,
                         float fx, float fy, float fz) {
  float lx0 = gl_mix(vxo_lookup_density_brick(v, ix, iy, iz),
                     vxo_lookup_density_brick(v, ix + 1, iy, iz), fx);
  float lx1 = gl_mix(vxo_lookup_density_brick(v, ix, iy + 1, iz),
                     vxo_lookup_density_brick(v, ix + 1, iy + 1, iz), fx);
  float hx0 = gl_mix(vxo_lookup_density_brick(v, ix, iy, iz + 1),
                     vxo_lookup_density_brick(v, ix + 1, iy, iz + 1), fx);
  float hx1 = gl_mix(vxo_lookup_density_brick(v, ix, iy + 1, iz + 1),
                     vxo_lookup_density_brick(v, ix + 1, iy + 1, iz + 1), fx);
  return density_scale * gl_mix(gl_mix(lx0, lx1, fy), gl_mix(hx0, hx1, fy), fz);
}

float vxo_lookup_density_trilinear(const VxoVolume* v, float density_scale, float px, float py,
                                   float pz) {
  float qx = px - 0.5f, qy = py - 0.5f, qz = pz - 0.5f;
  float flx = floorf(qx), fly = floorf(qy), flz = floorf(qz);
  float fx = qx - flx, fy = qy - fly, fz = qz - flz; /* fract */
  int32_t ix = f2i(flx), iy = f2i(fly), iz = f2i(flz);
  return vxo_trilinear_cell(v, density_scale, ix, iy, iz, fx, fy, fz);
}

/* common.glsl:50-53 */
float vxo_lookup_majorant(const VxoVolume* v, float density_scale, float px, float py, float pz,
                          int32_t mip) {
  int32_t sh = 3 + mip;
  int32_t bx = f2i(floorf(px)) >> sh, by = f2i(floorf(py)) >> sh, bz = f2i(floorf(pz)) >> sh;
  float r, g;
  fetch_range(v, bx, by, bz, mip, &r, &g);
  return density_scale * r;
}

/* common.glsl:78-83; texture() with NEAREST + CLAMP_TO_EDGE (viewer.ts:386-389) selects
 * texel min(floor(d*L), L-1) for d in [0,1]; d<0 cannot pass the range test when
 * sample_range.x >= 0, but is clamped to texel 0 for completeness. */
void vxo_lookup_transfer(const float* tf, uint32_t L, const float sr[2], float d, float out[4]) {
  if (d < sr[0] || d > sr[1]) { /* NaN fails both tests and falls through: [build] texel 0 */
    out[0] = out[1] = out[2] = out[3] = 0.0f;
    return;
  }
  int32_t i = f2i(floorf(d * (float)L));
  if (i < 0) i = 0;
  if (i > (int32_t)L - 1) i = (int32_t)L - 1;
  memcpy(out, tf + 4 * (size_t)i, 16);
}

/* common.glsl:9-32 */
static void stochastic_tricubic_filter(v3 ipos, uint32_t s[4], int32_t tap[3]) {
  float q[3] = {ipos.x - 0.5f, ipos.y - 0.5f, ipos.z - 0.5f};
  int32_t ii[3];
  float t[3], t2[3];
  for (int c = 0; c < 3; ++c) {
    ii[c] = f2i(floorf(q[c]));
    t[c] = q[c] - (float)ii[c];
    t2[c] = t[c] * t[c];
  }
  const float sixth = 1.0f / 6.0f;
  float w[3], sum[3];
  int32_t idx[3] = {0, 0, 0};
  /* first tap: (1/6)*(-t*t2 + 3*t2 - 3*t + 1) */
  for (int c = 0; c < 3; ++c) {
    w[c] = sixth * (fmaf(-3.0f, t[c], fmaf(3.0f, t2[c], -t[c] * t2[c])) + 1.0f);
    sum[c] = w[c];
  }
  float r[3];
  /* second tap: (1/6)*(3*t*t2 - 6*t2 + 4) */
  for (int c = 0; c < 3; ++c) {
    w[c] = sixth * (fmaf(-6.0f, t2[c], 3.0f * t[c] * t2[c]) + 4.0f);
    sum[c] = w[c] + sum[c];
  }
  r[0] = vxo_rng(s); r[1] = vxo_rng(s); r[2] = vxo_rng(s);
  for (int c = 0; c < 3; ++c)
    if (r[c] < w[c] / gl_max(1e-3f, sum[c])) idx[c] = 1;
  /* third tap: (1/6)*(-3*t*t2 + 3*t2 + 3*t + 1) */
  for (int c = 0; c < 3; ++c) {
    w[c] = sixth * (fmaf(3.0f, t[c], fmaf(3.0f, t2[c], -3.0f * t[c] * t2[c])) + 1.0f);
    sum[c] = w[c] + sum[c];
  }
  r[0] = vxo_rng(s); r[1] = vxo_rng(s); r[2] = vxo_rng(s);
  for (int c = 0; c < 3; ++c)
    if (r[c] < w[c] / gl_max(1e-3f, sum[c])) idx[c] = 2;
  /* fourth tap: (1/6)*t*t2 */
  for (int c = 0; c < 3; ++c) {
    w[c] = sixth * t[c] * t2[c];
    sum[c] = w[c] + sum[c];
  }
  r[0] = vxo_rng(s); r[1] = vxo_rng(s); r[2] = vxo_rng(s);
  for (int c = 0; c < 3; ++c)
    if (r[c] < w[c] / gl_max(1e-3f, sum[c])) idx[c] = 3;
  for (int c = 0; c < 3; ++c) tap[c] = ii[c] + idx[c] - 1;
}

/* ------------------------------------------------------------------------------------ */
/* per-render state                                                                      */

typedef struct {
  const VxParams* p;
  const VxoVolume* v;
  const float* tf;
  uint32_t tf_len;
  uint32_t frame;
  VxoCounters c;
  const uint32_t* skip_bits; /* NULL: skipping off */
  int skip_level;
  uint32_t skip_dims[3];
  const VxoEnvironment* env; /* NULL: directional light (u_use_env < 1) */
} Ctx;

typedef struct { v3 o, d; } Ray;

/* common.glsl:56-58,72-76 */
static float lookup_density_stochastic(Ctx* k, v3 ipos, uint32_t s[4]) {
  int32_t tap[3];
  stochastic_tricubic_filter(ipos, s, tap);
  /* lookup_density(vec3(tap)): floor of an integer-valued float is itself */
  return k->p->volume_density_scale * vxo_lookup_density_brick(k->v, tap[0], tap[1], tap[2]);
}
/* the transfer function of a SAMPLE (counted: tf_samples = samples inside the sample range) */
static inline void lookup_transfer(Ctx* k, float d, float out[4]) {
  if (!(d < k->p->sample_range[0] || d > k->p->sample_range[1])) k->c.tf_samples++;
  vxo_lookup_transfer(k->tf, k->tf_len, k->p->sample_range, d, out);
}
static inline float trilinear(Ctx* k, v3 ip) {
  return vxo_lookup_density_trilinear(k->v, k->p->volume_density_scale, ip.x, ip.y, ip.z);
}

/* utils.glsl:61-69 */
static int ray_box_intersection(Ray r, const float bmin[3], const float bmax[3], float* near,
                                float* far) {
  float ix = 1.0f / r.d.x, iy = 1.0f / r.d.y, iz = 1.0f / r.d.z;
  float lox = (bmin[0] - r.o.x) * ix, loy = (bmin[1] - r.o.y) * iy, loz = (bmin[2] - r.o.z) * iz;
  float hix = (bmax[0] - r.o.x) * ix, hiy = (bmax[1] - r.o.y) * iy, hiz = (bmax[2] - r.o.z) * iz;
  float tminx = gl_min(lox, hix), tminy = gl_min(loy, hiy), tminz = gl_min(loz, hiz);
  float tmaxx = gl_max(lox, hix), tmaxy = gl_max(loy, hiy), tmaxz = gl_max(loz, hiz);
  *near = gl_max(0.0f, gl_max(tminx, gl_max(tminy, tminz)));
  *far = gl_min(tmaxx, gl_min(tmaxy, tmaxz));
  return *near <= *far;
}

/* raymarch.glsl:31-32 etc.: index-space origin and (non-normalised) direction */
static inline void to_index(Ctx* k, Ray r, v3* ipos, v3* idir) {
  float a[4], b[4];
  mat4_mul(k->p->density_transform_inv, r.o.x, r.o.y, r.o.z, 1.0f, a);
  mat4_mul(k->p->density_transform_inv, r.d.x, r.d.y, r.d.z, 0.0f, b);
  *ipos = V3(a[0], a[1], a[2]);
  *idir = V3(b[0], b[1], b[2]);
}

/* ---- RAYMARCH mode: sampling/raymarch.glsl ---- */
#define RAYMARCH_STEPS 64

/* raymarch.glsl:8-23 */
static float transmittance_raymarch(Ctx* k, Ray ray, uint32_t s[4]) {
  float near, far;
  if (!ray_box_intersection(ray, k->p->volume_aabb_min, k->p->volume_aabb_max, &near, &far))
    return 1.0f;
  v3 ipos, idir;
  to_index(k, ray, &ipos, &idir);
  float dt = (far - near) / (float)RAYMARCH_STEPS;
  near = fmaf(vxo_rng(s), dt, near);
  float tau = 0.0f;
  for (int i = 0; i < RAYMARCH_STEPS; ++i) {
    float t = gl_min(fmaf((float)i, dt, near), far);
    float d = lookup_density_stochastic(k, madd3(ipos, t, idir), s);
    float rgba[4];
    lookup_transfer(k, d * k->p->volume_inv_maj, rgba);
    tau = fmaf(rgba[3] * k->p->volume_maj, dt, tau);
    k->c.samples++;
  }
  return expf(-tau);
}

/* raymarch.glsl:25-55 */
static int sample_volume_raymarch(Ctx* k, Ray ray, float* t_out, v3* throughput, uint32_t s[4]) {
  float near, far;
  if (!ray_box_intersection(ray, k->p->volume_aabb_min, k->p->volume_aabb_max, &near, &far))
    return 0;
  v3 ipos, idir;
  to_index(k, ray, &ipos, &idir);
  float tau_target = -logf(1.0f - vxo_rng(s));
  float dt = (far - near) / (float)RAYMARCH_STEPS;
  near = fmaf(vxo_rng(s), dt, near);
  float tau = 0.0f;
  for (int i = 0; i < RAYMARCH_STEPS; ++i) {
    float t = gl_min(fmaf((float)i, dt, near), far);
    *t_out = t;
    float d = lookup_density_stochastic(k, madd3(ipos, t, idir), s);
    float rgba[4];
    lookup_transfer(k, d * k->p->volume_inv_maj, rgba);
    tau = fmaf(rgba[3] * k->p->volume_maj, dt, tau);
    k->c.samples++;
    if (tau >= tau_target) {
      throughput->x *= rgba[0] * k->p->volume_albedo[0];
      throughput->y *= rgba[1] * k->p->volume_albedo[1];
      throughput->z *= rgba[2] * k->p->volume_albedo[2];
      return 1;
    }
  }
  return 0;
}

/* ---- default mode: sampling/dda.glsl ---- */
#define MIP_START 3.0f
#define MIP_SPEED_UP 0.25f
#define MIP_SPEED_DOWN 2.0f
#define DDA_MAX_STEPS 100u /* dda.glsl:18 */

/* dda.glsl:11-16 */
static float step_dda(v3 pos, v3 inv_dir, int32_t mip) {
  float dim = (float)(8 << mip);
  float inv_dim = 1.0f / dim;
  float ox = (inv_dir.x >= 0.0f) ? dim + 0.5f : -0.5f;
  float oy = (inv_dir.y >= 0.0f) ? dim + 0.5f : -0.5f;
  float oz = (inv_dir.z >= 0.0f) ? dim + 0.5f : -0.5f;
  float tx = ((floorf(pos.x * inv_dim) * dim + ox) - pos.x) * inv_dir.x;
  float ty = ((floorf(pos.y * inv_dim) * dim + oy) - pos.y) * inv_dir.y;
  float tz = ((floorf(pos.z * inv_dim) * dim + oz) - pos.z) * inv_dir.z;
  return gl_min(tx, gl_min(ty, tz));
}

static inline float local_majorant(Ctx* k, v3 curr, int32_t mip) {
  float rgba[4];
  float m = vxo_lookup_majorant(k->v, k->p->volume_density_scale, curr.x, curr.y, curr.z, mip);
  vxo_lookup_transfer(k->tf, k->tf_len, k->p->sample_range, m * k->p->volume_inv_maj, rgba);   /* a majorant, not a sample */
  return k->p->volume_maj * rgba[3];
}

/* dda.glsl:21-62 */
static float transmittance_dda(Ctx* k, Ray ray, uint32_t s[4]) {
  float near, far;
  if (!ray_box_intersection(ray, k->p->volume_aabb_min, k->p->volume_aabb_max, &near, &far))
    return 1.0f;
  v3 ipos, idir;
  to_index(k, ray, &ipos, &idir);
  v3 ri = V3(1.0f / idir.x, 1.0f / idir.y, 1.0f / idir.z);
  float t = near + 1e-6f, Tr = 1.0f, tau = -logf(1.0f - vxo_rng(s)), mip = MIP_START;
  uint32_t step = 0;
  while (t < far && (step++ < DDA_MAX_STEPS)) {
    v3 curr = madd3(ipos, t, idir);
    int32_t m = f2i(roundf(mip)); /* [build] round half away from zero (quirk Q12) */
    float majorant = local_majorant(k, curr, m);
    float dt = step_dda(curr, ri, m);
    k->c.skip_steps++;
    t += dt;
    tau = fmaf(-majorant, dt, tau);
    mip = gl_min(mip + MIP_SPEED_UP, 3.0f);
    if (tau > 0.0f) continue;
    t += tau / majorant;
    if (t >= far) break;
    float rgba[4];
    lookup_transfer(k, trilinear(k, madd3(ipos, t, idir)) * k->p->volume_inv_maj, rgba);
    k->c.samples++;
    float d = k->p->volume_maj * rgba[3];
    if (vxo_rng(s) * majorant < d) {
      Tr *= gl_max(0.0f, 1.0f - k->p->volume_maj / majorant); /* quirk Q9 */
      if (Tr < 0.1f) {
        float prob = 1.0f - Tr;
        if (vxo_rng(s) < prob) return 0.0f;
        Tr /= 1.0f - prob;
      }
    }
    tau = -logf(1.0f - vxo_rng(s));
    mip = gl_max(0.0f, mip - MIP_SPEED_DOWN);
  }
  return Tr;
}

/* dda.glsl:65-98 (lookup_emission is a stub returning 0, common.glsl:87-89) */
static int sample_volume_dda(Ctx* k, Ray ray, float* t_out, v3* throughput, uint32_t s[4]) {
  float near, far;
  if (!ray_box_intersection(ray, k->p->volume_aabb_min, k->p->volume_aabb_max, &near, &far))
    return 0;
  v3 ipos, idir;
  to_index(k, ray, &ipos, &idir);
  v3 ri = V3(1.0f / idir.x, 1.0f / idir.y, 1.0f / idir.z);
  float t = near + 1e-6f;
  float tau = -logf(1.0f - vxo_rng(s)), mip = MIP_START;
  int hit = 0;
  while (t < far) {
    v3 curr = madd3(ipos, t, idir);
    int32_t m = f2i(roundf(mip));
    float majorant = local_majorant(k, curr, m);
    float dt = step_dda(curr, ri, m);
    k->c.skip_steps++;
    t += dt;
    tau = fmaf(-majorant, dt, tau);
    mip = gl_min(mip + MIP_SPEED_UP, 3.0f);
    if (tau > 0.0f) continue;
    t += tau / majorant;
    if (t >= far) break;
    float rgba[4];
    lookup_transfer(k, trilinear(k, madd3(ipos, t, idir)) * k->p->volume_inv_maj, rgba);
    k->c.samples++;
    float d = k->p->volume_maj * rgba[3];
    if (vxo_rng(s) * majorant < d) {
      throughput->x *= k->p->volume_albedo[0];
      throughput->y *= k->p->volume_albedo[1];
      throughput->z *= k->p->volume_albedo[2];
      throughput->x *= rgba[0];
      throughput->y *= rgba[1];
      throughput->z *= rgba[2];
      hit = 1;
      break;
    }
    tau = -logf(1.0f - vxo_rng(s));
    mip = gl_max(0.0f, mip - MIP_SPEED_DOWN);
  }
  *t_out = t;
  return hit;
}

/* ---- NO_DDA mode: sampling/normal.glsl ---- */

/* normal.glsl:6-31 */
static float transmittance_simple(Ctx* k, Ray ray, uint32_t s[4]) {
  float near, far;
  if (!ray_box_intersection(ray, k->p->volume_aabb_min, k->p->volume_aabb_max, &near, &far))
    return 1.0f;
  v3 ipos, idir;
  to_index(k, ray, &ipos, &idir);
  float t = fmaf(-logf(1.0f - vxo_rng(s)), k->p->volume_inv_maj, near), Tr = 1.0f;
  while (t < far) {
    float rgba[4];
    lookup_transfer(k, trilinear(k, madd3(ipos, t, idir)) * k->p->volume_inv_maj, rgba);
    k->c.samples++;
    float d = k->p->volume_maj * rgba[3];
    Tr *= fmaf(-d, k->p->volume_inv_maj, 1.0f);
    if (Tr < 0.1f) {
      float prob = 1.0f - Tr;
      if (vxo_rng(s) < prob) return 0.0f;
      Tr /= 1.0f - prob;
    }
    t = fmaf(-logf(1.0f - vxo_rng(s)), k->p->volume_inv_maj, t);
  }
  return Tr;
}

/* normal.glsl:33-57 */
static int sample_volume_simple(Ctx* k, Ray ray, float* t_out, v3* throughput, uint32_t s[4]) {
  float near, far;
  if (!ray_box_intersection(ray, k->p->volume_aabb_min, k->p->volume_aabb_max, &near, &far))
    return 0;
  v3 ipos, idir;
  to_index(k, ray, &ipos, &idir);
  float t = fmaf(-logf(1.0f - vxo_rng(s)), k->p->volume_inv_maj, near);
  int hit = 0;
  while (t < far) {
    float rgba[4];
    lookup_transfer(k, trilinear(k, madd3(ipos, t, idir)) * k->p->volume_inv_maj, rgba);
    k->c.samples++;
    float d = k->p->volume_maj * rgba[3];
    float p_real = d * k->p->volume_inv_maj;
    if (vxo_rng(s) < p_real) {
      throughput->x *= rgba[0] * k->p->volume_albedo[0];
      throughput->y *= rgba[1] * k->p->volume_albedo[1];
      throughput->z *= rgba[2] * k->p->volume_albedo[2];
      hit = 1;
      break;
    }
    t = fmaf(-logf(1.0f - vxo_rng(s)), k->p->volume_inv_maj, t);
  }
  *t_out = t;
  return hit;
}

/* sampling.glsl:11-44 */
static int sample_volume(Ctx* k, Ray ray, float* t, v3* thr, uint32_t s[4]) {
  switch (k->p->render_mode) {
    case VX_MODE_NO_DDA: return sample_volume_simple(k, ray, t, thr, s);
    case VX_MODE_RAYMARCH: return sample_volume_raymarch(k, ray, t, thr, s);
    default: return sample_volume_dda(k, ray, t, thr, s);
  }
}
static float transmittance(Ctx* k, Ray ray, uint32_t s[4]) {
  switch (k->p->render_mode) {
    case VX_MODE_NO_DDA: return transmittance_simple(k, ray, s);
    case VX_MODE_RAYMARCH: return transmittance_raymarch(k, ray, s);
    default: return transmittance_dda(k, ray, s);
  }
}

/* ---- utils.glsl / environment.glsl helpers ---- */
#define M_PI_F 3.14159265358979323846f
static const float INV_4PI = 1.0f / (4.0f * M_PI_F);

static inline float sqrf(float x) { return x * x; }
/* utils.glsl:100 */
static inline float luma(v3 c) {
  return dot3(c, V3(0.212671f, 0.715160f, 0.072169f));
}
/* utils.glsl:104 */
static inline float power_heuristic(float a, float b) { return sqrf(a) / (sqrf(a) + sqrf(b)); }
/* utils.glsl:121-124 */
static inline float phase_hg(float cos_t, float g) {
  float denom = fmaf(2.0f * g, cos_t, 1.0f + sqrf(g));
  return INV_4PI * (1.0f - sqrf(g)) / (denom * sqrtf(denom));
}
/* utils.glsl:106-114 */
static v3 align3(v3 N, v3 v) {
  v3 T;
  if (fabsf(N.x) > fabsf(N.y)) {
    float l = sqrtf(fmaf(N.z, N.z, N.x * N.x));
    T = V3(-N.z / l, 0.0f / l, N.x / l);
  } else {
    float l = sqrtf(fmaf(N.z, N.z, N.y * N.y));
    T = V3(0.0f / l, N.z / l, -N.y / l);
  }
  v3 B = cross3(N, T);
  v3 r = V3(fmaf(v.z, N.x, fmaf(v.y, B.x, v.x * T.x)), fmaf(v.z, N.y, fmaf(v.y, B.y, v.x * T.y)),
            fmaf(v.z, N.z, fmaf(v.y, B.z, v.x * T.z)));
  return normalize3(r);
}
/* utils.glsl:133-139 */
static v3 sample_phase_hg(v3 dir, float g, float u0, float u1) {
  float cos_t;
  if (fabsf(g) < 1e-4f) {
    cos_t = fmaf(-2.0f, u0, 1.0f);
  } else {
    float q = (1.0f - sqrf(g)) / fmaf(2.0f * g, u0, 1.0f - g);
    cos_t = ((1.0f + sqrf(g)) - sqrf(q)) / (2.0f * g);
  }
  float sin_t = sqrtf(gl_max(0.0f, 1.0f - sqrf(cos_t)));
  float phi = 2.0f * M_PI_F * u1;
  return align3(dir, V3(sin_t * cosf(phi), sin_t * sinf(phi), cos_t));
}

/* ---- environment map ------------------------------------------------------------------ */
uint32_t vxo_imp_offset(uint32_t level) {
  uint32_t o = 0;
  for (uint32_t k = 0; k < level; ++k) o += (VXO_IMP_DIM >> k) * (VXO_IMP_DIM >> k);
  return o;
}
/* environment.ts:30-32 UNPACK_FLIP_Y_WEBGL: source row 0 (top) becomes the last texel row */
void vxo_env_flip_rows(const float* src, uint32_t w, uint32_t h, float* dst) {
  for (uint32_t y = 0; y < h; ++y) memcpy(dst + (size_t)(h - 1 - y) * w * 4, src + (size_t)y * w * 4, (size_t)w * 16);
}
/* texture(u_envmap, uv): GL ES 3.0 section 3.8.10 LINEAR, wrap REPEAT / CLAMP_TO_EDGE
 * (environment.ts:22-26) */
static void env_texture(const float* tex, uint32_t w, uint32_t h, float u, float v, float rgb[3]) {
  float x = fmaf(u, (float)w, -0.5f), y = fmaf(v, (float)h, -0.5f);
  float fx = floorf(x), fy = floorf(y);
  float a = x - fx, b = y - fy;
  int32_t i0 = f2i(fx), j0 = f2i(fy);
  int32_t W = (int32_t)w, H = (int32_t)h;
  int32_t i1 = i0 + 1, j1 = j0 + 1;
  i0 %= W; if (i0 < 0) i0 += W;
  i1 %= W; if (i1 < 0) i1 += W;
  j0 = j0 < 0 ? 0 : (j0 > H - 1 ? H - 1 : j0);
  j1 = j1 < 0 ? 0 : (j1 > H - 1 ? H - 1 : j1);
  const float* t00 = tex + ((size_t)j0 * w + i0) * 4;
  const float* t10 = tex + ((size_t)j0 * w + i1) * 4;
  const float* t01 = tex + ((size_t)j1 * w + i0) * 4;
  const float* t11 = tex + ((size_t)j1 * w + i1) * 4;
  for (int c = 0; c < 3; ++c) {
    float lo = fmaf(t10[c], a, t00[c] * (1.0f - a)); /* mix(t00, t10, a) */
    float hi = fmaf(t11[c], a, t01[c] * (1.0f - a));
    rgb[c] = fmaf(hi, b, lo * (1.0f - b));
  }
}
void vxo_env_texture(const VxoEnvironment* e, float u, float v, float rgb[3]) {
  env_texture(e->texture, e->width, e->height, u, v, rgb);
}
/* envSetup.frag:26-41 dispatched over 512x512 with 8x8 samples per texel (environment.ts:9-10,
 * 49-56), then generateMipmap (environment.ts:58-60) */
void vxo_env_build_importance(const float* tex, uint32_t w, uint32_t h, float* pyr) {
  const int ns = 8;
  const float out_size = (float)(VXO_IMP_DIM * ns), inv_samples = 1.0f / (float)(ns * ns);
  for (uint32_t py = 0; py < VXO_IMP_DIM; ++py)
    for (uint32_t px = 0; px < VXO_IMP_DIM; ++px) {
      float imp = 0.0f;
      for (int y = 0; y < ns; ++y)
        for (int x = 0; x < ns; ++x) {
          float u = ((float)(px * ns) + ((float)x + 0.5f)) / out_size;
          float v = ((float)(py * ns) + ((float)y + 0.5f)) / out_size;
          float rgb[3];
          env_texture(tex, w, h, u, v, rgb);
          imp += luma(V3(rgb[0], rgb[1], rgb[2]));
        }
      pyr[(size_t)py * VXO_IMP_DIM + px] = imp * inv_samples;
    }
  for (uint32_t k = 1; k < VXO_IMP_LEVELS; ++k) {
    uint32_t n = VXO_IMP_DIM >> k, m = n * 2;
    const float* src = pyr + vxo_imp_offset(k - 1);
    float* dst = pyr + vxo_imp_offset(k);
    for (uint32_t y = 0; y < n; ++y)
      for (uint32_t x = 0; x < n; ++x) {
        float a = src[(size_t)(2 * y) * m + 2 * x], b = src[(size_t)(2 * y) * m + 2 * x + 1];
        float c = src[(size_t)(2 * y + 1) * m + 2 * x], d = src[(size_t)(2 * y + 1) * m + 2 * x + 1];
        dst[(size_t)y * n + x] = (((a + b) + c) + d) * 0.25f;
      }
  }
}
static inline float imp_fetch(const VxoEnvironment* e, int32_t x, int32_t y, int mip) {
  int32_t n = (int32_t)(VXO_IMP_DIM >> mip);
  if (x < 0 || y < 0 || x >= n || y >= n) return 0.0f; /* texelFetch out of range */
  return e->importance[vxo_imp_offset((uint32_t)mip) + (size_t)y * n + x];
}
/* environment.glsl:19-27 */
void vxo_env_lookup(const VxoEnvironment* e, float env_strength, const float dir[3], float rgb[3]) {
  float u = atan2f(dir[2], dir[0]) / (2.0f * M_PI_F) + 0.5f;
  float v = 1.0f - acosf(dir[1]) / M_PI_F;
  env_texture(e->texture, e->width, e->height, u, v, rgb);
  for (int c = 0; c < 3; ++c) rgb[c] = env_strength * rgb[c];
}
/* environment.glsl:29-79 */
void vxo_env_sample(const VxoEnvironment* e, float env_strength, float u0, float u1, float w_i[3],
                    float le_pdf[4]) {
  int32_t px = 0, py = 0;
  float sx = u0, sy = u1;
  for (int mip = (int)VXO_IMP_LEVELS - 2; mip >= 0; --mip) {
    px *= 2; py *= 2;
    float w[4] = {imp_fetch(e, px, py, mip), imp_fetch(e, px + 1, py, mip), imp_fetch(e, px, py + 1, mip),
                  imp_fetch(e, px + 1, py + 1, mip)};
    float q[2] = {w[0] + w[2], w[1] + w[3]};
    int off_x;
    float d = q[0] / gl_max(1e-8f, q[0] + q[1]);
    if (sx < d) { off_x = 0; sx = sx / d; }
    else { off_x = 1; sx = (sx - d) / (1.0f - d); }
    px += off_x;
    float ee = w[off_x] / q[off_x];
    if (sy < ee) { sy = sy / ee; }
    else { py += 1; sy = (sy - ee) / (1.0f - ee); }
  }
  const float inv_dim = 1.0f / (float)VXO_IMP_DIM;
  float uvx = ((float)px + sx) * inv_dim, uvy = ((float)py + sy) * inv_dim;
  float theta = gl_clamp(1.0f - uvy, 0.0f, 1.0f) * M_PI_F;
  float phi = (gl_clamp(uvx, 0.0f, 1.0f) * 2.0f - 1.0f) * M_PI_F;
  float sin_t = sinf(theta);
  w_i[0] = sin_t * cosf(phi); w_i[1] = cosf(theta); w_i[2] = sin_t * sinf(phi);
  float rgb[3];
  env_texture(e->texture, e->width, e->height, uvx, uvy, rgb);
  float avg_w = imp_fetch(e, 0, 0, (int)VXO_IMP_LEVELS - 1);
  float pdf = imp_fetch(e, px, py, 0) / avg_w;
  for (int c = 0; c < 3; ++c) le_pdf[c] = env_strength * rgb[c];
  le_pdf[3] = pdf * INV_4PI;
}
/* environment.glsl:82-86 */
float vxo_env_pdf(const VxoEnvironment* e, float env_strength, const float dir[3]) {
  float rgb[3];
  vxo_env_lookup(e, env_strength, dir, rgb);
  float avg_w = imp_fetch(e, 0, 0, (int)VXO_IMP_LEVELS - 1);
  return luma(V3(rgb[0], rgb[1], rgb[2])) / avg_w * INV_4PI;
}

/* environment.glsl:19-27.  Directional branch: [build] pow() of a negative base is undefined in
 * GLSL ES 3.00; the base is clamped at 0 first (quirk Q16). */
static v3 lookup_environment(Ctx* k, v3 dir) {
  if (k->p->use_env > 0 && k->env) {
    float d[3] = {dir.x, dir.y, dir.z}, rgb[3];
    vxo_env_lookup(k->env, k->p->env_strength, d, rgb);
    return V3(rgb[0], rgb[1], rgb[2]);
  }
  v3 nl = V3(-k->p->light_dir[0], -k->p->light_dir[1], -k->p->light_dir[2]);
  float c = gl_max(dot3(dir, nl), 0.0f);
  float s = gl_clamp(powf(c, 300.0f), 0.0f, 1.0f);
  float e = k->p->env_strength * fmaf(s, 4.0f, 0.01f);
  return V3(e, e, e);
}
/* environment.glsl:82-86 -- with the directional light [build] the pdf of hitting a delta light by
 * phase sampling is 0 (avg_w would come from an importance map that does not exist). */
static inline float pdf_environment(Ctx* k, v3 dir) {
  if (k->p->use_env > 0 && k->env) {
    float d[3] = {dir.x, dir.y, dir.z};
    return vxo_env_pdf(k->env, k->p->env_strength, d);
  }
  return 0.0f;
}

/* fragment.frag:79-124 */
static void trace_path(Ctx* k, Ray ray, uint32_t s[4], float out[4]) {
  v3 L = V3(0, 0, 0), thr = V3(1, 1, 1);
  int free_path = 1;
  uint32_t n_paths = 0;
  float t = 0.0f, f_p = 0.0f;
  const VxParams* p = k->p;
  while (1) {
    int hit = sample_volume(k, ray, &t, &thr, s);
    if (!hit) break;
    ray.o = madd3(ray.o, t, ray.d); /* :88 */
    /* :92 sample_environment(rng2(seed), w_i): two draws consumed even with the
       directional light (environment.glsl:30-33) */
    float e0 = vxo_rng(s), e1 = vxo_rng(s);
    v3 w_i = V3(-p->light_dir[0], -p->light_dir[1], -p->light_dir[2]);
    float Le3[3] = {p->env_strength * 4.01f, p->env_strength * 4.01f, p->env_strength * 4.01f}, pdf = 1.0f;
    if (p->use_env > 0 && k->env) {
      float wi[3], lp[4];
      vxo_env_sample(k->env, p->env_strength, e0, e1, wi, lp);
      w_i = V3(wi[0], wi[1], wi[2]);
      Le3[0] = lp[0]; Le3[1] = lp[1]; Le3[2] = lp[2]; pdf = lp[3];
    }
    if (pdf > 0.0f) {
      f_p = phase_hg(dot3(neg3(ray.d), w_i), p->volume_phase_g);
      float mis = p->show_environment > 0 ? power_heuristic(pdf, f_p) : 1.0f;
      Ray sr = {ray.o, w_i};
      float Tr = transmittance(k, sr, s);
      /* :97 L += throughput * mis_weight * f_p * Tr * Le_pdf.rgb / Le_pdf.w */
      L.x += thr.x * mis * f_p * Tr * Le3[0] / pdf;
      L.y += thr.y * mis * f_p * Tr * Le3[1] / pdf;
      L.z += thr.z * mis * f_p * Tr * Le3[2] / pdf;
    }
    if (++n_paths >= (uint32_t)p->bounces) { free_path = 0; break; } /* :101 */
    float rr = luma(thr); /* :103-108 */
    if (rr < 0.1f) {
      float prob = 1.0f - rr;
      if (vxo_rng(s) < prob) { free_path = 0; break; }
      float q = 1.0f - prob;
      thr = V3(thr.x / q, thr.y / q, thr.z / q);
    }
    float u0 = vxo_rng(s), u1 = vxo_rng(s); /* :111 */
    v3 sd = sample_phase_hg(ray.d, p->volume_phase_g, u0, u1);
    f_p = phase_hg(dot3(neg3(ray.d), sd), p->volume_phase_g);
    ray.d = sd;
  }
  if (free_path && p->show_environment > 0) { /* :117-121 */
    v3 Le = lookup_environment(k, ray.d);
    float mis = n_paths > 0u ? power_heuristic(f_p, pdf_environment(k, ray.d)) : 1.0f;
    L.x = fmaf(thr.x * mis, Le.x, L.x);
    L.y = fmaf(thr.y * mis, Le.y, L.y);
    L.z = fmaf(thr.z * mis, Le.z, L.z);
  }
  out[0] = L.x; out[1] = L.y; out[2] = L.z;
  out[3] = gl_clamp((float)n_paths, 0.0f, 1.0f);
}


/* ---- [build] exact empty-space skipping ------------------------------------------------- */
int vxo_skip_level(const VxoVolume* v) {
  for (int g = 1; g <= 3; ++g) {
    uint64_t n = 1;
    for (int a = 0; a < 3; ++a) n *= (uint64_t)(v->index_extent[a] >> (3 + g)) + 1u;
    if (n <= 65536u) return g;
  }
  return 3;
}
void vxo_skip_dims(const VxoVolume* v, int level, uint32_t dims[3]) {
  for (int a = 0; a < 3; ++a) dims[a] = (v->index_extent[a] >> (3 + level)) + 1u;
}
/* TF bin i can produce no opacity: alpha 0, or wholly outside the sample range (one-bin margin) */
static int bin_dead(const VxParams* p, const float* tf, uint32_t L, int32_t i) {
  if (tf[4 * (size_t)i + 3] == 0.0f) return 1;
  float lf = (float)L;
  if ((float)(i + 2) / lf < p->sample_range[0]) return 1;
  if ((float)(i - 1) / lf > p->sample_range[1]) return 1;
  return 0;
}
/* every density in [lo, hi] maps to dead bins (bins I(lo)-1 .. I(hi)+1) */
static int range_transparent(const VxParams* p, const float* tf, uint32_t L, float lo, float hi) {
  float lf = (float)L;
  int32_t a = f2i(floorf(((lo * p->volume_density_scale) * p->volume_inv_maj) * lf)) - 1;
  int32_t b = f2i(floorf(((hi * p->volume_density_scale) * p->volume_inv_maj) * lf)) + 1;
  if (a < 0) a = 0;
  if (b > (int32_t)L - 1) b = (int32_t)L - 1;
  for (int32_t i = a; i <= b; ++i)
    if (!bin_dead(p, tf, L, i)) return 0;
  return 1;
}
void vxo_build_skip_mask(const VxParams* p, const VxoVolume* v, const float* tf, uint32_t L, int level,
                         uint32_t* bits) {
  uint32_t md[3];
  vxo_skip_dims(v, level, md);
  size_t n = (size_t)md[0] * md[1] * md[2];
  memset(bits, 0, ((n + 31) / 32) * 4);
  size_t nb = (size_t)v->range_size[0] * v->range_size[1] * v->range_size[2];
  uint8_t* bt = (uint8_t*)malloc(nb ? nb : 1); /* per-brick transparency */
  for (size_t i = 0; i < nb; ++i)
    bt[i] = (uint8_t)range_transparent(p, tf, L, vxo_f16_to_f32(v->range[2 * i + 1]), vxo_f16_to_f32(v->range[2 * i]));
  int zero_ok = range_transparent(p, tf, L, 0.0f, 0.0f);
  int32_t w = 1 << level; /* bricks per macro cell and axis */
  for (uint32_t mz = 0; mz < md[2]; ++mz)
    for (uint32_t my = 0; my < md[1]; ++my)
      for (uint32_t mx = 0; mx < md[0]; ++mx) {
        int32_t lo[3] = {(int32_t)mx * w - 1, (int32_t)my * w - 1, (int32_t)mz * w - 1};
        int empty = 1;
        for (int32_t bz = lo[2]; bz <= lo[2] + w && empty; ++bz)
          for (int32_t by = lo[1]; by <= lo[1] + w && empty; ++by)
            for (int32_t bx = lo[0]; bx <= lo[0] + w && empty; ++bx) {
              if (bx < 0 || by < 0 || bz < 0 || (uint32_t)bx >= v->range_size[0] ||
                  (uint32_t)by >= v->range_size[1] || (uint32_t)bz >= v->range_size[2])
                empty = zero_ok;
              else
                empty = bt[buf_index(v->range_size, (uint32_t)bx, (uint32_t)by, (uint32_t)bz)];
            }
        if (empty) {
          size_t i = ((size_t)mz * md[1] + my) * md[0] + mx;
          bits[i >> 5] |= 1u << (i & 31);
        }
      }
  free(bt);
}
/* is trilinear cell c (= floor of the cell-frame position) in a macro cell flagged empty? */
static inline int cell_is_skipped(const Ctx* k, int32_t c0, int32_t c1, int32_t c2) {
  if (!k->skip_bits) return 0;
  int sh = 3 + k->skip_level;
  int32_t cx = c0 + 1, cy = c1 + 1, cz = c2 + 1;
  if (cx < 0 || cy < 0 || cz < 0) return 0;
  uint32_t mx = (uint32_t)cx >> sh, my = (uint32_t)cy >> sh, mz = (uint32_t)cz >> sh;
  if (mx >= k->skip_dims[0] || my >= k->skip_dims[1] || mz >= k->skip_dims[2]) return 0;
  size_t i = ((size_t)mz * k->skip_dims[1] + my) * k->skip_dims[0] + mx;
  return (k->skip_bits[i >> 5] >> (i & 31)) & 1u;
}

/* ---- [build] deterministic DVR = E[RAYMARCH, bounces 1, no shadow term] ------------- */
/* SURVEY.md section 8 row A12 / Appendix A.5.  Loop body restates raymarch.glsl:38-52
 * with lookup_density_trilinear (common.glsl:61-69) in place of the stochastic tap and the
 * early-out on accumulated optical depth.                                               */
/* The march contract of rounds 1-2, kept as a second path: t_k = fma(k, dt, t0), a sample exists while
 * t_k < far (and k < max_steps), its position is fma(t_k, idir, ipos) and the cell frame of A5 is that minus 1/2.  The same
 * line, walked with one more rounding per axis; vxo_set_dvr_march(1) selects it.  tests/test_tolerance_envelope.py pins
 * the shipped contract against it: per-ray sample counts within +-1, images within the stated tolerance. */
static int g_dvr_walk_t = 0;
void vxo_set_dvr_march(int walk_t) { g_dvr_walk_t = walk_t; }
int vxo_get_dvr_march(void) { return g_dvr_walk_t; }

static void dvr_pixel(Ctx* k, Ray ray, float start_offset, int phong, float out[4]) {
  const VxParams* p = k->p;
  float near, far;
  v3 C = V3(0, 0, 0);
  float T = 1.0f;
  int hit = ray_box_intersection(ray, p->volume_aabb_min, p->volume_aabb_max, &near, &far);
  if (hit) {
    k->c.rays++;
    v3 ipos, idir;
    to_index(k, ray, &ipos, &idir);
    float dt = p->dvr_step_voxels / sqrtf(dot3(idir, idir));
    float t0 = fmaf(start_offset, dt, near);
    float tau = 0.0f, kf = 0.0f;
    /* [build] march contract: the ray has n = min(ceil((far - t0) / dt), max_steps) samples
       (none unless the quotient is positive); sample k sits at q = fma(k, dq, q0) in the cell frame of A5
       (position - 1/2), dq = dt * idir, q0 = fma(t0, idir, ipos) - 1/2, per axis.  (The first form walked
       t_k = fma(k, dt, t0) and formed fma(t_k, idir, ipos) - 1/2: one more rounding and five more operations per
       sample for the same line.) */
    const float xq = (far - t0) / dt;
    const float nf = (xq > 0.0f) ? fminf(ceilf(xq), (float)p->dvr_max_steps) : 0.0f;
    const v3 dq = V3(dt * idir.x, dt * idir.y, dt * idir.z);
    const v3 q0 = V3(fmaf(t0, idir.x, ipos.x) - 0.5f, fmaf(t0, idir.y, ipos.y) - 0.5f, fmaf(t0, idir.z, ipos.z) - 0.5f);
    v3 nl = V3(-p->light_dir[0], -p->light_dir[1], -p->light_dir[2]);
    v3 hv = V3(0, 0, 0);
    if (phong) hv = normalize3(sub3(nl, ray.d)); /* Blinn half vector of l and v = -dir */
    const int walk_t = g_dvr_walk_t;
    const float n_steps = walk_t ? (float)p->dvr_max_steps : nf;
    for (; kf < n_steps; kf += 1.0f) {
      float qx, qy, qz;
      if (walk_t) {                                  /* the rounds 1-2 contract */
        const float t = fmaf(kf, dt, t0);
        if (!(t < far)) break;
        qx = fmaf(t, idir.x, ipos.x) - 0.5f; qy = fmaf(t, idir.y, ipos.y) - 0.5f; qz = fmaf(t, idir.z, ipos.z) - 0.5f;
      } else {
        qx = fmaf(kf, dq.x, q0.x); qy = fmaf(kf, dq.y, q0.y); qz = fmaf(kf, dq.z, q0.z);
      }
      const float flx = floorf(qx), fly = floorf(qy), flz = floorf(qz);
      const float fx = qx - flx, fy = qy - fly, fz = qz - flz;
      const int32_t cx = f2i(flx), cy = f2i(fly), cz = f2i(flz);
      if (cell_is_skipped(k, cx, cy, cz)) { k->c.skip_steps++; continue; } /* alpha would be exactly 0 */
      /* lookup_density_trilinear, common.glsl:61-69, from the cell and fractions of q */
      float d = vxo_trilinear_cell(k->v, p->volume_density_scale, cx, cy, cz, fx, fy, fz);
      float rgba[4];
      lookup_transfer(k, d * p->volume_inv_maj, rgba);
      k->c.samples++;
      if (rgba[3] > 0.0f) {
        if (phong) {
          k->c.grad_samples++;
          /* [build] central differences of A5 one voxel either side of the sample along each axis (6 extra
             trilinear look-ups, BASELINE config 4), taken in the sample's own cell frame: the shifted look-ups
             read the cells c +- e with the sample's fractions -- fract((p +- 1) - 0.5) equals fract(p - 0.5)
             mathematically, and forming it from the rounded p +- 1 would only add rounding noise.  World
             gradient = g_i * Minv_ii */
          const float ds = p->volume_density_scale;
          float gx = vxo_trilinear_cell(k->v, ds, cx + 1, cy, cz, fx, fy, fz) - vxo_trilinear_cell(k->v, ds, cx - 1, cy, cz, fx, fy, fz);
          float gy = vxo_trilinear_cell(k->v, ds, cx, cy + 1, cz, fx, fy, fz) - vxo_trilinear_cell(k->v, ds, cx, cy - 1, cz, fx, fy, fz);
          float gz = vxo_trilinear_cell(k->v, ds, cx, cy, cz + 1, fx, fy, fz) - vxo_trilinear_cell(k->v, ds, cx, cy, cz - 1, fx, fy, fz);
          v3 g = V3(gx * p->density_transform_inv[0], gy * p->density_transform_inv[5],
                    gz * p->density_transform_inv[10]);
          float g2 = dot3(g, g);
          if (g2 > 1e-12f) {
            v3 n = scale3(g, -1.0f / sqrtf(g2));
            float ndl = gl_max(0.0f, dot3(n, nl));
            float ndh = gl_max(0.0f, dot3(n, hv));
            float diff = fmaf(p->phong_kd, ndl, p->phong_ka);
            float spec = p->phong_ks * powf(ndh, p->phong_shininess);
            rgba[0] = fmaf(rgba[0], diff, spec);
            rgba[1] = fmaf(rgba[1], diff, spec);
            rgba[2] = fmaf(rgba[2], diff, spec);
          }
        }
        tau = fmaf(rgba[3] * p->volume_maj, dt, tau);
        float Tn = expf(-tau);
        float dT = T - Tn;
        C.x = fmaf(dT, rgba[0], C.x);
        C.y = fmaf(dT, rgba[1], C.y);
        C.z = fmaf(dT, rgba[2], C.z);
        T = Tn;
        if (tau >= p->dvr_ert_tau) { T = 0.0f; break; }
      }
    }
  }
  v3 L = V3(C.x * p->dvr_gain[0], C.y * p->dvr_gain[1], C.z * p->dvr_gain[2]);
  if (p->show_environment > 0 && T > 0.0f) {
    v3 Le = lookup_environment(k, ray.d);
    L.x = fmaf(T, Le.x, L.x);
    L.y = fmaf(T, Le.y, L.y);
    L.z = fmaf(T, Le.z, L.z);
  }
  out[0] = L.x; out[1] = L.y; out[2] = L.z;
  out[3] = hit ? 1.0f : 0.0f;
}

/* ---- camera: fragment.frag:57-65 + utils.glsl:23-40 with the inverses hoisted -------- */
static Ray setup_world_ray(const VxParams* p, float tex_x, float tex_y, float rx, float ry) {
  float x_off = fmaf(rx, 2.0f, -1.0f) * (1.0f / (float)p->res[0]);
  float y_off = fmaf(ry, 2.0f, -1.0f) * (1.0f / (float)p->res[1]);
  float sx = tex_x + x_off, sy = tex_y + y_off;
  if (p->camera_ortho) {
    /* [build] orthographic camera (BASELINE config 1; the reference camera is perspective only,
       scene.ts:65-72): the ray starts at the unprojected near-plane point of the pixel and runs along
       the camera's -z axis: o = inverse(view) * inverse(proj) * (ndc.xy, -1, 1), d = normalize(inverse(view)
       * (0,0,-1,0)).  Same mat4 * vec4 fma chains and perspective divides as the perspective branch. */
    float np_[4], wo[4], wd[4];
    mat4_mul(p->camera_proj_inv, fmaf(sx, 2.0f, -1.0f), fmaf(sy, 2.0f, -1.0f), -1.0f, 1.0f, np_);
    mat4_mul(p->camera_view_inv, np_[0] / np_[3], np_[1] / np_[3], np_[2] / np_[3], 1.0f, wo);
    mat4_mul(p->camera_view_inv, 0.0f, 0.0f, -1.0f, 0.0f, wd);
    Ray ro;
    ro.o = V3(wo[0] / wo[3], wo[1] / wo[3], wo[2] / wo[3]);
    ro.d = normalize3(V3(wd[0], wd[1], wd[2]));
    return ro;
  }
  /* cameraWorldPos */
  float cw[4];
  mat4_mul(p->camera_view_inv, 0.0f, 0.0f, 0.0f, 1.0f, cw);
  v3 cam = V3(cw[0] / cw[3], cw[1] / cw[3], cw[2] / cw[3]);
  /* cameraWorldDir */
  float vp[4], wp[4];
  mat4_mul(p->camera_proj_inv, fmaf(sx, 2.0f, -1.0f), fmaf(sy, 2.0f, -1.0f), 0.0f, 1.0f, vp);
  float vx = vp[0] / vp[3], vy = vp[1] / vp[3], vz = vp[2] / vp[3];
  mat4_mul(p->camera_view_inv, vx, vy, vz, 1.0f, wp);
  v3 world = V3(wp[0] / wp[3], wp[1] / wp[3], wp[2] / wp[3]);
  Ray r;
  r.o = cam;
  r.d = normalize3(sub3(world, cam));
  return r;
}

static inline float sanitize1(float x) { return (x != x || isinf(x)) ? 0.0f : x; }

/* fragment.frag:128-158 for one pixel */
static void shade_pixel(Ctx* k, int32_t px, int32_t py, float result[4], Ray* ray_out) {
  const VxParams* p = k->p;
  uint32_t s[4];
  vxo_seed_xoshiro(vxo_pixel_seed((uint32_t)px, (uint32_t)py, (uint32_t)p->res[0], k->frame), s);
  float tex_x = ((float)px + 0.5f) / (float)p->res[0];
  float tex_y = ((float)py + 0.5f) / (float)p->res[1];
  /* :146 (rng2 + rng2)/2 -- always drawn, also in debugHits mode */
  float a0 = vxo_rng(s), a1 = vxo_rng(s), b0 = vxo_rng(s), b1 = vxo_rng(s);
  float jx = (a0 + b0) / 2.0f, jy = (a1 + b1) / 2.0f;
  int dvr = p->render_mode == VX_MODE_DVR || p->render_mode == VX_MODE_DVR_PHONG;
  if (dvr && !p->dvr_jitter) { jx = 0.5f; jy = 0.5f; }
  Ray ray = setup_world_ray(p, tex_x, tex_y, jx, jy);
  if (ray_out) *ray_out = ray;
  if (p->debug_hits) { /* :147-153 */
    float near, far;
    if (ray_box_intersection(ray, p->volume_aabb_min, p->volume_aabb_max, &near, &far)) {
      v3 hit_min = madd3(ray.o, near, ray.d); /* utils.glsl:70-84 (near >= 0 always) */
      result[0] = (hit_min.x - p->volume_aabb_min[0]) / (p->volume_aabb_max[0] - p->volume_aabb_min[0]);
      result[1] = (hit_min.y - p->volume_aabb_min[1]) / (p->volume_aabb_max[1] - p->volume_aabb_min[1]);
      result[2] = (hit_min.z - p->volume_aabb_min[2]) / (p->volume_aabb_max[2] - p->volume_aabb_min[2]);
      result[3] = 1.0f;
      k->c.rays++;
    } else {
      v3 bg = lookup_environment(k, ray.d); /* get_background_color, u_hide_envmap == 0 */
      result[0] = bg.x; result[1] = bg.y; result[2] = bg.z; result[3] = 1.0f;
    }
    return;
  }
  if (dvr) {
    /* draw order mirrors raymarch.glsl:28-30: tau_target slot, then start jitter */
    float u_unused = vxo_rng(s);
    float u_start = vxo_rng(s);
    (void)u_unused;
    float off = p->dvr_jitter ? u_start : 0.5f;
    dvr_pixel(k, ray, off, p->render_mode == VX_MODE_DVR_PHONG, result);
  } else {
    float near, far;
    if (ray_box_intersection(ray, p->volume_aabb_min, p->volume_aabb_max, &near, &far)) k->c.rays++;
    trace_path(k, ray, s, result);
  }
  for (int i = 0; i < 4; ++i) result[i] = sanitize1(result[i]); /* :155, utils.glsl:96-98 */
}

int vxo_render(const VxParams* p, uint32_t frame_index, float sample_weight, const VxoVolume* v,
               const float* tf, uint32_t tf_len, const float* prev, float* out, int32_t x0,
               int32_t x1, int32_t y0, int32_t y1, VxoCounters* counters) {
  return vxo_render_env(p, frame_index, sample_weight, v, tf, tf_len, NULL, prev, out, x0, x1, y0, y1, counters);
}

static int render_rect(const VxParams* p, uint32_t frame_index, float sample_weight, const VxoVolume* v,
                       const float* tf, uint32_t tf_len, const VxoEnvironment* env, const float* prev,
                       float* out, uint32_t* ray_samples, int32_t x0, int32_t x1, int32_t y0, int32_t y1,
                       VxoCounters* counters);

int vxo_render_env(const VxParams* p, uint32_t frame_index, float sample_weight, const VxoVolume* v,
                   const float* tf, uint32_t tf_len, const VxoEnvironment* env, const float* prev,
                   float* out, int32_t x0, int32_t x1, int32_t y0, int32_t y1, VxoCounters* counters) {
  return render_rect(p, frame_index, sample_weight, v, tf, tf_len, env, prev, out, NULL, x0, x1, y0, y1, counters);
}

/* the same, and the number of samples every pixel's paths evaluated (samples + skipped steps) into
   ray_samples[py * res.x + px] -- for tests that compare march contracts ray by ray */
int vxo_render_ray_samples(const VxParams* p, uint32_t frame_index, const VxoVolume* v, const float* tf,
                           uint32_t tf_len, const VxoEnvironment* env, float* out, uint32_t* ray_samples,
                           int32_t x0, int32_t x1, int32_t y0, int32_t y1, VxoCounters* counters) {
  if (!ray_samples) return 1;
  return render_rect(p, frame_index, 0.0f, v, tf, tf_len, env, NULL, out, ray_samples, x0, x1, y0, y1, counters);
}

static int render_rect(const VxParams* p, uint32_t frame_index, float sample_weight, const VxoVolume* v,
                       const float* tf, uint32_t tf_len, const VxoEnvironment* env, const float* prev,
                       float* out, uint32_t* ray_samples, int32_t x0, int32_t x1, int32_t y0, int32_t y1,
                       VxoCounters* counters) {
  if (!p || !v || !tf || !out || tf_len == 0) return 1;
  if (p->use_env > 0 && !env) return 2;
  Ctx k;
  memset(&k, 0, sizeof k);
  k.p = p; k.v = v; k.tf = tf; k.tf_len = tf_len; k.frame = frame_index; k.env = env;
  uint32_t* mask = NULL;
  if (p->dvr_skip_empty && (p->render_mode == VX_MODE_DVR || p->render_mode == VX_MODE_DVR_PHONG) &&
      !p->debug_hits) {
    k.skip_level = vxo_skip_level(v);
    vxo_skip_dims(v, k.skip_level, k.skip_dims);
    size_t n = (size_t)k.skip_dims[0] * k.skip_dims[1] * k.skip_dims[2];
    mask = (uint32_t*)malloc(((n + 31) / 32) * 4);
    vxo_build_skip_mask(p, v, tf, tf_len, k.skip_level, mask);
    k.skip_bits = mask;
  }
  int32_t W = p->res[0];
  float w = sample_weight;
  for (int32_t py = y0; py < y1; ++py)
    for (int32_t px = x0; px < x1; ++px) {
      float r[4];
      const uint64_t before = k.c.samples + k.c.skip_steps;
      shade_pixel(&k, px, py, r, NULL);
      if (ray_samples) ray_samples[(size_t)py * W + px] = (uint32_t)(k.c.samples + k.c.skip_steps - before);
      size_t o = ((size_t)py * W + px) * 4;
      /* :158 out = (w*prev + (1-w)*result).rgb, alpha 1 */
      for (int c = 0; c < 3; ++c) {
        float pv = (prev && w != 0.0f) ? prev[o + c] : 0.0f;
        out[o + c] = fmaf(1.0f - w, r[c], w * pv);
      }
      out[o + 3] = 1.0f;
      k.c.pixels++;
    }
  if (counters) *counters = k.c;
  free(mask);
  return 0;
}

void vxo_primary_ray(const VxParams* p, uint32_t frame_index, int32_t px, int32_t py,
                     float origin[3], float dir[3]) {
  Ctx k;
  memset(&k, 0, sizeof k);
  VxParams q = *p;
  q.debug_hits = 1;
  VxoVolume dummy;
  memset(&dummy, 0, sizeof dummy);
  k.p = &q; k.v = &dummy; k.frame = frame_index;
  float r[4];
  Ray ray;
  shade_pixel(&k, px, py, r, &ray);
  origin[0] = ray.o.x; origin[1] = ray.o.y; origin[2] = ray.o.z;
  dir[0] = ray.d.x; dir[1] = ray.d.y; dir[2] = ray.d.z;
}

/* blit.frag:17-35 */
static inline float hable(float x) {
  const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
  float num = fmaf(D, E, x * fmaf(C, B, A * x));
  float den = fmaf(D, F, x * (A * x + B));
  return num / den - E / F;
}
void vxo_blit(const float* accum, uint32_t n, float exposure, float gamma, uint8_t* rgba8,
              float* rgba_f32) {
  float white = hable(11.2f);
  float ig = 1.0f / gamma;
  for (uint32_t i = 0; i < n; ++i) {
    float o[4];
    for (int c = 0; c < 3; ++c) o[c] = powf(hable(exposure * accum[4 * i + c]) / white, ig);
    o[3] = accum[4 * i + 3];
    for (int c = 0; c < 4; ++c) {
      if (rgba_f32) rgba_f32[4 * i + c] = o[c];
      if (rgba8) { /* GL float -> unorm8: round(clamp(f,0,1)*255) */
        float f = o[c];
        if (f != f) f = 0.0f;
        f = f < 0.0f ? 0.0f : (f > 1.0f ? 1.0f : f);
        rgba8[4 * i + c] = (uint8_t)f2i(floorf(fmaf(f, 255.0f, 0.5f)));
      }
    }
  }
}
