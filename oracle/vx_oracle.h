/*
 * vx_oracle.h -- CPU oracle for the Volxel raymarch hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker.  Nothing under volxel_amd/ links, imports or calls it.
 *
 * PARITY UNPINNED: the reference (Volxel/Volxel @ 2025-12-26) ships no tests, golden
 * images, fixtures or known-answer vectors for this path, and
 * none of its three languages (GLSL ES 3.00, TypeScript, Rust->wasm) can be built or run
 * in the build image.  This file is a from-scratch scalar restatement that follows the
 * cited reference lines; the only available "pins" are the canonical TEA / Wang-hash /
 * IEEE binary16 definitions and agreement with the independent NumPy restatement in
 * oracle/np_oracle.py (RNG, brick codec, trilinear, DVR, and the three reference render modes with
 * trace_path pixel by pixel: bit-identical on the test scenes).
 */
#ifndef VX_ORACLE_H
#define VX_ORACLE_H

#include <stdint.h>
#include "../include/volxel_hip.h" /* VxParams: the uniform block is the shared contract */

#ifdef __cplusplus
extern "C" {
#endif

/* the brick-grid textures exactly as uploaded at viewer.ts:1106-1142 */
typedef struct VxoVolume {
  const uint32_t* indirection; /* RGB10_A2UI texels, brick.rs:30-35                     */
  uint32_t ind_size[3];
  const uint16_t* range;       /* RG16F texels: u16[2i]=R=max, u16[2i+1]=G=min           */
  uint32_t range_size[3];
  const uint8_t* atlas;        /* R8 unorm                                               */
  uint32_t atlas_size[3];
  const uint16_t* mips[3];     /* GL mip levels 1..3 of the range texture                */
  uint32_t mip_size[3][3];
  uint32_t index_extent[3];    /* padded extent = brick_count*8, brick.rs:236-238        */
} VxoVolume;

typedef struct VxoCounters {
  uint64_t samples;
  uint64_t rays;
  uint64_t pixels;
  uint64_t skip_steps;
  uint64_t grad_samples;
  uint64_t tf_samples;   /* samples whose density lay inside the sample range (common.glsl:79 not taken) */
} VxoCounters;

/* ---- RNG (shaders/random.glsl) ---- */
uint32_t vxo_tea(uint32_t v0, uint32_t v1, uint32_t n);
uint32_t vxo_wang(uint32_t x);
void vxo_seed_xoshiro(uint32_t seed, uint32_t state[4]);
uint32_t vxo_xoshiro_next(uint32_t state[4]);
float vxo_rng(uint32_t state[4]);
uint32_t vxo_pixel_seed(uint32_t px, uint32_t py, uint32_t res_x, uint32_t frame);

/* ---- binary16 (half 2.7.1, used at brick.rs:19-28) ---- */
uint16_t vxo_f32_to_f16(float x);
float vxo_f16_to_f32(uint16_t h);

/* ---- brick grid (dicom_preprocessor/src/brick.rs, buf3d.rs, dicom.rs) ---- */
void vxo_brick_count(const uint32_t dims[3], uint32_t out[3]);
/* returns number of allocated (non-constant) bricks, or -1 on "Exceeded max brick count".
 * Caller allocates: indirection/range = bc.x*bc.y*bc.z u32 (range as packed u32
 * (min16<<16)|max16, brick.rs:19-23), atlas = (bc*8) bytes (unpruned), mips[k] =
 * (bc>>(k+1))^3 u32. */
int64_t vxo_brick_construct(const uint16_t* voxels, const uint32_t dims[3], uint16_t max_value,
                            uint32_t* indirection, uint32_t* range, uint8_t* atlas,
                            uint32_t* mip0, uint32_t* mip1, uint32_t* mip2,
                            uint32_t atlas_size_out[3]);
float vxo_dicom_lookup(const uint16_t* voxels, const uint32_t dims[3], uint16_t max_value,
                       uint32_t x, uint32_t y, uint32_t z);
float vxo_brick_lookup(const VxoVolume* v, uint32_t x, uint32_t y, uint32_t z);
void vxo_histogram_gradient(const uint32_t* hist, uint32_t n, int32_t* smoothed, uint32_t* gmin,
                            uint32_t* gmax);

/* ---- shader-side lookups (shaders/sampling/common.glsl) ---- */
float vxo_lookup_density_brick(const VxoVolume* v, int32_t x, int32_t y, int32_t z);
float vxo_trilinear_cell(const VxoVolume* v, float density_scale, int32_t ix, int32_t iy, int32_t iz,
                         float fx, float fy, float fz);
float vxo_lookup_density_trilinear(const VxoVolume* v, float density_scale, float px, float py,
                                   float pz);
float vxo_lookup_majorant(const VxoVolume* v, float density_scale, float px, float py, float pz,
                          int32_t mip);
void vxo_lookup_transfer(const float* tf, uint32_t tf_len, const float sample_range[2], float d,
                         float out[4]);

/* ---- environment map + importance map (representation/environment.ts, shaders/envSetup.frag,
 *      shaders/environment.glsl).  texture = RGBA32F texels in GL orientation (row 0 = bottom, i.e.
 *      after UNPACK_FLIP_Y_WEBGL, environment.ts:30-32); importance = mip pyramid of the 512x512 R32F
 *      map, level k at offset vxo_imp_offset(k), (512>>k)^2 floats, level 9 = 1 texel.
 *      [build] texture(): LINEAR on level 0 (MAX_LEVEL 0), REPEAT in s, CLAMP_TO_EDGE in t, weights
 *      are the exact fp32 fractions (GL leaves the weight precision to the implementation);
 *      generateMipmap = 2x2 box, ((a+b)+c+d)*0.25. */
#define VXO_IMP_DIM 512u
#define VXO_IMP_LEVELS 10u
#define VXO_IMP_FLOATS 349525u
typedef struct VxoEnvironment {
  const float* texture; /* width*height*4 */
  uint32_t width, height;
  const float* importance; /* VXO_IMP_FLOATS */
} VxoEnvironment;
uint32_t vxo_imp_offset(uint32_t level);
void vxo_env_flip_rows(const float* top_first, uint32_t w, uint32_t h, float* gl_rows);
void vxo_env_texture(const VxoEnvironment* e, float u, float v, float rgb[3]);
void vxo_env_build_importance(const float* texture, uint32_t w, uint32_t h, float* pyramid);
/* environment.glsl:35-79 / :19-27 / :82-86 with u_use_env = 1 (env_strength applied) */
void vxo_env_sample(const VxoEnvironment* e, float env_strength, float u0, float u1, float w_i[3],
                    float le_pdf[4]);
void vxo_env_lookup(const VxoEnvironment* e, float env_strength, const float dir[3], float rgb[3]);
float vxo_env_pdf(const VxoEnvironment* e, float env_strength, const float dir[3]);

/* vxo_render with an environment (params.use_env = 1 needs one; NULL = directional light only) */
int vxo_render_env(const VxParams* p, uint32_t frame_index, float sample_weight, const VxoVolume* v,
                   const float* tf, uint32_t tf_len, const VxoEnvironment* env, const float* prev,
                   float* out, int32_t x0, int32_t x1, int32_t y0, int32_t y1, VxoCounters* counters);

/* the same with per-pixel sample counts (samples + skipped steps of the pixel's paths) into
 * ray_samples[py * res.x + px]; sample_weight 0 */
int vxo_render_ray_samples(const VxParams* p, uint32_t frame_index, const VxoVolume* v, const float* tf,
                           uint32_t tf_len, const VxoEnvironment* env, float* out, uint32_t* ray_samples,
                           int32_t x0, int32_t x1, int32_t y0, int32_t y1, VxoCounters* counters);

/* [build] DVR march contract: 0 = shipped (n = ceil((far - t0) / dt) samples at q = fma(k, dq, q0)), 1 = the contract of
 * rounds 1-2 (t_k = fma(k, dt, t0) while t_k < far, position fma(t_k, idir, ipos) - 1/2).  Process-wide switch, set
 * between renders; tests/test_tolerance_envelope.py pins one against the other. */
void vxo_set_dvr_march(int walk_t);
int vxo_get_dvr_march(void);

/* ---- full fragment program (shaders/fragment.frag main) over a pixel rectangle ----
 * out / prev: res.x*res.y*4 floats, row 0 = bottom.  Only pixels in [x0,x1) x [y0,y1) are
 * written.  prev may be NULL when sample_weight == 0. */
int vxo_render(const VxParams* p, uint32_t frame_index, float sample_weight, const VxoVolume* v,
               const float* tf, uint32_t tf_len, const float* prev, float* out, int32_t x0,
               int32_t x1, int32_t y0, int32_t y1, VxoCounters* counters);

/* [build] exact empty-space skipping of VX_MODE_DVR.  Macro-cell lattice at
 * level g: cell c = floor(p-0.5) belongs to macro cell (c+1) >> (3+g); dims[a] =
 * (index_extent[a] >> (3+g)) + 1.  A macro cell is EMPTY iff every brick that can hold a tap of
 * one of its cells (and the value 0 outside the grid) is transparent under the current TF /
 * sample range.  vxo_skip_level picks the smallest g in 1..3 whose lattice has <= 65536 cells.
 * bits: one bit per macro cell, x fastest, word = index >> 5. */
int vxo_skip_level(const VxoVolume* v);
void vxo_skip_dims(const VxoVolume* v, int level, uint32_t dims[3]);
void vxo_build_skip_mask(const VxParams* p, const VxoVolume* v, const float* tf, uint32_t tf_len,
                         int level, uint32_t* bits);

/* primary ray of a pixel (fragment.frag:57-65,140-146), for ray-gen tests */
void vxo_primary_ray(const VxParams* p, uint32_t frame_index, int32_t px, int32_t py,
                     float origin[3], float dir[3]);

/* blit.frag:17-35 */
void vxo_blit(const float* accum, uint32_t n_pixels, float exposure, float gamma, uint8_t* rgba8,
              float* rgba_f32_or_null);

#ifdef __cplusplus
}
#endif
#endif
