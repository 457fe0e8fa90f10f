/*
 * volxel_hip.h -- C ABI of libvolxel_hip.so, the MI355X (gfx950) drop-in for the
 * render boundary of Volxel's volxel-3d-viewer.
 *
 * The reference drives its hot path (the per-pixel volume ray loop of
 * volxel-3d-viewer/src/shaders/fragment.frag) through a set of WebGL2 calls issued by
 * volxel-3d-viewer/src/viewer.ts.  Every entry point below replaces one group of those
 * calls; the citation says which.  All paths are relative to the reference repository.
 *
 * Conventions
 *   - plain C, no exceptions, no callbacks; every call returns an int status
 *     (VX_OK == 0) and vx_last_error() returns the message of the last failure, which
 *     the host turns into the `throw new Error(...)` of viewer.ts:797-816;
 *   - one context is used from one thread at a time (the reference is single threaded,
 *     viewer.ts:1160);
 *   - host pointers are read synchronously and may be dropped by the caller on return
 *     (the texImage3D contract of viewer.ts:1106-1142);
 *   - matrices are 16 floats, column major (gl-matrix / math.gl convention);
 *   - framebuffers are RGBA32F, row 0 = bottom row (GL origin, vertex.vert:9).
 */
#ifndef VOLXEL_HIP_H
#define VOLXEL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VX_OK 0
#define VX_ERR_INVALID 1      /* bad argument / call order                                */
#define VX_ERR_DEVICE 2       /* HIP runtime failure (message has hipGetErrorString)      */
#define VX_ERR_NO_VOLUME 3    /* render before vx_upload_volume (viewer.ts:1081)          */
#define VX_ERR_NO_DEVICE 4    /* no gfx950 device visible: the product path never falls
                                 back to a CPU implementation                            */

/* render-mode define of the fragment shader (viewer.ts:66-70,771-787, fragment.frag:3). */
enum VxRenderMode {
  VX_MODE_DEFAULT = 0,  /* hierarchical DDA + null collisions   sampling/dda.glsl        */
  VX_MODE_NO_DDA = 1,   /* delta / ratio tracking               sampling/normal.glsl     */
  VX_MODE_RAYMARCH = 2, /* 64-step stochastic march             sampling/raymarch.glsl   */
  VX_MODE_DVR = 3,      /* [build] deterministic front-to-back compositing = E[RAYMARCH],
                           SURVEY.md section 8 row A12 -- the BASELINE.json headline loop */
  VX_MODE_DVR_PHONG = 4 /* [build] DVR + central-difference gradient + Phong (config 4)  */
};

/* device layout of the brick grid the trilinear look-ups sample (vx_set_layout) */
enum VxLayout {
  VX_LAYOUT_REFERENCE = 0, /* the three reference textures, linear buffers (common.glsl:35-43) */
  VX_LAYOUT_CELLQUAD = 1,  /* MI355X native: apron bricks of pre-decoded fp32 xy-quads (18 bytes per
                              voxel), two 16-byte gathers per trilinear look-up; the DVR kernel keeps
                              batches of 4 steps in flight per wave                              */
  VX_LAYOUT_BRICKF32 = 2,  /* MI355X native: 8^3 bricks decoded to fp32, 2 KiB contiguous each (4 bytes
                              per voxel); the DVR / Phong kernel stages the window of voxels a wave is
                              marching through into LDS and takes every tap from there           */
  VX_LAYOUT_AUTO = 3,      /* default: each render mode on the layout its kernels are fastest on -- DVR and
                              DVR + Phong on BRICKF32 (built at upload), the path-traced reference modes on
                              CELLQUAD (built the first time such a mode is rendered); volumes beyond the index
                              range of a layout fall back to REFERENCE for the modes concerned       */
  VX_LAYOUT_BRICKU8 = 4    /* MI355X native, opt-in: the 8^3 bricks of BRICKF32 kept as the atlas' 8-bit codes
                              (1 byte per voxel, 512 B per brick) + {min, max - min} per brick; the DVR / Phong
                              kernel decodes them with A4's own fma while it stages a window into LDS -- the same
                              bits as BRICKF32 at a quarter of the memory, a few per cent slower (the march is
                              bound by the vector ALUs, not by HBM).  The path-traced modes sample the
                              reference textures under this layout.                                     */
};

/*
 * The uniform block of fragment.frag / utils.glsl / environment.glsl, one POD.
 * Field-for-field the values viewer.ts:1295-1357 and scene.ts:53-56 bind; the host side
 * (volxel_amd/renderer.py, js/viewer.js) fills it exactly like bindUniforms() does.
 * Every member is 4 bytes wide, no padding.
 */
typedef struct VxParams {
  /* utils.glsl:20-21, set at scene.ts:53-56 */
  float camera_view[16];
  float camera_proj[16];
  /* inverse(camera_view) / inverse(camera_proj): the shader inverts per fragment
     (utils.glsl:24,29,35); hoisted to the host (quirk Q11). */
  float camera_view_inv[16];
  float camera_proj_inv[16];
  /* [build] 0: the reference's perspective ray (utils.glsl:23-40; scene.ts:65-72 is perspective
     only).  1: orthographic -- parallel rays (BASELINE config 1): origin = the unprojected near-plane
     point inverse(view)*inverse(proj)*(ndc.xy,-1,1), direction = normalize(inverse(view)*(0,0,-1,0));
     camera_proj is then a gl-matrix ortho() matrix.                                              */
  int32_t camera_ortho;

  /* fragment.frag:22, viewer.ts:1319-1320 (already clipped by volumeClipMin/Max) */
  float volume_aabb_min[3];
  float volume_aabb_max[3];
  /* fragment.frag:24-30, viewer.ts:1321-1327 */
  float volume_min;
  float volume_maj;
  float volume_inv_maj;
  float volume_albedo[3];
  float volume_phase_g;
  float volume_density_scale;
  /* fragment.frag:34-35, viewer.ts:1329-1331 */
  float density_transform[16];
  float density_transform_inv[16];
  /* fragment.frag:43, viewer.ts:1343 */
  float sample_range[2];

  /* environment.glsl:7-16,  viewer.ts:1303,1338-1340, environment.ts:82-84 */
  float light_dir[3];
  float env_strength;
  int32_t show_environment;
  int32_t use_env; /* 0: directional light (environment.glsl:30-33); 1: the uploaded environment
                      map (environment.glsl:35-79), needs vx_upload_environment first           */
  int32_t bounces;

  /* fragment.frag:44-51, viewer.ts:1351-1356 */
  int32_t res[2]; /* u_res; also the render size (quirk Q2 dropped)                       */
  int32_t debug_hits;

  int32_t render_mode; /* enum VxRenderMode */

  /* [build] parameters of VX_MODE_DVR*; shared verbatim by oracle and kernel            */
  float dvr_step_voxels; /* march step in index-space voxels (BASELINE config: 0.5)      */
  float dvr_ert_tau;     /* early ray termination once optical depth tau >= this
                            (= -ln(eps) for a transmittance threshold eps)               */
  int32_t dvr_jitter;    /* 1: sub-pixel + start jitter from the RNG like the reference
                            (fragment.frag:146, raymarch.glsl:30); 0: pixel centre,
                            start offset 0.5 step                                        */
  int32_t dvr_max_steps; /* samples per ray at most, 0 .. 2^24 (the step index is an fp32 value:
                            t_k = fma(k, dt, t0)); vx_set_params refuses more            */
  int32_t dvr_skip_empty; /* 1: exact empty-space skipping -- samples whose macro cell (16..64
                             voxels) can only see TF-transparent bricks are
                             not evaluated (their alpha is exactly 0) and not counted          */
  float dvr_gain[3];     /* albedo * mis * f_p * Le / pdf  (fragment.frag:94-97), host-computed */
  /* Phong terms of VX_MODE_DVR_PHONG */
  float phong_ka, phong_kd, phong_ks, phong_shininess;

  /* image-space sharding: this context renders the tiles
     t with t % shard_count == shard_rank, tile = shard_tile x shard_tile pixels        */
  int32_t shard_rank;
  int32_t shard_count;
} VxParams;

/* exact work counters of the launches since the last vx_reset_counters */
typedef struct VxCounters {
  uint64_t samples;      /* volume sample evaluations (density lookup + TF + accumulate)  */
  uint64_t rays;         /* primary rays that hit the clipped AABB                        */
  uint64_t pixels;       /* pixels written                                                */
  uint64_t skip_steps;   /* DDA / empty-space steps (not samples)                         */
  uint64_t grad_samples; /* samples that also evaluated the 6-tap gradient (DVR_PHONG)    */
  uint64_t lane_slots;   /* 64 x wave iterations of the DVR march loop (samples / lane_slots
                            = SIMD lane utilisation); 0 for kernels that do not count it   */
  uint64_t launches;     /* render-kernel launches (one launch may cover several frames)   */
  uint64_t frames;       /* accumulation frames rendered                                  */
  double kernel_ms;      /* sum of HIP-event durations of those launches                  */
  double last_kernel_ms; /* duration of the most recent launch                            */
  uint64_t gathers;      /* 16-byte-per-lane gather instructions (global_load_dwordx4 wave
                            instructions) issued by the tuned DVR kernels; 0 for the others  */
  uint64_t lds_reads;    /* LDS tap reads (ds_read wave instructions) of the LDS-tile kernels  */
  double merge_ms;       /* sum of HIP-event durations of the running-mean blend kernels that
                            follow multi-frame launches (fragment.frag:158 applied in order)  */
  uint32_t min_launch_frames; /* smallest / largest number of accumulation frames one launch  */
  uint32_t max_launch_frames; /* actually covered (what ran, not what was requested)           */
  uint64_t tf_samples;   /* samples whose density lay inside the sample range: the ones that fetch a
                            transfer-function entry (common.glsl:78-83) and enter the composite  */
  uint64_t active_lane_slots; /* of lane_slots, the slots whose lane did work -- counted by the event-batched path
                            kernels (default / no_dda), whose lanes may wait for an event pass; 0 elsewhere
                            (for the DVR kernels samples / lane_slots is the lane utilisation)          */
} VxCounters;

typedef struct VxContext VxContext;

#define VX_SHARD_TILE 64 /* pixels per side of one sharding tile */

/* ---- lifecycle: replaces canvas.getContext("webgl2") + program/FBO/texture creation
 *      (viewer.ts:221-414).  device_id = HIP ordinal.  Fails with VX_ERR_NO_DEVICE
 *      when no GPU is present.  */
int vx_create(int device_id, VxContext** out_ctx);
void vx_destroy(VxContext* ctx);
/* message of the last failed call on ctx (ctx may be NULL for vx_create failures) */
const char* vx_last_error(const VxContext* ctx);

/* run all work of this context on an existing HIP stream (hipStream_t passed as void*);
 * NULL = the context's own stream.  Lets a torch/RCCL host order copies and collectives. */
int vx_set_stream(VxContext* ctx, void* hip_stream);

/* ---- volume upload: replaces setupFromGrid's four texImage3D groups
 *      (viewer.ts:1106-1142); arguments are field-for-field WasmWorkerMessageDicomReturn
 *      (common.ts:37-55).  `range` and the mips are the LE u16 stream [max,min] per brick
 *      (brick.rs:19-23,357-359).  n_mips must be 3 (brick.rs:13).  */
int vx_upload_volume(VxContext* ctx,
                     const uint32_t* indirection, const uint32_t indirection_size[3],
                     const uint16_t* range, const uint32_t range_size[3],
                     const uint8_t* atlas, const uint32_t atlas_size[3],
                     int n_mips, const uint16_t* const* mip_data, const uint32_t (*mip_size)[3],
                     const uint32_t index_extent[3]);

/* facts of the last vx_upload_volume on this context: wall seconds (copies + device-side layout build,
 * both inside the call), host bytes moved over PCIe, and whether the atlas could be pinned in place
 * (hipHostRegister) so that the copy engine read it directly -- the "pin/upload volumes to HBM" step.
 * The atlas goes in chunks of whole 8-slice layers and the layout of the brick layers a chunk completes is
 * built behind it on a second stream.  Any out pointer may be NULL. */
int vx_upload_stats(VxContext* ctx, double* seconds, uint64_t* host_bytes, int* pinned);

/* the same straight from a native brick grid (volxel_brick.h): a C / Rust host that built the grid
 * with vxb_read_dicoms_to_grid or vxb_build_from_u16 uploads it without the copy-out of
 * worker.ts:19-58.  The grid stays owned by the caller (vxb_free afterwards). */
struct VxBrickGrid;
int vx_upload_brick_grid(VxContext* ctx, const struct VxBrickGrid* grid);

/* select the device layout the trilinear modes sample from (default VX_LAYOUT_AUTO);
 * takes effect at the next vx_upload_volume or immediately if a volume is resident. */
int vx_set_layout(VxContext* ctx, int layout);

/* ---- transfer function: replaces changeTransferFunc's texImage2D (viewer.ts:1147-1153);
 *      rgba = length x 4 floats, sampled NEAREST + CLAMP_TO_EDGE (viewer.ts:386-389). */
int vx_upload_transfer(VxContext* ctx, const float* rgba, uint32_t length);

/* ---- environment map: replaces `new Environment(gl, env)` (representation/environment.ts:15-61,
 *      viewer.ts:1076-1077): rgba = width*height*4 floats with row 0 = TOP, exactly the `floats` of
 *      WasmWorkerMessageEnvReturn (the library applies the UNPACK_FLIP_Y_WEBGL of environment.ts:30-32);
 *      builds the 512x512 importance map (shaders/envSetup.frag, 8x8 taps per texel) and its mip
 *      chain on the device.  Needed before VxParams.use_env = 1.  Passing rgba = NULL removes it. */
int vx_upload_environment(VxContext* ctx, const float* rgba, uint32_t width, uint32_t height);
/* test hook: the importance pyramid, 349525 floats (levels 0..9 of the 512^2 map back to back) */
int vx_debug_read_importance(VxContext* ctx, float* out);

/* ---- uniforms: replaces bindUniforms + Camera.bindAsUniforms (viewer.ts:1295-1357,
 *      scene.ts:53-56). */
int vx_set_params(VxContext* ctx, const VxParams* params);

/* ---- framebuffers: replaces resizeFramebuffersToCanvas / the two RGBA32F ping-pong
 *      FBOs (viewer.ts:294-324).  Clears the accumulation. */
int vx_resize(VxContext* ctx, uint32_t width, uint32_t height);

/* ---- one accumulation sample: replaces gl.drawArrays(TRIANGLE_STRIP,0,4) of the
 *      path-tracing program (viewer.ts:1208-1211) including the running-mean blend
 *      out = w*prev + (1-w)*result (fragment.frag:158); frame_index = u_frame_index,
 *      sample_weight = u_sample_weight (viewer.ts:1351,1356).  Asynchronous on the
 *      context's stream. */
int vx_render_frame(VxContext* ctx, uint32_t frame_index, float sample_weight);

/* The same for `count` consecutive accumulation frames (weights[i] = u_sample_weight of frame
 * first_frame + i), with up to `in_flight` (<= 64) of them rendered by one kernel launch, each into its
 * own result buffer.  Accumulation frames are independent given their index; only the running mean is
 * ordered and it is applied afterwards, in order -- the accumulator is bit-identical to `count` calls of
 * vx_render_frame.  Needs in_flight x (framebuffer + counters) of extra device memory.
 * [build] no reference counterpart: WebGL2 draws are serialised. */
int vx_render_frames(VxContext* ctx, uint32_t first_frame, uint32_t count, const float* weights, int in_flight);

/* ---- multi-GPU load balance (no counterpart in the reference).  By default tile t of the 64x64 tile grid
 *      (row-major) belongs to shard t % shard_count.  vx_set_tile_order installs another dealing order:
 *      position pos holds tile perm[pos] and belongs to shard pos % shard_count (local index pos /
 *      shard_count), so every shard still owns the same number of tiles.  perm must be a permutation of
 *      0..n_tiles-1 and identical on all ranks; NULL restores the default.  The accumulator is cleared:
 *      restart the accumulation (frame 0) afterwards.  vx_probe_tile_costs fills costs[t] with a cost
 *      estimate of every tile of the image (DVR samples of 64 probe rays per tile, current volume /
 *      transfer function / params) -- the same numbers on every rank, so sorting them gives every rank the
 *      same order without communication. */
int vx_probe_tile_costs(VxContext* ctx, uint32_t* costs, uint32_t n_tiles);
int vx_set_tile_order(VxContext* ctx, const uint32_t* perm, uint32_t n_tiles);

/* current framebuffer size (what vx_read_accum / vx_read_display will write) */
int vx_render_size(VxContext* ctx, uint32_t* width, uint32_t* height);

/* ---- synchronise: replaces gl.finish() (viewer.ts:1214,1289) */
int vx_finish(VxContext* ctx);

/* ---- readback of the accumulation buffer (what blit.frag samples as u_result),
 *      width*height*4 floats, row 0 = bottom.  Synchronises. */
int vx_read_accum(VxContext* ctx, float* rgba_out);
/* ---- display pass: replaces the blit program (blit.frag:17-35, viewer.ts:1259-1265):
 *      Hable tonemap + gamma, RGBA8, width*height*4 bytes.  Synchronises. */
int vx_read_display(VxContext* ctx, uint8_t* rgba8_out, float exposure, float gamma);
/* ---- the same pass drawn to a canvas of another size, as the viewer does while the low-resolution
 *      preview (resolutionFactor 0.33, viewer.ts:1167-1188) is up: the blit samples u_result with
 *      NEAREST filtering (viewer.ts:310-311), i.e. canvas pixel (x,y) shows render pixel
 *      (floor((x+0.5)*w/out_w), floor((y+0.5)*h/out_h)).  out_w*out_h*4 bytes.  Synchronises. */
int vx_read_display_scaled(VxContext* ctx, uint8_t* rgba8_out, uint32_t out_w, uint32_t out_h,
                           float exposure, float gamma);

/* device-side views for a zero-copy host (torch / RCCL gather): the tile-major slab this
 * shard owns (floats = vx_slab_floats) and a de-tiling pass from a gathered set of slabs. */
int vx_slab_info(VxContext* ctx, uint64_t* slab_floats, uint32_t* tiles_per_shard);
int vx_slab_device_ptr(VxContext* ctx, void** dev_ptr);
/* gathered = shard_count slabs back to back (device pointer); writes row-major W*H*4 floats
 * to image_out (device pointer). */
int vx_detile(VxContext* ctx, const void* gathered_dev, void* image_out_dev);

/* ---- counters: the benchmark harness of viewer.ts:1213-1252 measures wall time around
 *      gl.finish(); here the kernel is bracketed by HIP events on its own stream. */
int vx_get_counters(VxContext* ctx, VxCounters* out);
int vx_reset_counters(VxContext* ctx);

/* library / device facts for logs (viewer.ts:225-242 device record) */
int vx_device_info(VxContext* ctx, char* name_out, uint32_t name_cap, uint32_t* cu_count,
                   uint64_t* hbm_bytes);
const char* vx_version(void);

/* test hook: the integer RNG of shaders/random.glsl evaluated ON THE DEVICE (rows A1/A2), one thread per
 * output word.  op 0: out[i] = tea(a[i], b[i], 32) (random.glsl:41-51); op 1: out[i] = wangHash(a[i]) (:59-66);
 * op 2: out[0..n) = the first n xoshiro128pp_next words of seedXoshiro(a[0]) (:69-94, quirk Q1);
 * op 3: the same stream as rng() floats, bit patterns (:103-106);
 * op 4: out[i] = how many of the 256 draws r = (256 * (a[0] + i) + j) / 2^24, j = 0..255, give a free-flight logarithm
 * (the library's -log(1 - r), normal.glsl:13,28 / dda.glsl:28,58 / raymarch.glsl:26) that differs in any bit from
 * -logf(1 - r): a[0] = 0, n = 65536 covers every value rng() can return.  a / b / out are host pointers.  */
int vx_debug_rng(VxContext* ctx, int op, const uint32_t* a, const uint32_t* b, uint32_t n, uint32_t* out);

/* measurement hook (no reference counterpart): what the vector L1 of this device sustains for the
 * gather shape of the cellquad DVR march -- wave instructions of 16 bytes per lane whose 64 lanes form `lines`
 * groups of consecutive lanes, each group inside one L1-resident 128-byte line, the groups using `distinct`
 * (<= lines) different lines in turn; nothing else in the loop, 20 waves per CU, 8 gathers in flight per wave.
 * Returns the cost in clocks per gather instruction per CU at the device's nominal clock (clock_khz_out).
 * bench.py calls it for the roofline.l1 block with lines = the march's line look-ups per gather over its
 * 4-lane groups and distinct = its distinct lines per gather over the whole wave. */
int vx_probe_gather_rate(VxContext* ctx, uint32_t lines, uint32_t distinct, double* clk_per_gather_out,
                         uint32_t* clock_khz_out);
/* measurement hook: what the vector ALUs of this device sustain: clocks (nominal clock) per wave64 VALU instruction per
 * SIMD, from independent v_fma_f32 chains at 8 waves per SIMD.  bench.py prices the instruction count of the LDS-window
 * DVR kernel with it (roofline.issue). */
int vx_probe_valu_rate(VxContext* ctx, double* clk_per_instruction_out, uint32_t* clock_khz_out);
/* measurement hook: re-march frame `frame_index` with the current volume / params and count, per gather
 * instruction of the tuned cellquad DVR kernel, the distinct 128-byte lines its active lanes address
 * (whole wave) and the line look-ups of its 16 groups of 4 consecutive lanes (what the L1 tag pipe sees).
 * Nothing is written to the accumulator.  out3 = {gather instructions, distinct lines, quad look-ups}. */
int vx_probe_gather_spread(VxContext* ctx, uint32_t frame_index, uint64_t out3[3]);

/* test hook (no reference counterpart): the device's R8-unorm decode table, 256 floats */
int vx_debug_unorm_table(VxContext* ctx, float* out256);
/* test hook: the host-built empty-space bitmask (pure CPU; bits_out may be NULL to query level/dims) */
int vx_debug_build_skip_mask(const uint32_t* range_packed, const uint32_t brick_count[3], const float* tf_rgba,
                             uint32_t tf_len, const VxParams* params, uint32_t* bits_out, uint32_t* level_out,
                             uint32_t dims_out[3]);

#ifdef __cplusplus
}
#endif
#endif /* VOLXEL_HIP_H */
