/*
 * volxel_napi.c -- thin Node N-API shim over the C ABI of libvolxel_hip.so
 * (include/volxel_hip.h, include/volxel_brick.h).  One JS function per C entry point;
 * typed arrays are passed without copying; a non-zero status becomes a thrown JS Error
 * carrying vx_last_error(), so the host's handleError contract (viewer.ts:797-816) holds.
 * Plain C against <node_api.h> (N-API v3+, Node >= 10).
 */
#include <node_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/volxel_brick.h"
#include "../../include/volxel_hip.h"

#define NAPI_OK(call)                                                    \
  do {                                                                   \
    if ((call) != napi_ok) {                                             \
      napi_throw_error(env, NULL, "volxel_napi: N-API call failed: " #call); \
      return NULL;                                                       \
    }                                                                    \
  } while (0)

static napi_value throw_msg(napi_env env, const char* msg) {
  napi_throw_error(env, NULL, msg && *msg ? msg : "volxel_hip: unknown error");
  return NULL;
}

static int get_args(napi_env env, napi_callback_info info, size_t n, napi_value* argv) {
  size_t argc = n;
  if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) != napi_ok || argc < n) {
    napi_throw_type_error(env, NULL, "volxel_napi: wrong number of arguments");
    return 0;
  }
  return 1;
}

/* The JS handle wraps a small box, not the context itself: destroy() empties the box, so a call made after
 * dispose() (or a second destroy) finds NULL and throws instead of touching freed memory. */
typedef struct Handle {
  VxContext* ctx;
} Handle;

static Handle* get_handle(napi_env env, napi_value v) {
  void* p = NULL;
  if (napi_get_value_external(env, v, &p) != napi_ok || !p) {
    napi_throw_type_error(env, NULL, "volxel_napi: expected a context handle");
    return NULL;
  }
  return (Handle*)p;
}

static VxContext* get_ctx(napi_env env, napi_value v) {
  Handle* h = get_handle(env, v);
  if (!h) return NULL;
  if (!h->ctx) {
    napi_throw_error(env, NULL, "volxel_napi: the context has been destroyed");
    return NULL;
  }
  return h->ctx;
}

/* number of elements of [x, y, z] multiplied out, saturating */
static uint64_t prod3(const uint32_t v[3]) { return (uint64_t)v[0] * v[1] * v[2]; }

/* typed array -> pointer + element count (+ checks the element type) */
static int typed(napi_env env, napi_value v, napi_typedarray_type want, void** data, size_t* len) {
  napi_typedarray_type t;
  napi_value ab;
  size_t off;
  if (napi_get_typedarray_info(env, v, &t, len, data, &ab, &off) != napi_ok || t != want) {
    napi_throw_type_error(env, NULL, "volxel_napi: typed array of the wrong element type");
    return 0;
  }
  return 1;
}

static int u32x3(napi_env env, napi_value arr, uint32_t out[3]) {
  for (uint32_t i = 0; i < 3; ++i) {
    napi_value e;
    if (napi_get_element(env, arr, i, &e) != napi_ok || napi_get_value_uint32(env, e, &out[i]) != napi_ok) {
      napi_throw_type_error(env, NULL, "volxel_napi: expected [x, y, z]");
      return 0;
    }
  }
  return 1;
}

static napi_value prop(napi_env env, napi_value obj, const char* name) {
  napi_value v = NULL;
  napi_get_named_property(env, obj, name, &v);
  return v;
}

static void ctx_finalize(napi_env env, void* data, void* hint) {
  (void)env; (void)hint;
  Handle* h = (Handle*)data;
  if (!h) return;
  if (h->ctx) vx_destroy(h->ctx); /* a handle dropped without destroy() still releases the device memory */
  free(h);
}

/* create(deviceId) -> handle */
static napi_value n_create(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (!get_args(env, info, 1, a)) return NULL;
  int32_t dev = 0;
  NAPI_OK(napi_get_value_int32(env, a[0], &dev));
  VxContext* c = NULL;
  if (vx_create(dev, &c) != VX_OK) return throw_msg(env, vx_last_error(NULL));
  Handle* box = (Handle*)malloc(sizeof(Handle));
  if (!box) {
    vx_destroy(c);
    return throw_msg(env, "volxel_napi: out of memory");
  }
  box->ctx = c;
  napi_value h;
  if (napi_create_external(env, box, ctx_finalize, NULL, &h) != napi_ok) {
    vx_destroy(c);
    free(box);
    return throw_msg(env, "volxel_napi: napi_create_external failed");
  }
  return h;
}

/* destroy(handle): idempotent */
static napi_value n_destroy(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (!get_args(env, info, 1, a)) return NULL;
  Handle* h = get_handle(env, a[0]);
  if (!h) return NULL;
  if (h->ctx) vx_destroy(h->ctx);
  h->ctx = NULL;
  return NULL;
}

/* uploadVolume(ctx, msg): msg = WasmWorkerMessageDicomReturn (common.ts:37-55) */
static napi_value n_upload_volume(napi_env env, napi_callback_info info) {
  napi_value a[2];
  if (!get_args(env, info, 2, a)) return NULL;
  VxContext* c = get_ctx(env, a[0]);
  if (!c) return NULL;
  napi_value m = a[1];
  void *ind, *rng, *atl;
  size_t n, n_ind, n_rng, n_atl;
  uint32_t is[3], rs[3], as[3], ext[3];
  if (!typed(env, prop(env, m, "indirection"), napi_uint32_array, &ind, &n_ind)) return NULL;
  if (!typed(env, prop(env, m, "range"), napi_uint16_array, &rng, &n_rng)) return NULL;
  if (!typed(env, prop(env, m, "atlas"), napi_uint8_array, &atl, &n_atl)) return NULL;
  if (!u32x3(env, prop(env, m, "indirectionSize"), is) || !u32x3(env, prop(env, m, "rangeSize"), rs) ||
      !u32x3(env, prop(env, m, "atlasSize"), as) || !u32x3(env, prop(env, m, "indexExtent"), ext))
    return NULL;
  /* the C ABI takes raw pointers: the typed arrays must really hold what the size fields promise */
  if ((uint64_t)n_ind < prod3(is)) return throw_msg(env, "uploadVolume: indirection shorter than indirectionSize");
  if ((uint64_t)n_rng < 2u * prod3(rs)) return throw_msg(env, "uploadVolume: range shorter than 2 * rangeSize");
  if ((uint64_t)n_atl < prod3(as)) return throw_msg(env, "uploadVolume: atlas shorter than atlasSize");
  napi_value mips = prop(env, m, "rangeMipmaps");
  uint32_t nm = 0;
  NAPI_OK(napi_get_array_length(env, mips, &nm));
  const uint16_t* mp[3] = {0, 0, 0};
  uint32_t ms[3][3];
  if (nm > 3) nm = 3;
  for (uint32_t k = 0; k < nm; ++k) {
    napi_value e;
    void* d;
    NAPI_OK(napi_get_element(env, mips, k, &e));
    if (!typed(env, prop(env, e, "mipmap"), napi_uint16_array, &d, &n)) return NULL;
    mp[k] = (const uint16_t*)d;
    if (!u32x3(env, prop(env, e, "stride"), ms[k])) return NULL;
    if ((uint64_t)n < 2u * prod3(ms[k])) return throw_msg(env, "uploadVolume: range mipmap shorter than 2 * stride");
  }
  if (vx_upload_volume(c, (const uint32_t*)ind, is, (const uint16_t*)rng, rs, (const uint8_t*)atl, as, (int)nm, mp,
                       (const uint32_t(*)[3])ms, ext) != VX_OK)
    return throw_msg(env, vx_last_error(c));
  return NULL;
}

static napi_value n_upload_transfer(napi_env env, napi_callback_info info) {
  napi_value a[3];
  if (!get_args(env, info, 3, a)) return NULL;
  VxContext* c = get_ctx(env, a[0]);
  if (!c) return NULL;
  void* d;
  size_t n;
  uint32_t len;
  if (!typed(env, a[1], napi_float32_array, &d, &n)) return NULL;
  NAPI_OK(napi_get_value_uint32(env, a[2], &len));
  if (n < (size_t)len * 4) return throw_msg(env, "uploadTransfer: data shorter than length*4");
  if (vx_upload_transfer(c, (const float*)d, len) != VX_OK) return throw_msg(env, vx_last_error(c));
  return NULL;
}

/* uploadEnvironment(ctx, Float32Array rgba | null, width, height): `new Environment(gl, env)` */
static napi_value n_upload_environment(napi_env env, napi_callback_info info) {
  napi_value a[4];
  if (!get_args(env, info, 4, a)) return NULL;
  VxContext* c = get_ctx(env, a[0]);
  if (!c) return NULL;
  napi_valuetype t;
  NAPI_OK(napi_typeof(env, a[1], &t));
  if (t == napi_null || t == napi_undefined) {
    if (vx_upload_environment(c, NULL, 0, 0) != VX_OK) return throw_msg(env, vx_last_error(c));
    return NULL;
  }
  void* d;
  size_t n;
  uint32_t w, h;
  if (!typed(env, a[1], napi_float32_array, &d, &n)) return NULL;
  NAPI_OK(napi_get_value_uint32(env, a[2], &w));
  NAPI_OK(napi_get_value_uint32(env, a[3], &h));
  if (n < (size_t)w * h * 4) return throw_msg(env, "uploadEnvironment: data shorter than width*height*4");
  if (vx_upload_environment(c, (const float*)d, w, h) != VX_OK) return throw_msg(env, vx_last_error(c));
  return NULL;
}

/* setParams(ctx, ArrayBuffer holding one VxParams) */
static napi_value n_set_params(napi_env env, napi_callback_info info) {
  napi_value a[2];
  if (!get_args(env, info, 2, a)) return NULL;
  VxContext* c = get_ctx(env, a[0]);
  if (!c) return NULL;
  void* d;
  size_t n;
  NAPI_OK(napi_get_arraybuffer_info(env, a[1], &d, &n));
  if (n != sizeof(VxParams)) return throw_msg(env, "setParams: buffer is not sizeof(VxParams)");
  if (vx_set_params(c, (const VxParams*)d) != VX_OK) return throw_msg(env, vx_last_error(c));
  return NULL;
}

static napi_value n_sizeof_params(napi_env env, napi_callback_info info) {
  (void)info;
  napi_value v;
  NAPI_OK(napi_create_uint32(env, (uint32_t)sizeof(VxParams), &v));
  return v;
}

static napi_value n_resize(napi_env env, napi_callback_info info) {
  napi_value a[3];
  if (!get_args(env, info, 3, a)) return NULL;
  VxContext* c = get_ctx(env, a[0]);
  if (!c) return NULL;
  uint32_t w, h;
  NAPI_OK(napi_get_value_uint32(env, a[1], &w));
  NAPI_OK(napi_get_value_uint32(env, a[2], &h));
  if (vx_resize(c, w, h) != VX_OK) return throw_msg(env, vx_last_error(c));
  return NULL;
}

static napi_value n_set_layout(napi_env env, napi_callback_info info) {
  napi_value a[2];
  if (!get_args(env, info, 2, a)) return NULL;
  VxContext* c = get_ctx(env, a[0]);
  if (!c) return NULL;
  int32_t l;
  NAPI_OK(napi_get_value_int32(env, a[1], &l));
  if (vx_set_layout(c, l) != VX_OK) return throw_msg(env, vx_last_error(c));
  return NULL;
}

static napi_value n_render_frame(napi_env env, napi_callback_info info) {
  napi_value a[3];
  if (!get_args(env, info, 3, a)) return NULL;
  VxContext* c = get_ctx(env, a[0]);
  if (!c) return NULL;
  uint32_t f;
  double w;
  NAPI_OK(napi_get_value_uint32(env, a[1], &f));
  NAPI_OK(napi_get_value_double(env, a[2], &w));
  if (vx_render_frame(c, f, (float)w) != VX_OK) return throw_msg(env, vx_last_error(c));
  return NULL;
}

/* renderFrames(ctx, firstFrame, Float32Array weights, inFlight): weights.length accumulation frames,
 * up to inFlight of them per launch (vx_render_frames) */
static napi_value n_render_frames(napi_env env, napi_callback_info info) {
  napi_value a[4];
  if (!get_args(env, info, 4, a)) return NULL;
  VxContext* c = get_ctx(env, a[0]);
  if (!c) return NULL;
  uint32_t f;
  int32_t in_flight;
  void* w;
  size_t n;
  NAPI_OK(napi_get_value_uint32(env, a[1], &f));
  if (!typed(env, a[2], napi_float32_array, &w, &n)) return NULL;
  NAPI_OK(napi_get_value_int32(env, a[3], &in_flight));
  if (n > 0xffffffffu) return throw_msg(env, "renderFrames: too many frames");
  if (vx_render_frames(c, f, (uint32_t)n, (const float*)w, in_flight) != VX_OK) return throw_msg(env, vx_last_error(c));
  return NULL;
}

/* probeTileCosts(ctx, Uint32Array out): cost estimate of every 64x64 tile (vx_probe_tile_costs) */
static napi_value n_probe_tile_costs(napi_env env, napi_callback_info info) {
  napi_value a[2];
  if (!get_args(env, info, 2, a)) return NULL;
  VxContext* c = get_ctx(env, a[0]);
  if (!c) return NULL;
  void* d;
  size_t n;
  if (!typed(env, a[1], napi_uint32_array, &d, &n)) return NULL;
  if (vx_probe_tile_costs(c, (uint32_t*)d, (uint32_t)n) != VX_OK) return throw_msg(env, vx_last_error(c));
  return NULL;
}

/* setTileOrder(ctx, Uint32Array perm | null): dealing order of the tiles over the shards (vx_set_tile_order) */
static napi_value n_set_tile_order(napi_env env, napi_callback_info info) {
  napi_value a[2];
  if (!get_args(env, info, 2, a)) return NULL;
  VxContext* c = get_ctx(env, a[0]);
  if (!c) return NULL;
  napi_valuetype t;
  NAPI_OK(napi_typeof(env, a[1], &t));
  void* d = NULL;
  size_t n = 0;
  if (t != napi_null && t != napi_undefined && !typed(env, a[1], napi_uint32_array, &d, &n)) return NULL;
  if (vx_set_tile_order(c, (const uint32_t*)d, (uint32_t)n) != VX_OK) return throw_msg(env, vx_last_error(c));
  return NULL;
}

/* deviceInfo(ctx) -> { name, computeUnits, hbmBytes } */
static napi_value n_device_info(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (!get_args(env, info, 1, a)) return NULL;
  VxContext* c = get_ctx(env, a[0]);
  if (!c) return NULL;
  char name[256];
  uint32_t cus = 0;
  uint64_t hbm = 0;
  if (vx_device_info(c, name, sizeof name, &cus, &hbm) != VX_OK) return throw_msg(env, vx_last_error(c));
  napi_value o, v;
  NAPI_OK(napi_create_object(env, &o));
  NAPI_OK(napi_create_string_utf8(env, name, NAPI_AUTO_LENGTH, &v));
  NAPI_OK(napi_set_named_property(env, o, "name", v));
  NAPI_OK(napi_create_uint32(env, cus, &v));
  NAPI_OK(napi_set_named_property(env, o, "computeUnits", v));
  NAPI_OK(napi_create_double(env, (double)hbm, &v));
  NAPI_OK(napi_set_named_property(env, o, "hbmBytes", v));
  return o;
}

static napi_value n_finish(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (!get_args(env, info, 1, a)) return NULL;
  VxContext* c = get_ctx(env, a[0]);
  if (!c) return NULL;
  if (vx_finish(c) != VX_OK) return throw_msg(env, vx_last_error(c));
  return NULL;
}

static napi_value n_read_accum(napi_env env, napi_callback_info info) {
  napi_value a[2];
  if (!get_args(env, info, 2, a)) return NULL;
  VxContext* c = get_ctx(env, a[0]);
  if (!c) return NULL;
  void* d;
  size_t n;
  uint32_t w = 0, h = 0;
  if (!typed(env, a[1], napi_float32_array, &d, &n)) return NULL;
  vx_render_size(c, &w, &h);
  if (n < (size_t)w * h * 4) return throw_msg(env, "readAccum: buffer smaller than width*height*4 floats");
  if (vx_read_accum(c, (float*)d) != VX_OK) return throw_msg(env, vx_last_error(c));
  return NULL;
}

static napi_value n_read_display(napi_env env, napi_callback_info info) {
  napi_value a[4];
  if (!get_args(env, info, 4, a)) return NULL;
  VxContext* c = get_ctx(env, a[0]);
  if (!c) return NULL;
  void* d;
  size_t n;
  double ex, ga;
  if (!typed(env, a[1], napi_uint8_array, &d, &n)) return NULL;
  NAPI_OK(napi_get_value_double(env, a[2], &ex));
  NAPI_OK(napi_get_value_double(env, a[3], &ga));
  uint32_t w = 0, h = 0;
  vx_render_size(c, &w, &h);
  if (n < (size_t)w * h * 4) return throw_msg(env, "readDisplay: buffer smaller than width*height*4 bytes");
  if (vx_read_display(c, (uint8_t*)d, (float)ex, (float)ga) != VX_OK) return throw_msg(env, vx_last_error(c));
  return NULL;
}

/* readDisplayScaled(ctx, Uint8Array out, outW, outH, exposure, gamma): the blit to a canvas */
static napi_value n_read_display_scaled(napi_env env, napi_callback_info info) {
  napi_value a[6];
  if (!get_args(env, info, 6, a)) return NULL;
  VxContext* c = get_ctx(env, a[0]);
  if (!c) return NULL;
  void* d;
  size_t n;
  uint32_t ow, oh;
  double ex, ga;
  if (!typed(env, a[1], napi_uint8_array, &d, &n)) return NULL;
  NAPI_OK(napi_get_value_uint32(env, a[2], &ow));
  NAPI_OK(napi_get_value_uint32(env, a[3], &oh));
  NAPI_OK(napi_get_value_double(env, a[4], &ex));
  NAPI_OK(napi_get_value_double(env, a[5], &ga));
  if (n < (size_t)ow * oh * 4) return throw_msg(env, "readDisplayScaled: buffer smaller than outW*outH*4 bytes");
  if (vx_read_display_scaled(c, (uint8_t*)d, ow, oh, (float)ex, (float)ga) != VX_OK)
    return throw_msg(env, vx_last_error(c));
  return NULL;
}

static napi_value n_get_counters(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (!get_args(env, info, 1, a)) return NULL;
  VxContext* c = get_ctx(env, a[0]);
  if (!c) return NULL;
  VxCounters k;
  if (vx_get_counters(c, &k) != VX_OK) return throw_msg(env, vx_last_error(c));
  napi_value o, v;
  NAPI_OK(napi_create_object(env, &o));
#define PUT(name, val)                                      \
  NAPI_OK(napi_create_double(env, (double)(val), &v));      \
  NAPI_OK(napi_set_named_property(env, o, name, v));
  PUT("samples", k.samples) PUT("rays", k.rays) PUT("pixels", k.pixels) PUT("skipSteps", k.skip_steps)
  PUT("gradSamples", k.grad_samples) PUT("tfSamples", k.tf_samples) PUT("activeLaneSlots", k.active_lane_slots) PUT("laneSlots", k.lane_slots) PUT("launches", k.launches) PUT("frames", k.frames)
  PUT("kernelMs", k.kernel_ms) PUT("lastKernelMs", k.last_kernel_ms) PUT("gathers", k.gathers)
  PUT("ldsReads", k.lds_reads) PUT("mergeMs", k.merge_ms) PUT("minLaunchFrames", k.min_launch_frames)
  PUT("maxLaunchFrames", k.max_launch_frames)
#undef PUT
  return o;
}

static napi_value n_reset_counters(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (!get_args(env, info, 1, a)) return NULL;
  VxContext* c = get_ctx(env, a[0]);
  if (c && vx_reset_counters(c) != VX_OK) return throw_msg(env, vx_last_error(c));
  return NULL;
}

static napi_value n_version(napi_env env, napi_callback_info info) {
  (void)info;
  napi_value v;
  NAPI_OK(napi_create_string_utf8(env, vx_version(), NAPI_AUTO_LENGTH, &v));
  return v;
}

/* ---- preprocessor: buildBrickGrid(Uint16Array voxels, [x,y,z], [sx,sy,sz], maxValue, threads)
 *      -> WasmWorkerMessageDicomReturn-shaped object (worker.ts:19-58 copies every buffer out
 *      and frees the grid; so does this) */
static napi_value make_typed(napi_env env, napi_typedarray_type t, const void* src, size_t count, size_t esz) {
  napi_value ab, ta;
  void* dst = NULL;
  if (napi_create_arraybuffer(env, count * esz, &dst, &ab) != napi_ok) return NULL;
  if (count) memcpy(dst, src, count * esz);
  if (napi_create_typedarray(env, t, count, ab, 0, &ta) != napi_ok) return NULL;
  return ta;
}
static napi_value make_u32x3(napi_env env, const uint32_t s[3]) {
  napi_value arr, e;
  napi_create_array_with_length(env, 3, &arr);
  for (uint32_t i = 0; i < 3; ++i) {
    napi_create_uint32(env, s[i], &e);
    napi_set_element(env, arr, i, e);
  }
  return arr;
}

static napi_value grid_to_object(napi_env env, VxBrickGrid* g);

/* readDicomsToGrid(Array<Uint8Array> files, threads): the wasm export of lib.rs:193-202 as called
 * at worker.ts:101-104 */
static napi_value n_read_dicoms_to_grid(napi_env env, napi_callback_info info) {
  napi_value a[2];
  if (!get_args(env, info, 2, a)) return NULL;
  uint32_t n = 0;
  int32_t threads;
  bool is_arr = false;
  NAPI_OK(napi_is_array(env, a[0], &is_arr));
  if (!is_arr) return throw_msg(env, "readDicomsToGrid: expected an array of Uint8Array");
  NAPI_OK(napi_get_array_length(env, a[0], &n));
  NAPI_OK(napi_get_value_int32(env, a[1], &threads));
  const uint8_t** ptrs = (const uint8_t**)calloc(n ? n : 1, sizeof(*ptrs));
  uint64_t* sizes = (uint64_t*)calloc(n ? n : 1, sizeof(*sizes));
  for (uint32_t i = 0; i < n; ++i) {
    napi_value e;
    void* p;
    size_t len;
    if (napi_get_element(env, a[0], i, &e) != napi_ok || !typed(env, e, napi_uint8_array, &p, &len)) {
      free(ptrs);
      free(sizes);
      return NULL;
    }
    ptrs[i] = (const uint8_t*)p;
    sizes[i] = len;
  }
  VxBrickGrid* g = NULL;
  int rc = vxb_read_dicoms_to_grid(ptrs, sizes, n, threads, &g);
  free(ptrs);
  free(sizes);
  if (rc != VXB_OK) return throw_msg(env, vxb_last_error());
  return grid_to_object(env, g);
}

static napi_value n_build_brick_grid(napi_env env, napi_callback_info info) {
  napi_value a[5];
  if (!get_args(env, info, 5, a)) return NULL;
  void* vox;
  size_t n;
  uint32_t dims[3], maxv;
  int32_t threads;
  float sp[3];
  if (!typed(env, a[0], napi_uint16_array, &vox, &n) || !u32x3(env, a[1], dims)) return NULL;
  for (uint32_t i = 0; i < 3; ++i) {
    napi_value e;
    double d;
    NAPI_OK(napi_get_element(env, a[2], i, &e));
    NAPI_OK(napi_get_value_double(env, e, &d));
    sp[i] = (float)d;
  }
  NAPI_OK(napi_get_value_uint32(env, a[3], &maxv));
  NAPI_OK(napi_get_value_int32(env, a[4], &threads));
  if (n != (size_t)dims[0] * dims[1] * dims[2]) return throw_msg(env, "buildBrickGrid: voxel count != x*y*z");
  VxBrickGrid* g = NULL;
  if (vxb_build_from_u16((const uint16_t*)vox, dims, sp, (uint16_t)maxv, threads, &g) != VXB_OK)
    return throw_msg(env, vxb_last_error());
  return grid_to_object(env, g);
}

static napi_value grid_to_object(napi_env env, VxBrickGrid* g) {
  uint32_t is[3], rs[3], as[3], ext[3];
  vxb_indirection_size(g, is);
  vxb_range_size(g, rs);
  vxb_atlas_size(g, as);
  vxb_index_extent(g, ext);
  size_t nb = (size_t)is[0] * is[1] * is[2];
  napi_value o, v;
  napi_create_object(env, &o);
  napi_create_string_utf8(env, "return_dicom", NAPI_AUTO_LENGTH, &v);
  napi_set_named_property(env, o, "type", v);
  napi_set_named_property(env, o, "indirectionSize", make_u32x3(env, is));
  napi_set_named_property(env, o, "rangeSize", make_u32x3(env, rs));
  napi_set_named_property(env, o, "atlasSize", make_u32x3(env, as));
  napi_set_named_property(env, o, "indexExtent", make_u32x3(env, ext));
  napi_set_named_property(env, o, "indirection", make_typed(env, napi_uint32_array, vxb_indirection_data(g), nb, 4));
  napi_set_named_property(env, o, "range", make_typed(env, napi_uint16_array, vxb_range_data(g), nb * 2, 2));
  napi_set_named_property(env, o, "atlas",
                          make_typed(env, napi_uint8_array, vxb_atlas_data(g), (size_t)as[0] * as[1] * as[2], 1));
  float t[16];
  vxb_transform(g, t);
  napi_set_named_property(env, o, "transform", make_typed(env, napi_float32_array, t, 16, 4));
  uint32_t hl = vxb_histogram_len(g);
  napi_set_named_property(env, o, "histogram", make_typed(env, napi_uint32_array, vxb_histogram(g), hl, 4));
  napi_set_named_property(env, o, "histogramGradient",
                          make_typed(env, napi_int32_array, vxb_histogram_gradient(g), hl, 4));
  napi_value pair, e;
  napi_create_array_with_length(env, 2, &pair);
  napi_create_uint32(env, vxb_histogram_gradient_min(g), &e);
  napi_set_element(env, pair, 0, e);
  napi_create_uint32(env, vxb_histogram_gradient_max(g), &e);
  napi_set_element(env, pair, 1, e);
  napi_set_named_property(env, o, "histogramGradientRange", pair);
  napi_create_array_with_length(env, 2, &pair);
  napi_create_double(env, vxb_minorant(g), &e);
  napi_set_element(env, pair, 0, e);
  napi_create_double(env, vxb_majorant(g), &e);
  napi_set_element(env, pair, 1, e);
  napi_set_named_property(env, o, "minMaj", pair);
  napi_value mips;
  uint32_t nm = vxb_range_mipmaps(g);
  napi_create_array_with_length(env, nm, &mips);
  for (uint32_t k = 0; k < nm; ++k) {
    uint32_t st[3];
    vxb_range_mipmap_stride(g, k, st);
    napi_value mo;
    napi_create_object(env, &mo);
    napi_set_named_property(env, mo, "mipmap",
                            make_typed(env, napi_uint16_array, vxb_range_mipmap(g, k), (size_t)st[0] * st[1] * st[2] * 2, 2));
    napi_set_named_property(env, mo, "stride", make_u32x3(env, st));
    napi_set_element(env, mips, k, mo);
  }
  napi_set_named_property(env, o, "rangeMipmaps", mips);
  napi_create_uint32(env, vxb_brick_counter(g), &e);
  napi_set_named_property(env, o, "brickCounter", e);
  vxb_free(g); /* worker.ts:54 */
  return o;
}

static napi_value init(napi_env env, napi_value exports) {
  static const struct { const char* name; napi_callback fn; } fns[] = {
      {"create", n_create}, {"destroy", n_destroy}, {"uploadVolume", n_upload_volume},
      {"uploadTransfer", n_upload_transfer}, {"uploadEnvironment", n_upload_environment}, {"setParams", n_set_params}, {"sizeofParams", n_sizeof_params},
      {"resize", n_resize}, {"setLayout", n_set_layout}, {"renderFrame", n_render_frame}, {"renderFrames", n_render_frames},
      {"probeTileCosts", n_probe_tile_costs}, {"setTileOrder", n_set_tile_order}, {"deviceInfo", n_device_info}, {"finish", n_finish},
      {"readAccum", n_read_accum}, {"readDisplay", n_read_display},
      {"readDisplayScaled", n_read_display_scaled}, {"getCounters", n_get_counters},
      {"resetCounters", n_reset_counters}, {"version", n_version}, {"buildBrickGrid", n_build_brick_grid},
      {"readDicomsToGrid", n_read_dicoms_to_grid}};
  for (size_t i = 0; i < sizeof fns / sizeof fns[0]; ++i) {
    napi_value f;
    if (napi_create_function(env, fns[i].name, NAPI_AUTO_LENGTH, fns[i].fn, NULL, &f) != napi_ok ||
        napi_set_named_property(env, exports, fns[i].name, f) != napi_ok) {
      napi_throw_error(env, NULL, "volxel_napi: export failed");
      return NULL;
    }
  }
  return exports;
}

NAPI_MODULE(volxel_napi, init)
