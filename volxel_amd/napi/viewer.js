'use strict';
/**
 * Headless JavaScript host over the N-API shim: keeps the public surface of the reference's
 * render core (class Volxel3DDicomRenderer, volxel-3d-viewer/src/viewer.ts) for the calls on
 * the hot path -- setupFromGrid (viewer.ts:1080-1145), changeTransferFunc (:1147-1153),
 * restartRendering (:1155-1181), bindUniforms (:1295-1357), render (:1183-1293), renderMode
 * (:1442-1452), restoreSettings (:704-713) -- and of utils/data.ts (generateTransferFunction).
 * The reference is TypeScript + math.gl in a browser; the build image has Node 12 without tsc,
 * so this is plain ES2019 with the typings in index.d.ts.  Everything that touches pixels goes
 * through libvolxel_hip.so; there is no JavaScript fallback.
 */
const fs = require('fs');
const path = require('path');
const native = require('./volxel_napi.node');

const RenderMode = Object.freeze({ default: 0, no_dda: 1, raymarch: 2, dvr: 3, dvr_phong: 4 });
const LOW_RES_DURATION = 5; // viewer.ts:132

// ---- VxParams field table, parsed from the C header (single source of truth) --------------
function parseParamsLayout() {
  const text = fs.readFileSync(path.join(__dirname, '..', '..', 'include', 'volxel_hip.h'), 'utf8')
    .replace(/\/\*[\s\S]*?\*\//g, '');
  const body = /typedef\s+struct\s+VxParams\s*\{([\s\S]*?)\}\s*VxParams\s*;/.exec(text)[1];
  const fields = {};
  let off = 0;
  for (const decl of body.split(';')) {
    const m = /^\s*(\w+)\s+([\s\S]+)$/.exec(decl);
    if (!m) continue;
    for (const item of m[2].split(',')) {
      const a = /^\s*(\w+)\s*(?:\[(\d+)\])?\s*$/.exec(item);
      const n = a[2] ? parseInt(a[2], 10) : 1;
      fields[a[1]] = { type: m[1], offset: off, count: n };
      off += 4 * n;
    }
  }
  if (off !== native.sizeofParams()) throw new Error('VxParams layout mismatch between header and library');
  return { fields, size: off };
}
const LAYOUT = parseParamsLayout();

class ParamsBlock {
  constructor() { this.buffer = new ArrayBuffer(LAYOUT.size); this.view = new DataView(this.buffer); }
  set(name, value) {
    const f = LAYOUT.fields[name];
    if (!f) throw new Error(`unknown uniform ${name}`);
    const vals = (typeof value === 'number') ? [value] : Array.from(value);
    if (vals.length !== f.count) throw new Error(`uniform ${name} expects ${f.count} values`);
    vals.forEach((v, i) => {
      const o = f.offset + 4 * i;
      if (f.type === 'float') this.view.setFloat32(o, v, true);
      else if (f.type === 'uint32_t') this.view.setUint32(o, v, true);
      else this.view.setInt32(o, v, true);
    });
  }
}

// ---- column-major 4x4 helpers in doubles (gl-matrix conventions used through math.gl) ------
const M = {
  identity: () => [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1],
  mul(a, b) { // a * b
    const o = new Array(16).fill(0);
    for (let c = 0; c < 4; ++c) for (let r = 0; r < 4; ++r)
      for (let k = 0; k < 4; ++k) o[4 * c + r] += a[4 * k + r] * b[4 * c + k];
    return o;
  },
  scale: (s) => [s[0], 0, 0, 0, 0, s[1], 0, 0, 0, 0, s[2], 0, 0, 0, 0, 1],
  translate: (t) => [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, t[0], t[1], t[2], 1],
  apply(m, v) { return [0, 1, 2, 3].map(r => m[r] * v[0] + m[4 + r] * v[1] + m[8 + r] * v[2] + m[12 + r] * v[3]); },
  lookAt(eye, center, up) { // scene.ts:58-64
    let z = [eye[0] - center[0], eye[1] - center[1], eye[2] - center[2]];
    if (z.every(c => Math.abs(c) < 1e-6)) return M.identity();
    const n = (v) => { const l = Math.hypot(v[0], v[1], v[2]); return l ? v.map(c => c / l) : [0, 0, 0]; };
    const cross = (a, b) => [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]];
    const dot = (a, b) => a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
    z = n(z);
    const x = n(cross(up, z));
    const y = n(cross(z, x));
    return [x[0], y[0], z[0], 0, x[1], y[1], z[1], 0, x[2], y[2], z[2], 0, -dot(x, eye), -dot(y, eye), -dot(z, eye), 1];
  },
  perspective(fovy, aspect, near, far) { // scene.ts:65-72
    const f = 1 / Math.tan(fovy / 2), nf = 1 / (near - far);
    return [f / aspect, 0, 0, 0, 0, f, 0, 0, 0, 0, (far + near) * nf, -1, 0, 0, 2 * far * near * nf, 0];
  },
  ortho(left, right, bottom, top, near, far) { // gl-matrix mat4.ortho; [build]: the reference camera is perspective only
    const lr = 1 / (left - right), bt = 1 / (bottom - top), nf = 1 / (near - far);
    return [-2 * lr, 0, 0, 0, 0, -2 * bt, 0, 0, 0, 0, 2 * nf, 0, (left + right) * lr, (top + bottom) * bt, (far + near) * nf, 1];
  },
  invert(m) { // Gauss-Jordan with partial pivoting on the 4x4
    const a = [0, 1, 2, 3].map(r => [m[r], m[4 + r], m[8 + r], m[12 + r], +(r === 0), +(r === 1), +(r === 2), +(r === 3)]);
    for (let c = 0; c < 4; ++c) {
      let p = c;
      for (let r = c + 1; r < 4; ++r) if (Math.abs(a[r][c]) > Math.abs(a[p][c])) p = r;
      if (a[p][c] === 0) throw new Error('singular matrix');
      [a[c], a[p]] = [a[p], a[c]];
      const d = a[c][c];
      for (let k = 0; k < 8; ++k) a[c][k] /= d;
      for (let r = 0; r < 4; ++r) if (r !== c) { const f = a[r][c]; for (let k = 0; k < 8; ++k) a[r][k] -= f * a[c][k]; }
    }
    const o = new Array(16);
    for (let r = 0; r < 4; ++r) for (let c = 0; c < 4; ++c) o[4 * c + r] = a[r][4 + c];
    return o;
  },
  f32: (m) => Array.from(new Float32Array(m)),
};

// ---- utils/data.ts:21-60 ---------------------------------------------------------------
function generateTransferFunction(colors, generatedSteps = 128) {
  if (colors.length < 1) throw new Error('At least one color stop required');
  const stops = colors.slice().sort((a, b) => a.stop - b.stop);
  if (stops.some(s => s.stop < 0 || s.stop > 1)) throw new Error('ColorStop outside stop range');
  const out = [];
  let cur = -1;
  for (let i = 0; i < generatedSteps; ++i) {
    const pos = i / generatedSteps;
    if (cur < 0) {
      if (stops[0].stop >= pos) { cur = 0; out.push(stops[0].color); } else out.push([0, 0, 0, 0]);
      continue;
    }
    const next = stops[cur + 1];
    if (!next) { out.push(stops[cur].color); continue; }
    const w = (pos - stops[cur].stop) / (next.stop - stops[cur].stop);
    if (w >= 1) { out.push(next.color); cur++; continue; }
    out.push(stops[cur].color.map((v, k) => (1 - w) * v + w * next.color[k]));
  }
  return { data: new Float32Array(out.flat()), length: generatedSteps };
}

// math.gl Quaternion / Vector3 pieces used by the orbit camera (gl-matrix quat.setAxisAngle, quat.multiply, vec3.transformQuat)
const Q = {
  axis: (a, rad) => { const s = Math.sin(rad * 0.5); return [a[0] * s, a[1] * s, a[2] * s, Math.cos(rad * 0.5)]; },
  mul: (a, b) => [a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1], a[1] * b[3] + a[3] * b[1] + a[2] * b[0] - a[0] * b[2],
    a[2] * b[3] + a[3] * b[2] + a[0] * b[1] - a[1] * b[0], a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2]],
  rot: (v, q) => {
    const cross = (a, b) => [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]];
    const uv = cross(q, v), uuv = cross(q, uv);
    return [0, 1, 2].map(i => v[i] + 2 * q[3] * uv[i] + 2 * uuv[i]);
  },
};
const V = {
  sub: (a, b) => a.map((x, i) => x - b[i]), add: (a, b) => a.map((x, i) => x + b[i]), scale: (a, s) => a.map(x => x * s),
  len: a => Math.sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]), norm: a => V.scale(a, 1 / V.len(a)),
  cross: (a, b) => [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]],
};

class Camera { // representation/scene.ts
  // orthoHalfHeight ([build], BASELINE config 1): null = the reference's perspective camera
  constructor(distance = 1) { this.view = [0, 0, 0]; this.pos = [0, 0, -distance]; this.orthoHalfHeight = null; this.yaw = 0; this.pitch = 0; }
  rotateAroundView(by) { // scene.ts:15-33
    this.yaw += -by[0]; this.pitch += by[1];
    const maxPitch = Math.PI / 2 - 0.01;
    this.pitch = Math.min(Math.max(this.pitch, -maxPitch), maxPitch);
    const qYaw = Q.axis([0, 1, 0], this.yaw);
    const right = V.norm(Q.rot([1, 0, 0], qYaw));
    const orientation = Q.mul(Q.axis(right, this.pitch), qYaw);
    this.pos = V.add(V.scale(Q.rot([0, 0, -1], orientation), V.len(V.sub(this.pos, this.view))), this.view);
  }
  zoom(by) { // scene.ts:35-40
    const dir = V.sub(this.pos, this.view), n = V.len(dir);
    if (n * by <= 0.1 || n * by >= 10) return false;
    this.pos = V.add(V.scale(dir, by), this.view);
    return true;
  }
  translateOnPlane(by) { // scene.ts:42-47
    const dir = V.sub(this.pos, this.view), right = V.norm(V.cross(dir, [0, 1, 0])), localUp = V.norm(V.cross(dir, right));
    this.translate(V.add(V.scale(right, by[0] * 5), V.scale(localUp, -by[1] * 5)));
  }
  translate(by) { this.pos = V.add(this.pos, by); this.view = V.add(this.view, by); } // scene.ts:49-52
  viewMatrix() { return M.lookAt(this.pos, this.view, [0, 1, 0]); }
  projMatrix(aspect, fov = Math.PI / 3) {
    if (this.orthoHalfHeight !== null) {
      const h = this.orthoHalfHeight;
      return M.ortho(-h * aspect, h * aspect, -h, h, 0.1, 1000);
    }
    return M.perspective(fov, aspect, 0.1, 1000);
  }
}

/** representation/environment.ts: base map = width x height RGBA floats, row 0 = TOP */
class Environment {
  constructor(floats, width, height, strength = 1) {
    if (floats.length !== width * height * 4) throw new Error('Environment: floats must hold width*height RGBA texels');
    this.floats = floats; this.width = width; this.height = height; this.strength = strength;
  }
  static default() { // environment.ts:102-130, 8x6 checkerboard with a bright upper third
    const width = 8, height = 6, d = new Float32Array(width * height * 4);
    for (let y = 0; y < height; ++y) {
      const top = y < Math.floor(height / 3);
      for (let x = 0; x < width; ++x) {
        const light = ((x + y) & 1) === 0, val = top ? (light ? 3 : 0.9) : (light ? 0.1 : 0.0), i = (y * width + x) * 4;
        d[i] = d[i + 1] = d[i + 2] = val; d[i + 3] = 1;
      }
    }
    return new Environment(d, width, height);
  }
}

// ---- container I/O behind the load methods (not on the hot path; what Node's own zlib / fs can decode) ------------
/** zip.rs:36-115: files in archive order, at most one directory entry, every later file directly inside it */
function readZipSlices(buf) {
  const zlib = require('zlib');
  const fail = (kind, msg) => { throw new Error(`${kind}: ${msg === undefined ? 'No Message Specified' : msg}`); };
  let eocd = -1;
  for (let i = buf.length - 22; i >= Math.max(0, buf.length - 65557); --i) if (buf.readUInt32LE(i) === 0x06054b50) { eocd = i; break; }
  if (eocd < 0) fail('ExtractFailed', 'invalid Zip archive: Could not find central directory end');
  const total = buf.readUInt16LE(eocd + 10);
  let p = buf.readUInt32LE(eocd + 16);
  if (total < 1) fail('NoFiles');
  let directory = null;
  const out = [];
  for (let k = 0; k < total; ++k) {
    if (buf.readUInt32LE(p) !== 0x02014b50) fail('ExtractFailed', 'invalid central directory entry');
    const method = buf.readUInt16LE(p + 10), csize = buf.readUInt32LE(p + 20), nlen = buf.readUInt16LE(p + 28),
      xlen = buf.readUInt16LE(p + 30), clen = buf.readUInt16LE(p + 32), lho = buf.readUInt32LE(p + 42);
    const name = buf.toString('utf8', p + 46, p + 46 + nlen);
    p += 46 + nlen + xlen + clen;
    const norm = require('path').posix.normalize(name);
    if (name.startsWith('/') || norm.startsWith('..')) fail('ExtractFailed', 'No enclosed name was able to be found');
    if (name.endsWith('/')) {
      if (directory !== null) fail('MoreThanOneFolder');
      directory = norm.replace(/\/$/, '');
      continue;
    }
    if (directory !== null && require('path').posix.dirname(norm) !== directory) fail('MoreThanOneFolder');
    const dataAt = lho + 30 + buf.readUInt16LE(lho + 26) + buf.readUInt16LE(lho + 28);
    const raw = buf.slice(dataAt, dataAt + csize);
    if (method === 0) out.push(new Uint8Array(raw));
    else if (method === 8) out.push(new Uint8Array(zlib.inflateRawSync(raw)));
    else fail('ExtractFailed', `unsupported compression method ${method}`);
  }
  if (!out.length) fail('NoFiles', 'No dicom data collected');
  return out;
}
/** worker.ts:115-126 fetch(url) + exportResponseBytes: a path or a file:// URL (this Node has no fetch, the machines no network) */
function fetchBytes(url) {
  const fs = require('fs');
  if (/^file:\/\//.test(url)) return fs.readFileSync(new (require('url').URL)(url));
  if (/^[a-z][a-z0-9+.-]*:\/\//i.test(url)) throw new Error(`fetch is not available in this host: ${url} (read the bytes and call the *Bytes / loadEnv method)`);
  return fs.readFileSync(url);
}
/** hdr.rs:23-36: encoded environment map -> { floats, width, height } with row 0 = top.  Radiance RGBE only. */
function decodeEnvironment(b) {
  if (b.length >= 4 && b.readUInt32LE(0) === 0x01312f76) throw new Error("OpenEXR decode is outside this build's scope (hdr.rs uses the `image` crate's exr decoder): decode the map elsewhere and pass {width, height, floats} to setupEnv()");
  const head = b.toString('latin1', 0, 10);
  if (!head.startsWith('#?RADIANCE') && !head.startsWith('#?RGBE')) throw new Error('unrecognised environment map format (expected Radiance RGBE; decoded floats go to setupEnv())');
  let pos = b.indexOf('\n\n') + 2;
  if (!b.toString('latin1', 0, pos).includes('FORMAT=32-bit_rle_rgbe')) throw new Error('Radiance map: only FORMAT=32-bit_rle_rgbe is supported');
  const eol = b.indexOf('\n', pos), dims = b.toString('latin1', pos, eol).trim().split(/\s+/);
  if (dims.length !== 4 || dims[0] !== '-Y' || dims[2] !== '+X') throw new Error('Radiance map: only the standard -Y h +X w orientation is supported');
  const h = parseInt(dims[1], 10), w = parseInt(dims[3], 10);
  let p = eol + 1;
  const px = new Uint8Array(w * h * 4);
  for (let y = 0; y < h; ++y) {
    if (w >= 8 && w < 32768 && b[p] === 2 && b[p + 1] === 2 && ((b[p + 2] << 8) | b[p + 3]) === w) {
      p += 4;
      for (let ch = 0; ch < 4; ++ch) for (let x = 0; x < w;) {
        let n = b[p++];
        const run = n > 128; if (run) n -= 128;
        // a zero-length run never advances; a run past the scanline or the buffer is a corrupt file (the image crate errors too)
        if (n === 0 || x + n > w || p + (run ? 1 : n) > b.length) throw new Error('Radiance map: corrupt run-length data');
        if (run) { const v = b[p++]; for (let k = 0; k < n; ++k) px[(y * w + x + k) * 4 + ch] = v; } else { for (let k = 0; k < n; ++k) px[(y * w + x + k) * 4 + ch] = b[p++]; }
        x += n;
      }
    } else { if (p + 4 * w > b.length) throw new Error('Radiance map: pixel data ends early'); for (let k = 0; k < 4 * w; ++k) px[y * w * 4 + k] = b[p++]; }
  }
  const floats = new Float32Array(w * h * 4);
  for (let i = 0; i < w * h; ++i) {
    const e = px[4 * i + 3], s = e > 0 ? Math.pow(2, e - 136) : 0;
    floats[4 * i] = Math.fround(px[4 * i] * s); floats[4 * i + 1] = Math.fround(px[4 * i + 1] * s); floats[4 * i + 2] = Math.fround(px[4 * i + 2] * s); floats[4 * i + 3] = 1;
  }
  return { floats, width: w, height: h };
}

let workerFactory = null;
/** registerVolxelComponents (viewer.ts:1455-1462): keeps the worker factory and names the component classes.  The four
 *  widget elements are browser UI without a counterpart here; the renderer calls the native preprocessor in-process. */
function registerVolxelComponents(worker) {
  workerFactory = worker || null;
  return { 'volxel-3d-viewer': Volxel3DDicomRenderer };
}

class Volxel3DDicomRenderer {
  /** width, height: the canvas.  lowResPreview reproduces the viewer's interactive sizing
   *  (settings.resolutionFactor and the 0.33 ramp of viewer.ts:1167-1188); off = full-size frames. */
  constructor({ width = 1920, height = 1080, device = 0, layout, lowResPreview = false } = {}) {
    this.ctx = native.create(device);            // throws when no GPU is visible
    this.canvasWidth = width; this.canvasHeight = height;
    this.width = width; this.height = height;    // current render size
    this.lowResPreview = lowResPreview; this.resolutionFactor = 1.0; // viewer.ts:131
    this.settings = { // viewer.ts:147-163 + [build] DVR parameters
      densityMultiplier: 1, maxSamples: 2000, debugHits: false, volumeClipMin: [0, 0, 0], volumeClipMax: [1, 1, 1],
      showEnvironment: true, useEnv: true, lightDir: [-1, -1, -1].map(v => v / Math.sqrt(3)), syncLightDir: false,
      bounces: 3, gamma: 2.2, exposure: 5.5, sampleRange: [0, 1], renderMode: 'default', resolutionFactor: 1,
      dvrStepVoxels: 0.5, dvrErtEpsilon: 1e-4, dvrJitter: false, dvrMaxSteps: 1 << 20, dvrSkipEmpty: true, phong: [0.3, 0.7, 0.4, 32],
    };
    this.camera = new Camera(1);                  // viewer.ts:418
    this.environment = null;
    this.frameIndex = 0;
    this.densityScale = 1;
    this.volume = null;
    if (layout !== undefined) native.setLayout(this.ctx, layout);
    native.resize(this.ctx, width, height);
    this.setEnvironment(Environment.default());   // viewer.ts:372-374
    const tf = generateTransferFunction([{ color: [1, 1, 1, 0], stop: 0 }, { color: [1, 1, 1, 1], stop: 1 }]);
    this.changeTransferFunc(tf.data, tf.length);  // viewer.ts:377-385
  }
  setEnvironment(env) { // viewer.ts:1073-1078; null removes the map (directional light only)
    if (env) native.uploadEnvironment(this.ctx, env.floats, env.width, env.height);
    else native.uploadEnvironment(this.ctx, null, 0, 0);
    this.environment = env || null;
    this.restartRendering();
  }
  get envStrength() { return this.environment ? this.environment.strength : (this._envStrength === undefined ? 1 : this._envStrength); }
  set envStrength(v) { this._envStrength = v; if (this.environment) this.environment.strength = v; }
  dispose() { if (this.ctx) { native.destroy(this.ctx); this.ctx = null; } }

  get renderMode() { return this.settings.renderMode; }
  set renderMode(to) {
    if (!(to in RenderMode)) throw new Error(`Unrecognized render mode provided: ${to}`);
    this.settings.renderMode = to; this.restartRendering();
  }

  /** restartFromFiles (viewer.ts:963-975): DICOM slice paths (or Uint8Arrays) in stacking order */
  restartFromFiles(files, threads = 0) {
    const fs = require('fs');
    this.restartFromBytes(files.map(f => (typeof f === 'string' ? new Uint8Array(fs.readFileSync(f)) : f)), threads);
  }
  /** restartFromZip (viewer.ts:977-989): a ZIP of DICOM slices (Buffer / Uint8Array / path), folder rule of zip.rs:54-70 */
  restartFromZip(zip, threads = 0) {
    const fs = require('fs');
    this.restartFromBytes(readZipSlices(typeof zip === 'string' ? fs.readFileSync(zip) : Buffer.from(zip.buffer || zip, zip.byteOffset || 0, zip.byteLength)), threads);
  }
  /** restartFromZipUrl (viewer.ts:991-1003, worker.ts:115-118) */
  restartFromZipUrl(url, threads = 0) { this.restartFromZip(fetchBytes(url), threads); }
  /** restartFromURLs (viewer.ts:1005-1017, worker.ts:120-123): one slice per URL, in the order given */
  restartFromURLs(urls, threads = 0) { this.restartFromBytes(urls.map(u => new Uint8Array(fetchBytes(u))), threads); }
  /** loadEnv (viewer.ts:1019-1033, worker.ts:77-90, hdr.rs:23-36): Radiance RGBE is decoded, OpenEXR is refused */
  loadEnv(bytes) { this.setupEnv(decodeEnvironment(Buffer.from(bytes.buffer || bytes, bytes.byteOffset || 0, bytes.byteLength))); }
  /** loadEnvFromUrl (viewer.ts:1035-1040) */
  loadEnvFromUrl(url) { this.loadEnv(fetchBytes(url)); }
  /** maybeSyncLight (viewer.ts:789-795): lightDir = -(view - pos), normalised by the direction widget it is assigned to
   *  (cubeDirection.ts:177-206).  Called after a camera rotation and by the backlight toggle -- not by restoreSettings,
   *  exactly as in the reference. */
  maybeSyncLight() {
    if (!this.settings.syncLightDir) return;
    const d = V.sub(this.camera.pos, this.camera.view), n = V.len(d);
    if (n > 0) this.settings.lightDir = V.scale(d, 1 / n);
  }
  rotateCamera(by) { this.camera.rotateAroundView(by); this.maybeSyncLight(); this.restartRendering(); } // viewer.ts:443-449
  get syncLightDir() { return this.settings.syncLightDir; }
  set syncLightDir(on) { this.settings.syncLightDir = !!on; this.maybeSyncLight(); this.restartRendering(); } // viewer.ts:543-551
  /** setupEnv (viewer.ts:1073-1078): { width, height, floats } with row 0 = top */
  setupEnv(env) { this.setEnvironment(new Environment(env.floats, env.width, env.height)); }
  /** the same once the files are in memory: one Uint8Array per slice (worker.ts:101-104) */
  restartFromBytes(files, threads = 0) {
    this.setupFromGrid(native.readDicomsToGrid(files, threads));
  }

  /** the same for an already decoded u16 stack */
  restartFromVoxels(voxels, dims, spacing = [1, 1, 1], maxValue = 0, threads = 0) {
    this.setupFromGrid(native.buildBrickGrid(voxels, dims, spacing, maxValue, threads));
  }

  setupFromGrid(grid) { // viewer.ts:1080-1145
    this.densityScale = 1;
    this.settings.volumeClipMax = [1, 1, 1]; this.settings.volumeClipMin = [0, 0, 0];
    const g = Array.from(grid.transform);
    const e = grid.indexExtent;
    const lo = M.apply(g, [0, 0, 0, 1]).slice(0, 3), hi = M.apply(g, [e[0], e[1], e[2], 1]).slice(0, 3);
    const ext = hi.map((v, i) => v - lo[i]);
    const size = Math.max(ext[0], Math.max(ext[1], ext[2]));
    let t = M.identity();
    if (size !== 1) {
      t = M.mul(M.scale([1 / size, 1 / size, 1 / size]), M.translate(lo.map((v, i) => -v - ext[i] * 0.5)));
      this.densityScale *= size;
    }
    this.volume = { grid: { transform: g, indexExtent: e, minMaj: grid.minMaj }, transform: t };
    native.uploadVolume(this.ctx, grid);
    this.restartRendering();
  }

  changeTransferFunc(data, length) { // viewer.ts:1147-1153
    native.uploadTransfer(this.ctx, data, length);
    this.frameIndex = 0;
  }
  restartRendering() {                          // viewer.ts:1155-1181
    if (this.lowResPreview) { this.resolutionFactor = 0.33; this.resizeFramebuffersToCanvas(); }
    this.frameIndex = 0;
  }
  resizeFramebuffersToCanvas() {                // viewer.ts:925-949
    const f = this.lowResPreview ? this.resolutionFactor * this.settings.resolutionFactor : 1.0;
    const w = Math.max(1, Math.floor(this.canvasWidth * f)), h = Math.max(1, Math.floor(this.canvasHeight * f));
    if (w !== this.width || h !== this.height) { this.width = w; this.height = h; native.resize(this.ctx, w, h); }
  }

  restoreSettings(s) { // viewer.ts:704-713 (+ transfer/display/lighting closures)
    if (s.version !== 'v3') throw new Error(`Unsupported Settings Format Version: ${s.version}`);
    this.settings.densityMultiplier = s.transfer.densityMultiplier;
    this.settings.sampleRange = s.transfer.histogramRange.slice();
    if (s.transfer.transfer.type === 'color_stops') {
      const tf = generateTransferFunction(s.transfer.transfer.colors);
      this.changeTransferFunc(tf.data, tf.length);
    } else {
      this.changeTransferFunc(new Float32Array(s.transfer.transfer.colors.flat()), s.transfer.transfer.colors.length);
    }
    Object.assign(this.settings, {
      bounces: s.display.bounces, maxSamples: s.display.samples, gamma: s.display.gamma, exposure: s.display.exposure,
      debugHits: s.display.debugHits, renderMode: s.display.renderMode, resolutionFactor: s.display.resolutionFactor,
      showEnvironment: s.lighting.showEnv, useEnv: s.lighting.useEnv, syncLightDir: s.lighting.syncLightDir,
      lightDir: s.lighting.lightDir.slice(), volumeClipMax: s.other.clipMax.slice(), volumeClipMin: s.other.clipMin.slice(),
    });
    this.envStrength = s.lighting.envStrength;
    this.camera.pos = s.other.cameraPos.slice(); this.camera.view = s.other.cameraLookAt.slice();
    this.restartRendering();
  }

  bindUniforms() { // viewer.ts:1295-1357 + scene.ts:53-56
    if (!this.volume) throw new Error('Trying to bind uniforms without a volume.');
    const p = new ParamsBlock(), s = this.settings;
    const view = this.camera.viewMatrix(), proj = this.camera.projMatrix(this.width / this.height);
    p.set('camera_view', view); p.set('camera_proj', proj);
    p.set('camera_view_inv', M.invert(M.f32(view))); p.set('camera_proj_inv', M.invert(M.f32(proj)));
    p.set('camera_ortho', this.camera.orthoHalfHeight !== null ? 1 : 0);
    const combined = M.mul(this.volume.transform, this.volume.grid.transform); // volume.ts:14-16
    const e = this.volume.grid.indexExtent;
    const lo = M.apply(combined, [0, 0, 0, 1]), hi = M.apply(combined, [e[0], e[1], e[2], 1]);
    p.set('volume_aabb_min', [0, 1, 2].map(i => lo[i] + (hi[i] - lo[i]) * s.volumeClipMin[i]));
    p.set('volume_aabb_max', [0, 1, 2].map(i => lo[i] + (hi[i] - lo[i]) * s.volumeClipMax[i]));
    const [mn, maj] = this.volume.grid.minMaj, k = this.densityScale * s.densityMultiplier;
    p.set('volume_min', mn * k); p.set('volume_maj', maj * k); p.set('volume_inv_maj', 1 / (maj * k));
    p.set('volume_albedo', [0.9, 0.9, 0.9]); p.set('volume_phase_g', 0); p.set('volume_density_scale', k);
    p.set('density_transform', combined); p.set('density_transform_inv', M.invert(combined));
    p.set('sample_range', s.sampleRange);
    p.set('light_dir', s.lightDir); p.set('env_strength', this.envStrength);
    p.set('show_environment', s.showEnvironment ? 1 : 0); p.set('use_env', s.useEnv && this.environment ? 1 : 0); p.set('bounces', s.bounces);
    p.set('res', [this.width, this.height]); p.set('debug_hits', s.debugHits ? 1 : 0);
    p.set('render_mode', RenderMode[s.renderMode]);
    p.set('dvr_step_voxels', s.dvrStepVoxels); p.set('dvr_ert_tau', -Math.log(s.dvrErtEpsilon));
    p.set('dvr_jitter', s.dvrJitter ? 1 : 0); p.set('dvr_max_steps', s.dvrMaxSteps); p.set('dvr_skip_empty', s.dvrSkipEmpty ? 1 : 0);
    const f = Math.fround;
    const fp = f(f(1) / f(f(4) * f(Math.PI)));                         // utils.glsl:121-124, g = 0
    const mis = s.showEnvironment ? f(f(1) / f(f(1) + f(fp * fp))) : f(1); // utils.glsl:104
    const gain = f(f(f(f(0.9) * mis) * fp) * f(f(this.envStrength) * f(4.01)));
    p.set('dvr_gain', [gain, gain, gain]);
    p.set('phong_ka', s.phong[0]); p.set('phong_kd', s.phong[1]); p.set('phong_ks', s.phong[2]);
    p.set('phong_shininess', s.phong[3]);
    p.set('shard_rank', 0); p.set('shard_count', 1);
    native.setParams(this.ctx, p.buffer);
    this.params = p;
    return p;
  }

  /** frames accumulation samples, up to inFlight of them per launch (vx_render_frames: same bits as one launch per
   *  frame, 1.6x the speed at 32) */
  render(frames = 1, inFlight = 32) { // viewer.ts:1183-1293
    const weight = f => (f < LOW_RES_DURATION ? 0 : (f - LOW_RES_DURATION) / (f - LOW_RES_DURATION + 1)); // viewer.ts:1356
    let bound = false;
    for (let i = 0; i < frames && this.frameIndex <= this.settings.maxSamples;) {
      if (this.lowResPreview && this.frameIndex >= LOW_RES_DURATION && this.resolutionFactor !== 1.0) {
        this.resolutionFactor = 1.0; this.resizeFramebuffersToCanvas(); bound = false; // viewer.ts:1185-1188
      }
      if (!bound) { this.bindUniforms(); bound = true; }
      const f = this.frameIndex;
      const ramp = this.lowResPreview && f < LOW_RES_DURATION;   // the size changes at LOW_RES_DURATION
      let n = Math.min(frames - i, this.settings.maxSamples + 1 - f);
      if (ramp) n = Math.min(n, LOW_RES_DURATION - f);
      if (inFlight > 1 && n > 1) {
        const w = new Float32Array(n);
        for (let k = 0; k < n; ++k) w[k] = weight(f + k);
        native.renderFrames(this.ctx, f, w, inFlight);
      } else {
        n = 1;
        native.renderFrame(this.ctx, f, weight(f));
      }
      this.frameIndex += n; i += n;
    }
  }
  /** startBenchmark (viewer.ts:856-890) + the result record of viewer.ts:1229-1241.  `volumes` maps an
   *  entry's "zip" string to a brick-grid message (container I/O is outside the path). */
  startBenchmark(collection, volumes = {}) {
    if (!collection || !Array.isArray(collection.sharedSettings) || !Array.isArray(collection.benchmarks)) throw new Error('Malformed benchmark collection.');
    const results = [], savedPreview = this.lowResPreview;
    this.lowResPreview = true;                    // the viewer's framebuffer sizing is part of the scenario
    try {
      for (const b of collection.benchmarks) {
        if (b.zip !== undefined) {
          if (!volumes[b.zip]) throw new Error(`benchmark volume not provided: ${b.zip}`);
          this.setupFromGrid(volumes[b.zip]);
        }
        if (b.env !== undefined) throw new Error('benchmark entry asks for an environment map URL; pass the decoded map with setEnvironment() before the run');
        this.restoreSettings(typeof b.settings === 'number' ? collection.sharedSettings[b.settings] : b.settings);
        if (b.renderMode) this.renderMode = b.renderMode;
        this.restartRendering();
        let total = 0;
        while (this.frameIndex <= this.settings.maxSamples) {   // viewer.ts:1194
          const t0 = process.hrtime.bigint();
          this.render(1); this.finish();                          // viewer.ts:1213-1218
          total += Number(process.hrtime.bigint() - t0) / 1e6;
        }
        const rf = this.settings.resolutionFactor;
        results.push(JSON.parse(JSON.stringify({
          name: b.name, settings: this.settings, timePerSample: total / this.frameIndex, totalTime: total,
          viewport: [0, 0, rf * this.canvasWidth, rf * this.canvasHeight],
          device: { platform: process.platform, userAgent: `node ${process.version} / volxel_hip ${native.version()}`,
            hardwareConcurrency: require('os').cpus().length, screen: { width: this.canvasWidth, height: this.canvasHeight, pixelRatio: 1 },
            gpu: { vendor: 'AMD', renderer: this.deviceInfo().name.trim(), version: 'HIP' } },
          timestamp: new Date(),
        })));
      }
    } finally {
      this.lowResPreview = savedPreview;
      if (!savedPreview) { this.resolutionFactor = 1.0; this.resizeFramebuffersToCanvas(); }
    }
    return results;
  }
  /** multi-GPU hosts: cost estimate of every 64x64 tile / dealing order of the tiles (volxel_hip.h) */
  probeTileCosts() {
    this.bindUniforms();
    const n = Math.ceil(this.width / 64) * Math.ceil(this.height / 64), out = new Uint32Array(n);
    native.probeTileCosts(this.ctx, out);
    return out;
  }
  setTileOrder(perm) { native.setTileOrder(this.ctx, perm || null); this.restartRendering(); }
  deviceInfo() { return native.deviceInfo(this.ctx); }
  finish() { native.finish(this.ctx); }
  readAccum() { const o = new Float32Array(this.width * this.height * 4); native.readAccum(this.ctx, o); return o; }
  readDisplay() { // the blit to the canvas (NEAREST, viewer.ts:310-311,1253-1265)
    const o = new Uint8Array(this.canvasWidth * this.canvasHeight * 4);
    native.readDisplayScaled(this.ctx, o, this.canvasWidth, this.canvasHeight, this.settings.exposure, this.settings.gamma);
    return o;
  }
  counters() { return native.getCounters(this.ctx); }
  resetCounters() { native.resetCounters(this.ctx); }
}

module.exports = { Volxel3DDicomRenderer, Environment, VolxelRenderMode: RenderMode, generateTransferFunction, Camera, native,
  registerVolxelComponents, readZipSlices, decodeEnvironment, fetchBytes, getWorkerFactory: () => workerFactory };
