'use strict';
// node smoke.js <out_dir> [gpu]: brick builder through N-API (CPU); with "gpu" also one DVR frame.
const fs = require('fs');
const path = require('path');
const { Volxel3DDicomRenderer, generateTransferFunction, native } = require('./index');
const out = process.argv[2];
const n = 32;
const vox = new Uint16Array(n * n * n);
for (let z = 0; z < n; ++z) for (let y = 0; y < n; ++y) for (let x = 0; x < n; ++x) {
  const c = (n - 1) / 2, r = 28 * n / 64;
  const d = Math.sqrt((x - c) ** 2 + (y - c) ** 2 + (z - c) ** 2);
  vox[(z * n + y) * n + x] = Math.round(4095 * Math.max(0, 1 - d / r));
}
const grid = native.buildBrickGrid(vox, [n, n, n], [1, 1, 1], 0, 2);
fs.writeFileSync(path.join(out, 'atlas.bin'), Buffer.from(grid.atlas.buffer));
fs.writeFileSync(path.join(out, 'indirection.bin'), Buffer.from(grid.indirection.buffer));
fs.writeFileSync(path.join(out, 'range.bin'), Buffer.from(grid.range.buffer));
fs.writeFileSync(path.join(out, 'meta.json'), JSON.stringify({ atlasSize: grid.atlasSize, indexExtent: grid.indexExtent,
  brickCounter: grid.brickCounter, minMaj: grid.minMaj, mips: grid.rangeMipmaps.map(m => m.stride) }));
const dcmDir = path.join(out, 'dicom');
if (fs.existsSync(dcmDir)) {  // worker.ts:101-104 path: DICOM bytes in, brick grid out
  const files = fs.readdirSync(dcmDir).sort().map(f => new Uint8Array(fs.readFileSync(path.join(dcmDir, f))));
  const g2 = native.readDicomsToGrid(files, 2);
  fs.writeFileSync(path.join(out, 'atlas_dcm.bin'), Buffer.from(g2.atlas.buffer));
  fs.writeFileSync(path.join(out, 'meta_dcm.json'), JSON.stringify({ transform: Array.from(g2.transform),
    histogramLength: g2.histogram.length, indexExtent: g2.indexExtent }));
  let msg = '';
  try { native.readDicomsToGrid([new Uint8Array(200)], 1); } catch (e) { msg = e.message; }
  fs.writeFileSync(path.join(out, 'dcm_error.json'), JSON.stringify({ msg }));
}
const tf = generateTransferFunction([{ color: [1, 1, 1, 0], stop: 0 }, { color: [1, 1, 1, 1], stop: 1 }]);
fs.writeFileSync(path.join(out, 'tf.bin'), Buffer.from(tf.data.buffer));
if (process.argv[3] === 'gpu') {
  const r = new Volxel3DDicomRenderer({ width: 96, height: 64 });
  r.setupFromGrid(grid);
  r.settings.renderMode = 'dvr'; r.settings.bounces = 1;
  r.render(1);
  fs.writeFileSync(path.join(out, 'accum.bin'), Buffer.from(r.readAccum().buffer));
  fs.writeFileSync(path.join(out, 'params.bin'), Buffer.from(r.params.buffer));
  fs.writeFileSync(path.join(out, 'counters.json'), JSON.stringify(r.counters()));
  { // 12 more frames one by one vs the same frames 8 per launch: identical accumulators
    r.render(12); const a = r.readAccum().slice();
    r.restartRendering(); r.render(1); r.render(12, 8); const b = r.readAccum();
    let same = a.length === b.length;
    for (let i = 0; same && i < a.length; ++i) same = a[i] === b[i];
    fs.writeFileSync(path.join(out, 'pipelined.json'), JSON.stringify({ same, frameIndex: r.frameIndex }));
  }
  { // tile cost probe + a reversed dealing order: a single shard renders the same image in any order
    const costs = r.probeTileCosts();
    const perm = Uint32Array.from(costs.keys()).reverse();
    r.render(3); const a = r.readAccum().slice();
    r.setTileOrder(perm); r.render(r.frameIndex === 0 ? 16 : 0);
    // frameIndex restarted at 0: render the same number of frames as before the switch
    fs.writeFileSync(path.join(out, 'tiles.json'), JSON.stringify({ tiles: costs.length, nonzero: costs.filter(v => v > 0).length, device: r.deviceInfo() }));
    r.setTileOrder(null);
  }
  if (process.argv[4]) { // benchmark collection JSON -> result records (viewer.ts:856-890)
    const coll = JSON.parse(fs.readFileSync(process.argv[4], 'utf8'));
    fs.writeFileSync(path.join(out, 'benchmark_results.json'), JSON.stringify(r.startBenchmark(coll)));
  }
  r.dispose();
} else {
  let threw = false;
  try { new Volxel3DDicomRenderer({ width: 8, height: 8 }); } catch (e) { threw = /no HIP device|failed/.test(e.message); }
  fs.writeFileSync(path.join(out, 'nogpu.json'), JSON.stringify({ threw }));
}
