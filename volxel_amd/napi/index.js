'use strict';
// mirrors volxel-3d-viewer/src/index.ts:1-4 for the parts on the hot path
module.exports = require('./viewer');
