#!/bin/bash
# usage (GPU box): tools/fuse_modes_test.sh -- every render mode with the running mean applied in the render kernel (launches of exactly 32 frames)
# against result slabs + merge_results (VX_DVR_FUSE=0): image hashes, then kernel + blend ms per frame of the reference modes
for f in 0 1; do echo "=== hashes, 40 frames, 32 in flight, VX_DVR_FUSE=$f"; IMG_HASH_FRAMES=40 IMG_HASH_INFLIGHT=32 VX_DVR_FUSE=$f timeout -k 10 200 python tools/img_hash.py 2>&1 | grep -v "^$" | tail -6; done
for i in 1 2; do for f in 0 1; do echo "--- pass $i FUSE=$f"; VX_DVR_FUSE=$f timeout -k 10 300 python tools/mode_probe.py --fpl 32 1 2>/dev/null | cut -c1-100; done; done
