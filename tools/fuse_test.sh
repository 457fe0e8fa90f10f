#!/bin/bash
# usage (GPU box): tools/fuse_test.sh -- the running mean applied inside the LDS-window DVR kernel (MultiOut::fuse, launches of exactly 32 / 64
# frames) against the result slabs + merge_results (VX_DVR_FUSE=0): image hashes at 32 and 64 frames in flight, then kernel and blend time
for n in 32 64; do for f in 0 1; do echo "=== hashes, $n frames in flight, VX_DVR_FUSE=$f"; IMG_HASH_FRAMES=$((n + 8)) IMG_HASH_INFLIGHT=$n VX_DVR_FUSE=$f timeout -k 10 120 python tools/img_hash.py 2>&1 | grep "^dvr"; done; done
for i in 1 2 3; do for f in 0 1; do echo -n "pass $i FUSE=$f: "; VX_DVR_FUSE=$f timeout -k 10 200 python tools/fpl_sweep.py --jitter 1 --fpl 32 --frames 640 2>&1 | grep 'frames/launch' | cut -c1-110; done; done
