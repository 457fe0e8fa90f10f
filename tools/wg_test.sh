export IMG_HASH_FRAMES=32 IMG_HASH_INFLIGHT=32
echo "=== hashes VX_DVR_WG=0"; VX_DVR_WG=0 timeout -k 10 120 python tools/img_hash.py 2>&1 | grep "^dvr" 
echo "=== hashes VX_DVR_WG=1"; VX_DVR_WG=1 timeout -k 10 120 python tools/img_hash.py 2>&1 | grep "^dvr" || exit 1
for i in 1 2; do for w in 0 1; do echo -n "pass $i WG=$w: "; VX_DVR_WG=$w timeout -k 10 200 python tools/fpl_sweep.py --jitter 1 --fpl 32,64 --frames 640 2>&1 | grep 'frames/launch' | cut -c1-100; done; done
