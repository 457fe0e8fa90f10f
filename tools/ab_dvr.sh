#!/bin/bash
# usage (GPU box): tools/ab_dvr.sh <variant|cur> ...  -- the headline DVR kernel (config 3, jitter on, 32 frames per launch, 640
# frames) with each library variant volxel_amd/libvolxel_hip_<variant>.so, three passes back to back (DVFS noise: compare
# within a pass), then the image hashes of every mode (bitwise A/B)
for i in 1 2 3; do
for v in "$@"; do
  if [ $v = cur ]; then unset VOLXEL_HIP_LIB; else export VOLXEL_HIP_LIB=$PWD/volxel_amd/libvolxel_hip_$v.so; fi
  echo -n "pass $i $v: "
  timeout -k 10 200 python tools/fpl_sweep.py --jitter 1 --fpl 32 --frames 640 2>&1 | grep 'frames/launch' | cut -c1-90
done; done
for v in "$@"; do
  if [ $v = cur ]; then unset VOLXEL_HIP_LIB; else export VOLXEL_HIP_LIB=$PWD/volxel_amd/libvolxel_hip_$v.so; fi
  echo "=== hashes $v"
  timeout -k 10 200 python tools/img_hash.py 2>/dev/null
done
