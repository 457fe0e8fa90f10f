#!/bin/bash
# usage (GPU box): tools/ab_all.sh <variant|cur> ...  -- DVR / Phong (fpl_sweep, phong_probe) and the reference modes (mode_probe) with
# each library variant volxel_amd/libvolxel_hip_<variant>.so, twice, back to back
for i in 1 2; do
for v in "$@"; do
  if [ $v = cur ]; then unset VOLXEL_HIP_LIB; else export VOLXEL_HIP_LIB=$PWD/volxel_amd/libvolxel_hip_$v.so; fi
  echo "=== $v (pass $i)"
  timeout -k 10 200 python tools/fpl_sweep.py --jitter 1 --fpl 32 --frames 512 2>&1 | grep 'frames/launch' | cut -c1-70
  timeout -k 10 200 python tools/phong_probe.py 2>/dev/null | grep "fpl 16" | cut -c1-60
  timeout -k 10 300 python tools/mode_probe.py --fpl 32 1 2>/dev/null | cut -c1-50
done; done
