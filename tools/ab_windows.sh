#!/bin/bash
# usage (GPU box): tools/ab_windows.sh <variant|cur> ... -- windows per frame, lane utilisation and ms per frame of the DVR kernel per variant
for v in "$@"; do
  if [ $v = cur ]; then unset VOLXEL_HIP_LIB; else export VOLXEL_HIP_LIB=$PWD/volxel_amd/libvolxel_hip_$v.so; fi
  echo "=== $v"
  timeout -k 10 200 python tools/fpl_probe.py 2>&1 | grep "dvr fpl"
done
