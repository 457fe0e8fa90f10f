#!/bin/bash
# A/B of library variants on the Phong workload (config 4): tools/ab_phong.sh <variant|cur> ...
for i in 1 2; do
for v in "$@"; do
  if [ $v = cur ]; then unset VOLXEL_HIP_LIB; else export VOLXEL_HIP_LIB=volxel_amd/libvolxel_hip_$v.so; fi
  echo "$v $(python tools/phong_probe.py 2 2>&1 | grep 'dvr_phong  jitter 1 fpl 16' | cut -c1-50)"
done; done
