#!/bin/bash
# A/B of library variants back to back: tools/ab_probe.sh <variant|cur> ...   (variants: tools/variant_build.sh)
for i in 1 2; do
for v in "$@"; do
  if [ $v = cur ]; then unset VOLXEL_HIP_LIB; else export VOLXEL_HIP_LIB=volxel_amd/libvolxel_hip_$v.so; fi
  echo "$v $(python tools/fpl_sweep.py --jitter 1 --fpl 32 --frames 512 2>&1 | grep 'frames/launch' | cut -c1-60)"
done; done
