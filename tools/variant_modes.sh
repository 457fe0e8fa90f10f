#!/bin/bash
# usage (GPU box): tools/variant_modes.sh "<mode_probe args>" name...   -- tools/mode_probe.py with each library variant
ARGS=$1; shift
for v in "$@"; do
  if [ "$v" = default ]; then unset VOLXEL_HIP_LIB; else export VOLXEL_HIP_LIB=$PWD/volxel_amd/libvolxel_hip_$v.so; fi
  echo "=== $v"
  timeout -k 10 300 python tools/mode_probe.py $ARGS 2>/dev/null
done
