#!/bin/bash
# runs tools/spill_probe.py with every library variant given (names as built by tools/variant_build.sh)
for v in "$@"; do
  if [ "$v" = default ]; then unset VOLXEL_HIP_LIB; else export VOLXEL_HIP_LIB=$PWD/volxel_amd/libvolxel_hip_$v.so; fi
  echo "=== $v"
  timeout -k 10 180 python tools/spill_probe.py 0 1 2 || echo "FAILED $v rc=$?"
done
