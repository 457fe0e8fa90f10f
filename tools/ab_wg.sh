#!/bin/bash
# usage (GPU box): tools/ab_wg.sh <variant|cur> ... -- the shared-window DVR kernel (VX_DVR_WG=1) per library variant: lane utilisation,
# wave-windows per frame, ms per frame (tools/fpl_probe.py), against the shipped wave-private kernel (VX_DVR_WG=0) of the in-tree build
echo "=== cur, wave-private windows"; VX_DVR_WG=0 timeout -k 10 200 python tools/fpl_probe.py 2>&1 | grep "dvr fpl 32"
for v in "$@"; do
  if [ $v = cur ]; then unset VOLXEL_HIP_LIB; else export VOLXEL_HIP_LIB=$PWD/volxel_amd/libvolxel_hip_$v.so; fi
  echo "=== $v, shared window"; VX_DVR_WG=1 timeout -k 10 200 python tools/fpl_probe.py 2>&1 | grep "dvr fpl 32"
done
unset VOLXEL_HIP_LIB
echo "=== cur, wave-private windows (again)"; VX_DVR_WG=0 timeout -k 10 200 python tools/fpl_probe.py 2>&1 | grep "dvr fpl 32"
