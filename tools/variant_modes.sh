#!/bin/bash
# usage (on the GPU box): tools/variant_modes.sh variants/*.so -- tools/mode_bench.py with each library variant
cp volxel_amd/libvolxel_hip.so /tmp/base.so
for v in base "$@"; do
  if [ "$v" != base ]; then cp "$v" volxel_amd/libvolxel_hip.so; else cp /tmp/base.so volxel_amd/libvolxel_hip.so; fi
  echo "== $v"
  timeout -k 10 300 python tools/mode_bench.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  %-50s %.4f ms' % (d['case'], d['ms_per_frame']))"
done
cp /tmp/base.so volxel_amd/libvolxel_hip.so
