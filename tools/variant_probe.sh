#!/bin/bash
# usage (on the GPU box): tools/variant_probe.sh variants/*.so -- runs bench.py with each library variant
cp volxel_amd/libvolxel_hip.so /tmp/base.so
for v in base "$@"; do
  if [ "$v" != base ]; then cp "$v" volxel_amd/libvolxel_hip.so; else cp /tmp/base.so volxel_amd/libvolxel_hip.so; fi
  echo -n "$v: "
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-skip-variant --no-mode-variants 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_kernel_ms'])"
done
cp /tmp/base.so volxel_amd/libvolxel_hip.so
