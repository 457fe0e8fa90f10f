#!/bin/bash
# usage (on the GPU box): tools/variant_sweep.sh name... -- bench.py (lean) with each library variant built by
# tools/variant_build.sh (`default` = the in-tree library); prints value, kernel ms per launch, lane utilisation
for v in "$@"; do
  if [ "$v" = default ]; then unset VOLXEL_HIP_LIB; else export VOLXEL_HIP_LIB=$PWD/volxel_amd/libvolxel_hip_$v.so; fi
  echo -n "$v: "
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-skip-variant --no-mode-variants --no-side-measurements ${BENCH_ARGS:-} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['avg_kernel_ms'], d['config'].get('lane_utilisation'))"
done
