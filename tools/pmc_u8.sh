#!/bin/bash
# HBM bytes and instruction counts of the LDS-window kernel on brickf32 against bricku8 (config-5 workload by default):
# kernel trace + three counter passes per layout (counters only, one group per run) -> gpurun_out/prof/<tag>_l<layout>/summary.txt
# usage (through gpurun): bash tools/pmc_u8.sh <tag> [bench args...]
set -u
TAG=${1:-u8}; shift || true
if [ $# -eq 0 ]; then set -- --volume 1024 --width 3840 --height 2160 --steps 4 --warmup 1; fi
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
LEAN="--no-cpu-baseline --no-skip-variant --no-mode-variants --no-side-measurements --no-cold"
for LAY in 2 4; do
  OUT=gpurun_out/prof/${TAG}_l$LAY
  mkdir -p "$OUT"
  ARGS="$* --layout $LAY"
  echo "bench.py $ARGS" > "$OUT/command.txt"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 bench.py $ARGS $LEAN > "$OUT/bench_kt.json" 2> "$OUT/bench_kt.err" || { tail -5 "$OUT/bench_kt.err"; exit 1; }
  pass() {
    local name=$1; shift
    timeout -k 10 400 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py $ARGS $LEAN > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err" || { echo "pass $name failed"; tail -5 "$OUT/bench_$name.err"; return 1; }
    echo "layout $LAY pass $name ok"
  }
  pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU &&
  pass fetch FETCH_SIZE &&
  pass write WRITE_SIZE
  python3 tools/pmc_summary.py "$OUT" render_dvr_lds > "$OUT/summary.txt"
  tail -12 "$OUT/summary.txt"
done
