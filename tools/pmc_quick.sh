#!/bin/bash
# The two SQ counter passes of tools/pmc_profile.sh only (instruction counts, waits, clock), no kernel trace of other sizes:
# usage: tools/pmc_quick.sh <tag> [bench args...]
set -u
TAG=${1:-quick}; shift || true
if [ $# -eq 0 ]; then set -- --gpus 1 --steps 20 --warmup 5; fi
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof/$TAG
mkdir -p "$OUT"
echo "bench.py $*" > "$OUT/command.txt"
LEAN="--no-cpu-baseline --no-skip-variant --no-mode-variants --no-side-measurements --no-cold"
ARGS="$*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 bench.py $ARGS $LEAN > "$OUT/bench_kt.json" 2> "$OUT/bench_kt.err" || { tail -5 "$OUT/bench_kt.err"; exit 1; }
pass() {
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py $ARGS $LEAN > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err" || { echo "pass $name failed"; tail -5 "$OUT/bench_$name.err"; return 1; }
  echo "pass $name ok"
}
pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU &&
pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE
python3 tools/pmc_summary.py "$OUT" render_dvr_lds > "$OUT/summary.txt"
tail -40 "$OUT/summary.txt"
