#!/bin/bash
# Collects rocprofv3 evidence for bench.py on the GPU box (run through gpurun):
#   kernel-trace stats  -> gpurun_out/prof/<tag>/kt      (the command exactly as given)
#   PMC passes (counters only, one group per run; never combined with tracing domains), the same command
#   plus the switches that drop the side measurements, so that every pass launches the same timed kernels
# usage: tools/pmc_profile.sh <tag> [bench args...]      default args = the driver's: --gpus 1 --steps 20 --warmup 5
set -u
TAG=${1:-r02}; shift || true
if [ $# -eq 0 ]; then set -- --gpus 1 --steps 20 --warmup 5; fi
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof/$TAG
mkdir -p "$OUT"
echo "bench.py $*" > "$OUT/command.txt"
LEAN="--no-cpu-baseline --no-skip-variant --no-mode-variants --no-side-measurements --no-cold"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 bench.py "$@" --no-cpu-baseline --no-cold > "$OUT/bench_kt.json" 2> "$OUT/bench_kt.err" || { tail -5 "$OUT/bench_kt.err"; exit 1; }
pass() {  # name counters...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py $ARGS $LEAN > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err" || { echo "pass $name failed"; tail -5 "$OUT/bench_$name.err"; return 1; }
  echo "pass $name ok"
}
ARGS="$*"
pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU &&
pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE &&
pass tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum &&
pass tcp2 TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TD_TD_BUSY_sum TCP_TCC_READ_REQ_LATENCY_sum &&
pass tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum &&
pass fetch FETCH_SIZE &&
pass write WRITE_SIZE
find "$OUT" -name "*.csv" | wc -l
