#!/bin/bash
# Runs bench.py in the configurations quoted in DESIGN.md section 6 and writes the lines, each preceded by
# its command, to gpurun_out/bench_lines.txt (copied to profiles/rNN_bench_lines.txt by hand).
# usage (through gpurun): bash tools/bench_lines.sh "<one-line description of the build>"
set -u
OUT=gpurun_out/bench_lines.txt
mkdir -p gpurun_out
{
  echo "# bench.py lines, 1x MI355X (gpurun), ${1:-current build}."
  echo "# One JSON line per run, exactly as printed; the command precedes each line."
} > "$OUT"
run() {  # note, args...
  local note=$1; shift
  echo >> "$OUT"
  echo "## python bench.py $*   ($note)" >> "$OUT"
  timeout -k 10 400 python bench.py "$@" 2>> gpurun_out/bench_lines.err | grep '^{' >> "$OUT" || { echo "FAILED: $*"; return 1; }
  echo "ok: $*"
}
LEAN="--no-cpu-baseline --no-mode-variants --no-skip-variant"
run "the driver's command" --gpus 1 --steps 20 --warmup 5 &&
run "32 frames per launch: 4 launches" --steps 128 --warmup 5 --no-cpu-baseline &&
run "64 frames per launch" --steps 128 --warmup 5 --frames-per-launch 64 $LEAN &&
run "cellquad gather kernel" --gpus 1 --steps 20 --warmup 5 --layout 1 $LEAN &&
run "BASELINE config 5 workload, whole frame on one GPU" --volume 1024 --width 3840 --height 2160 --steps 20 --warmup 5 --no-cpu-baseline --no-mode-variants &&
run "config 5 workload, cellquad gather kernel" --volume 1024 --width 3840 --height 2160 --steps 20 --warmup 5 --layout 1 $LEAN &&
run "RCCL gather path with one rank" --gpus 1 --steps 20 --warmup 5 --force-gather $LEAN &&
run "the driver's command without device preconditioning: the chip still at its idle clocks" --gpus 1 --steps 20 --warmup 5 --precondition-ms 0 $LEAN
