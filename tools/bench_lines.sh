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
run "the driver's command: 20 steps of 32 accumulation frames, one launch each" --gpus 1 --steps 20 --warmup 5 &&
run "64 frames per launch" --steps 20 --warmup 5 --frames-per-launch 64 $LEAN &&
run "one frame per step, the step of rounds 1-2: a single 20-frame launch is timed" --steps 20 --warmup 5 --frames-per-step 1 $LEAN &&
run "the same after 150 ms of device preconditioning (value) and before it (value_cold)" --steps 20 --warmup 5 --frames-per-step 1 --precondition-ms 150 $LEAN &&
run "cellquad gather kernel" --gpus 1 --steps 20 --warmup 5 --layout 1 $LEAN &&
run "bricku8: 8-bit bricks decoded while a window is staged (opt-in)" --gpus 1 --steps 20 --warmup 5 --layout 4 $LEAN &&
run "BASELINE config 5 workload, whole frame on one GPU" --volume 1024 --width 3840 --height 2160 --steps 20 --warmup 5 --no-cpu-baseline --no-mode-variants &&
run "config 5 workload, cellquad gather kernel" --volume 1024 --width 3840 --height 2160 --steps 10 --warmup 2 --layout 1 $LEAN &&
run "RCCL gather path with one rank" --gpus 1 --steps 20 --warmup 5 --force-gather $LEAN
