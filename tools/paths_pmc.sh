#!/bin/bash
# lane utilisation of the reference modes: one-pixel-per-lane kernel vs the LDS re-packed path kernel (counters only)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof/r02_paths; mkdir -p $OUT
for kern in generic packed; do
  for mode in default no_dda; do
    VX_PATHS_KERNEL=$kern timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE \
       --output-format csv -d $OUT/${kern}_$mode -- python3 tools/mode_profile.py $mode 1 > $OUT/${kern}_$mode.log 2>&1 || echo "failed $kern $mode"
    tail -1 $OUT/${kern}_$mode.log
  done
done
