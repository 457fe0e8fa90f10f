#!/bin/bash
# rocprofv3 counter passes (counters only) of the three reference render modes on config 3 -> gpurun_out/prof/<tag>;
# summary: python tools/modes_summary.py gpurun_out/prof/<tag> > profiles/<name>.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r02_modes}
OUT=gpurun_out/prof/$TAG; mkdir -p $OUT
for mode in default no_dda raymarch; do
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE \
     --output-format csv -d $OUT/sq_$mode -- python3 tools/mode_profile.py $mode 1 > $OUT/sq_$mode.log 2>&1 || echo "failed sq $mode"
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TD_TD_BUSY_sum SQ_WAIT_INST_ANY \
     --output-format csv -d $OUT/mem_$mode -- python3 tools/mode_profile.py $mode 1 > $OUT/mem_$mode.log 2>&1 || echo "failed mem $mode"
  tail -n 1 $OUT/sq_$mode.log
done
