#!/bin/bash
# rocprofv3 evidence for BASELINE config 4 (DVR + gradient + Blinn-Phong, render_dvr_lds<16,true,.>): kernel trace with
# stats, then counter passes (counters only) -> gpurun_out/prof/<tag>; summary: python tools/phong_summary.py <dir>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r02_phong}
OUT=gpurun_out/prof/$TAG; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 tools/mode_profile.py dvr_phong 1 > $OUT/kt.log 2>&1 || echo "failed kt"
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE \
   --output-format csv -d $OUT/sq -- python3 tools/mode_profile.py dvr_phong 1 > $OUT/sq.log 2>&1 || echo "failed sq"
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY \
   --output-format csv -d $OUT/mem -- python3 tools/mode_profile.py dvr_phong 1 > $OUT/mem.log 2>&1 || echo "failed mem"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 tools/mode_profile.py dvr_phong 1 > $OUT/fetch.log 2>&1 || echo "failed fetch"
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 tools/mode_profile.py dvr_phong 1 > $OUT/write.log 2>&1 || echo "failed write"
grep -h "ms/frame" $OUT/*.log
