#!/bin/bash
# rocprofv3 counter pass (counters only) of `default` and `no_dda` on config 3 with the one-pixel-per-lane kernel and with
# the event-batched kernel (VX_PATHS_KERNEL=events) -> gpurun_out/prof/<tag>; printed: VALU instructions and busy figures
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r03_events}
OUT=gpurun_out/prof/$TAG; mkdir -p $OUT
for kern in generic events; do
  export VX_PATHS_KERNEL=$kern
  for mode in default no_dda; do
    timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE \
       --output-format csv -d $OUT/${kern}_$mode -- python3 tools/mode_profile.py $mode 1 > $OUT/${kern}_$mode.log 2>&1 || echo "failed $kern $mode"
    tail -n 1 $OUT/${kern}_$mode.log
  done
done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
for d in sorted(glob.glob(os.path.join(root, "*_*"))):
    if not os.path.isdir(d): continue
    best = defaultdict(dict)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "render_" in row["Kernel_Name"]:
                best[(int(row["Grid_Size"]), int(row["Dispatch_Id"]))][row["Counter_Name"]] = float(row["Counter_Value"])
                best[(int(row["Grid_Size"]), int(row["Dispatch_Id"]))]["_k"] = row["Kernel_Name"].split("(")[0] + " vgpr " + row["VGPR_Count"] + " scratch " + row["Scratch_Size"]
    if not best: continue
    v = best[max(best)]
    clk = v["GRBM_GUI_ACTIVE"] / 8
    print(f"{os.path.basename(d):16s} {v['_k']}: VALU insts {v['SQ_INSTS_VALU']:.3e}, useful lanes {v['SQ_THREAD_CYCLES_VALU'] / (64 * v['SQ_ACTIVE_INST_VALU']):.2f}, "
          f"VALU issue {v['SQ_INSTS_VALU'] * 2 / (clk * 1024):.2f}, waiting {v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES']:.2f}, LDS insts {v['SQ_INSTS_LDS']:.3e}, VMEM rd {v['SQ_INSTS_VMEM_RD']:.3e}, clocks {clk:.3e}")
PY
