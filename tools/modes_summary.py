#!/usr/bin/env python3
"""Summary of tools/modes_pmc.sh: per mode the counters of the largest render launch and what follows from them.
usage: python tools/modes_summary.py gpurun_out/prof/<tag> > profiles/<name>.txt"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
print(f"# rocprofv3 --pmc (two passes per mode, counters only) of the reference render modes on config 3, 1080p, default")
print(f"# environment map, bounces 1, 16 frames per launch (tools/modes_pmc.sh, tools/mode_profile.py); values of the")
print(f"# largest launch; ms per frame from the library's HIP events in the same runs")
for mode in ("default", "no_dda", "raymarch"):
    vals, meta = {}, {}
    for grp in ("sq", "mem"):
        best = defaultdict(dict)
        for f in glob.glob(os.path.join(root, f"{grp}_{mode}", "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if "render_" not in row["Kernel_Name"]:
                    continue
                best[(int(row["Grid_Size"]), int(row["Dispatch_Id"]))][row["Counter_Name"]] = float(row["Counter_Value"])
                meta[(int(row["Grid_Size"]), int(row["Dispatch_Id"]))] = (row["Kernel_Name"].split("(")[0], row["VGPR_Count"], row["Scratch_Size"],
                                                                         int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        if best:
            k = max(best)
            vals.update(best[k])
            if grp == "sq":
                km = meta[k]
                grid = k[0]
    log = os.path.join(root, f"sq_{mode}.log")
    line = [ln.strip() for ln in open(log) if ln.startswith(mode)] if os.path.exists(log) else []
    print(f"\n## {mode}   {line[-1] if line else ''}")
    if not vals:
        print("(no counters)")
        continue
    print(f"kernel {km[0]}, grid {grid} threads, VGPR_Count {km[1]}, scratch {km[2]} B, {km[3] / 1e6:.2f} ms under the counter pass")
    for name in sorted(vals):
        print(f"{name:36s} {vals[name]:18.1f}")
    d = []
    if "SQ_THREAD_CYCLES_VALU" in vals and "SQ_ACTIVE_INST_VALU" in vals:
        d.append(f"lane slots of issued VALU instructions doing work {vals['SQ_THREAD_CYCLES_VALU'] / (64 * vals['SQ_ACTIVE_INST_VALU'] * 4) * 4:.2f}"
                 if False else f"lane slots of issued VALU instructions doing work {vals['SQ_THREAD_CYCLES_VALU'] / (64 * vals['SQ_ACTIVE_INST_VALU']):.2f}")
    if "SQ_WAIT_ANY" in vals and "SQ_WAVE_CYCLES" in vals:
        d.append(f"wave cycles waiting {vals['SQ_WAIT_ANY'] / vals['SQ_WAVE_CYCLES']:.2f}")
    if "SQ_INSTS_VALU" in vals and "GRBM_GUI_ACTIVE" in vals:
        # GRBM_GUI_ACTIVE sums the 8 XCDs; 1024 SIMDs, one wave64 VALU instruction per 2 clocks each
        clk = vals["GRBM_GUI_ACTIVE"] / 8.0
        d.append(f"VALU issue {vals['SQ_INSTS_VALU'] * 2.0 / (clk * 1024):.2f} of the kernel's clocks (2 clk per wave64 instruction, 1024 SIMDs)")
    if "TD_TD_BUSY_sum" in vals and "GRBM_GUI_ACTIVE" in vals:
        d.append(f"TD busy {vals['TD_TD_BUSY_sum'] / (vals['GRBM_GUI_ACTIVE'] / 8.0 * 256):.2f}")
    print("# derived: " + "; ".join(d))
