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
print(f"# environment map, bounces 1, 32 frames per launch (tools/modes_pmc.sh, tools/mode_profile.py); values of the")
print(f"# largest launch; ms per frame from the library's HIP events in the same runs")
json_out = None
if "--json" in sys.argv:
    i = sys.argv.index("--json"); json_out = sys.argv[i + 1]; src_name = sys.argv[i + 2]
    import json
    db = {"_note": "lane utilisation of the reference render modes on the bench scene (config 3: 512^3, 1920x1080, bounces 1, default "
                   "environment map): SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) of the rocprofv3 PMC passes of tools/modes_pmc.sh "
                   "(32 frames per launch: the running mean is applied in the kernel, 2 pixels x 32 frames per wave); bench.py copies these into config.other_modes when the scene matches (source named per entry)"}
    try:
        old_db = json.load(open(json_out))
    except Exception:
        old_db = {}
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
    if json_out and "SQ_THREAD_CYCLES_VALU" in vals and "SQ_ACTIVE_INST_VALU" in vals:
        e = dict(old_db.get(mode, {}))          # keeps the L1 / L2 / HBM figures of tools/modes_l2.sh, if any
        e.update({"width": 1920, "height": 1080, "volume": 512,
                  "lane_utilisation": round(vals["SQ_THREAD_CYCLES_VALU"] / (64 * vals["SQ_ACTIVE_INST_VALU"]), 3),
                  "valu_issue_of_clocks": (round(vals["SQ_INSTS_VALU"] * 2.0 / (vals["GRBM_GUI_ACTIVE"] / 8.0 * 1024), 3)
                                           if "GRBM_GUI_ACTIVE" in vals else None),
                  "kernel": km[0].replace("kernel void ", ""), "source": src_name})
        db[mode] = e
if json_out:
    json.dump(db, open(json_out, "w"), indent=1)
