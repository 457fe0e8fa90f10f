#!/usr/bin/env python3
"""Summary of tools/phong_pmc.sh: python tools/phong_summary.py gpurun_out/prof/<tag> > profiles/<name>.txt"""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
KERN = "render_dvr_lds<16, true"
print("# rocprofv3 evidence for BASELINE config 4 (config 3 + central-difference gradient + Blinn-Phong): kernel")
print("# vx::render_dvr_lds<16, true, false>, 1080p, jitter on, 16 frames per launch (tools/phong_pmc.sh, tools/mode_profile.py)")
for f in glob.glob(os.path.join(root, "*.log")):
    for ln in open(f):
        if ln.startswith("dvr_phong"):
            print("# " + os.path.basename(f) + ": " + ln.strip())
for f in glob.glob(os.path.join(root, "kt", "**", "*kernel_stats.csv"), recursive=True):
    print("\n## kernel-trace --stats")
    print(open(f).read().strip())
by = defaultdict(list)
for f in glob.glob(os.path.join(root, "kt", "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if KERN in row.get("Kernel_Name", ""):
            by[int(row["Grid_Size_X"]) if "Grid_Size_X" in row else int(row.get("Grid_Size", 0))].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
print("\n## kernel trace of the Phong kernel by launch size (threads): calls, mean ms (16-frame launches = the largest grid)")
for g, v in sorted(by.items()):
    print(f"grid {g:10d}: {len(v):3d} calls, mean {sum(v)/len(v):8.4f} ms = {sum(v)/len(v)/max(1, g // min(by)):.4f} ms per frame")
vals = {}
for grp in ("sq", "mem", "fetch", "write"):
    best = defaultdict(dict)
    for f in glob.glob(os.path.join(root, grp, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if KERN in row["Kernel_Name"]:
                best[(int(row["Grid_Size"]), int(row["Dispatch_Id"]))][row["Counter_Name"]] = float(row["Counter_Value"])
    if best:
        vals.update(best[max(best)])
print("\n## PMC counters of one 16-frame launch")
for k in sorted(vals):
    print(f"{k:36s} {vals[k]:18.1f}")
d = []
if "SQ_THREAD_CYCLES_VALU" in vals:
    d.append(f"lane slots of issued VALU instructions doing work {vals['SQ_THREAD_CYCLES_VALU'] / (64 * vals['SQ_ACTIVE_INST_VALU']):.2f}")
if "SQ_INSTS_VALU" in vals and "GRBM_GUI_ACTIVE" in vals:
    d.append(f"VALU issue {vals['SQ_INSTS_VALU'] * 2.0 / (vals['GRBM_GUI_ACTIVE'] / 8.0 * 1024):.2f} of the kernel's clocks (2 clk per wave64 instruction, 1024 SIMDs)")
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    d.append(f"HBM traffic 2 x FETCH_SIZE + WRITE_SIZE = {(2 * vals['FETCH_SIZE'] + vals['WRITE_SIZE']) * 1024 / 1e9:.2f} GB per 16-frame launch")
print("# derived: " + "; ".join(d))
