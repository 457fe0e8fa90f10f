#!/usr/bin/env python3
"""Summarise the rocprofv3 output of tools/pmc_profile.sh into one text file for profiles/.
usage: tools/pmc_summary.py gpurun_out/prof/<tag> <kernel-substring> > profiles/<name>.txt"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root, kern = sys.argv[1], sys.argv[2]
print(f"# rocprofv3 summary for kernels matching '{kern}' under {root}")
def newest(pattern):
    fs = glob.glob(pattern, recursive=True)
    return [max(fs, key=os.path.getmtime)] if fs else []

for f in newest(os.path.join(root, "kt", "**", "*kernel_stats.csv")):
    print("\n## kernel-trace --stats (" + os.path.relpath(f, root) + ")")
    print(open(f).read().strip())
for f in newest(os.path.join(root, "kt", "**", "*kernel_trace.csv")):
    by = defaultdict(list)
    for row in csv.DictReader(open(f)):
        if kern in row.get("Kernel_Name", ""):
            by[int(row["Grid_Size_X"]) if "Grid_Size_X" in row else int(row.get("Grid_Size", 0))].append(
                (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
    print("\n## kernel trace of '" + kern + "' by launch size (threads in x): calls, mean ms, min, max")
    for g, v in sorted(by.items()):
        print(f"grid {g:10d}: {len(v):3d} calls, mean {sum(v)/len(v):8.4f} ms, min {min(v):8.4f}, max {max(v):8.4f}")
bj = os.path.join(root, "bench_kt.json")
if os.path.exists(bj) and os.path.getsize(bj):
    print("\n## bench.py line of the traced run\n" + open(bj).read().strip())
print("\n## PMC counters: per-dispatch mean over the timed dispatches of the kernel")
for d in sorted(glob.glob(os.path.join(root, "*"))):
    name = os.path.basename(d)
    if name == "kt" or not os.path.isdir(d):
        continue
    acc = defaultdict(list)
    for f in newest(os.path.join(d, "**", "*counter_collection.csv")):
        rows = [r for r in csv.DictReader(open(f)) if kern in r.get("Kernel_Name", "")]
        # only the full-size launches (bench.py default: 64 frames per launch); the first frames of a run are
        # launched one by one while the launch order is being built
        gmax = max((int(r["Grid_Size"]) for r in rows), default=0)
        for row in rows:
            if int(row["Grid_Size"]) == gmax:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(f"{name:6s} {k:40s} mean {sum(v)/len(v):18.1f}  (n={len(v)}, grid {gmax})")

# ---- HBM traffic per launch (gfx950 rule: 2 x FETCH_SIZE + WRITE_SIZE), for bench.py's roofline.traffic
if len(sys.argv) > 3:
    import re
    vals = {}
    for d in ("fetch", "write"):
        acc = []
        for f in newest(os.path.join(root, d, "**", "*counter_collection.csv")):
            rows = [r for r in csv.DictReader(open(f)) if kern in r.get("Kernel_Name", "")
                    and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE")]
            gmax = max((int(r["Grid_Size"]) for r in rows), default=0)
            acc += [float(r["Counter_Value"]) for r in rows if int(r["Grid_Size"]) == gmax]
        vals[d] = sum(acc) / len(acc) if acc else None
    if vals.get("fetch") and vals.get("write"):
        out = {"cellquad": {"width": 1920, "height": 1080, "volume": 512, "fetch_size_kb": vals["fetch"],
                            "write_size_kb": vals["write"],
                            "hbm_bytes_per_launch": int((2 * vals["fetch"] + vals["write"]) * 1024),
                            "source": sys.argv[3], "frames_per_launch": int(sys.argv[4]) if len(sys.argv) > 4 else 64}}
        json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "traffic.json"), "w"), indent=1)
        print(f"\n## traffic.json: 2 x {vals['fetch']:.0f} KB + {vals['write']:.0f} KB per launch")
