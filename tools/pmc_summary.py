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
        for row in csv.DictReader(open(f)):
            if kern in row.get("Kernel_Name", ""):
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in sorted(acc.items()):
        v = v[2:] if len(v) > 4 else v      # drop warm-up dispatches
        print(f"{name:6s} {k:40s} mean {sum(v)/len(v):18.1f}  (n={len(v)})")
