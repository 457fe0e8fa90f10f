#!/usr/bin/env python3
"""Summarise the rocprofv3 output of tools/pmc_profile.sh into one text file for profiles/.
usage: tools/pmc_summary.py gpurun_out/prof/<tag> <kernel-substring> [profiles/<name>.txt] > profiles/<name>.txt
With the third argument the HBM traffic per launch (2 x FETCH_SIZE + WRITE_SIZE, the gfx950 rule of
MI355X_MICROARCH.md section HBM) is also recorded in profiles/traffic.json for bench.py's roofline.traffic,
keyed by the launch shape the traced bench line reports."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root, kern = sys.argv[1], sys.argv[2]
print(f"# rocprofv3 summary for kernels matching '{kern}' under {root}")
cmd = os.path.join(root, "command.txt")
if os.path.exists(cmd):
    print("# command: python3 " + open(cmd).read().strip())


def newest(pattern):
    fs = glob.glob(pattern, recursive=True)
    return [max(fs, key=os.path.getmtime)] if fs else []


for f in newest(os.path.join(root, "kt", "**", "*kernel_stats.csv")):
    print("\n## kernel-trace --stats (" + os.path.relpath(f, root) + ")")
    print(open(f).read().strip())
timed_grid = None
by = defaultdict(list)
for f in newest(os.path.join(root, "kt", "**", "*kernel_trace.csv")):
    for row in csv.DictReader(open(f)):
        if kern in row.get("Kernel_Name", ""):
            by[int(row["Grid_Size_X"]) if "Grid_Size_X" in row else int(row.get("Grid_Size", 0))].append(
                (int(row["Start_Timestamp"]), (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6))
    by = defaultdict(list, {g: [d for _, d in sorted(v)] for g, v in by.items()})   # durations in launch order
    print("\n## kernel trace of '" + kern + "' by launch size (threads in x): calls, mean ms, min, max")
    for g, v in sorted(by.items()):
        print(f"grid {g:10d}: {len(v):3d} calls, mean {sum(v)/len(v):8.4f} ms, min {min(v):8.4f}, max {max(v):8.4f}")
line = None
bj = os.path.join(root, "bench_kt.json")
if os.path.exists(bj) and os.path.getsize(bj):
    txt = open(bj).read().strip()
    print("\n## bench.py line of the traced run\n" + txt)
    try:
        line = json.loads(txt.splitlines()[-1])
    except Exception:
        line = None
# the launches bench.py timed have frames_per_launch x (blocks of one frame) x 256 threads
fpl = line["roofline"]["frames_per_launch"] if line else None
if line and by:
    g1 = min(by)
    gt = g1 * fpl if g1 * fpl in by else max(by)
    v = by[gt]
    # launches of this shape in order: [cold pass], warm-up launches, the timed ones, then side measurements
    n_timed = int(line["roofline"]["launches"])
    fps = int(line["config"].get("frames_per_step", 1))
    n_warm = -(-max(int(line["warmup"]) * fps, 2) // fpl)
    # the first warm-up call of a fresh context renders its first two frames one by one (they build the launch order)
    # and the rest of its frames in a smaller launch: one launch fewer of the full size
    n_warm_full = max(n_warm - 1, 0) if fpl > 2 else n_warm
    first = n_warm_full + ((n_warm + n_timed) if line.get("value_cold") else 0)
    tv = v[first:first + n_timed] if len(v) >= first + n_timed else v
    print("\n## the timed launches")
    print(f"The --stats average above mixes every launch of the kernel: the warm-up steps (the first one runs on a chip at")
    print(f"its idle clocks), other frames per launch, the single-frame side measurement.  In launch order the {len(v)} launches of")
    print(f"{fpl} frames = grid {gt} are {n_warm_full} warm-up + {n_timed} timed + the rest; the {len(tv)} timed ones: kernel trace mean "
          f"{sum(tv)/len(tv):.4f} ms (min {min(tv):.4f}, max {max(tv):.4f}); bench.py (HIP events, same run) "
          f"roofline.avg_kernel_ms {line['roofline']['avg_kernel_ms']}")
    by[gt] = tv
print("\n## PMC counters: per-dispatch mean over the timed dispatches of the kernel")
means = {}
for d in sorted(glob.glob(os.path.join(root, "*"))):
    name = os.path.basename(d)
    if name == "kt" or not os.path.isdir(d):
        continue
    acc = defaultdict(list)
    gsel = 0
    for f in newest(os.path.join(d, "**", "*counter_collection.csv")):
        rows = [r for r in csv.DictReader(open(f)) if kern in r.get("Kernel_Name", "")]
        grids = sorted({int(r["Grid_Size"]) for r in rows})
        if not grids:
            continue
        # the single-frame launches (launch-order refresh, first frames) have the smallest grid; the timed
        # launches cover frames_per_launch frames
        gsel = grids[0] * fpl if (fpl and grids[0] * fpl in grids) else grids[-1]
        for row in rows:
            if int(row["Grid_Size"]) == gsel:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in sorted(acc.items()):
        means[k] = sum(v) / len(v)
        print(f"{name:6s} {k:40s} mean {sum(v)/len(v):18.1f}  (n={len(v)}, grid {gsel})")

if len(sys.argv) > 3 and line and "FETCH_SIZE" in means and "WRITE_SIZE" in means:
    cfg = line["config"]
    w, h = [int(x) for x in cfg["workload"].split(", ")[1].split("x")]
    vol = int(cfg["workload"].split(": ")[1].split("^")[0])
    entry = {"width": w, "height": h, "volume": vol, "frames_per_launch": fpl, "dvr_jitter": cfg["dvr_jitter"],
             "fetch_size_kb": means["FETCH_SIZE"], "write_size_kb": means["WRITE_SIZE"],
             "hbm_bytes_per_launch": int((2 * means["FETCH_SIZE"] + means["WRITE_SIZE"]) * 1024),
             "traced_kernel_ms": (sum(by[gsel]) / len(by[gsel])) if by.get(gsel) else None,
             "effective_clock_mhz": (means["GRBM_GUI_ACTIVE"] / 8.0 / (sum(by[gsel]) / len(by[gsel]) * 1e-3) / 1e6)
                                    if (by.get(gsel) and means.get("GRBM_GUI_ACTIVE")) else None,
             "sq_insts_valu_per_launch": means.get("SQ_INSTS_VALU"), "sq_insts_salu_per_launch": means.get("SQ_INSTS_SALU"),
             "sq_insts_lds_per_launch": means.get("SQ_INSTS_LDS"), "sq_insts_vmem_rd_per_launch": means.get("SQ_INSTS_VMEM_RD"),
             "source": sys.argv[3] + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of the same command; "
                                     "2 x FETCH_SIZE per the MI355X_MICROARCH.md HBM note)"}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "traffic.json")
    try:
        db = json.load(open(path))
        if not isinstance(db.get(cfg["layout"], []), list):
            db = {}
    except Exception:
        db = {}
    key = ("width", "height", "volume", "frames_per_launch", "dvr_jitter")
    rows = [e for e in db.get(cfg["layout"], []) if any(e.get(k) != entry[k] for k in key)]
    db[cfg["layout"]] = rows + [entry]
    json.dump(db, open(path, "w"), indent=1)
    print(f"\n## traffic.json: 2 x {means['FETCH_SIZE']:.0f} KB + {means['WRITE_SIZE']:.0f} KB per {fpl}-frame launch")
