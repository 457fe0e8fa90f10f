#!/bin/bash
# rocprofv3 counter passes of the L2 / fabric side of the reference render modes on config 3 -> gpurun_out/prof/<tag>
# (FETCH_SIZE, TCC hits / misses / requests, TCP -> TCC requests and their latency); counters only, one pass each
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-modes_l2}; shift
OUT=gpurun_out/prof/$TAG; mkdir -p $OUT
for mode in "$@"; do
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$mode -- python3 tools/mode_profile.py $mode 1 > $OUT/fetch_$mode.log 2>&1 || echo "failed fetch $mode"
  timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d $OUT/tcc_$mode -- python3 tools/mode_profile.py $mode 1 > $OUT/tcc_$mode.log 2>&1 || echo "failed tcc $mode"
  timeout -k 10 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/tcp_$mode -- python3 tools/mode_profile.py $mode 1 > $OUT/tcp_$mode.log 2>&1 || echo "failed tcp $mode"
  tail -n 1 $OUT/tcp_$mode.log
  python3 - "$OUT" "$mode" <<'PY'
import csv, glob, sys
out, mode = sys.argv[1], sys.argv[2]
for kind in ("fetch", "tcc", "tcp"):
    best = {}
    for f in glob.glob(f"{out}/{kind}_{mode}/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        gen = [r for r in rows if "render_generic" in r["Kernel_Name"]]
        if not gen: continue
        big = max(int(r["Grid_Size"]) for r in gen)
        disp = {}
        for r in gen:
            if int(r["Grid_Size"]) == big:
                disp.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
        last = disp[sorted(disp, key=int)[-1]]
        best.update(last)
    for k, v in sorted(best.items()): print(f"{mode:9s} {k:34s} {v:.0f}")
PY
done
