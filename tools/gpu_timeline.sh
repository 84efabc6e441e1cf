#!/bin/bash
# Kernel timeline of a bench run (rocprofv3 --kernel-trace): start offsets, durations, gaps.
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out/timeline
rm -rf "$OUT"; mkdir -p "$OUT"
for i in ${RUNS:-1 2 3}; do
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/r$i" -- python3 bench.py --no-cpu-baseline > "$OUT/r$i.log" 2>&1 || { echo "trace failed"; tail -3 "$OUT/r$i.log"; exit 1; }
  python3 - "$OUT/r$i" <<'PY'
import csv, glob, sys
rows=[r for f in glob.glob(sys.argv[1]+"/**/*kernel_trace.csv", recursive=True) for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
big=[i for i,r in enumerate(rows) if "render_persistent" in r["Kernel_Name"]]
t0=int(rows[big[0]]["Start_Timestamp"])
prev_end=None
for i,r in enumerate(rows):
    if i < big[0]-1: continue
    s=(int(r["Start_Timestamp"])-t0)/1e6; e=(int(r["End_Timestamp"])-t0)/1e6
    gap = (s-prev_end) if prev_end is not None else 0
    if e-s > 0.3 or gap > 1.0:
        print("  %9.2f ms  dur %8.2f  gap_before %7.2f  %s" % (s, e-s, gap, r["Kernel_Name"][:60]))
    prev_end=e
PY
  grep -o '"ms_per_step": [0-9.]*' "$OUT/r$i.log"
done
