#!/bin/bash
# Kernel trace of the display cadence with render-ahead: which kernels fill the time between the estimator launches.
#   tools/gpu_trace_progressive.sh <tag> [ahead]
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r03q}/trace_progressive
AHEAD=${2:-80}
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 tools/progressive_bench.py --spp 10 --updates 96 --ahead $AHEAD --reference-spp 0 > "$OUT/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
tail -3 "$OUT/trace.log"
F=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)
cat "$F" | cut -c1-200
T=$(find "$OUT/trace" -name "*kernel_trace.csv" | head -1)
python3 - "$T" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 40 dispatches: name, duration, gap to the previous one's end
prev = None
out = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out.append((r["Kernel_Name"][:40], (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0))
    prev = e
for name, dur, gap in out[-60:]:
    print(f"{name:40s} {dur:10.1f} us   gap before {gap:8.1f} us")
PY
