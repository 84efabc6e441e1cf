#!/bin/bash
# The dataset pipeline (`cloudtrace collect` = Tasks::collect, Tasks.cpp:114-155) on one GPU: where one scene setup's time
# goes, what K setups in flight buy (`--jobs K`), and the kernel statistics of one setup.
#   gpurun -- 'bash tools/gpu_collect_bench.sh <tag> [volume=256] [setups=8]'
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
TAG=${1:-collect}; N=${2:-256}; SETUPS=${3:-8}
OUT=gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
CLI=deepestscatter_amd/host/cloudtrace
LIST="$OUT/setups.txt"
for i in $(seq 1 $SETUPS); do
  L=Side; [ $((i % 2)) -eq 0 ] && L=Back
  echo "procedural:$N:$((1000 + i)) $L 7000" >> "$LIST"
done
for K in ${JOBS:-1 2 4 8}; do
  timeout -k 10 900 $CLI collect @"$LIST" --jobs $K --out "$OUT/tables_k$K" ${COLLECT_EXTRA:-} > "$OUT/collect_k$K.log" 2>&1 || { echo "collect --jobs $K failed"; tail -5 "$OUT/collect_k$K.log"; exit 1; }
  grep collect_totals "$OUT/collect_k$K.log"
  for T in ScatterSample Result DisneyDescriptor SceneSetup; do cmp "$OUT/tables_k$K/$T.flat" "$OUT/tables_k1/$T.flat" || echo "TABLE $T DIFFERS at --jobs $K"; done
done
grep -h collect_timings "$OUT/collect_k1.log" | head -3
rm -rf "$OUT"/tables_k*
head -1 "$LIST" > "$OUT/one.txt"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $CLI collect @"$OUT/one.txt" --out "$OUT/tables_p" ${COLLECT_EXTRA:-} > "$OUT/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
rm -rf "$OUT/tables_p"
STATS=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)
cp "$STATS" "$OUT/kernel_stats_$N.csv"
python3 - "$OUT/kernel_stats_$N.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("%-60s calls %5s total %9.3f ms avg %9.3f ms" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
rm -rf "$OUT/trace"
