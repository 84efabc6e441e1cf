#!/bin/bash
# What ct_create costs at a given volume size: kernel statistics of creating one handle (shadow volume, bricks, clearance).
#   gpurun -- 'bash tools/gpu_create_profile.sh [volume=512]'
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
N=${1:-512}; OUT=gpurun_out/create_$N; rm -rf "$OUT"; mkdir -p "$OUT"
cat > /tmp/create_one.py <<PY
import sys, time
sys.path.insert(0, "$GRAFT_REPO_ROOT")
import deepestscatter_amd as ds
tex = ds.make_procedural_cloud($N)
for i in range(2):
    t0 = time.perf_counter()
    tr = ds.CloudTracer(tex, width=64, height=64, mode=0)
    tr.synchronize()
    print("ct_create + sync: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
    tr.close()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 /tmp/create_one.py > "$OUT/log.txt" 2>&1 || { tail -5 "$OUT/log.txt"; exit 1; }
grep "ct_create" "$OUT/log.txt"
STATS=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)
cp "$STATS" "$OUT/kernel_stats.csv"; rm -rf "$OUT/trace"
python3 - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print("%-60s calls %4s avg %9.3f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
