#!/bin/bash
# Per-XCD job queues over N image regions: speed AND L2 misses of the headline launch (bench.py with its live PMC passes).
#   REGIONS="8 16 64" tools/gpu_xcd_regions_pmc.sh <tag>
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
OUT=gpurun_out/${1:-r03s}
mkdir -p "$OUT"
for R in ${REGIONS:-8 16 64}; do
  CT_XCD_QUEUES=1 CT_XCD_REGIONS=$R timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-delta-leg --no-progressive-leg > "$OUT/bench_xcd_regions_$R.json" 2> /dev/null || { echo "regions $R FAILED"; continue; }
  python3 - "$OUT/bench_xcd_regions_$R.json" $R <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r = d["roofline"]
print("regions %s: %.1f Msamples/s, launch %.2f ms, L2 misses per launch %.3e, hit rate %.3f" % (sys.argv[2], d["value"], r["avg_launch_ms"], r.get("tcc_miss_per_full_launch") or 0, r.get("l2_hit_rate") or 0))
PY
done
