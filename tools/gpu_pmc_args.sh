#!/bin/bash
# L2 / fetch counters of the estimator kernel for a given bench command line (ARGS).
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out/pmc_args
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/a" -- python3 bench.py --no-cpu-baseline ${ARGS:-} > "$OUT/a.log" 2>&1 || { echo "pmc failed"; tail -5 "$OUT/a.log"; exit 1; }
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/b" -- python3 bench.py --no-cpu-baseline ${ARGS:-} > "$OUT/b.log" 2>&1 || { echo "pmc failed"; tail -5 "$OUT/b.log"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, sys, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "render_persistent" in r["Kernel_Name"]:
            acc[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
for d in sorted(acc):
    c = acc[d]
    h, m, f = c.get("TCC_HIT_sum", 0), c.get("TCC_MISS_sum", 0), c.get("FETCH_SIZE", 0)
    print("dispatch %d: hit %.3g miss %.3g hit-rate %.3f  miss x128B %.1f GB  FETCH_SIZE x2 %.1f GB" % (d, h, m, h / max(h + m, 1), m * 128 / 1e9, f * 2 * 1024 / 1e9))
PY
grep -o '"value": [0-9.]*\|"avg_launch_ms": [0-9.]*' "$OUT/a.log" | tr '\n' ' '; echo
