#!/bin/bash
# L2 hit/miss of the estimator kernel for two settings of an env knob (A/B), one PMC pass each.
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out/pmc_l2
rm -rf "$OUT"; mkdir -p "$OUT"
for V in ${VALUES:-1 0}; do
  export ${KNOB:-CT_XCD_QUEUES}=$V
  timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d "$OUT/v$V" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/v$V.log" 2>&1 || { echo "pmc failed"; tail -5 "$OUT/v$V.log"; exit 1; }
  echo "== ${KNOB:-CT_XCD_QUEUES}=$V"
  python3 - "$OUT/v$V" <<'PY'
import csv, sys, glob
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "render_persistent" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("  %-22s %s" % (k, ["%.4g" % x for x in v]))
if "TCC_HIT_sum" in acc:
    for h, m in zip(acc["TCC_HIT_sum"], acc["TCC_MISS_sum"]):
        print("  hit rate %.3f" % (h / (h + m)))
PY
done
