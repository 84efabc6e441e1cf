#!/bin/bash
# Texture-addresser / vector-L1 counters of the estimator kernels (is the kernel bound by the number of vector memory
# instructions?): busy shares of TA and TCP, stalls between them, L1 accesses and read wavefronts, for MARCH and DELTA.
#   tools/gpu_pmc_ta.sh [out dir under gpurun_out]
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out/${1:-pmc_ta}; rm -rf "$OUT"; mkdir -p "$OUT"
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-delta-leg --no-progressive-leg --no-pmc-traffic"
PASSES=("TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum")
for EST in 0 1; do
  i=0
  for P in "${PASSES[@]}"; do
    i=$((i + 1))
    timeout -k 10 300 rocprofv3 --pmc $P --output-format csv -d "$OUT/e${EST}_p$i" -- python3 bench.py $ARGS --estimator $EST > "$OUT/e${EST}_p$i.log" 2>&1 || { echo "pass $i of estimator $EST failed"; tail -3 "$OUT/e${EST}_p$i.log"; }
  done
done
python3 - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import csv, sys, glob
from collections import defaultdict
for est, name in ((0, "render_persistent_kernel"), (1, "render_delta_kernel")):
    acc = defaultdict(lambda: defaultdict(float))
    for f in glob.glob(sys.argv[1] + f"/e{est}_p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if name in r["Kernel_Name"]:
                acc[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    print(name)
    for c in sorted(acc):
        v = acc[c]
        big = max(v.values())
        print("  %-40s %s" % (c, " ".join("%.4g" % v[d] for d in sorted(v) if v[d] > 0.2 * big)))
PY
