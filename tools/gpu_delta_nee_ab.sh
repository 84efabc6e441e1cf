#!/bin/bash
# Round 4: the DELTA kernel's three fetch layouts (render_delta_kernel<.., NEE>, CT_DELTA_NEE=0|1|2): parity tests of each,
# then bench.py --estimator 1 alternating.  Output: gpurun_out/<tag>/delta_nee_ab.log
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-r04b}; mkdir -p "$OUT"
LOG="$OUT/delta_nee_ab.log"; : > "$LOG"
for NEE in ${VARIANTS:-1 2}; do
  echo "== parity, CT_DELTA_NEE=$NEE" | tee -a "$LOG"
  CT_DELTA_NEE=$NEE timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_parity_gaps.py -q -m gpu -x -k "delta" 2>&1 | tail -3 | tee -a "$LOG"
  [ "${PIPESTATUS[0]}" = 0 ] || { echo "parity FAILED for NEE=$NEE" | tee -a "$LOG"; exit 1; }
done
for round in 1 2; do
  for NEE in ${BENCH_VARIANTS:-0 1 2}; do
    CT_DELTA_NEE=$NEE python bench.py --estimator 1 --steps ${STEPS:-6} --no-cpu-baseline --no-pmc-traffic --no-delta-leg --no-progressive-leg 2>/dev/null > "$OUT/delta_nee${NEE}_$round.json" || { echo "bench failed NEE=$NEE" | tee -a "$LOG"; exit 1; }
    python - "$OUT/delta_nee${NEE}_$round.json" $NEE <<'PY' | tee -a "$LOG"
import json, sys
d = json.load(open(sys.argv[1]))
print("NEE", sys.argv[2], "DELTA", round(d["value"], 1), "Msamples/s", round(d["roofline"]["avg_launch_ms"], 2), "ms per launch")
PY
  done
done
