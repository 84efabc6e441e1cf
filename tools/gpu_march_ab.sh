#!/bin/bash
# MARCH parity tests + the bench (no CPU leg, no PMC) under two settings of an environment knob:
#   KNOB=CT_NEE_CACHE gpurun -- 'bash tools/gpu_march_ab.sh'
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "bit_exact or knobs or differential or fetch_counters or free_space or adversarial" 2>&1 | tail -2 || exit 1
for V in 0 1 0 1; do
  env ${KNOB:-CT_NEE_CACHE}=$V python bench.py --steps ${STEPS:-6} --no-cpu-baseline --no-pmc-traffic --no-delta-leg ${BENCH_EXTRA:-} 2>/dev/null > /tmp/b.json
  python - $V <<'PY'
import json, sys
d = json.load(open('/tmp/b.json')); r = d["roofline"]
print("knob", sys.argv[1], round(d["value"], 1), "Msamples/s", round(r["avg_launch_ms"], 2), "ms per launch; fetches per sample", round(r["fetches_per_sample"], 2))
PY
done
