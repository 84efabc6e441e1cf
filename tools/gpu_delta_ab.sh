#!/bin/bash
# DELTA as the main estimator, twice (quick A/B after a kernel change) + its parity tests.
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "delta or knobs or differential" 2>&1 | tail -2 || exit 1
for i in 1 2; do
  python bench.py --estimator 1 --steps ${STEPS:-8} --no-cpu-baseline --no-pmc-traffic --no-delta-leg 2>/dev/null > /tmp/b.json
  python - <<'PY'
import json
d=json.load(open('/tmp/b.json'))
print("DELTA", round(d["value"],1), "Msamples/s", round(d["roofline"]["avg_launch_ms"],2), "ms per launch")
PY
done
