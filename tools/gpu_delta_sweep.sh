#!/bin/bash
# DELTA schedule sweep: scatter threshold x burst length (bench.py --estimator 1).
cd "$GRAFT_REPO_ROOT"
for SM in ${SCATTER_MINS:-12 16 20 24}; do for B in ${BURSTS:-2 3}; do
  CT_SCATTER_MIN=$SM CT_MARCH_BURST=$B python bench.py --estimator 1 --steps 6 --no-cpu-baseline --no-pmc-traffic --no-delta-leg 2>/dev/null > /tmp/b.json
  python - $SM $B <<'PY'
import json, sys
d = json.load(open('/tmp/b.json'))
print("scatter_min", sys.argv[1], "burst", sys.argv[2], round(d["value"], 1))
PY
done; done
