#!/bin/bash
# Round 4: the DELTA kernel with a tracking burst right behind every scatter phase (CT_DELTA_FUSE=1) against the separate
# iterations (0), under a few schedule settings.  Knob test first (results never change).
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-r04h}; mkdir -p "$OUT"; LOG="$OUT/delta_fuse_ab.log"; : > "$LOG"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "scheduler_knobs and 1" 2>&1 | tail -2 | tee -a "$LOG"
for round in 1 2; do
SETS=${SETS:-0:16:2 1:16:2 1:8:2 1:24:2 1:16:3 1:32:2 0:16:2 1:16:2}
for SET in $SETS; do
  IFS=: read F SM MB <<< "$SET"
  CT_DELTA_FUSE=$F CT_SCATTER_MIN=$SM CT_MARCH_BURST=$MB python bench.py --estimator 1 --steps ${STEPS:-5} --no-cpu-baseline --no-pmc-traffic --no-delta-leg --no-progressive-leg 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('fuse $F scatter_min $SM burst $MB:', round(d['value'], 1), 'Msamples/s', round(d['roofline']['avg_launch_ms'], 2), 'ms')" | tee -a "$LOG"
done
[ -n "$ONE_ROUND" ] && break
done
