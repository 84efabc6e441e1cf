#!/bin/bash
# Job sizes for render-ahead launches (80 subframes): does a finer job list (more jobs per pixel group in flight) pay there?
#   ENVS="X=0|CT_JOB_MAX=4|CT_JOB_MAX=8 CT_JOB_WORK=8" tools/gpu_ahead_jobs_ab.sh <tag>
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
OUT=gpurun_out/${1:-r03v}; mkdir -p "$OUT"
IFS='|' read -ra RUNS <<< "${ENVS:-X=0}"
for E in "${RUNS[@]}"; do
  echo "== $E"
  env $E timeout -k 10 200 python tools/progressive_bench.py --spp 10 --updates 96 --ahead ${AHEAD:-80} --reference-spp 0 2>/dev/null | grep -v summary | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('  ahead %3d: %.3f ms per update, %.0f Msamples/s' % (d['render_ahead'], d['ms_per_update'], d['Msamples_per_s']))
"
done
