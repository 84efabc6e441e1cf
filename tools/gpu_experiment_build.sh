#!/bin/bash
# Timing experiments that need a different BUILD (results are wrong on purpose): FLAGS="-DCT_EXPERIMENT_ONE_LOAD" etc.
#   FLAGSETS="-DCT_EXPERIMENT_ONE_LOAD|-DCT_EXPERIMENT_NO_NEE_LOAD" ARGS="--no-delta-leg" bash tools/gpu_experiment_build.sh
# The library is rebuilt on the GPU box for every flag set and the normal build is restored at the end.
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
: > gpurun_out/experiment.log
IFS='|' read -ra SETS <<< "${FLAGSETS:- }"
for F in "${SETS[@]}"; do
  echo "== build flags: $F" >> gpurun_out/experiment.log
  CT_EXTRA_FLAGS="$F" python -m deepestscatter_amd.build --force > /dev/null 2>&1 || { echo "build failed" >> gpurun_out/experiment.log; continue; }
  timeout -k 10 300 python bench.py --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline --no-pmc-traffic ${ARGS:-} 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('value %.1f Msamples/s  ms/step %.2f  launch_ms %.2f  lookups/sample %.1f  fetches/sample %.2f  scatter events/launch %.3e' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r.get('lookups_per_sample', 0), r.get('fetches_per_sample', 0), r.get('counters_per_launch', {}).get('scatter_events', 0)) + ('  DELTA %.1f' % d['delta_estimator']['value'] if d.get('delta_estimator') else ''))
" >> gpurun_out/experiment.log || echo FAILED >> gpurun_out/experiment.log
done
python -m deepestscatter_amd.build --force > /dev/null 2>&1
cat gpurun_out/experiment.log
