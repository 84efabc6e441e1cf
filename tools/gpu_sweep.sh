#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/sweep.log
for S in ${SWEEP:-16 64}; do
  echo "== spp-per-step $S ${EXTRA_ENV}" >> gpurun_out/sweep.log
  env ${EXTRA_ENV} timeout -k 10 300 python bench.py --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline --no-pmc-traffic --spp-per-step $S 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('value %.1f Msamples/s  ms/step %.2f  launch_ms %.2f  frac %.3f  lookups/s %.3g' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac'], r['lookups_per_s']))
        if 'scheduler_stats' in r: print(json.dumps(r['scheduler_stats']))
" >> gpurun_out/sweep.log
done
cat gpurun_out/sweep.log
