#!/bin/bash
# bench.py under several environment settings: ENVS="A=1 B=2|A=3" (| separates runs), ARGS = extra bench args
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/envbench.log
IFS='|' read -ra RUNS <<< "${ENVS:-X=0}"
for E in "${RUNS[@]}"; do
  echo "== $E ${ARGS:-}" >> gpurun_out/envbench.log
  env $E timeout -k 10 300 python bench.py --steps ${STEPS:-4} --warmup 1 --no-cpu-baseline --no-pmc-traffic ${ARGS:-} 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('value %.1f Msamples/s  ms/step %.2f  launch_ms %.2f  frac %.3f' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac']))
" >> gpurun_out/envbench.log || { echo FAILED >> gpurun_out/envbench.log; }
done
cat gpurun_out/envbench.log
