#!/bin/bash
# Sweep of the march-burst scheduler knobs (schedule only: results are bit-identical).
# CONFIGS="burst:scatter:idle ..." ; S = spp per step.
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/burst.log
if [ -n "${PYTEST_K:-}" ]; then
  timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "$PYTEST_K" > gpurun_out/pytest_gpu.log 2>&1
  rc=$?; tail -3 gpurun_out/pytest_gpu.log >> gpurun_out/burst.log
  [ $rc -ne 0 ] && { cat gpurun_out/burst.log; exit $rc; }
fi
for C in ${CONFIGS:-1:65:65 4:16:16}; do
  IFS=: read B SC ID <<< "$C"
  echo "== burst $B scatter $SC idle $ID ${EXTRA_ENV}" >> gpurun_out/burst.log
  env CT_MARCH_BURST=$B CT_BURST_SCATTER=$SC CT_BURST_IDLE=$ID ${EXTRA_ENV} timeout -k 10 200 python bench.py --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline --no-pmc-traffic --spp-per-step ${S:-128} 2>gpurun_out/burst_stderr.log | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('value %.1f Msamples/s  ms/step %.2f  launch_ms %.2f  frac %.3f' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac']))
        if 'scheduler_stats' in r: print(json.dumps(r['scheduler_stats']))
" >> gpurun_out/burst.log || { echo "FAILED" >> gpurun_out/burst.log; cat gpurun_out/burst.log; exit 1; }
grep -h "cloudtrace\]" gpurun_out/burst_stderr.log | tail -2 >> gpurun_out/burst.log || true
done
cat gpurun_out/burst.log
