#!/bin/bash
# Generic A/B of library builds (python -m deepestscatter_amd.build --variant ...): parity tests of the first library, then the
# bench line of every library, alternating, ROUNDS times.   tools/gpu_lib_ab.sh <out dir> <estimator> <lib> [<lib> ...]
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$1; EST=$2; shift 2; mkdir -p "$OUT"; LOG="$OUT/lib_ab.log"; : >> "$LOG"
if [ -z "$SKIP_TESTS" ]; then
  CT_LIBRARY=$1 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_parity_gaps.py -q -m gpu -x -k "${TESTS:-delta or knobs or differential or continuation}" -p no:cacheprovider 2>&1 | tail -1 | sed "s/^/$1: /" | tee -a "$LOG"
  [ "${PIPESTATUS[0]}" = 0 ] || exit 1
fi
for round in $(seq 1 ${ROUNDS:-2}); do
  for LIB in "$@"; do
    CT_LIBRARY=$LIB python bench.py --estimator $EST --steps ${STEPS:-6} --no-cpu-baseline --no-pmc-traffic --no-delta-leg --no-progressive-leg ${BENCH_ARGS} 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$LIB est $EST:', round(d['value'], 1), 'Msamples/s', round(d['roofline']['avg_launch_ms'], 2), 'ms per launch')" | tee -a "$LOG"
  done
done
