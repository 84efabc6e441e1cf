#!/bin/bash
# Round 4: fused scheduler iterations (a scatter phase followed by the march / tracking burst in one iteration).
#   DELTA: libcloudtrace.so (fused, the default) vs libcloudtrace_nofuse.so, with a few scatter thresholds;
#   MARCH: libcloudtrace.so (separate iterations) vs libcloudtrace_mfuse.so.
# Parity of the non-default builds first (the knob tests: results never change).
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-r04i}; mkdir -p "$OUT"; LOG="$OUT/fuse_ab.log"; : > "$LOG"
for LIB in libcloudtrace.so libcloudtrace_mfuse.so; do
  CT_LIBRARY=$LIB timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "scheduler_knobs or north_star or differential" -p no:cacheprovider 2>&1 | tail -1 | sed "s/^/$LIB: /" | tee -a "$LOG"
done
run() {  # lib estimator scatter_min("-" = default) label
  local extra=""; [ "$3" != "-" ] && extra="CT_SCATTER_MIN=$3"
  env CT_LIBRARY=$1 $extra python bench.py --estimator $2 --steps ${STEPS:-5} --no-cpu-baseline --no-pmc-traffic --no-delta-leg --no-progressive-leg 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$4:', round(d['value'], 1), 'Msamples/s', round(d['roofline']['avg_launch_ms'], 2), 'ms per launch')" | tee -a "$LOG"
}
for round in 1 2; do
  run libcloudtrace_nofuse.so 1 - "DELTA separate iterations"
  run libcloudtrace.so 1 - "DELTA fused (default)"
  run libcloudtrace.so 1 8 "DELTA fused, scatter_min 8"
  run libcloudtrace.so 1 12 "DELTA fused, scatter_min 12"
  run libcloudtrace.so 0 - "MARCH separate iterations (default)"
  run libcloudtrace_mfuse.so 0 - "MARCH fused"
done
