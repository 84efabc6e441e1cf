#!/bin/bash
# A/B of environment settings: ENVS="A=1 B=2|A=3" (| separates runs; spaces separate variables)
set -o pipefail
: > gpurun_out/ab.log
IFS='|' read -ra RUNS <<< "${ENVS:-CT_XCD_QUEUES=0|CT_XCD_QUEUES=1}"
for E in "${RUNS[@]}"; do
  EXTRA_ENV="$E" CONFIGS="${CONFIGS:-8:48:32}" bash tools/gpu_burst_sweep.sh > /dev/null || exit 1
  cat gpurun_out/burst.log >> gpurun_out/ab.log
done
cat gpurun_out/ab.log
