set -o pipefail
: > gpurun_out/ab.log
for E in "CT_SHARED_DEPTH=64" "CT_SHARED_DEPTH=16" "CT_SHARED_DEPTH=256" "CT_SHARED_DEPTH=4" "CT_XCD_QUEUES=0" "CT_STATS=1"; do
  EXTRA_ENV="$E" CONFIGS="8:48:32" bash tools/gpu_burst_sweep.sh > /dev/null || exit 1
  cat gpurun_out/burst.log >> gpurun_out/ab.log
done
cat gpurun_out/ab.log
