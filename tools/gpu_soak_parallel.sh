#!/bin/bash
# Armed soak (CT_DEBUG_INVARIANTS=1: NaN-filled scratch, path conservation, alpha check) in several processes at once:
#   SEEDS="2001 2002 2003 2004" CASES=6000 SCRIPT=soak.py bash tools/gpu_soak_parallel.sh <tag>
# Every process logs to gpurun_out/soak_<tag>_<seed>.log; the last lines are collected in gpurun_out/soak_<tag>.txt.
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
TAG=${1:-r02}
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-4}
pids=()
for S in ${SEEDS:-2001 2002 2003 2004}; do
  timeout -k 10 ${LIMIT:-1000} python tools/${SCRIPT:-soak.py} $S ${CASES:-6000} > gpurun_out/soak_${TAG}_$S.log 2>&1 &
  pids+=($!)
done
# a progress line a minute while they run (gpurun takes silence for a hang)
while :; do
  alive=0
  for p in "${pids[@]}"; do kill -0 $p 2>/dev/null && alive=1; done
  [ $alive = 0 ] && break
  sleep 45
  for S in ${SEEDS:-2001 2002 2003 2004}; do tail -1 gpurun_out/soak_${TAG}_$S.log | cut -c1-120; done
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
{ echo "# ${SCRIPT:-soak.py}, seeds ${SEEDS:-2001 2002 2003 2004}, ${CASES:-6000} cases each, CT_DEBUG_INVARIANTS=1";
  for S in ${SEEDS:-2001 2002 2003 2004}; do echo "seed $S: $(grep -c 'MISMATCH\|INVARIANT' gpurun_out/soak_${TAG}_$S.log) reports; $(tail -1 gpurun_out/soak_${TAG}_$S.log)"; done; } > gpurun_out/soak_${TAG}.txt
cat gpurun_out/soak_${TAG}.txt
exit $rc
