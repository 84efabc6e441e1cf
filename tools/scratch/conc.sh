#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/conc; CLI=deepestscatter_amd/host/cloudtrace
for K in 1 3 6; do
  T0=$(date +%s.%N)
  PIDS=""
  for i in $(seq 1 $K); do
    timeout -k 10 400 $CLI collect procedural:256:$((100+i)) --out /tmp/t_$K_$i > gpurun_out/conc/k${K}_$i.log 2>&1 &
    PIDS="$PIDS $!"
  done
  for p in $PIDS; do wait $p || echo "fail $p"; done
  T1=$(date +%s.%N)
  echo "K=$K wall $(echo "$T1 - $T0" | bc) s"
  grep -h collect_timings gpurun_out/conc/k${K}_*.log | sed 's/.*radiance_ms": \([0-9.]*\).*updates": \([0-9]*\).*/radiance_ms \1 updates \2/'
done
